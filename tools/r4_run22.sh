set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mt in 2 1; do echo "== SSBEV_DF_MT=$mt"; SSBEV_DF_MT=$mt timeout 300 python tools/wino_df_probe.py 10 2>&1 | grep -v amdgpu | cut -c1-118; done
