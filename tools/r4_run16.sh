set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "wino" 2>&1 | tail -3
timeout 300 python tools/wino_df_probe.py 10 2>&1 | grep -v amdgpu | cut -c1-130
