set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SSBEV_PRECISION=bf16 ATEN_DEPTH=3 timeout 600 python tools/aten_sites.py 500000 2>&1 | grep -v amdgpu | grep -i "_to_copy\|copy_\|clone\|large ATen" | head -50
