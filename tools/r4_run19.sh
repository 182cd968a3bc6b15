set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "tap_split or winograd_h_full" 2>&1 | tail -8
timeout 300 python tools/wgrad_dh_probe.py 2>&1 | grep -v amdgpu
