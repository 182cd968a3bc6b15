set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" 2>&1 | tail -4
timeout 300 python tools/pw32_probe.py 2>&1 | grep -v amdgpu
