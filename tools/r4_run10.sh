set -u
out=gpurun_out/${1:-r4l}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "tap_split or winograd_h_full" 2>&1 | tail -12
timeout 300 python tools/tapdh_probe.py 2>&1 | grep -v amdgpu | tee $out/tapdh_probe.txt
for gpc in 6 8 9 12 24; do SSBEV_TAPDH_GPC=$gpc timeout 120 python tools/taph_gpc_probe.py 2>&1 | grep -v amdgpu | sed "s/^/gpc $gpc: /" | tee -a $out/tapdh_probe.txt; done
