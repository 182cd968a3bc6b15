set -u
out=gpurun_out/${1:-r4o}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fusion.py -q -x -k "bri or attention or gemm or softmax" 2>&1 | tail -3
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 600 python tools/layer_table.py kitti_d192 3 2>&1 | grep -v amdgpu > $out/layer_table.txt; grep "bri\|total timed" $out/layer_table.txt | cut -c1-150
