set -u
out=gpurun_out/${1:-r4q}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 600 python tools/layer_table.py kitti_d192 3 2>&1 | grep -v amdgpu > $out/layer_table.txt; grep "winoDF\|total timed" $out/layer_table.txt | cut -c1-150
