set -u
out=gpurun_out/${1:-r4v}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 600 python tools/layer_table.py kitti_d192 3 2>&1 | grep -v amdgpu > $out/layer_table.txt; grep "k111\|total timed" $out/layer_table.txt | cut -c1-150
for v in 0 1; do
SSBEV_PW32=$v timeout 600 python bench.py --steps 10 --warmup 4 --cpu-sample none --skip-forward-extra --skip-serial-replay 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fp32 B=1 SSBEV_PW32=$v', round(d['ms_per_step'],2), 'ms')"
done
