set -u
out=gpurun_out/${1:-r4n}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" 2>&1 | tail -3
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 600 python tools/layer_table.py kitti_d192 3 2>&1 | grep -v amdgpu > $out/layer_table.txt; head -45 $out/layer_table.txt | cut -c1-150
timeout 600 python bench.py --steps 10 --warmup 4 --cpu-sample none --skip-forward-extra --skip-serial-replay 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fp32 B=1', round(d['ms_per_step'],2), 'ms', d['roofline']['kernel'], round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_us'],1))"
