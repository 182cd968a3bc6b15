set -u
out=gpurun_out/${1:-r4m}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fusion.py -q -x 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3
for v in 0 1; do
SSBEV_TAPDH=$v timeout 600 python bench.py --steps 10 --warmup 4 --cpu-sample none --skip-forward-extra --skip-serial-replay 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fp32 B=1 SSBEV_TAPDH=$v', round(d['ms_per_step'],2), 'ms', d['roofline']['kernel'], round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_us'],1))"
done
