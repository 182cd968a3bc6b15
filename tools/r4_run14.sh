set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 0 1; do for v in 0 1; do
SSBEV_WGRAD_STREAM=$w SSBEV_VT_STREAMS=$v timeout 600 python bench.py --steps 10 --warmup 4 --cpu-sample none --skip-forward-extra --skip-serial-replay 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fp32 B=1 WGRAD_STREAM=$w VT_STREAMS=$v', round(d['ms_per_step'],2), 'ms', d['roofline']['kernel'], round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_us'],1))"
done; done
