set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "0 0" "9 0" "23 0" "0 12" "9 12" "12 24"; do set -- $cfg
env $( [ $1 != 0 ] && echo SSBEV_TAPDH_GPC=$1 ) $( [ $2 != 0 ] && echo SSBEV_WGRAD_DH_GPC=$2 ) timeout 600 python bench.py --steps 10 --warmup 4 --cpu-sample none --skip-forward-extra --skip-serial-replay 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('tapdh gpc $1 wgrad_dh gpc $2:', round(d['ms_per_step'],2), 'ms', round(d['roofline']['avg_launch_us'],1))"
done
