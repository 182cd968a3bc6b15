set -u
out=gpurun_out/${1:-r4r}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fusion.py -q -x -k "norm or gn or bn" 2>&1 | tail -3
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs_x -o k -- python bench.py --steps 6 --warmup 2 --cpu-sample none --skip-forward-extra --skip-serial-replay > /dev/null 2>&1
cp $(find /tmp/profs_x -name "*kernel_stats.csv" | head -1) $out/kernel_stats_serial.csv
python tools/prof_summary.py $out/kernel_stats_serial.csv 8 60 > $out/summary_serial.txt 2>&1
grep -i "finalize\|GroupNorm\|total kernel" $out/summary_serial.txt | cut -c1-150
timeout 600 python bench.py --steps 10 --warmup 4 --cpu-sample none --skip-forward-extra --skip-serial-replay 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fp32 B=1', round(d['ms_per_step'],2), 'ms')"
