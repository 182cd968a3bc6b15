# Baseline of a tree on the GPU box: GPU tests, fp32 / bf16 bench lines, serial-schedule kernel roll-up (fp32).  usage: bash tools/r4_base.sh <tag> [notests]
set -u
tag=${1:-r4j}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 10 --warmup 4 --cpu-sample none --skip-forward-extra 2>$out/bench.err | tail -1 > $out/bench_line.json
python -c "
import json; d=json.loads(open('$out/bench_line.json').read()); print('fp32 B=1', d['ms_per_step'], d['value'], d['roofline']['frac'])" || tail -20 $out/bench.err
for b in 1 2; do
timeout 600 python bench.py --steps 8 --warmup 5 --cpu-sample none --precision bf16 --batch $b --skip-forward-extra --skip-serial-replay 2>$out/bench_b$b.err | tail -1 > $out/bench_line_bf16_b$b.json
python -c "
import json; d=json.loads(open('$out/bench_line_bf16_b$b.json').read()); print('bf16 storage B=$b', d['ms_per_step'], d['value'])" || tail -20 $out/bench_b$b.err
done
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs_$tag -o k -- python bench.py --steps 6 --warmup 2 --cpu-sample none --skip-forward-extra --skip-serial-replay > /dev/null 2>&1
cp $(find /tmp/profs_$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats_serial.csv
python tools/prof_summary.py $out/kernel_stats_serial.csv 8 60 > $out/summary_serial.txt 2>&1
head -40 $out/summary_serial.txt
if [ "${2:-}" != "notests" ]; then
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $out/pytest_gpu.txt
tail -4 $out/pytest_gpu.txt
fi
