set -u
out=gpurun_out/${1:-r4e}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_bf16_storage.py -q 2>&1 | tail -8 > $out/pytest_bf16s.txt; tail -4 $out/pytest_bf16s.txt
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fusion.py -q -x 2>&1 | tail -8 > $out/pytest_kernels.txt; tail -4 $out/pytest_kernels.txt
for b in 1 2; do
timeout 600 python bench.py --steps 8 --warmup 5 --cpu-sample none --precision bf16 --batch $b --skip-forward-extra --skip-serial-replay 2>$out/bench_b$b.err | tail -1 > $out/bench_line_bf16_b$b.json
python -c "
import json; d=json.loads(open('$out/bench_line_bf16_b$b.json').read()); print('bf16 storage B=$b', d['ms_per_step'], d['value'])" || tail -20 $out/bench_b$b.err
done
timeout 600 python bench.py --steps 10 --warmup 4 --cpu-sample none --skip-forward-extra --skip-serial-replay 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fp32 B=1', d['ms_per_step'], d['value'])"
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b2 -o k -- python bench.py --steps 6 --warmup 6 --cpu-sample none --skip-forward-extra --skip-serial-replay --precision bf16 --batch 2 > $out/bench_prof.log 2>&1
cp $(find /tmp/prof_b2 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_bf16_b2.csv
python tools/prof_summary.py $out/kernel_stats_bf16_b2.csv 12 45 > $out/summary_bf16_b2.txt 2>&1
head -16 $out/summary_bf16_b2.txt | cut -c1-180
python tools/prof_summary.py $out/kernel_stats_bf16_b2.csv 12 200 "gn_|gn2_|bn_" | cut -c1-200
