set -u
out=gpurun_out/r4a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 1 2; do
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$b -o k -- python bench.py --steps 6 --warmup 6 --cpu-sample none --skip-forward-extra --skip-serial-replay --precision bf16 --batch $b > $out/bench_bf16_b$b.log 2>&1
cp $(find /tmp/prof_b$b -name "*kernel_stats.csv" | head -1) $out/kernel_stats_bf16_b$b.csv
python tools/prof_summary.py $out/kernel_stats_bf16_b$b.csv 12 70 > $out/summary_bf16_b$b.txt 2>&1
done
timeout 300 python bench.py --steps 8 --warmup 5 --cpu-sample none --precision bf16 --batch 2 --skip-forward-extra --skip-serial-replay 2>/dev/null | tail -1 > $out/bench_line_bf16_b2.json
timeout 300 python bench.py --steps 8 --warmup 5 --cpu-sample none --precision bf16 --batch 1 --skip-forward-extra --skip-serial-replay 2>/dev/null | tail -1 > $out/bench_line_bf16_b1.json
head -14 $out/summary_bf16_b2.txt
