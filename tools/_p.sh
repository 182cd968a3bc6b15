cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5u; mkdir -p $out
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb -o k -- python bench.py --steps 6 --warmup 4 --cpu-sample none --skip-forward-extra --skip-serial-replay --precision bf16 --batch 2 > gpurun_out/r5u/prof_bf16.log 2>&1
cp $(find /tmp/profb -name "*kernel_stats.csv" | head -1) $out/kernel_stats_bf16b2_serial.csv
python tools/prof_summary.py $out/kernel_stats_bf16b2_serial.csv 10 70 > $out/summary_bf16b2_serial.txt 2>&1
