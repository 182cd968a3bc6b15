# serial-schedule kernel roll-up of the bf16 storage mode: usage bash tools/r4_prof_bf16.sh <tag> [batch]
set -u
out=gpurun_out/${1:-r4u}
b=${2:-2}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb_$1 -o k -- python bench.py --steps 6 --warmup 4 --cpu-sample none --skip-forward-extra --skip-serial-replay --precision bf16 --batch $b > /dev/null 2>&1
cp $(find /tmp/profb_$1 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_bf16_b${b}_serial.csv
python tools/prof_summary.py $out/kernel_stats_bf16_b${b}_serial.csv 10 70 > $out/summary_bf16_b${b}_serial.txt 2>&1
head -60 $out/summary_bf16_b${b}_serial.txt
