# serial-schedule kernel roll-up of the fp32 step (side streams off): usage bash tools/r4_prof_serial.sh <tag>
set -u
out=gpurun_out/${1:-r4t}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs_$1 -o k -- python bench.py --steps 6 --warmup 2 --cpu-sample none --skip-forward-extra --skip-serial-replay > /dev/null 2>&1
cp $(find /tmp/profs_$1 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_serial.csv
python tools/prof_summary.py $out/kernel_stats_serial.csv 8 70 > $out/summary_serial.txt 2>&1
head -16 $out/summary_serial.txt
