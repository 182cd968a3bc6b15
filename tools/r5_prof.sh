# kernel roll-up of the fp32 step under rocprofv3: usage bash tools/r5_prof.sh <tag> <serial|streams> [ENV=..] -- extra bench args
set -u
tag=$1; mode=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
envs=""; [ "$mode" = "serial" ] && envs="SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0"
env $envs "$@" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs_$tag$mode -o k -- python bench.py --steps 6 --warmup 2 --cpu-sample none --skip-forward-extra --skip-serial-replay > /dev/null 2>$out/prof_$mode.err
cp $(find /tmp/profs_$tag$mode -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$mode.csv
python tools/prof_summary.py $out/kernel_stats_$mode.csv 8 70 > $out/summary_$mode.txt 2>&1
head -16 $out/summary_$mode.txt
