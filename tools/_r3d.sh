mkdir -p gpurun_out/r3d; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fusion.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r3d/t1.txt; cat gpurun_out/r3d/t1.txt
for v in "1 0" "0 0" "1 1" "0 1" "1 1" "1 0"; do set -- $v; SSBEV_GRAD_SLOTS=$1 SSBEV_VT_STREAMS=$2 python bench.py --steps 10 --warmup 3 --cpu-sample none --skip-forward-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"slots=$1 streams=$2\", d[\"ms_per_step\"])"; done | tee gpurun_out/r3d/ab.txt
for sl in 1 0; do
SSBEV_GRAD_SLOTS=$sl SSBEV_VT_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s$sl -o k -- python bench.py --steps 6 --warmup 2 --cpu-sample none --skip-forward-extra > /dev/null 2>&1
cp $(find /tmp/prof_s$sl -name "*kernel_stats.csv" | head -1) gpurun_out/r3d/kernel_stats_slots$sl.csv
python tools/prof_summary.py gpurun_out/r3d/kernel_stats_slots$sl.csv 8 70 > gpurun_out/r3d/summary_slots$sl.txt
done
head -14 gpurun_out/r3d/summary_slots1.txt; head -14 gpurun_out/r3d/summary_slots0.txt
