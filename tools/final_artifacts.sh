#!/bin/bash
# Regenerate the judged artifacts of a round from ONE tree on the GPU box: usage (inside gpurun) bash tools/final_artifacts.sh r2z
# Writes gpurun_out/<tag>/*; copy what is to be judged into profiles/<tag>_*.
set -u
tag=${1:-r2z}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?" >> $out/smoke.txt
timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample none --precision bf16 2>/dev/null | tail -1 > $out/bench_line_bf16.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o k -- python bench.py --steps 6 --warmup 2 --cpu-sample none --skip-forward-extra > $out/bench_under_rocprof.log 2>&1
cp $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python tools/prof_summary.py $out/kernel_stats.csv 8 60 > $out/summary.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${tag}_$c -o p -- python bench.py --steps 2 --warmup 1 --cpu-sample none --skip-forward-extra > /dev/null 2>&1
  cp $(find /tmp/pmc_${tag}_$c -name "*counter_collection.csv" | head -1) /tmp/pmc_${tag}_$c.csv
done
python tools/pmc_traffic.py /tmp/pmc_${tag}_FETCH_SIZE.csv /tmp/pmc_${tag}_WRITE_SIZE.csv $out/pmc_traffic.json > $out/pmc_traffic.txt 2>&1
timeout 600 python tools/stream_probe.py > $out/stream_probe.txt 2>&1
timeout 600 python tools/layer_table.py kitti_d192 3 2>&1 | grep -v amdgpu > $out/layer_table.txt
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $out/pytest_gpu.txt
tail -3 $out/pytest_gpu.txt; head -12 $out/summary.txt; python -c "
import json; d=json.loads(open('$out/bench_line.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline'])"
