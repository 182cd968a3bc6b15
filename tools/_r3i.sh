mkdir -p gpurun_out/r3i; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample none 2>/dev/null | tail -1 > gpurun_out/r3i/bench.json
python -c "
import json; d=json.load(open('gpurun_out/r3i/bench.json')); print(d['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'replay', {k:(round(v['frac'],3), round(v['avg_launch_us'],1)) for k,v in d['roofline_serial_replay'].items()}, 'step', d['step_roofline']['frac'], 'fwd', d['forward_only']['ms_per_step'])"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_i -o k -- python bench.py --steps 6 --warmup 2 --cpu-sample none --skip-forward-extra --skip-serial-replay > /dev/null 2>&1
cp $(find /tmp/prof_i -name "*kernel_stats.csv" | head -1) gpurun_out/r3i/kernel_stats.csv
cp $(find /tmp/prof_i -name "*kernel_trace.csv" | head -1) /tmp/kernel_trace.csv
python tools/prof_summary.py gpurun_out/r3i/kernel_stats.csv 8 70 > gpurun_out/r3i/summary.txt
head -16 gpurun_out/r3i/summary.txt
head -1 /tmp/kernel_trace.csv
python tools/overlap_report.py /tmp/kernel_trace.csv 8 | tee gpurun_out/r3i/overlap.txt
