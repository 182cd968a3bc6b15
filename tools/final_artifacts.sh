#!/bin/bash
# Regenerate the judged artifacts of a round from ONE tree on the GPU box: usage (inside gpurun) bash tools/final_artifacts.sh r3z
# Writes gpurun_out/<tag>/*; copy what is to be judged into profiles/<tag>_*.  Every step runs under `timeout`.
set -u
tag=${1:-r6z}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?" >> $out/smoke.txt
SSBEV_BENCH_DETAIL=$out/bench_detail.json timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err
timeout 300 python bench.py --steps 10 --warmup 5 --cpu-sample none --precision bf16 2>/dev/null | tail -1 > $out/bench_line_bf16.json
timeout 300 python bench.py --steps 8 --warmup 5 --cpu-sample none --precision bf16 --batch 2 --skip-forward-extra 2>/dev/null | tail -1 > $out/bench_line_bf16_b2.json
# bf16 storage mode (configs[3]): serial-schedule kernel roll-up at B = 2, layer table at B = 1, PMC traffic of its kernels at B = 1
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb_$tag -o k -- python bench.py --steps 6 --warmup 6 --cpu-sample none --skip-forward-extra --skip-serial-replay --precision bf16 --batch 2 > /dev/null 2>&1
cp $(find /tmp/profb_$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats_bf16_b2_serial.csv
python tools/prof_summary.py $out/kernel_stats_bf16_b2_serial.csv 12 60 > $out/summary_bf16_b2_serial.txt 2>&1
SSBEV_PRECISION=bf16 SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 600 python tools/layer_table.py kitti_d192 3 2>&1 | grep -v amdgpu > $out/layer_table_bf16.txt
for c in FETCH_SIZE WRITE_SIZE; do
  SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcb_${tag}_$c -o p -- python bench.py --steps 2 --warmup 2 --cpu-sample none --skip-forward-extra --skip-serial-replay --precision bf16 > /dev/null 2>&1
  cp $(find /tmp/pmcb_${tag}_$c -name "*counter_collection.csv" | head -1) /tmp/pmcb_${tag}_$c.csv
done
python tools/pmc_traffic.py /tmp/pmcb_${tag}_FETCH_SIZE.csv /tmp/pmcb_${tag}_WRITE_SIZE.csv $out/pmc_traffic_bf16.json > $out/pmc_traffic_bf16.txt 2>&1
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcmb_$tag -o p -- python bench.py --steps 2 --warmup 2 --cpu-sample none --skip-forward-extra --skip-serial-replay --precision bf16 > /dev/null 2>&1
python tools/pmc_mfma.py $(find /tmp/pmcmb_$tag -name "*counter_collection.csv" | head -1) > $out/pmc_mfma_bf16.txt 2>&1
# (a) the timed schedule (two streams): per-kernel averages agree with `roofline` of the bench line
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o k -- python bench.py --steps 6 --warmup 2 --cpu-sample none --skip-forward-extra --skip-serial-replay > $out/bench_under_rocprof.log 2>&1
cp $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python tools/prof_summary.py $out/kernel_stats.csv 8 60 > $out/summary.txt 2>&1
python tools/overlap_report.py $(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1) 8 > $out/overlap.txt 2>&1
# (b) the serial schedule (side streams off): per-kernel averages agree with `roofline_serial_replay`
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs_$tag -o k -- python bench.py --steps 6 --warmup 2 --cpu-sample none --skip-forward-extra --skip-serial-replay > /dev/null 2>&1
cp $(find /tmp/profs_$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats_serial.csv
python tools/prof_summary.py $out/kernel_stats_serial.csv 8 60 > $out/summary_serial.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${tag}_$c -o p -- python bench.py --steps 2 --warmup 1 --cpu-sample none --skip-forward-extra --skip-serial-replay > /dev/null 2>&1
  cp $(find /tmp/pmc_${tag}_$c -name "*counter_collection.csv" | head -1) /tmp/pmc_${tag}_$c.csv
done
python tools/pmc_traffic.py /tmp/pmc_${tag}_FETCH_SIZE.csv /tmp/pmc_${tag}_WRITE_SIZE.csv $out/pmc_traffic.json > $out/pmc_traffic.txt 2>&1
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcm_$tag -o p -- python bench.py --steps 2 --warmup 1 --cpu-sample none --skip-forward-extra --skip-serial-replay > /dev/null 2>&1
python tools/pmc_mfma.py $(find /tmp/pmcm_$tag -name "*counter_collection.csv" | head -1) > $out/pmc_mfma.txt 2>&1
timeout 600 python tools/stream_probe.py > $out/stream_probe.txt 2>&1
timeout 300 python tools/tapdh_probe.py 2>&1 | grep -v amdgpu > $out/tapdh_probe.txt
timeout 300 python tools/wgrad_dh_probe.py 2>&1 | grep -v amdgpu > $out/wgrad_dh_probe.txt
timeout 300 python tools/pool_prepare_probe.py 2>&1 | grep -v amdgpu > $out/pool_prepare_probe.txt
timeout 300 python tools/gather_split_probe.py 2>&1 | grep -v amdgpu > $out/gather_split_probe.txt
timeout 300 python tools/pw32_probe.py 2>&1 | grep -v amdgpu > $out/pw32_probe.txt
timeout 300 python tools/wino_df_probe.py 20 2>&1 | grep -v amdgpu > $out/wino_df_probe.txt
timeout 300 python tools/host_time_probe.py 2>&1 | grep -v amdgpu > $out/host_time.txt
bash tools/sq_counters.sh $tag > /dev/null 2>&1
SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 600 python tools/layer_table.py kitti_d192 3 2>&1 | grep -v amdgpu > $out/layer_table.txt
timeout 600 python tools/bucket_timeline.py 64 300 2>&1 | grep -v amdgpu > $out/bucket_timeline.txt
# configs[4]: one whole-step roofline object per ablation mode
python - > $out/ablation_roofline.json <<PYEOF
import json, subprocess, sys
res = {}
for mode in ("full", "stereo_only", "bev_only"):
    import os
    env = dict(os.environ, SSBEV_BENCH_DETAIL="/tmp/ablation_detail.json")
    out = subprocess.run([sys.executable, "bench.py", "--steps", "8", "--warmup", "3", "--cpu-sample", "none", "--skip-forward-extra",
                          "--skip-serial-replay", "--ablation", mode], capture_output=True, text=True, timeout=600, env=env).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    sr = json.load(open("/tmp/ablation_detail.json"))["step_roofline"]      # the per-group table lives in the detail file
    res[mode] = {"ms_per_step": d["ms_per_step"], "voxels_per_s": d["value"], "step_roofline": {k: sr[k] for k in ("floor_ms", "frac", "operator_floor_ms", "operator_frac")},
                 "groups_ms_floor": {k: round(v["floor_ms_per_step"], 3) for k, v in sr["groups"].items()},
                 "roofline_kernel": d["roofline"] and {k: d["roofline"][k] for k in ("kernel", "frac", "avg_launch_us", "launches_per_step")}}
print(json.dumps({"_doc": "BASELINE configs[4]: fp32, kitti_d192, B = 1, fwd + bwd, one MI355X; step_roofline as in bench.py", "modes": res}, indent=1))
PYEOF
for cfgline in "--config kitti_d112" "--batch 2" "--ablation stereo_only" "--ablation bev_only"; do
  echo "$cfgline: $(timeout 300 python bench.py --steps 8 --warmup 3 --cpu-sample none --skip-forward-extra --skip-serial-replay $cfgline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms/step', round(d['value']/1e6,2), 'M voxels/s')")" >> $out/other_configs.txt
done
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $out/pytest_gpu.txt
tail -3 $out/pytest_gpu.txt; head -14 $out/summary.txt; cat $out/other_configs.txt; python -c "
import json; d=json.loads(open('$out/bench_line.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('alone_on_device'), d['cpu_baseline'])"
