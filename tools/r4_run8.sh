set -u
out=gpurun_out/${1:-r4i}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_bf16_storage.py -q 2>&1 | tail -30 > $out/pytest_bf16s.txt; tail -30 $out/pytest_bf16s.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -k "bf16" 2>&1 | tail -4
timeout 300 python tools/bf16_grad_probe.py 2>&1 | grep -v amdgpu > $out/bf16_grad_probe.txt
for b in 1 2; do
timeout 600 python bench.py --steps 8 --warmup 5 --cpu-sample none --precision bf16 --batch $b --skip-forward-extra --skip-serial-replay 2>$out/bench_b$b.err | tail -1 > $out/bench_line_bf16_b$b.json
python -c "
import json; d=json.loads(open('$out/bench_line_bf16_b$b.json').read()); print('bf16 storage B=$b', d['ms_per_step'], d['value'])" || tail -20 $out/bench_b$b.err
done
SSBEV_WGRING16=0 timeout 600 python bench.py --steps 8 --warmup 5 --cpu-sample none --precision bf16 --batch 2 --skip-forward-extra --skip-serial-replay 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('B=2 ring wgrad off', d['ms_per_step'])"
SSBEV_PRECISION=bf16 SSBEV_WGRAD_STREAM=0 SSBEV_VT_STREAMS=0 timeout 600 python tools/layer_table.py kitti_d192 3 2>&1 | grep -v amdgpu > $out/layer_table_bf16.txt; head -30 $out/layer_table_bf16.txt | cut -c1-170
