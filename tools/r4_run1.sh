set -u
out=gpurun_out/r4b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/micro/tr_probe.hip -o /tmp/tr_probe 2>/dev/null && /tmp/tr_probe > $out/tr_probe.txt 2>&1
head -20 $out/tr_probe.txt
timeout 1200 python -m pytest tests/test_gpu_bf16_storage.py -x -q 2>&1 | tail -25 > $out/pytest_bf16s.txt; cat $out/pytest_bf16s.txt
timeout 900 python -m pytest tests/test_gpu_fusion.py -q -k "flat_gradient or used_twice or side_streams" 2>&1 | tail -8 > $out/pytest_advice.txt; cat $out/pytest_advice.txt
