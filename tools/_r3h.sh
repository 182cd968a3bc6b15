mkdir -p gpurun_out/r3h; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "tap or conv3d_family or wgrad" 2>&1 | tail -3
timeout 300 python tools/taph_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3h/taph_probe.txt
for i in 1 2; do timeout 120 python bench.py --steps 10 --warmup 3 --cpu-sample none --skip-forward-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"], d[\"roofline\"][\"frac\"], d[\"roofline\"].get(\"avg_launch_us\"))"; done
