mkdir -p gpurun_out/r3k; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fusion.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python tools/taph_probe.py 2>&1 | grep -v amdgpu | grep "hint 0" | tee gpurun_out/r3k/taph_probe.txt
timeout 300 python tools/wino_df_probe.py 2>&1 | grep -v amdgpu | tail -12 | tee gpurun_out/r3k/wino_df_probe.txt
timeout 300 python tools/gemm_probe.py 2>&1 | grep -v amdgpu | grep -E "bri|aspp|fpn k4|2-D" | tee gpurun_out/r3k/gemm_probe.txt
for i in 1 2; do timeout 120 python bench.py --steps 10 --warmup 3 --cpu-sample none --skip-forward-extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:(round(v['frac'],3), round(v['avg_launch_us'],1)) for k,v in d['roofline_serial_replay'].items()})"; done
