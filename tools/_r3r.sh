cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "depth_fused" 2>&1 | tail -2
for v in 2 1; do echo "NW2_MT=$v"; SSBEV_DF_NW2_MT=$v timeout 300 python tools/wino_df_probe.py 2>&1 | grep -v amdgpu | grep "64-> 64"; done
for v in 2 1 2 1; do SSBEV_DF_NW2_MT=$v timeout 120 python bench.py --steps 10 --warmup 3 --cpu-sample none --skip-forward-extra --skip-serial-replay 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NW2_MT=$v', d['ms_per_step'])"; done
