cd /root/repo
timeout 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
SSBEV_NORM_CAT=1 timeout 300 python bench.py --steps 20 --warmup 5 --skip-serial-replay 2>/dev/null | python -c "import sys,json; print('norm_cat=1', json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"
SSBEV_NORM_CAT=0 timeout 300 python bench.py --steps 20 --warmup 5 --skip-serial-replay 2>/dev/null | python -c "import sys,json; print('norm_cat=0', json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"
done
