cd /root/repo
for i in 1 2; do
SSBEV_TAPH_ALIGNED=1 timeout 300 python bench.py --steps 20 --warmup 5 --skip-serial-replay 2>/dev/null | python -c "import sys,json; print('aligned=1', json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"
SSBEV_TAPH_ALIGNED=0 timeout 300 python bench.py --steps 20 --warmup 5 --skip-serial-replay 2>/dev/null | python -c "import sys,json; print('aligned=0', json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"
done
