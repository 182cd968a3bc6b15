cd /root/repo
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --skip-serial-replay 2>/dev/null | python -c "import sys,json; print('fused tails on ', json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"
SSBEV_OCC_TAIL=0 SSBEV_DEPTH_BCE=0 SSBEV_GEOM_FUSED=0 timeout 300 python bench.py --steps 20 --warmup 5 --skip-serial-replay 2>/dev/null | python -c "import sys,json; print('fused tails off', json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"
done
