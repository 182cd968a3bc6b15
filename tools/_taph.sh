cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv" 2>&1 | tail -3
for v in "SSBEV_TAPH_ALIGNED=0" "SSBEV_TAPH_ALIGNED=1" "SSBEV_TAPH_ALIGNED=18" "SSBEV_TAPH_ALIGNED=20" "SSBEV_TAPH_ALIGNED=16" "SSBEV_TAPH_ALIGNED=0" "SSBEV_TAPH_ALIGNED=1"; do
echo "== $v"; env $v timeout 120 python tools/taph_gpc_probe.py 2>/dev/null | tail -1
done
for v in "SSBEV_TAPH_ALIGNED=1"; do
for c in TCC_EA0_RDREQ_sum; do
rm -rf /tmp/pmc; env $v timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc -o p -- python tools/taph_gpc_probe.py > /dev/null 2>&1
f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
python - "$f" "$v" "$c" <<'PY'
import csv,sys,collections
tot=collections.Counter(); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv_taph' in r['Kernel_Name']:
        tot[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in tot: print(sys.argv[2], k, tot[k]/n[k], n[k])
PY
done
done
