set -u
out=gpurun_out/${1:-r4k}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "pool or lift or bev or voxel" 2>&1 | tail -3
for items in 8 16 32; do
echo "== items $items" | tee -a $out/pool_prepare_probe.txt
rm -rf /tmp/prof_pp
SSBEV_POOL_ITEMS=$items timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pp -o k -- python tools/pool_prepare_probe.py one 2>&1 | grep -v amdgpu | grep kitti | tee -a $out/pool_prepare_probe.txt
python - <<PY | tee -a $out/pool_prepare_probe.txt
import csv, glob
f = glob.glob('/tmp/prof_pp/**/*kernel_stats.csv', recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    if 'csr_' in r['Name'] or 'fillBuffer' in r['Name']:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
        if 'csr_' in r['Name']: tot += float(r['AverageNs'])/1e3
print('sum of csr kernel averages', round(tot, 1), 'us')
PY
done
