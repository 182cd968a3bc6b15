cd /root/repo
for v in "SSBEV_GWC_BWD_UNR=1" "SSBEV_GWC_BWD_UNR=2"; do
echo "== $v"
env SSBEV_GWC_BWD_PROF=1 $v timeout 300 python tools/stream_probe.py 20 2>&1 | grep "fused\|chunk "
done
