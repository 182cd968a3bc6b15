cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -k "gwc" 2>&1 | tail -3
for v in "SSBEV_GWC_BWD=2" "SSBEV_GWC_BWD=3" "SSBEV_GWC_BWD=3 SSBEV_GWC_BWD_UNR=2" "SSBEV_GWC_BWD=3 SSBEV_GWC_BWD_UNR=1" "SSBEV_GWC_BWD=3 SSBEV_GWC_BWD_RUNCOST=2" "SSBEV_GWC_BWD=3 SSBEV_GWC_BWD_RUNCOST=3" "SSBEV_GWC_BWD=3 SSBEV_GWC_BWD_RUNCOST=8" "SSBEV_GWC_BWD=3 SSBEV_GWC_BWD_WGS=512" "SSBEV_GWC_BWD=3 SSBEV_GWC_BWD_WGS=192" "SSBEV_GWC_FWD_WGS=240" "SSBEV_GWC_FWD_WGS=480 SSBEV_GWC_FWD_RUNCOST=2" "SSBEV_GWC_FWD_WGS=480 SSBEV_GWC_FWD_RUNCOST=8" "SSBEV_GWC_FWD_WGS=480 SSBEV_GWC_FWD_THREADS=512"; do
echo "== $v"; env $v timeout 120 python tools/stream_probe.py 30 2>/dev/null | grep "gwc_warp_fwd\|fused"
done
