cd /root/repo
ATEN_DEPTH=3 ATEN_ROWS=2000 timeout 600 python tools/aten_sites.py 1 > gpurun_out/aten_deep2.txt 2>&1
