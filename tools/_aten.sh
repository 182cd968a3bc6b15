cd /root/repo
ATEN_DEPTH=4 ATEN_ROWS=2000 timeout 600 python tools/aten_sites.py 1 > gpurun_out/aten_deep.txt 2>&1
