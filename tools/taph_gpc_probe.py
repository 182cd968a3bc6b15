"""GPU probe: conv_taph_kernel on the 32 -> 32 cost-volume layer; SSBEV_TAPH_GPC overrides the chunk length (row pairs per
workgroup) that launch_conv_taph's cost model would pick."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F
D, H, W = 192, 48, 160
x = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.03
with torch.no_grad():
    for _ in range(3):
        y = F.conv3d(x, w, None, 1, 1)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            y = F.conv3d(x, w, None, 1, 1)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20)
print(os.environ.get("SSBEV_TAPH_GPC"), " ".join(f"{t*1e3:.3f}" for t in ts))
