"""Time the Winograd forward pipeline with (a) the own LDS-streaming batched GEMM, (b) rocBLAS bmm, (c) the depth-fused kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for ci, co, (D, H, W) in ((128, 128, (128, 128, 16)), (384, 192, (128, 128, 16)), (192, 384, (128, 128, 16)), (256, 256, (64, 64, 8)),
                          (512, 512, (32, 32, 4)), (128, 128, (48, 12, 40))):
    x = torch.randn(1, ci, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.02
    res = []
    for own, fused in ((True, False), (False, False), (False, True)):
        F.WINO_OWN_GEMM, F.WINO_DEPTH_FUSED = own, fused
        with torch.no_grad():
            res.append(timeit(lambda: F.conv3d(x, w, None, 1, 1)))
    F.WINO_OWN_GEMM, F.WINO_DEPTH_FUSED = True, False
    with torch.no_grad():
        ref = F.conv3d(x, w, None, 1, 1)
        F.WINO_OWN_GEMM = False
        err = (F.conv3d(x, w, None, 1, 1) - ref).abs().max().item()
    print(f"{ci}->{co} @{D}x{H}x{W}: own GEMM {res[0]*1e3:6.2f} ms | rocBLAS {res[1]*1e3:6.2f} ms | depth-fused {res[2]*1e3:6.2f} ms | own-vs-rocBLAS maxdiff {err:.2e}", flush=True)
