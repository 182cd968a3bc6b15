"""Tile-hint probe for the stride-2 / transposed cost-volume layers (fwd and dgrad separately)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
CASES = [("conv 32->64 s2", 32, 64, (192, 48, 160), False), ("deconv 64->32 s2", 64, 32, (96, 24, 80), True),
         ("conv 64->128 s2", 64, 128, (96, 24, 80), False), ("deconv 128->64 s2", 128, 64, (48, 12, 40), True)]
for name, ci, co, (D, H, W), tr in CASES:
    x = torch.randn(1, ci, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    w = torch.randn((ci, co, 3, 3, 3) if tr else (co, ci, 3, 3, 3), device="cuda") * 0.05
    f = (lambda: F.conv_transpose3d(x, w, None, 2, 1, 1)) if tr else (lambda: F.conv3d(x, w, None, 2, 1))
    y = f(); go = torch.randn_like(y)
    row = []
    for hint in (0, 111, 112, 114, 121, 122, 124, 211, 212, 214, 221, 222, 224, 910, 920):
        F.TILE_HINT = hint
        try:
            tf = timeit(f)
            def bw():
                x.grad = None; f().backward(go)
            tb = timeit(bw, 3) - tf
            row.append(f"{hint}:{tf*1e3:.2f}/{tb*1e3:.2f}")
        except Exception as e:
            row.append(f"{hint}:ERR")
    F.TILE_HINT = 0
    print(f"{name:18s} fwd/dgrad ms  " + " ".join(row), flush=True)
