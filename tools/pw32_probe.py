"""GPU probe: conv_pw32_kernel (1x1x1, 32 -> 32 on the 192 x 48 x 160 cost volume) against the generic gather kernel (tile hint 8),
forward and data gradient; HBM rate = (read + write of the 189 MB tensors) / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F

D, H, W = 192, 48, 160
x = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(32, 32, 1, 1, 1, device="cuda") * 0.2
b = torch.randn(32, device="cuda")
go = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
nbytes = 2.0 * x.numel() * 4
ref = gref = None
for hint in (8, 0, 8, 0):
    F.TILE_HINT = hint
    xq = x.clone().requires_grad_(True)
    y = F.conv3d(xq, w, b, 1, 0)
    (gx,) = torch.autograd.grad(y, xq, go, retain_graph=True)
    torch.cuda.synchronize()
    def med(fn):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, e in ev:
            a.record(); fn(); e.record()
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(e) for a, e in ev)[5] * 1e-3
    with torch.no_grad():
        tf = med(lambda: F.conv3d(x, w, b, 1, 0))
    td = med(lambda: torch.autograd.grad(y, xq, go, retain_graph=True))
    if ref is None:
        ref, gref = y.detach().clone(), gx.clone()
    print(f"hint {hint}: fwd {tf * 1e6:6.1f} us ({nbytes / tf / 1e12:.2f} TB/s)  dgrad {td * 1e6:6.1f} us   maxdiff fwd {(y - ref).abs().max().item():.2e} "
          f"dgrad {(gx - gref).abs().max().item():.2e}")
F.TILE_HINT = 0
