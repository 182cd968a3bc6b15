"""GPU probe: conv_tapdh_kernel (F(2,3) along d and h, library default) against conv_taph_kernel (h only, tile hint 4) and
conv_tap_kernel (direct, tile hint 6) on the 32 -> 32 cost-volume layer (192 x 48 x 160), forward and data gradient."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F

D, H, W = 192, 48, 160
x = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.03
bias = torch.randn(32, device="cuda")
fl = 2.0 * D * H * W * 27 * 32 * 32
ref = None
for hint in (6, 4, 0, 4, 0):
    F.TILE_HINT = hint
    with torch.no_grad():
        y = F.conv3d(x, w, bias, 1, 1)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); y = F.conv3d(x, w, bias, 1, 1); b.record()
        torch.cuda.synchronize()
        dt = sorted(a.elapsed_time(b) for a, b in ev)[5] * 1e-3
    if ref is None:
        ref = y
    print(f"fwd  hint {hint}: {dt * 1e3:.3f} ms  {fl / dt / 1e12:.1f} TF/s (operator)  maxdiff {(y - ref).abs().max().item():.2e} "
          f"of {ref.abs().max().item():.2f}")
gref = None
go = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
for hint in (6, 4, 0):
    F.TILE_HINT = hint
    xq = x.clone().requires_grad_(True)
    y = F.conv3d(xq, w, bias, 1, 1)
    y.backward(go)
    torch.cuda.synchronize()
    if gref is None:
        gref = xq.grad.clone()
    print(f"dgrad hint {hint}: maxdiff {(xq.grad - gref).abs().max().item():.2e} of {gref.abs().max().item():.2f}")
F.TILE_HINT = 0
