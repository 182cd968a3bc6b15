"""GPU probe: weight gradient of the 32 -> 32 cost-volume layer (192 x 48 x 160): wgrad_tapdh_kernel (F(2,3) along d and h, library
default) against wgrad_lds_kernel<.., WINO> (h only, tile hint 4) and the plain wgrad_lds_kernel (tile hint 6)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F

D, H, W = 192, 48, 160
x = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
w = (torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.03).requires_grad_(True)
go = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
fl = 2.0 * D * H * W * 27 * 32 * 32
ref = None
for hint in (6, 4, 0, 4, 0):
    F.TILE_HINT = hint
    y = F.conv3d(x, w, None, 1, 1)
    torch.autograd.grad(y, w, go, retain_graph=True)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); (gw,) = torch.autograd.grad(y, w, go, retain_graph=True); b.record()
    torch.cuda.synchronize()
    dt = sorted(a.elapsed_time(b) for a, b in ev)[5] * 1e-3
    if ref is None:
        ref = gw.clone()
    print(f"wgrad hint {hint}: {dt * 1e3:.3f} ms  {fl / dt / 1e12:.1f} TF/s (operator)  maxdiff {(gw - ref).abs().max().item():.2e} of {ref.abs().max().item():.1f}")
F.TILE_HINT = 0
