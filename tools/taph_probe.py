"""GPU probe: conv_taph_kernel (F(2,3) along h) against conv_tap_kernel (tile hint 6) on the 32 -> 32 cost-volume layer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F

D, H, W = 192, 48, 160
x = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.03
ref = None
for hint in (6, 0, 6, 0):
    F.TILE_HINT = hint
    with torch.no_grad():
        y = F.conv3d(x, w, None, 1, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            y = F.conv3d(x, w, None, 1, 1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    if ref is None:
        ref = y
    fl = 2.0 * D * H * W * 27 * 32 * 32
    print(f"hint {hint}: {dt * 1e3:.3f} ms  {fl / dt / 1e12:.1f} TF/s (operator)  maxdiff {(y - ref).abs().max().item():.2e} "
          f"of {ref.abs().max().item():.2f}")
F.TILE_HINT = 0

# weight gradient of the same layer: wgrad_lds_kernel<..., WINO> against the plain variant (tile hint 6)
xg = x.detach()
wq = w.clone().requires_grad_(True)
go = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
gref = None
for hint in (6, 0, 6, 0):
    F.TILE_HINT = hint
    def run():
        wq.grad = None
        F.conv3d(xg, wq, None, 1, 1).backward(go)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    if gref is None:
        gref = wq.grad.clone()
    print(f"fwd+wgrad hint {hint}: {dt * 1e3:.3f} ms  maxdiff {(wq.grad - gref).abs().max().item():.2e} of {gref.abs().max().item():.1f}")
F.TILE_HINT = 0
