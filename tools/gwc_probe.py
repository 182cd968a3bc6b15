"""Cost-volume kernels in isolation at the KITTI size (B=1, 64 channels, 32 groups, D=192, 48x160): forward / backward time
and achieved HBM bandwidth against the 8 TB/s peak."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F
L = torch.randn(1, 64, 48, 160, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
R = torch.randn(1, 64, 48, 160, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
calib = torch.tensor([398.0], device="cuda")
def fwd():
    return F.gwc_warp(L, R, calib, 192, 32, True)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
with torch.no_grad():
    tf = t(fwd)
v = fwd(); g = torch.randn_like(v)
def bwd():
    L.grad = R.grad = None
    v.backward(g, retain_graph=True)
tb = t(bwd)
mb = v.numel() * 4 / 1e6
print(f"gwc_warp fwd {tf:.1f} us = {mb / tf:.2f} TB/s written ({mb:.0f} MB, {mb / tf / 8 * 100:.0f} % of the 8 TB/s HBM peak); "
      f"bwd (both views) {tb:.1f} us = {2 * mb / tb:.2f} TB/s read")
