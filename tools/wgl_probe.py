"""GPU probe: weight gradient of the 32 -> 32 cost-volume layer (wgrad_lds_kernel<1,1,1,4,1,WINO>); SSBEV_WGL_WSEG / SSBEV_WGL_DEBUG
select / print the plan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F
D, H, W = 192, 48, 160
x = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
w = (torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.03).requires_grad_(True)
go = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
def fwd():
    with torch.no_grad():
        return F.conv3d(x, w, None, 1, 1)
def both():
    w.grad = None
    F.conv3d(x, w, None, 1, 1).backward(go)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
tf = min(t(fwd) for _ in range(3)); tb = min(t(both) for _ in range(3))
print(os.environ.get("SSBEV_WGL_WSEG"), f"fwd {tf*1e3:.3f} ms  wgrad {1e3*(tb-tf):.3f} ms")
