"""Isolated timing of the row-softmax kernels on the BRI attention matrix shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereoscene_amd import functional as F, capi
T = 7680
x = torch.randn(1, T, T, device="cuda")
g = torch.randn(1, T, T, device="cuda")
spare = torch.randn(1, T, T, device="cuda")      # pushes x / g out of the 256 MB MALL between launches
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(n):
        spare.add_(1.0)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
y = torch.softmax(x, -1)
print(f"fwd in place      {t(lambda: F.softmax_rows_(x)):8.1f} us")
print(f"bwd in place      {t(lambda: F.softmax_rows_bwd_(y, g)):8.1f} us")
lib = capi.load()
out = torch.empty_like(g)
print(f"bwd out of place  {t(lambda: lib.ssbev_softmax_rows_bwd(capi.ptr(y), capi.ptr(g), capi.ptr(out), T, T, capi.stream())):8.1f} us")
print(f"aten bwd          {t(lambda: torch._softmax_backward_data(g, y, -1, torch.float32)):8.1f} us")
print(f"aten copy         {t(lambda: out.copy_(g)):8.1f} us")
