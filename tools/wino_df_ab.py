"""GPU A/B of the depth-fused Winograd kernels (wino_df_kernel forward / data gradient, wino_dfw_kernel weight gradient) between two
builds of the library: seeded inputs, HIP-event medians, CRC of the result bytes.  Layer shapes of the step, in the model's grid
orientation (short axis last).
    python tools/wino_df_ab.py [path/to/libssbev_hip.so]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import capi
if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
from stereoscene_amd import functional as F


def crc(t):
    return zlib.crc32(t.detach().contiguous().cpu().numpy().tobytes())


def timed(fn, iters=12):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[iters // 2] * 1e3


print("library:", capi.LIB_PATH)
for Cin, Cout, D, H, W in ((128, 128, 128, 128, 16), (384, 192, 128, 128, 16), (256, 256, 64, 64, 8), (64, 64, 96, 24, 80), (128, 128, 48, 12, 40)):
    torch.manual_seed(11)
    x = torch.randn(1, Cin, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.02).requires_grad_(True)
    with torch.no_grad():
        y0 = F.conv3d(x, w, None, 1, 1)
        t_f = timed(lambda: F.conv3d(x, w, None, 1, 1))
    y = F.conv3d(x, w, None, 1, 1)
    go = torch.randn_like(y)
    t_d = timed(lambda: torch.autograd.grad(y, x, go, retain_graph=True))
    t_w = timed(lambda: torch.autograd.grad(y, w, go, retain_graph=True))
    gx, gw = torch.autograd.grad(y, (x, w), go)
    print(f"{Cin:3d}->{Cout:3d} @ {D}x{H}x{W}: fwd {t_f:7.1f} us  dgrad {t_d:7.1f}  wgrad {t_w:7.1f} | crc {crc(y0):08x} {crc(gx):08x} {crc(gw):08x}", flush=True)
