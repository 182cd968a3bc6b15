"""GPU A/B of conv_tap2_kernel / conv_tap2up_kernel between two builds of the library: seeded inputs, median of 20 launches, CRC of
the result bytes (a schedule change must leave every bit alone).  Hourglass layers of the cost-volume stack at kitti_d192.
    python tools/tap2_ab.py [path/to/libssbev_hip.so]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import capi
if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
from stereoscene_amd import functional as F


def crc(t):
    return zlib.crc32(t.detach().contiguous().cpu().numpy().tobytes())


def timed(fn, iters=20):
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[iters // 2] * 1e3


def cl(t):
    return t.contiguous(memory_format=torch.channels_last_3d)


def case(Cf, Cc, D, H, W, bias):
    torch.manual_seed(7)
    x = cl(torch.randn(1, Cf, D, H, W, device="cuda"))
    w = torch.randn(Cc, Cf, 3, 3, 3, device="cuda") * 0.03
    bd = torch.randn(Cc, device="cuda") if bias else None
    xc = cl(torch.randn(1, Cc, D // 2, H // 2, W // 2, device="cuda"))
    wt = torch.randn(Cc, Cf, 3, 3, 3, device="cuda") * 0.03
    bu = torch.randn(Cf, device="cuda") if bias else None
    with torch.no_grad():
        y = F.conv3d(x, w, bd, 2, 1)
        t_down = timed(lambda: F.conv3d(x, w, bd, 2, 1))
        yu = F.conv_transpose3d(xc, wt, bu, 2, 1, 1)
        t_up = timed(lambda: F.conv_transpose3d(xc, wt, bu, 2, 1, 1))
    xq = x.clone().requires_grad_(True)
    yq = F.conv3d(xq, w, bd, 2, 1)
    gq = cl(torch.randn_like(yq))
    t_dg = timed(lambda: torch.autograd.grad(yq, xq, gq, retain_graph=True))
    (gx,) = torch.autograd.grad(yq, xq, gq, retain_graph=True)
    xcq = xc.clone().requires_grad_(True)
    yuq = F.conv_transpose3d(xcq, wt, bu, 2, 1, 1)
    guq = cl(torch.randn_like(yuq))
    t_udg = timed(lambda: torch.autograd.grad(yuq, xcq, guq, retain_graph=True))
    (gxc,) = torch.autograd.grad(yuq, xcq, guq, retain_graph=True)
    print(f"{Cf}<->{Cc} @ {D}x{H}x{W}: down fwd {t_down:.1f} us, its dgrad {t_dg:.1f} | up fwd {t_up:.1f} us, its dgrad {t_udg:.1f} | "
          f"crc {crc(y):08x} {crc(gx):08x} {crc(yu):08x} {crc(gxc):08x}", flush=True)


def case_pw(Ci, Co, D, H, W, bias):
    """1x1x1 convolution (conv_pw32_kernel): forward with bias + ReLU epilogue, data gradient, and a forked input whose second data
    gradient accumulates into the first one's buffer"""
    torch.manual_seed(21)
    x = cl(torch.randn(1, Ci, D, H, W, device="cuda")).requires_grad_(True)
    w = torch.randn(Co, Ci, 1, 1, 1, device="cuda") * 0.1
    w2 = torch.randn(Co, Ci, 1, 1, 1, device="cuda") * 0.1
    b = torch.randn(Co, device="cuda") if bias else None
    with torch.no_grad():
        y = F.conv3d(x, w, b, 1, 0)
        t_f = timed(lambda: F.conv3d(x, w, b, 1, 0))
    yq = F.conv3d(x, w, b, 1, 0)
    gq = cl(torch.randn_like(yq))
    t_d = timed(lambda: torch.autograd.grad(yq, x, gq, retain_graph=True))
    (gx,) = torch.autograd.grad(yq, x, gq, retain_graph=True)
    xa, xb = F.fork(x * 1.0, 2)
    yf = torch.relu(F.conv3d(xa, w, b, 1, 0)) + F.conv3d(xb, w2, None, 1, 0)
    (gf,) = torch.autograd.grad(yf, x, gq)
    print(f"1x1x1 {Ci}->{Co} @ {D}x{H}x{W}: fwd {t_f:.1f} us, dgrad {t_d:.1f} | crc {crc(y):08x} {crc(gx):08x} {crc(gf):08x}", flush=True)


print("library:", capi.LIB_PATH)
case_pw(32, 32, 192, 48, 160, True)
case_pw(32, 32, 192, 48, 160, False)
case_pw(32, 18, 20, 12, 37, True)
case_pw(30, 32, 20, 12, 37, False)
case(32, 64, 192, 48, 160, False)
case(32, 64, 192, 48, 160, True)
case(32, 64, 22, 10, 38, True)
case(24, 48, 10, 6, 34, False)
