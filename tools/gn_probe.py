"""GroupNorm / BatchNorm operator timings per shape (fwd, bwd; finalize in the statistics tail on / off).
usage: python tools/gn_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F

SHAPES = [("gn", 1, 32, 2, (192, 48, 160)), ("gn", 1, 64, 2, (96, 24, 80)), ("gn", 1, 128, 2, (48, 12, 40)),
          ("gn", 1, 128, 32, (128, 128, 16)), ("gn", 1, 256, 32, (64, 64, 8)), ("bn", 1, 640, 640, (1, 48, 160)), ("bn", 1, 32, 32, (192, 48, 160))]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for kind, B, C, G, sp in SHAPES:
    x = torch.randn((B, C) + sp, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    w = torch.rand(C, device="cuda") + 0.5
    b = torch.randn(C, device="cuda")
    go = torch.randn_like(x)
    row = []
    for tail in (False, True):
        if hasattr(F, "GN_TAIL"):
            F.GN_TAIL = tail
        elif tail:
            continue

        def fwd():
            with torch.no_grad():
                if kind == "gn":
                    return F.group_norm(x, G, w, b, 1e-5, relu=True)
                return F.batch_norm_train(x, w, b, 1e-5, relu=True)[0]

        xr = x.detach().requires_grad_(True)
        wr, br = w.detach().requires_grad_(True), b.detach().requires_grad_(True)

        def both():
            y = F.group_norm(xr, G, wr, br, 1e-5, relu=True) if kind == "gn" else F.batch_norm_train(xr, wr, br, 1e-5, relu=True)[0]
            y.backward(go)
            xr.grad = wr.grad = br.grad = None

        tf, tb = timeit(fwd), timeit(both)
        row.append(f"tail={int(tail)}: fwd {tf:7.1f} us  fwd+bwd {tb:7.1f} us")
    mb = x.numel() * 4 / 1e6
    print(f"{kind} C={C:4d} G={G:4d} {mb:7.1f} MB  " + "   ".join(row), flush=True)
