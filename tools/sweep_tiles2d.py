"""Tile sweep for the 2-D (DepthNet / stereo feature) convolutions of the hot path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F
L2D = [("depthnet 640->640 k3", 1, 640, 640, 3, 1, 1), ("aspp 640->640 k3 d12", 1, 640, 640, 3, 12, 12),
       ("aspp 3200->640 k1", 1, 3200, 640, 1, 0, 1), ("stereo 640->128 k3 B2", 2, 640, 128, 3, 1, 1),
       ("ctx 640->128 k1", 1, 640, 128, 1, 0, 1), ("depth 640->192 k1", 1, 640, 192, 1, 0, 1)]
def timeit(fn, n=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for name, B, ci, co, k, p, dil in L2D:
    if len(sys.argv) > 1 and sys.argv[1] not in name: continue
    x = torch.randn(B, ci, 48, 160, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(co, ci, k, k, device="cuda") * 0.02
    f = lambda: F.conv2d(x, w, None, 1, p, dil)
    y = f(); go = torch.randn_like(y)
    flops = 2.0 * B * 48 * 160 * co * ci * k * k
    row = {}
    for mt, nt in ((1, 1), (2, 4), (2, 2), (1, 2), (1, 4), (1, 5), (3, 1)):
        for qu in (1, 2, 4):
            F.TILE_HINT = mt * 100 + nt * 10 + qu
            tf = timeit(f)
            def bw():
                x.grad = None; f().backward(go)
            tb = timeit(bw, 2) - tf
            row[F.TILE_HINT] = (flops / tf / 1e12, flops / max(tb, 1e-9) / 1e12)
    F.TILE_HINT = 0
    auto = flops / timeit(f) / 1e12
    bf = max(row.items(), key=lambda kv: kv[1][0]); bb = max(row.items(), key=lambda kv: kv[1][1])
    print(f"{name:24s} auto {auto:5.1f} | fwd best {bf[0]} {bf[1][0]:6.1f} TF | dgrad best {bb[0]} {bb[1][1]:6.1f} TF | " +
          " ".join(f"{h}:{a:.0f}/{b:.0f}" for h, (a, b) in sorted(row.items())), flush=True)
