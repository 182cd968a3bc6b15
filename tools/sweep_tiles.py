"""Sweep the gather kernel's register tilings (ssbev_conv_dims.tile_hint) per hot-path layer: fwd and dgrad."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F
from probe_conv import LAYERS

def timeit(fn, n=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

res = {}
for name, ci, co, (D, H, W), k, s, p, tr, op in LAYERS:
    if len(sys.argv) > 1 and sys.argv[1] not in name: continue
    x = torch.randn(1, ci, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    w = (torch.randn((ci, co, k, k, k) if tr else (co, ci, k, k, k), device="cuda") * 0.05)
    f = (lambda: F.conv_transpose3d(x, w, None, s, p, op)) if tr else (lambda: F.conv3d(x, w, None, s, p))
    y = f(); go = torch.randn_like(y)
    flops = 2.0 * (y.numel() // co) * co * ci * k ** 3 if not tr else 2.0 * (x.numel() // ci) * ci * co * k ** 3
    row = {}
    wide = len(sys.argv) > 1 and not (len(sys.argv) > 2 and sys.argv[2] == "narrow")
    for mt, nt in (((2, 4), (2, 2), (2, 3), (2, 6), (1, 6), (1, 3), (1, 2)) if wide else ((4, 1), (2, 1), (1, 1), (2, 4), (2, 2), (1, 2))):
        for qu in (1, 2, 4):
            F.TILE_HINT = mt * 100 + nt * 10 + qu
            try:
                tf = timeit(f)
                def bw():
                    x.grad = None
                    f().backward(go)
                tb = timeit(bw, 2) - tf      # dgrad only (w has no grad)
            except Exception as e:
                continue
            row[F.TILE_HINT] = (flops / tf / 1e12, flops / max(tb, 1e-9) / 1e12)
    F.TILE_HINT = 0
    auto = flops / timeit(f) / 1e12
    bf = max(row.items(), key=lambda kv: kv[1][0]); bb = max(row.items(), key=lambda kv: kv[1][1])
    print(f"{name:34s} auto {auto:5.1f} | fwd best {bf[0]} {bf[1][0]:6.1f} TF | dgrad best {bb[0]} {bb[1][1]:6.1f} TF | " +
          " ".join(f"{h}:{a:.0f}/{b:.0f}" for h, (a, b) in sorted(row.items())), flush=True)
