import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from stereoscene_amd import functional as F
from probe_conv import LAYERS
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for name, ci, co, (D, H, W), k, s, p, tr, op in LAYERS:
    if "s2" not in name and "deconv 64" not in name and "deconv 128" not in name: continue
    x = torch.randn(1, ci, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    w = (torch.randn((ci, co, k, k, k) if tr else (co, ci, k, k, k), device="cuda") * 0.05)
    f = (lambda: F.conv_transpose3d(x, w, None, s, p, op)) if tr else (lambda: F.conv3d(x, w, None, s, p))
    y = f(); go = torch.randn_like(y)
    flops = 2.0 * (y.numel() // co) * co * ci * k ** 3 if not tr else 2.0 * (x.numel() // ci) * ci * co * k ** 3
    out = []
    for hint in (0, 910, 920):
        F.TILE_HINT = hint
        try:
            tf = timeit(f)
            def bw():
                x.grad = None; f().backward(go)
            tb = timeit(bw, 3) - tf
            out.append(f"{hint}: fwd {flops / tf / 1e12:5.1f} dgrad {flops / max(tb, 1e-9) / 1e12:5.1f}")
        except Exception as e:
            out.append(f"{hint}: {type(e).__name__}")
    F.TILE_HINT = 0
    print(f"{name:30s} " + " | ".join(out), flush=True)
