"""GPU probe: conv_tap2_kernel (stride-2 "down" gather on the LDS-ring / register-weights design) against the generic gather
kernel (tile hint 8) on the hourglass layers: conv 32 -> 64 forward and the data gradient of the transposed conv 64 -> 32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F


def timed(fn, iters=10):
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


for (D, H, W) in ((192, 48, 160), (112, 48, 160)):
    x = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(64, 32, 3, 3, 3, device="cuda") * 0.03
    fl = 2.0 * (D // 2) * (H // 2) * (W // 2) * 27 * 32 * 64
    ref = None
    for hint in (8, 0, 8, 0):
        F.TILE_HINT = hint
        with torch.no_grad():
            y = F.conv3d(x, w, None, 2, 1)
            t = timed(lambda: F.conv3d(x, w, None, 2, 1))
        ref = y if ref is None else ref
        print(f"conv 32->64 s2 @ {D}x{H}x{W}  hint {hint}: {t:.3f} ms  {fl / t / 1e9:6.1f} TF/s   maxdiff {(y - ref).abs().max().item():.2e} of {ref.abs().max().item():.2f}", flush=True)
    # data gradient of the transposed conv 64 -> 32 (input on the coarse grid)
    xc = torch.randn(1, 64, D // 2, H // 2, W // 2, device="cuda").contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    wt = torch.randn(64, 32, 3, 3, 3, device="cuda") * 0.03
    go = torch.randn(1, 32, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    ref = None
    for hint in (8, 0, 8, 0):
        F.TILE_HINT = hint
        yy = F.conv_transpose3d(xc, wt, None, 2, 1, 1)
        def run():
            xc.grad = None
            yy.backward(go, retain_graph=True)
        t = timed(run)
        ref = xc.grad.clone() if ref is None else ref
        print(f"deconv 64->32 dgrad @ {D}x{H}x{W}  hint {hint}: {t:.3f} ms  {fl / t / 1e9:6.1f} TF/s   maxdiff {(xc.grad - ref).abs().max().item():.2e} of {ref.abs().max().item():.2f}", flush=True)
    F.TILE_HINT = 0
    # the "up" gather: transposed conv 64 -> 32 forward and the data gradient of the conv 32 -> 64
    wt2 = torch.randn(64, 32, 3, 3, 3, device="cuda") * 0.03
    xc2 = xc.detach()
    ref = None
    for hint in (8, 0, 8, 0):
        F.TILE_HINT = hint
        with torch.no_grad():
            y = F.conv_transpose3d(xc2, wt2, None, 2, 1, 1)
            t = timed(lambda: F.conv_transpose3d(xc2, wt2, None, 2, 1, 1))
        ref = y if ref is None else ref
        print(f"deconv 64->32 fwd @ {D}x{H}x{W}  hint {hint}: {t:.3f} ms  {fl / t / 1e9:6.1f} TF/s   maxdiff {(y - ref).abs().max().item():.2e} of {ref.abs().max().item():.2f}", flush=True)
    xf = x.detach().requires_grad_(True)
    gc = torch.randn(1, 64, D // 2, H // 2, W // 2, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    ref = None
    for hint in (8, 0, 8, 0):
        F.TILE_HINT = hint
        yy = F.conv3d(xf, w, None, 2, 1)
        def run2():
            xf.grad = None
            yy.backward(gc, retain_graph=True)
        t = timed(run2)
        ref = xf.grad.clone() if ref is None else ref
        print(f"conv 32->64 dgrad @ {D}x{H}x{W}  hint {hint}: {t:.3f} ms  {fl / t / 1e9:6.1f} TF/s   maxdiff {(xf.grad - ref).abs().max().item():.2e} of {ref.abs().max().item():.2f}", flush=True)
    F.TILE_HINT = 0
