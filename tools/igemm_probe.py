"""conv_igemm_kernel vs conv_gather_kernel on the strided / transposed layers of the path (fwd + dgrad), HIP-event medians."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F


def timed(fn, iters=20):
    for _ in range(10):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


CASES = [("C", 64, 128, (96, 24, 80), 2, 1), ("T", 128, 64, (48, 12, 40), 2, 1), ("C", 128, 256, (128, 128, 16), 2, 1),
         ("C", 256, 512, (64, 64, 8), 2, 1), ("C", 32, 64, (192, 48, 160), 2, 1)]
tiles = [t for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])]
for kind, K, N, sp, st, dil in CASES:
    x = torch.randn((1, K) + sp, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    if kind == "C":
        w = torch.randn(N, K, 3, 3, 3, device="cuda") * 0.05
        fwd = lambda: F.conv3d(x, w, None, st, dil, dil)
    else:
        w = torch.randn(K, N, 3, 3, 3, device="cuda") * 0.05
        fwd = lambda: F.conv_transpose3d(x, w, None, st, 1, 1)
    with torch.no_grad():
        y = fwd()
    gy = torch.randn_like(y)
    xr = x.detach().requires_grad_(True)

    def dgrad():
        yy = (F.conv3d(xr, w, None, st, dil, dil) if kind == "C" else F.conv_transpose3d(xr, w, None, st, 1, 1))
        return torch.autograd.grad(yy, xr, gy)[0]

    gf = 2.0 * y.numel() / N * K * N * 27 / 1e9 if kind == "C" else 2.0 * x.numel() / K * K * N * 27 / 1e9
    row = []
    for mode, tile in [("0", "0")] + [("1", t) for t in tiles]:
        os.environ["SSBEV_IGEMM"] = mode
        if tile != "0":
            os.environ["SSBEV_IGEMM_TILE"] = tile
        else:
            os.environ.pop("SSBEV_IGEMM_TILE", None)
        try:
            with torch.no_grad():
                tf = timed(fwd)
            tb = timed(dgrad) - tf              # dgrad alone (the probe's backward re-runs the forward)
            row.append(f"{'gather' if mode == '0' else 'igemm ' + tile}: fwd {tf * 1e3:6.0f} us {gf / tf:5.1f} TF  dgrad {tb * 1e3:6.0f} us {gf / tb:5.1f} TF")
        except Exception as e:
            row.append(f"igemm {tile}: {type(e).__name__}")
    print(f"{kind} {K:3d}->{N:3d} {sp}  " + " | ".join(row), flush=True)
