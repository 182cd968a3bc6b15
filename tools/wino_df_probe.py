"""Wide 3x3x3 layers in isolation: depth-fused Winograd (csrc/winograd_fused.hip) vs the F(2x4x4) transforms + rocBLAS pipeline,
forward / data gradient / weight gradient, HIP-event medians.  usage: python tools/wino_df_probe.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stereoscene_amd import functional as F  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
# (cin, cout, D, H, W) as the MODEL runs them: the occupancy grid is [X, Y, Z] = (D, H, W) with the SHORT axis last (rounds 2-5 probed
# the encoder layers as 16 x 128 x 128 / 8 x 64 x 64 -- the transposed geometry, Thw = 1024 / 256 hw-tiles per plane instead of 128 / 32)
LAYERS = [(128, 128, 128, 128, 16), (384, 192, 128, 128, 16), (256, 256, 64, 64, 8), (64, 64, 96, 24, 80), (128, 128, 48, 12, 40)]


def timed(fn):
    for _ in range(2):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ITERS)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


for cin, cout, D, H, W in LAYERS:
    x = torch.randn(1, cin, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * (1.0 / (27 * cin)) ** 0.5
    gf = 2.0 * D * H * W * cin * cout * 27 / 1e9
    row = [f"{cin:3d}->{cout:3d} @ {D}x{H}x{W} ({gf:6.1f} GF)"]
    for df in (True, False):
        F.WINO_DF = df
        xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        with torch.no_grad():
            tf = timed(lambda: F.conv3d(xg, wg, None, 1, 1))
        y = F.conv3d(xg, wg, None, 1, 1)
        go = torch.randn_like(y)
        xd = x.clone().requires_grad_(True)
        yd = F.conv3d(xd, w, None, 1, 1)                      # data gradient only
        td = timed(lambda: torch.autograd.grad(yd, xd, go, retain_graph=True))
        yw = F.conv3d(x, wg, None, 1, 1)                      # weight gradient only
        tw = timed(lambda: torch.autograd.grad(yw, wg, go, retain_graph=True))
        row.append(f"{'DF ' if df else 'lib'} fwd {tf:6.3f} ms ({gf / tf:5.0f} TF/s eff)  dgrad {td:6.3f}  wgrad {tw:6.3f}")
        del y, yd, yw
    print("   |   ".join(row), flush=True)
    if os.environ.get("PROBE_STAGES"):
        F.WINO_DF = True
        xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        t = F.KernelTimer()
        F.KERNEL_TIMER = t
        for _ in range(5):
            y = F.conv3d(xg, wg, None, 1, 1)
            y.backward(torch.ones_like(y))
        F.KERNEL_TIMER = None
        for (fam, tag), d in sorted(t.by_tag().items(), key=lambda kv: kv[0][1]):
            ms = d["ms"] / d["launches"]
            tf = d["executed"] / d["launches"] / ms / 1e9 if d["executed"] else 0.0
            print(f"      {fam:22s} {tag:45s} {ms:7.3f} ms" + (f"  {tf:6.1f} TF/s executed" if tf else f"  {d['bytes'] / d['launches'] / ms / 1e9:6.2f} TB/s"))
