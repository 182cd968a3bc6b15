"""Own MFMA GEMMs (csrc/gemm.hip) vs rocBLAS on the path's shapes: HIP-event medians, TF/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F

def timed(fn, iters=30):
    for _ in range(15): fn()      # (clock ramp: the first ~10 launches after an idle gap run slower)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]

CASES = [("nn", 16, 1920, 640, 640, "2-D wino fwd"), ("tn", 16, 1920, 640, 640, "2-D wino wgrad"), ("tn", 64, 4096, 256, 256, "256 layer wgrad"), ("nn", 64, 4096, 256, 256, "256 layer fwd"), ("nn", 144, 128, 512, 512, "512 layer fwd"),
         ("tn", 144, 128, 512, 512, "512 layer wgrad"), ("tn", 1, 192, 7680, 7680, "bri energy"), ("nt", 1, 192, 7680, 7680, "bri out"),
         ("nn", 1, 192, 7680, 7680, "bri gVc"), ("nt", 1, 7680, 3200, 640, "aspp 3200->640"), ("nn", 1, 7680, 640, 3200, "aspp dgrad"),
         ("tn", 1, 7680, 640, 3200, "aspp wgrad"), ("nn", 1, 4096, 512, 8192, "fpn k4 fwd"), ("nt", 1, 4096, 8192, 512, "fpn k4 dgrad"),
         ("tn", 1, 4096, 512, 8192, "fpn k4 wgrad"), ("nn", 1, 262144, 128, 128, "fpn k1 fwd"), ("tn", 1, 262144, 128, 128, "fpn k1 wgrad"),
         ("nt", 1, 7680, 1440, 160, "dcn group"),
         ("nt", 1, 1474560, 32, 32, "1x1x1 32ch fwd"), ("nn", 1, 1474560, 32, 32, "1x1x1 32ch dgrad"), ("tn", 1, 1474560, 32, 32, "1x1x1 32ch wgrad"),
         ("nt", 1, 184320, 64, 64, "1x1x1 64ch fwd"), ("nn", 1, 184320, 64, 64, "1x1x1 64ch dgrad"), ("tn", 1, 184320, 64, 64, "1x1x1 64ch wgrad"),
         ("nt", 1, 262144, 128, 128, "input_proj fwd"), ("tn", 1, 262144, 128, 128, "input_proj wgrad"), ("nt", 1, 262144, 192, 20, "head 192->20")]
for form, bt, M, K, N, what in CASES:
    if form == "tn":      # rows = M (reduction), result K x N
        a = torch.randn(bt, M, K, device="cuda"); b = torch.randn(bt, M, N, device="cuda")
        own = lambda: F.gemm_tn(a, b); lib = lambda: torch.matmul(a.transpose(1, 2), b)
    elif form == "nt":
        a = torch.randn(bt, M, K, device="cuda"); b = torch.randn(bt, N, K, device="cuda")
        own = lambda: F.gemm_nt(a, b); lib = lambda: torch.matmul(a, b.transpose(1, 2))
    else:
        a = torch.randn(bt, M, K, device="cuda"); b = torch.randn(bt, K, N, device="cuda")
        own = lambda: F.gemm_nn(a, b); lib = lambda: torch.matmul(a, b)
    gf = 2.0 * bt * M * K * N / 1e9
    tl = min(timed(lib), 1e9); to = timed(own); tl = min(tl, timed(lib)); to = min(to, timed(own))
    print(f"{what:18s} {form} {bt:3d} x [{M} x {K} x {N}]  own {to:7.3f} ms {gf / to:6.1f} TF/s   rocBLAS {tl:7.3f} ms {gf / tl:6.1f} TF/s", flush=True)
