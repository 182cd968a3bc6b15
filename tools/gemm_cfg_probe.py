"""gemm_nn_kernel tile configurations (csrc/gemm.hip kNnCfgs, forced with SSBEV_GEMM_CFG) on the path's NN / NT shapes: HIP-event
medians, TF/s, max error vs torch.matmul; rocBLAS beside them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F


def timed(fn, iters=30):
    for _ in range(15):          # (clock ramp: the first ~10 launches after an idle gap run slower)
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


CASES = [("nn", 16, 1920, 640, 640, "2-D wino fwd"), ("nn", 16, 2016, 640, 640, "2-D wino polyphase"), ("nn", 144, 128, 512, 512, "512 layer fwd"),
         ("nn", 64, 4096, 256, 256, "256 layer"), ("nn", 1, 7680, 640, 3200, "aspp dgrad"), ("nt", 1, 7680, 3200, 640, "aspp 3200->640"),
         ("nt", 1, 192, 7680, 7680, "bri out"), ("nn", 1, 192, 7680, 7680, "bri gVc"), ("nn", 1, 4096, 512, 8192, "fpn k4 fwd")]
CFGS = [None] + [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5,6".split(","))]
for form, bt, M, K, N, what in CASES:
    a = torch.randn(bt, M, K, device="cuda")
    b = torch.randn(bt, N, K, device="cuda") if form == "nt" else torch.randn(bt, K, N, device="cuda")
    lib = (lambda: torch.matmul(a, b.transpose(1, 2))) if form == "nt" else (lambda: torch.matmul(a, b))
    own = (lambda: F.gemm_nt(a, b)) if form == "nt" else (lambda: F.gemm_nn(a, b))
    want = lib()
    gf = 2.0 * bt * M * K * N / 1e9
    tl = timed(lib)
    row = [f"rocBLAS {gf / tl:6.1f}"]
    for c in CFGS:
        if c is None:
            os.environ.pop("SSBEV_GEMM_CFG", None)
        else:
            os.environ["SSBEV_GEMM_CFG"] = str(c)
        try:
            got = own()
            err = (got - want).abs().max().item() / want.abs().max().item()
            t = timed(own)
            row.append(f"{'auto' if c is None else c}: {gf / t:6.1f}" + ("" if err < 1e-5 else f" ERR {err:.1e}"))
        except Exception as e:
            row.append(f"{c}: fail {type(e).__name__}")
    print(f"{what:18s} {form} {bt:3d} x [{M} x {K} x {N}]  " + "  ".join(row), flush=True)
