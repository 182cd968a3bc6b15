"""ssbev_gemm16_nn (conv_igemm16_kernel, batched plain products) vs torch.bmm (rocBLAS) in bf16 on the Winograd frequency shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F


def timed(fn, iters=30):
    for _ in range(15):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


CASES = [(16, 1920, 640, 640), (16, 3840, 640, 128), (64, 4096, 256, 256), (64, 512, 512, 512), (64, 2880, 128, 128), (16, 2016, 640, 640)]
tiles = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"]
for Bt, M, K, N in CASES:
    a = torch.randn(Bt, M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(Bt, K, N, device="cuda")
    b16 = b.to(torch.bfloat16)
    gf = 2.0 * Bt * M * K * N / 1e9
    row = [f"rocBLAS {gf / timed(lambda: torch.bmm(a, b16)):6.0f}", f"(+cast {gf / timed(lambda: torch.bmm(a, b.to(torch.bfloat16))):6.0f})"]
    for t in tiles:
        if t == "0":
            os.environ.pop("SSBEV_IGEMM_TILE", None)
        else:
            os.environ["SSBEV_IGEMM_TILE"] = t
        row.append(f"{t}: {gf / timed(lambda: F.gemm16_nn(a, b)):6.0f}")
    z = torch.randn(Bt, M, N, device="cuda").to(torch.bfloat16)
    row.append(f"| tn rocBLAS {gf / timed(lambda: torch.bmm(a.transpose(1, 2), z, out_dtype=torch.float32)):6.0f}  own {gf / timed(lambda: F.gemm16_tn(a, z)):6.0f}")
    print(f"{Bt:3d} x [{M} x {K} x {N}] TF/s  " + "  ".join(row), flush=True)
