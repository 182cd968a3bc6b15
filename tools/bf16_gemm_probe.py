"""Probe: batched Winograd-domain GEMM shapes on rocBLAS/hipBLASLt in fp32 vs bf16 (fp32 accumulate), MI355X."""
import time
import torch

def bench(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

shapes = [("128->128 fwd", 64, 32768, 128, 128), ("384->192 fwd", 64, 32768, 384, 192), ("64->64 fwd", 64, 23040, 64, 64),
          ("256->256 fwd", 64, 4096, 256, 256), ("640->640 2d", 16, 1920, 640, 640)]
for name, nb, M, K, N in shapes:
    a = torch.randn(nb, M, K, device="cuda"); b = torch.randn(nb, K, N, device="cuda")
    ah, bh = a.bfloat16(), b.bfloat16()
    gf = 2.0 * nb * M * K * N / 1e9
    t32 = bench(lambda: torch.bmm(a, b))
    t16 = bench(lambda: torch.bmm(ah, bh))
    try:
        t16o = bench(lambda: torch.bmm(ah, bh, out_dtype=torch.float32))
    except Exception as e:
        t16o = float("nan"); print("out_dtype unsupported:", type(e).__name__, str(e)[:100])
    # wgrad shape: [nb, K, M] x [nb, M, N]
    g = torch.randn(nb, M, N, device="cuda"); gh = g.bfloat16()
    w32 = bench(lambda: torch.bmm(a.transpose(1, 2), g))
    w16 = bench(lambda: torch.bmm(ah.transpose(1, 2), gh))
    try:
        w16o = bench(lambda: torch.bmm(ah.transpose(1, 2), gh, out_dtype=torch.float32))
    except Exception:
        w16o = float("nan")
    print(f"{name:14s} {gf:7.1f} GF  fwd fp32 {t32:6.3f} ms ({gf/t32:6.1f} TF/s)  bf16 {t16:6.3f} ({gf/t16:6.1f})  bf16->f32 {t16o:6.3f} | "
          f"wgrad fp32 {w32:6.3f}  bf16 {w16:6.3f}  bf16->f32 {w16o:6.3f}")
