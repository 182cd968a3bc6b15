"""GPU A/B of the own MFMA GEMMs (csrc/gemm.hip) between two builds of the library: seeded operands, HIP-event median of 30 launches,
CRC of the result bytes (a schedule change must leave every bit alone).  Shapes of the step (tools/gemm_probe.py).
    python tools/gemm_ab.py [path/to/libssbev_hip.so]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import capi
if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
from stereoscene_amd import functional as F


def crc(t):
    return zlib.crc32(t.detach().contiguous().cpu().numpy().tobytes())


def timed(fn, iters=30):
    for _ in range(15): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[iters // 2]


CASES = [("nn", 16, 1920, 640, 640, "2-D wino fwd"), ("tn", 16, 1920, 640, 640, "2-D wino wgrad"), ("nn", 36, 480, 640, 640, "2-D wino F(4,3) fwd"),
         ("tn", 36, 480, 640, 640, "2-D wino F(4,3) wgrad"), ("nn", 144, 128, 512, 512, "512 layer fwd"), ("tn", 144, 128, 512, 512, "512 layer wgrad"),
         ("tn", 1, 192, 7680, 7680, "bri energy"), ("nt", 1, 192, 7680, 7680, "bri out"), ("nn", 1, 192, 7680, 7680, "bri gVc"),
         ("nt", 1, 7680, 3200, 640, "aspp 3200->640"), ("nn", 1, 7680, 640, 3200, "aspp dgrad"), ("tn", 1, 7680, 640, 3200, "aspp wgrad"),
         ("nn", 1, 4096, 512, 8192, "fpn k4 fwd"), ("tn", 1, 4096, 512, 8192, "fpn k4 wgrad"), ("tn", 1, 262144, 128, 128, "fpn k1 wgrad"),
         ("tn", 1, 1474560, 32, 32, "1x1x1 32ch wgrad"), ("nt", 1, 262144, 128, 128, "input_proj fwd"), ("nn", 1, 333, 77, 129, "ragged")]
print("library:", capi.LIB_PATH)
for form, bt, M, K, N, what in CASES:
    torch.manual_seed(5)
    if form == "tn":
        a = torch.randn(bt, M, K, device="cuda"); b = torch.randn(bt, M, N, device="cuda"); own = lambda: F.gemm_tn(a, b)
    elif form == "nt":
        a = torch.randn(bt, M, K, device="cuda"); b = torch.randn(bt, N, K, device="cuda"); own = lambda: F.gemm_nt(a, b)
    else:
        a = torch.randn(bt, M, K, device="cuda"); b = torch.randn(bt, K, N, device="cuda"); own = lambda: F.gemm_nn(a, b)
    gf = 2.0 * bt * M * K * N / 1e9
    t = min(timed(own), timed(own))
    print(f"{what:22s} {form} {bt:3d} x [{M} x {K} x {N}]  {t * 1e3:8.1f} us {gf / t:6.1f} TF/s  crc {crc(own()):08x}", flush=True)
