import os, sys
sys.path.insert(0, "/root/repo")
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
for what, M, K, N in [("conv 32->64 s2 fwd", 184320, 864, 64), ("conv 64->128 s2 fwd", 23040, 1728, 128), ("conv 128->256 s2", 32768, 3456, 256),
                      ("deconv 128->64 class8", 23040, 1024, 64), ("deconv 128->64 class1", 23040, 128, 64), ("deconv 64->32 class4", 184320, 256, 32),
                      ("dgrad 32->64: class avg", 184320, 216, 32)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda")
    t = timed(lambda: F.gemm_nt(a, w))
    print(f"{what:26s} [{M} x {K} x {N}]  {t:7.3f} ms  {2.0*M*K*N/t/1e9:6.1f} TF/s")
