"""Time the six BRI products as (a) torch.bmm (rocBLAS) and (b) 1x1 convolutions on the ssbev MFMA kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F
T, D = 7680, 192
dev = "cuda"
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
Q = torch.randn(1, D, T, device=dev); K = torch.randn(1, D, T, device=dev)
att = torch.softmax(torch.randn(1, T, T, device=dev), -1); Vt = torch.randn(1, T, D, device=dev)
fl = 2.0 * T * T * D
# E = Q^T K   [T,T]
Qt = Q.transpose(1, 2).contiguous()
t = timeit(lambda: torch.bmm(Qt, K)); print(f"bmm  E=Q^T K        {t*1e3:6.3f} ms {fl/t/1e12:6.1f} TF/s")
x = Qt.view(1, 48, 160, D).permute(0, 3, 1, 2)                   # channels-last [1,D,48,160]
w = K[0].t().contiguous().view(T, D, 1, 1)
for hint in (0, 244, 224, 144, 154):
    F.TILE_HINT = hint
    try:
        t = timeit(lambda: F.conv2d(x, w)); print(f"conv E hint {hint:3d}     {t*1e3:6.3f} ms {fl/t/1e12:6.1f} TF/s")
    except Exception as e:
        print("hint", hint, "failed", e)
F.TILE_HINT = 0
# O = att Vt   [T,D]
t = timeit(lambda: torch.bmm(att, Vt)); print(f"bmm  O=att V'       {t*1e3:6.3f} ms {fl/t/1e12:6.1f} TF/s")
x2 = att.view(1, 48, 160, T).permute(0, 3, 1, 2)
w2 = Vt[0].t().contiguous().view(D, T, 1, 1)
for hint in (0, 224, 244, 212, 222):
    F.TILE_HINT = hint
    try:
        t = timeit(lambda: F.conv2d(x2, w2)); print(f"conv O hint {hint:3d}     {t*1e3:6.3f} ms {fl/t/1e12:6.1f} TF/s")
    except Exception as e:
        print("hint", hint, "failed", e)
F.TILE_HINT = 0
# gK = Q gE  [D,T]: (D x T) (T x T)
gE = torch.randn(1, T, T, device=dev)
t = timeit(lambda: torch.bmm(Q, gE)); print(f"bmm  gK=Q gE        {t*1e3:6.3f} ms {fl/t/1e12:6.1f} TF/s")
