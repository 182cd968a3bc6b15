"""Image branch at the KITTI size: EfficientNet-B7 + SECONDFPN on two 384 x 1280 views, forward and
forward+backward time, with the per-family HIP-event breakdown.  `python tools/image_branch_bench.py [--no-cp]`"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereoscene_amd import functional as F, model_zoo, synthetic as S  # noqa: E402
from stereoscene_amd import plugin  # noqa: E402,F401
from stereoscene_amd.registry import BACKBONES, NECKS  # noqa: E402

cfg = model_zoo.image_branch_cfg()
cfg["img_backbone"]["with_cp"] = "--no-cp" not in sys.argv
bb = BACKBONES.build(cfg["img_backbone"])
nk = NECKS.build(cfg["img_neck"])
S.fill_state_dict_(bb, "img_backbone.")
S.fill_state_dict_(nk, "img_neck.")
bb, nk = bb.cuda().train(), nk.cuda().train()
x = torch.randn(2, 3, 384, 1280, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)


def fwd():
    return nk(bb(x))[0]


def step():
    for p in list(bb.parameters()) + list(nk.parameters()):
        p.grad = None
    y = fwd()
    y.square().mean().backward()
    return y


def timeit(f, n=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


with torch.no_grad():
    t_f = timeit(fwd)
t_s = timeit(step)
y = fwd()
print(f"output {tuple(y.shape)}; forward {t_f:.2f} ms, forward+backward {t_s:.2f} ms (with_cp={cfg['img_backbone']['with_cp']}), "
      f"params {sum(p.numel() for p in bb.parameters()) / 1e6:.1f} M + {sum(p.numel() for p in nk.parameters()) / 1e6:.1f} M")
timer = F.KernelTimer()
F.KERNEL_TIMER = timer
step()
torch.cuda.synchronize()
F.KERNEL_TIMER = None
for fam, v in sorted(timer.summary().items(), key=lambda kv: -kv[1]["ms"]):
    tf = v["flops"] / max(v["ms"], 1e-9) / 1e9
    print(f"  {fam:22s} {v['launches']:5d} launches {v['ms']:8.2f} ms  {tf:8.1f} TF/s  {v['bytes'] / max(v['ms'], 1e-9) / 1e9:8.2f} TB/s")
print("top layers:")
for (fam, tag), v in sorted(timer.by_tag().items(), key=lambda kv: -kv[1]["ms"])[:28]:
    print(f"  {v['ms']:7.3f} ms {v['launches']:3d}x {v['flops'] / max(v['ms'], 1e-9) / 1e9:7.1f} TF/s {v['bytes'] / max(v['ms'], 1e-9) / 1e9:6.2f} TB/s  {fam:12s} {tag}")
