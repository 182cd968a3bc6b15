"""Per-layer table (ms/step, TF/s) of every conv launch of one fwd+bwd step, from the HIP-event spans of
stereoscene_amd.functional.KernelTimer.  Usage: python tools/layer_table.py [config] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F, model_zoo, synthetic as S

cfg = S.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "kitti_d192"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
model = model_zoo.build_detector(cfg).train()
smp = S.synthetic_sample(cfg, B=1, tag="bench0")
inputs = model_zoo.img_inputs_from_sample(smp)
gt = smp["gt_occ"].cuda()


def step():
    model.zero_grad(set_to_none=True)
    losses = model.forward_train(img_inputs=inputs, gt_occ=gt)
    sum(v for k, v in losses.items() if k.startswith("loss")).backward()


step()
F.KERNEL_TIMER = t = F.KernelTimer()
for _ in range(steps):
    step()
tab = t.by_tag()
F.KERNEL_TIMER = None
rows = sorted(tab.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for v in tab.values()) / steps
print(f"total timed-span time {tot:.1f} ms/step")
for (fam, tag), v in rows:
    ms = v["ms"] / steps
    print(f"{ms:7.2f} ms  {v['launches'] / steps:4.0f}x  {v['flops'] / v['ms'] / 1e9:6.1f} TF/s  "
          f"{v['flops'] / steps / 1e9:8.1f} GF  {v['bytes'] / v['ms'] / 1e9:6.2f} TB/s  {fam:12s} {tag}")
