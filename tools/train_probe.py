"""25 optimisation steps (fused AdamW + clip) of the full-size hot path on one synthetic sample: the loss must fall and stay finite."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from stereoscene_amd import model_zoo, synthetic as S
from stereoscene_amd.train import FlatAdamW, train_step
cfg = S.CONFIGS["kitti_d192"]
model = model_zoo.build_detector(cfg).train()
smp = S.synthetic_sample(cfg, B=1, tag="bench0")
inputs = model_zoo.img_inputs_from_sample(smp)
gt = smp["gt_occ"].cuda()
opt = FlatAdamW(model, lr=2e-4, weight_decay=0.01, max_grad_norm=35.0)
for it in range(25):
    losses = train_step(model, opt, inputs, gt)
    if it % 4 == 0 or it == 24:
        tot = sum(float(v) for k, v in losses.items() if k.startswith("loss"))
        print(it, round(tot, 4), {k: round(float(v), 4) for k, v in losses.items()}, flush=True)
        assert tot == tot
