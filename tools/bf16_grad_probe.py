"""bf16 storage mode vs fp32 mode: per-module relative L2 distance and cosine of the parameter gradients of one train step
(small config), and the same for an fp32 step whose INPUT features were rounded to bf16 (how much of the distance is the
network's own sensitivity to 2^-9 perturbations).  usage: python tools/bf16_grad_probe.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F, model_zoo, synthetic as S

cfg = S.CONFIGS[sys.argv[1]] if len(sys.argv) > 1 else S.CFG_S
model = model_zoo.build_detector(cfg).train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
smp = S.synthetic_sample(cfg, B=1, tag="bf16probe")
inputs = model_zoo.img_inputs_from_sample(smp)
gt = smp["gt_occ"].cuda()
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}


def step(mode, perturb=False):
    F.set_precision(mode)
    try:
        model.load_state_dict(sd0)
        model.zero_grad(set_to_none=True)
        inp = inputs
        if perturb:       # the stereo features rounded to bf16: an fp32 run with a 2^-9 input perturbation
            inp = [[t.to(torch.bfloat16).float() if (torch.is_tensor(t) and t.dim() >= 4 and t.is_floating_point()) else t for t in v]
                   if isinstance(v, (list, tuple)) else v for v in inputs]
        losses = model.forward_train(img_inputs=inp, gt_occ=gt)
        sum(v for k, v in losses.items() if k.startswith("loss")).backward()
        return ({k: float(v.detach()) for k, v in losses.items() if k.startswith("loss")},
                {n: p.grad.detach().double().clone() for n, p in model.named_parameters() if p.grad is not None})
    finally:
        F.set_precision("fp32")


l32, g32 = step("fp32")
l16, g16 = step("bf16")
lp, gp = step("fp32", perturb=True)
print("losses fp32", l32)
print("losses bf16", l16)
print("losses fp32, bf16-rounded inputs", lp)
groups = {}
for n in g32:
    parts = n.split(".")
    key = ".".join(parts[:3]) if parts[0] == "img_view_transformer" else ".".join(parts[:2])
    groups.setdefault(key, []).append(n)
print(f"{'module':58s} {'n':>3s}  {'bf16: relL2':>11s} {'cos':>7s} {'norm ratio':>10s}   {'perturbed fp32: relL2':>20s}")
for key, names in groups.items():
    a = torch.cat([g16[n].flatten() for n in names]); b = torch.cat([g32[n].flatten() for n in names]); c = torch.cat([gp[n].flatten() for n in names])
    if b.norm() < 1e-30:
        continue
    rel = ((a - b).norm() / b.norm()).item(); cos = (a @ b / (a.norm() * b.norm() + 1e-300)).item()
    relp = ((c - b).norm() / b.norm()).item()
    print(f"{key:58s} {len(names):3d}  {rel:11.3e} {cos:7.4f} {(a.norm() / b.norm()).item():10.4f}   {relp:20.3e}")
