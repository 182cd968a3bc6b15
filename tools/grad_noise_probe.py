"""Diagnostic: are GPU-vs-oracle gradient differences fp32 noise?  Compare both against an fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import path_ref as O
from stereoscene_amd import model_zoo, synthetic as S

cfg = S.CFG_T
model = model_zoo.build_detector(cfg).eval()
smp = S.synthetic_sample(cfg, B=2, tag="step")
inputs = model_zoo.img_inputs_from_sample(smp)
sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
losses = model.forward_train(img_inputs=inputs, gt_occ=smp["gt_occ"].cuda())
sum(v for k, v in losses.items() if k.startswith("loss")).backward()
trainable = {n for n, p in model.named_parameters() if p.requires_grad}


def oracle(dtype):
    cast = lambda t: t.to(dtype) if t.is_floating_point() else t
    sd = {k: (cast(v).clone().requires_grad_(True) if k in trainable else cast(v).clone()) for k, v in sd0.items()}
    geo_l = [cast(t) for t in smp["geo_l"]]; geo_r = [cast(t) for t in smp["geo_r"]]
    mlp_l, mlp_r = O.get_mlp_input(*geo_l), O.get_mlp_input(*geo_r)
    oin = [cast(smp["x_l"]), *geo_l, mlp_l, cast(smp["x_r"]), *geo_r, mlp_r, cast(smp["calib"])]
    ocfg = dict(D=model.img_view_transformer.D, numC_Trans=128, warp_align_corners=True,
                downsample=cfg["downsample"], dbound=cfg["dbound"])
    want, aux = O.forward_train(sd, oin, cast(smp["gt_depths"]), smp["gt_occ"], ocfg, train=False)
    sum(want.values()).backward()
    return sd, want

sd32, w32 = oracle(torch.float32)
rows = []
for n, p in model.named_parameters():
    if n in trainable and p.grad is not None and sd32[n].grad is not None:
        ref = sd32[n].grad
        sc = ref.abs().max().item() + 1e-12
        rows.append(((p.grad.cpu() - ref).abs().max().item() / sc, sc, n))
rows.sort(reverse=True)
for r in rows[:12]:
    print("gpu-vs-cpu32 rel %.3e scale %.3e %s" % r)
print({k: (float(losses[k]), float(v)) for k, v in w32.items()})
