"""Gradient-noise probe: L2-relative error of the parameter gradients of a full train step against the CPU oracle's
autograd, for the Winograd tile choices (F(4,3) everywhere / 3-D only / F(2,3))."""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import path_ref as O
from stereoscene_amd import model_zoo, synthetic as S, functional as F
def run():
    cfg = S.CFG_T
    model = model_zoo.build_detector(cfg); model.train(True)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
    smp = S.synthetic_sample(cfg, B=2, tag="step")
    inputs = model_zoo.img_inputs_from_sample(smp)
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    losses = model.forward_train(img_inputs=inputs, gt_occ=smp["gt_occ"].cuda())
    sum(v for k, v in losses.items() if k.startswith("loss")).backward()
    return model, sd0, smp
model, sd0, smp = run()
trainable = {n for n, p in model.named_parameters() if p.requires_grad}
sd = {k: (v.clone().requires_grad_(True) if k in trainable else v.clone()) for k, v in sd0.items()}
mlp_l, mlp_r = O.get_mlp_input(*smp["geo_l"]), O.get_mlp_input(*smp["geo_r"])
oin = [smp["x_l"], *smp["geo_l"], mlp_l, smp["x_r"], *smp["geo_r"], mlp_r, smp["calib"]]
ocfg = dict(D=model.img_view_transformer.D, numC_Trans=128, warp_align_corners=True, downsample=S.CFG_T["downsample"], dbound=S.CFG_T["dbound"])
want, aux = O.forward_train(sd, oin, smp["gt_depths"], smp["gt_occ"], ocfg, train=True, stats_out={})
sum(want.values()).backward()
import os
for f43, f2d, f444 in ((True, False, False), (True, False, True)):
    F.WINO_F43, F.WINO_F43_2D, F.WINO_F444 = f43, f2d, f444
    model, _, _ = run()
    rows = []
    for name, p in model.named_parameters():
        if name not in trainable or p.grad is None or sd[name].grad is None: continue
        ref = sd[name].grad
        if ref.abs().max().item() < 1e-8: continue
        rows.append((((p.grad.cpu() - ref).norm() / ref.norm()).item(), name))
    rows.sort(reverse=True)
    print(("F43" if f43 else "F23") + ("+2d" if f2d else "") + (f"+F444(cin>={F.WINO_F444_MIN_CIN})" if f444 else ""), [(round(a, 5), n.split("img_view_transformer.")[-1]) for a, n in rows[:5]])
