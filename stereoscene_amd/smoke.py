"""One tiny forward+backward of the hot path on cuda:0, checked against the CPU oracle
(called by __graft_entry__.smoke(); the oracle is only the checker here)."""
import torch


def run():
    from oracle import path_ref as O
    from . import model_zoo, synthetic as S
    cfg = S.CFG_T
    torch.manual_seed(0)
    model = model_zoo.build_detector(cfg).eval()      # eval: ASPP dropout off, BN running stats
    smp = S.synthetic_sample(cfg, B=1, tag="smoke")
    inputs = model_zoo.img_inputs_from_sample(smp)
    losses = model.forward_train(img_inputs=inputs, gt_occ=smp["gt_occ"].cuda())
    total = sum(v for k, v in losses.items() if k.startswith("loss"))
    total.backward()
    torch.cuda.synchronize()
    # oracle on the same inputs / weights
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    mlp_l = O.get_mlp_input(*smp["geo_l"])
    mlp_r = O.get_mlp_input(*smp["geo_r"])
    oin = [smp["x_l"], *smp["geo_l"], mlp_l, smp["x_r"], *smp["geo_r"], mlp_r, smp["calib"]]
    D = model.img_view_transformer.D
    ocfg = dict(D=D, numC_Trans=128, warp_align_corners=True, downsample=cfg["downsample"], dbound=cfg["dbound"])
    with torch.no_grad():
        want, _ = O.forward_train(sd, oin, smp["gt_depths"], smp["gt_occ"], ocfg, train=False)
    for k, v in want.items():
        got = float(losses[k])
        assert abs(got - float(v)) <= 2e-3 * max(1.0, abs(float(v))), (k, got, float(v))
    g = model.pts_bbox_head.occ_convs[0][0].weight.grad
    assert g is not None and torch.isfinite(g).all()
    print("smoke ok:", {k: round(float(v), 5) for k, v in losses.items()})
