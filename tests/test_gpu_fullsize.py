"""Full-size (BASELINE configs[1]: 48x160 features, D=192, 128x128x16 LSS grid) checks through size-independent
properties, plus direct oracle comparisons where the oracle still finishes in seconds (-m gpu)."""
import pytest
import torch
import torch.nn.functional as TF

from oracle import path_ref as O
from stereoscene_amd import functional as F
from stereoscene_amd import model_zoo, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_cost_volume_full_size_vs_oracle():
    B, C, H, W, D = 1, 64, 48, 160, 192
    L = S.hash_normal("fs/L", (B, C, H, W))
    R = S.hash_normal("fs/R", (B, C, H, W))
    calib = torch.tensor([393.8])
    got = F.gwc_warp(L.to(DEV), R.to(DEV), calib.to(DEV), D, 32, True).cpu()
    want = O.warp_volume(O.gwc_volume(L, R, D, 32), calib, 1, True)
    assert got.shape == want.shape == (1, 32, 192, 48, 160)
    assert (got - want).abs().max().item() < 2e-5


def test_cost_volume_backward_full_size_vs_oracle():
    """Backward of the fused cost volume at 192 x 48 x 160 (one read of the 189 MB gradient volume, both views): against the
    oracle's autograd on a 6-row slab (rows are independent), plus the adjoint identity <vol, go> == <L, gL> on the whole
    tensor (the operator is bilinear: linear in L for fixed R)."""
    B, C, H, W, D = 1, 64, 48, 160, 192
    L = S.hash_normal("fsb/L", (B, C, H, W))
    R = S.hash_normal("fsb/R", (B, C, H, W))
    go = S.hash_normal("fsb/go", (B, 32, D, H, W))
    calib = torch.tensor([393.8])
    Lg, Rg = L.to(DEV).requires_grad_(True), R.to(DEV).requires_grad_(True)
    vol = F.gwc_warp(Lg, Rg, calib.to(DEV), D, 32, True)
    vol.backward(go.to(DEV))
    rows = slice(20, 26)
    Lc, Rc = L[:, :, rows].clone().requires_grad_(True), R[:, :, rows].clone().requires_grad_(True)
    O.warp_volume(O.gwc_volume(Lc, Rc, D, 32), calib, 1, True).backward(go[:, :, :, rows])
    s = max(1.0, Lc.grad.abs().max().item())
    assert (Lg.grad[:, :, rows].cpu() - Lc.grad).abs().max().item() < 2e-5 * s
    assert (Rg.grad[:, :, rows].cpu() - Rc.grad).abs().max().item() < 2e-5 * s
    lhs = (vol.detach().double() * go.to(DEV).double()).sum().item()
    for x, g in ((Lg, Lg.grad), (Rg, Rg.grad)):
        rhs = (x.detach().double() * g.double()).sum().item()
        assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs))


def test_lift_splat_full_size_conserves_mass_and_is_bit_reproducible():
    cfg = S.CFG_K192
    gc = S.grid_config(cfg)
    dx, bx, nx = O.gen_dx_bx(gc["xbound"], gc["ybound"], gc["zbound"])
    fr = O.create_frustum(cfg["input_size"], 8, gc["dbound"])
    geom = O.get_geometry(fr, *S.kitti_calibration(1, 1280)[:6]).to(DEV)
    depth = torch.softmax(S.hash_normal("fs/depth", (1, 192, 48, 160), 2.0), 1).to(DEV)
    feat = S.hash_normal("fs/feat", (1, 128, 48, 160)).to(DEV)
    a = F.lift_splat(depth, feat, geom, bx, dx, nx)
    b = F.lift_splat(depth, feat, geom, bx, dx, nx)
    assert a.shape == (1, 128, 128, 128, 16) and torch.equal(a, b)
    # checksum of checksums: per-channel mass over the grid == per-channel mass over the kept frustum points
    vox = F.voxel_index(geom, bx, dx, nx).view(1, 192, 48, 160)
    kept = (vox >= 0).double()
    want = torch.einsum("bdhw,bchw->c", depth.double() * kept, feat.double())
    got = a.double().sum(dim=(0, 2, 3, 4))
    assert ((got - want).abs() / (want.abs() + 1.0)).max().item() < 1e-5
    assert int((vox >= 0).sum()) > 300000        # the KITTI-like calibration really fills the grid


@pytest.mark.parametrize("shape", [(384, 192, (128, 128, 16)), (32, 32, (192, 48, 160))])
def test_conv_full_size_linearity_and_sampled_values(shape):
    ci, co, sp = shape
    x1 = S.hash_normal("fs/x1", (1, ci) + sp).to(DEV)
    x2 = S.hash_normal("fs/x2", (1, ci) + sp).to(DEV)
    w = (S.hash_uniform("fs/w", (co, ci, 3, 3, 3), -1, 1) * (3.0 / (27 * ci)) ** 0.5).to(DEV)
    y1, y2 = F.conv3d(x1, w, None, 1, 1), F.conv3d(x2, w, None, 1, 1)
    y12 = F.conv3d(0.75 * x1 - 1.25 * x2, w, None, 1, 1)
    assert (y12 - (0.75 * y1 - 1.25 * y2)).abs().max().item() < 5e-5 * max(1.0, y1.abs().max().item())
    # a border window and an interior window against ATen on the CPU
    for sl in ((slice(0, 6), slice(0, 6), slice(0, 6)),
               (slice(sp[0] // 2, sp[0] // 2 + 6), slice(sp[1] - 6, sp[1]), slice(sp[2] // 2 - 3, sp[2] // 2 + 3))):
        lo = [max(s.start - 1, 0) for s in sl]
        hi = [min(s.stop + 1, n) for s, n in zip(sl, sp)]
        patch = x1[:, :, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].cpu()
        ref = TF.conv3d(patch, w.cpu(), None, 1, 1)
        off = [s.start - l for s, l in zip(sl, lo)]
        ref = ref[:, :, off[0]:off[0] + 6, off[1]:off[1] + 6, off[2]:off[2] + 6]
        got = y1[:, :, sl[0], sl[1], sl[2]].cpu()
        # windows that touch the true border keep zero padding on that side in both computations
        assert (got - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item())


def test_forward_is_bit_deterministic_at_full_size():
    model = model_zoo.build_detector(S.CFG_K192).eval()
    smp = S.synthetic_sample(S.CFG_K192, B=1, tag="fsdet")
    inputs = model_zoo.img_inputs_from_sample(smp)
    gt = smp["gt_occ"].to(DEV)
    with torch.no_grad():
        a = model.simple_test(None, inputs, gt_occ=gt)["output_voxels"]
        b = model.simple_test(None, inputs, gt_occ=gt)["output_voxels"]
    assert a.shape == (1, 20, 256, 256, 32)
    assert torch.equal(a, b)
    assert torch.isfinite(a).all()


# -------------------------------------------------------------------------------------------------------------------
# Assembled path at the BASELINE sizes against the CPU oracle (north-star gates: fp32 logits within 1e-3, depth
# distribution within 1e-4).  The oracle's full-size forward takes ~10 s on the GPU box's host with 32 threads.
# -------------------------------------------------------------------------------------------------------------------
def _oracle_inputs(smp):
    mlp_l, mlp_r = O.get_mlp_input(*smp["geo_l"]), O.get_mlp_input(*smp["geo_r"])
    return [smp["x_l"], *smp["geo_l"], mlp_l, smp["x_r"], *smp["geo_r"], mlp_r, smp["calib"]]


def _ocfg(cfg, D, ac):
    return dict(D=D, numC_Trans=128, warp_align_corners=ac, downsample=cfg["downsample"], dbound=cfg["dbound"])


def _coarse_outputs(model, inputs):
    voxel_feats, _, depth = model.extract_feat(None, img=inputs)
    return model.pts_bbox_head(voxel_feats=voxel_feats)["output_voxels"][0], depth


@pytest.mark.parametrize("name,ac", [("kitti_d192", True), ("kitti_d192", False), ("kitti_d112", True)])
def test_full_size_forward_logits_vs_oracle(name, ac):
    """kitti_d192 / kitti_d112 forward (eval mode, fill-by-key weights, both `warp` modes at the BASELINE size): coarse
    logits [1,20,128,128,16] <= 1e-3 max-abs, depth_prob <= 1e-4, BEV volume <= 1e-3 of its scale vs the oracle."""
    import torch
    cfg = S.CONFIGS[name]
    model = model_zoo.build_detector(cfg, warp_align_corners=ac).eval()
    smp = S.synthetic_sample(cfg, B=1, tag="fs" + name)
    inputs = model_zoo.img_inputs_from_sample(smp)
    with torch.no_grad():
        logits, depth = _coarse_outputs(model, inputs)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    nt = torch.get_num_threads()
    torch.set_num_threads(min(32, nt))       # ATen CPU gets slower beyond ~32 threads on the 256-thread host
    try:
        with torch.no_grad():
            _, aux = O.forward_train(sd, _oracle_inputs(smp), smp["gt_depths"], smp["gt_occ"],
                                     _ocfg(cfg, model.img_view_transformer.D, ac), train=False)
    finally:
        torch.set_num_threads(nt)
    assert logits.shape == aux["logits"].shape == (1, 20, 128, 128, 16)
    e_depth = (depth.cpu() - aux["depth_prob"]).abs().max().item()
    e_logit = (logits.cpu() - aux["logits"]).abs().max().item()
    agree = (logits.cpu().argmax(1) == aux["logits"].argmax(1)).float().mean().item()
    print(f"{name} ac={ac}: depth_prob max-abs {e_depth:.2e}, logits max-abs {e_logit:.2e} "
          f"(scale {aux['logits'].abs().max().item():.2f}), argmax agreement {agree:.6f}")
    assert e_depth < 1e-4
    assert e_logit < 1e-3
    assert agree > 0.9999


def test_full_size_forward_batch2_vs_oracle():
    """Two samples per GPU (the per-GPU shape of BASELINE configs[3]) at kitti_d112: per-sample calibration in the cost volume,
    the batch index of the voxel CSR, batch strides of every kernel -- logits and depth distribution vs the oracle."""
    cfg = S.CFG_K112
    model = model_zoo.build_detector(cfg).eval()
    smp = S.synthetic_sample(cfg, B=2, tag="fsb2")
    inputs = model_zoo.img_inputs_from_sample(smp)
    with torch.no_grad():
        logits, depth = _coarse_outputs(model, inputs)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    nt = torch.get_num_threads()
    torch.set_num_threads(min(32, nt))
    try:
        with torch.no_grad():
            _, aux = O.forward_train(sd, _oracle_inputs(smp), smp["gt_depths"], smp["gt_occ"],
                                     _ocfg(cfg, model.img_view_transformer.D, True), train=False)
    finally:
        torch.set_num_threads(nt)
    assert logits.shape == aux["logits"].shape == (2, 20, 128, 128, 16)
    e_depth = (depth.cpu() - aux["depth_prob"]).abs().max().item()
    e_logit = (logits.cpu() - aux["logits"]).abs().max().item()
    print(f"kitti_d112 B=2: depth_prob max-abs {e_depth:.2e}, logits max-abs {e_logit:.2e} (scale {aux['logits'].abs().max().item():.2f})")
    assert e_depth < 1e-4 and e_logit < 1e-3
    assert (logits[0] - logits[1]).abs().max().item() > 1e-2          # the two samples really differ


@pytest.mark.parametrize("cfg_name", ["kitti_d112", "kitti_d192"])
@pytest.mark.parametrize("mode", ["bev_only", "stereo_only"])
def test_full_size_ablation_modes_vs_oracle(mode, cfg_name):
    """BASELINE configs[4] at the KITTI size (the ablation modes were compared with the oracle only at the tiny
    config; VERDICT r5: at kitti_d112 only -- kitti_d192 is the BASELINE frustum): eval mode.  The reference has no ablation switch, so the expected lifted volume is composed from the
    oracle's own pieces -- its monocular (bev_only) / stereo (stereo_only) depth distribution pushed through its lift + splat;
    gates as for the full path: depth distribution 1e-4, BEV volume 1e-3 of its scale."""
    import torch
    cfg = S.CONFIGS[cfg_name]
    model = model_zoo.build_detector(cfg).eval()
    vt = model.img_view_transformer
    smp = S.synthetic_sample(cfg, B=1, tag="fsabl")
    sd = {k: v.detach().cpu().clone() for k, v in vt.state_dict().items()}
    oin = _oracle_inputs(smp)
    taps = {}
    nt = torch.get_num_threads()
    torch.set_num_threads(min(32, nt))
    try:
        with torch.no_grad():
            O.view_transformer(sd, "", oin, dict(D=vt.D, numC_Trans=128, warp_align_corners=True), taps=taps)
            dist_ref = taps["lss_volume"] if mode == "bev_only" else taps["stereo_volume"]
            bev_ref = O.lift_splat(dist_ref, taps["img_feat"], taps["geom"], sd["dx"], sd["bx"], sd["nx"])
    finally:
        torch.set_num_threads(nt)
    vt.ablation = mode
    try:
        with torch.no_grad():
            bev, dp = vt([t.cuda() for t in oin])
    finally:
        vt.ablation = "full"
    e_d = (dp.cpu() - dist_ref).abs().max().item()
    e_b = (bev.cpu() - bev_ref).abs().max().item()
    scale = bev_ref.abs().max().item()
    print(f"{cfg_name} ablation {mode}: depth distribution max-abs {e_d:.2e}, BEV volume max-abs {e_b:.2e} (scale {scale:.2f})")
    assert bev.shape == bev_ref.shape and e_d < 1e-4 and e_b < 1e-3 * max(1.0, scale)


@pytest.mark.parametrize("cfg_name,B", [("kitti_d192", 1), ("kitti_d112", 2), ("kitti_d192", 2)])
def test_full_size_bf16_mode_vs_oracle(cfg_name, B):
    """BASELINE configs[3] at the KITTI size: the bf16 STORAGE mode (round 4: bf16 activations between the layers, bf16 MFMA
    kernels of csrc/conv_bf16.hip, bf16 I/O in the norms and Winograd transforms; fp32 statistics, softmaxes, BRI, scatter and
    losses) against the fp32 ORACLE -- an error budget, not parity ("report max-abs and argmax agreement"): max-abs
    below 5 % of the logit scale, argmax agreement above 97 %.  kitti_d192 x B = 2 is configs[3]'s own per-GPU shape."""
    cfg = S.CONFIGS[cfg_name]
    model = model_zoo.build_detector(cfg).eval()
    smp = S.synthetic_sample(cfg, B=B, tag="fsbf16")
    inputs = model_zoo.img_inputs_from_sample(smp)
    F.set_precision("bf16")
    try:
        with torch.no_grad():
            logits, depth = _coarse_outputs(model, inputs)
    finally:
        F.set_precision("fp32")
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    nt = torch.get_num_threads()
    torch.set_num_threads(min(32, nt))
    try:
        with torch.no_grad():
            _, aux = O.forward_train(sd, _oracle_inputs(smp), smp["gt_depths"], smp["gt_occ"],
                                     _ocfg(cfg, model.img_view_transformer.D, True), train=False)
    finally:
        torch.set_num_threads(nt)
    scale = aux["logits"].abs().max().item()
    e_logit = (logits.float().cpu() - aux["logits"]).abs().max().item()
    agree = (logits.float().cpu().argmax(1) == aux["logits"].argmax(1)).float().mean().item()
    e_depth = (depth.float().cpu() - aux["depth_prob"]).abs().max().item()
    print(f"{cfg_name} B={B} bf16 mode vs fp32 oracle: logits max-abs {e_logit:.3f} (scale {scale:.2f}), argmax agreement {agree:.4f}, "
          f"depth_prob max-abs {e_depth:.2e}")
    assert e_logit > 1e-3, "bf16 mode did not engage"
    assert e_logit < 5e-2 * scale and agree > 0.97


_ORACLE_STEP = {}
_FLOOR = {}


def _oracle_step(cfg_name, ac, sd0, trainable, smp, D, perturb=0):
    """Oracle fwd+bwd of the step test (cached per case): (losses, aux, {name: grad}).  ``perturb`` = k multiplies both feature
    maps by (1 + k * 2^-23): a one-ulp relative change of the INPUT, used to measure the oracle's own gradient sensitivity."""
    key = (cfg_name, ac, perturb)
    if key in _ORACLE_STEP:
        return _ORACLE_STEP[key]
    import torch
    cfg = S.CONFIGS[cfg_name]
    sd = {k: (v.clone().requires_grad_(True) if k in trainable else v.clone()) for k, v in sd0.items()}
    oin = _oracle_inputs(smp)
    if perturb:
        f = 1.0 + perturb * 2.0 ** -23
        oin[0], oin[8] = oin[0] * f, oin[8] * f
    nt = torch.get_num_threads()
    torch.set_num_threads(min(32, nt))
    try:
        want, aux = O.forward_train(sd, oin, smp["gt_depths"], smp["gt_occ"], _ocfg(cfg, D, ac), train=True, stats_out={})
        sum(want.values()).backward()
    finally:
        torch.set_num_threads(nt)
    out = ({k: float(v.detach()) for k, v in want.items()}, {"logits": aux["logits"].detach()},
           {k: v.grad for k, v in sd.items() if k in trainable and v.grad is not None})
    if not perturb:                       # (perturbed runs are reduced to a small table by _floor_table: 360 MB of gradients each)
        _ORACLE_STEP[key] = out
    return out


def _l2_table(grads_a, grads_b):
    rows = {}
    for name, ref in grads_b.items():
        if name in grads_a and ref.abs().max().item() >= 1e-8:
            rows[name] = ((grads_a[name] - ref).norm() / ref.norm()).item()
    return rows


def _gpu_step(cfg_name, ac):
    import torch
    cfg = S.CONFIGS[cfg_name]
    model = model_zoo.build_detector(cfg, warp_align_corners=ac).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    smp = S.synthetic_sample(cfg, B=1, tag="fsstep")
    inputs = model_zoo.img_inputs_from_sample(smp)
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    seen = {}
    hook = model.pts_bbox_head.register_forward_hook(lambda mod, args, out: seen.update(logits=out["output_voxels"][0].detach()))
    losses = model.forward_train(img_inputs=inputs, gt_occ=smp["gt_occ"].to(DEV))
    hook.remove()
    sum(v for k, v in losses.items() if k.startswith("loss")).backward()
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if n in trainable and p.grad is not None}
    return model, smp, sd0, trainable, {k: float(v.detach()) for k, v in losses.items()}, seen["logits"].cpu(), grads


@pytest.mark.parametrize("cfg_name,ac", [("kitti_d112", True), ("kitti_d112", False), ("kitti_d192", True)])
def test_full_size_step_fwd_bwd_vs_oracle(cfg_name, ac):
    """One fwd+bwd step at the reference's own config (D=112, both `warp` modes) and at the BASELINE metric's config (D=192),
    256x256x32 grid, train mode (batch-stat BN, dropout off): the TRAIN-mode coarse logits, the 4 losses and every parameter
    gradient against the oracle's autograd."""
    model, smp, sd0, trainable, losses, logits, grads = _gpu_step(cfg_name, ac)
    want, aux, ograds = _oracle_step(cfg_name, ac, sd0, trainable, smp, model.img_view_transformer.D)
    e_logit = (logits - aux["logits"]).abs().max().item()
    print(f"{cfg_name} ac={ac} train-mode logits max-abs {e_logit:.2e} (scale {aux['logits'].abs().max().item():.2f})")
    assert logits.shape == aux["logits"].shape and e_logit < 1e-3
    for k, v in want.items():
        assert abs(losses[k] - v) < 1e-4 * max(1.0, abs(v)), (k, losses[k], v)
    rows = _l2_table(grads, ograds)
    worst = max((v, k) for k, v in rows.items())
    print(f"{cfg_name} ac={ac} step: {len(rows)} parameter gradients, worst L2-relative error {worst[0]:.3e} ({worst[1]})")
    # The gate is set against the ORACLE'S OWN response to perturbations of its inputs at rounding level, tensor by tensor: the
    # camera-aware MLP / SE parameters of the stereo branch and the dres* layers behind them sit at a floor of 0.5-2e-2 at this size
    # (ReLU / max sign flips), everything else at 1e-5..1e-3.  VERDICT / ADVICE r5: ONE perturbation is one noisy sample of that
    # floor -- it is now the MEAN response over four (x * (1 +- 2^-23), x * (1 +- 2^-22); table: tools/grad_gate_floor.py ->
    # profiles/r6_grad_gate_floor.txt).  A tensor may be at most three times its floor (+ 2e-3) away from the oracle; nothing may
    # be further than 3e-2; and only the tensors of GRAD_FLOOR_EXEMPT (the stereo branch, whose floor is above 1e-3) may be
    # further than the old absolute gate of 2e-2.
    floor = _floor_table(cfg_name, ac, sd0, trainable, smp, model.img_view_transformer.D, ograds)
    over = sorted(((rows[k] / (3.0 * floor[k] + 2e-3), k, rows[k], floor[k]) for k in rows), reverse=True)
    print(f"{cfg_name} ac={ac} worst distance / (3 floor + 2e-3): {over[0][0]:.2f} ({over[0][1]}: ours {over[0][2]:.3e}, floor {over[0][3]:.3e})")
    assert len(rows) > 150 and worst[0] < 3e-2, worst
    assert over[0][0] < 1.0, over[:3]
    above_old_gate = [k for k, v in rows.items() if v >= 2e-2]
    assert all(k.startswith(GRAD_FLOOR_EXEMPT) and floor[k] > 1e-3 for k in above_old_gate), above_old_gate


# tensors allowed above the absolute 2e-2 gradient gate (never above 3e-2), because the ORACLE itself moves by that much on them under a
# perturbation at rounding level: the camera-aware gate of the stereo feature net (VT:32-65) and the 3-D aggregation behind it
GRAD_FLOOR_EXEMPT = ("img_view_transformer.stereo_volume_net.feature_withcam.", "img_view_transformer.stereo_volume_net.dres")
FLOOR_PERTURBATIONS = (1, -1, 2, -2)           # input scaled by (1 + k * 2^-23)


def _floor_table(cfg_name, ac, sd0, trainable, smp, D, ograds, per_sample=None):
    """Mean L2-relative response of the oracle's parameter gradients to FLOOR_PERTURBATIONS of its inputs."""
    if per_sample is None and (cfg_name, ac) in _FLOOR:
        return _FLOOR[(cfg_name, ac)]
    acc = {}
    for k in FLOOR_PERTURBATIONS:
        _, _, pg = _oracle_step(cfg_name, ac, sd0, trainable, smp, D, perturb=k)
        t = _l2_table(pg, ograds)
        if per_sample is not None:
            per_sample[k] = t
        for name, v in t.items():
            acc[name] = acc.get(name, 0.0) + v / len(FLOOR_PERTURBATIONS)
    _FLOOR[(cfg_name, ac)] = acc
    return acc


def test_gradient_gate_vs_oracle_noise_floor():
    """"ReLU sign flips is asserted, not shown".  The same L2 statistic between the ORACLE and the oracle fed
    inputs that differ by one ulp (x * (1 + 2^-23)): if the reference's own arithmetic moves the dres0 / dres1 gradients by
    the same ~1e-2 under a perturbation at rounding level, the GPU path's distance to the oracle on those tensors is the
    noise floor of the problem and not a defect; everywhere the floor is low the GPU path has to be low as well."""
    cfg_name, ac = "kitti_d112", True
    model, smp, sd0, trainable, _losses, _logits, grads = _gpu_step(cfg_name, ac)
    D = model.img_view_transformer.D
    _, _, g0 = _oracle_step(cfg_name, ac, sd0, trainable, smp, D)
    floor = _floor_table(cfg_name, ac, sd0, trainable, smp, D, g0)      # mean of oracle(x (1 + k ulp)) vs oracle(x), k = +-1, +-2
    ours = _l2_table(grads, g0)             # GPU path vs oracle(x)
    names = sorted(ours, key=lambda k: -ours[k])
    lines = [f"{ours[k]:.3e}  floor {floor.get(k, float('nan')):.3e}  {k}" for k in names[:12]]
    print("GPU-vs-oracle L2 | oracle-vs-perturbed-oracle L2 (worst 12):\n" + "\n".join(lines))
    worst = names[0]
    assert "dres" in worst or "stereo_volume_net" in worst, worst
    # the tensors on which the GPU path is furthest from the oracle are the ones the oracle itself cannot pin down:
    # its own one-ulp response there is at least a third of our distance ...
    for k in names[:5]:
        assert floor[k] > ours[k] / 3.0, (k, ours[k], floor[k])
    # ... and wherever the oracle is stable (floor < 1e-4) the GPU path is within 2e-3
    stable = [k for k in ours if floor.get(k, 1.0) < 1e-4]
    assert len(stable) > 20
    bad = [(k, ours[k]) for k in stable if ours[k] > 2e-3]
    assert not bad, bad[:5]
