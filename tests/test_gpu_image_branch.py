"""GPU parity of the image branch: the HIP operators against ATen, the registry-built
CustomEfficientNet / SECONDFPN against the CPU oracle (which is pinned to the reference's own efficientnet.py by
tests/test_image_branch_oracle.py), forward and gradients."""
import pytest
import torch
import torch.nn.functional as TF

from conftest import load_golden
from oracle import image_branch_ref as IB
from stereoscene_amd import functional as F, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def maxdiff(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def cl(t):        # logical NCHW tensor in channels-last memory, as the branch produces them
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("case", [(2, 32, 9, 14, 3, 1), (2, 48, 10, 16, 3, 2), (1, 96, 11, 13, 5, 2), (2, 64, 7, 9, 5, 1),
                                  (1, 288, 24, 40, 5, 2), (1, 12, 1, 5, 3, 1)])
def test_depthwise_same_conv_matches_aten(case):
    B, Cc, H, W, k, s = case
    x = S.hash_normal(f"dw/x{case}", (B, Cc, H, W))
    w = S.hash_uniform(f"dw/w{case}", (Cc, 1, k, k), -1, 1)
    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    want = IB.conv_same(xc, wc, None, s, groups=Cc)
    go = S.hash_normal(f"dw/go{case}", tuple(want.shape))
    want.backward(go)
    xg, wg = cl(x).requires_grad_(True), w.to(DEV).requires_grad_(True)
    got = F.depthwise_conv2d_same(xg, wg, s)
    got.backward(cl(go))
    assert got.shape == want.shape
    assert maxdiff(got, want) < 1e-5 * max(1.0, want.abs().max().item())
    assert maxdiff(xg.grad, xc.grad) < 1e-5 * max(1.0, xc.grad.abs().max().item())
    assert maxdiff(wg.grad, wc.grad) < 2e-5 * max(1.0, wc.grad.abs().max().item())


def test_swish_and_se_pieces_match_aten():
    x = S.hash_normal("se/x", (2, 40, 6, 10), 2.0)
    gate = torch.sigmoid(S.hash_normal("se/g", (2, 40, 1, 1)))
    go = S.hash_normal("se/go", (2, 40, 6, 10))
    xc, gc = x.clone().requires_grad_(True), gate.clone().requires_grad_(True)
    want = (xc * torch.sigmoid(xc)) * gc + xc.mean((2, 3), keepdim=True)
    want.backward(go)
    xg, gg = cl(x).requires_grad_(True), gate.to(DEV).requires_grad_(True)
    got = F.chan_scale(F.swish(xg), gg) + F.global_avg_pool(xg)
    got.backward(cl(go))
    assert maxdiff(got, want) < 2e-6 * max(1.0, want.abs().max().item())
    assert maxdiff(xg.grad, xc.grad) < 5e-6 and maxdiff(gg.grad, gc.grad) < 2e-5
    t = S.hash_normal("se/odd", (3, 7))                   # sizes that are not a multiple of 4 take the tensor-op path
    assert maxdiff(F.swish(t.to(DEV)), t * torch.sigmoid(t)) < 1e-6


def _sd_from(g, tag, prefix):
    sd = {}
    for k, shp in g.items():
        if k.startswith(tag):
            key = k[len(tag):]
            dtype = torch.int64 if key.endswith("num_batches_tracked") else torch.float32
            t = torch.zeros(tuple(int(v) for v in shp), dtype=dtype)
            v = S.fill_value_for(prefix + key, t)
            sd[key] = t if v is None else v.to(dtype)
    return sd


def _build(arch, **kw):
    from stereoscene_amd import plugin  # noqa: F401
    from stereoscene_amd.registry import BACKBONES
    return BACKBONES.build(dict(type="CustomEfficientNet", arch=arch, **kw))


def test_efficientnet_b7_eval_vs_reference_fixture():
    """The config's backbone (arch b7, out_indices 2..6) on the HIP kernels against outputs of the reference's own
    efficientnet.py (same fill-by-key weights)."""
    g = load_golden("image_branch")
    net = _build("b7", drop_path_rate=0.2, out_indices=(2, 3, 4, 5, 6), with_cp=True)
    net.load_state_dict(_sd_from(g, "shape:", "img_backbone."))
    net = net.to(DEV).eval()
    with torch.no_grad():
        outs = net(cl(torch.from_numpy(g["x"])))
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g[f"b7_eval_{i}"])
        assert maxdiff(o, ref) < 1e-4 * max(1.0, ref.abs().max().item()), i


@pytest.mark.parametrize("with_cp", [False, True])
def test_efficientnet_b0_train_fwd_bwd_vs_oracle(with_cp):
    """Train mode (batch-stat BN, DropPath rate 0) with activation checkpointing on/off: outputs against the reference
    fixture, parameter and input gradients against the CPU oracle's autograd."""
    g = load_golden("image_branch")
    sd0 = _sd_from(g, "b0_shape:", "b0.")
    net = _build("b0", out_indices=(1, 3, 4), with_cp=with_cp)
    net.load_state_dict(sd0)
    net = net.to(DEV).train()
    x = torch.from_numpy(g["x"])
    xg = cl(x).requires_grad_(True)
    outs = net(xg)
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g[f"b0_train_{i}"])
        assert maxdiff(o, ref) < 2e-4 * max(1.0, ref.abs().max().item()), i
    gos = [S.hash_normal(f"b0/go{i}", tuple(o.shape)) for i, o in enumerate(outs)]
    sum((o * cl(go)).sum() for o, go in zip(outs, gos)).backward()
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd0.items()}
    xc = x.clone().requires_grad_(True)
    want = IB.efficientnet(sd, "", xc, arch="b0", out_indices=(1, 3, 4), train=True)
    sum((o * go).sum() for o, go in zip(want, gos)).backward()
    assert maxdiff(xg.grad, xc.grad) < 2e-3 * max(1.0, xc.grad.abs().max().item())
    checked = 0
    gmax = max(v.grad.abs().max().item() for v in sd.values() if v.requires_grad and v.grad is not None)
    for name, p in net.named_parameters():
        ref = sd[name].grad
        if ref is None or p.grad is None:
            continue
        scale = ref.abs().max().item()
        if scale < 1e-4 * gmax:
            # analytically-zero gradients (a per-channel shift in front of a conv + batch-stat BN: the bias of every
            # linear_conv BN that feeds an expand conv, SE biases behind saturated gates): rounding noise on both sides
            assert p.grad.abs().max().item() < 1e-3 * gmax, (name, scale, gmax)
            continue
        l2 = ((p.grad.cpu() - ref).norm() / ref.norm()).item()
        assert l2 < 2e-3, (name, l2, scale, gmax)
        checked += 1
    assert checked > 120
    if not with_cp:      # running statistics (with checkpointing the recomputation updates them twice, as upstream)
        new = net.state_dict()
        for k, v in g.items():
            if k.startswith("b0_stat:"):
                assert maxdiff(new[k[len("b0_stat:"):]], torch.from_numpy(v)) < 1e-4, k


def test_second_fpn_image_neck_vs_oracle():
    from stereoscene_amd.registry import NECKS
    from stereoscene_amd import plugin  # noqa: F401
    in_ch, strides = [48, 80, 224, 640, 2560], [0.5, 1, 2, 4, 4]
    neck = NECKS.build(dict(type="SECONDFPN", in_channels=in_ch, upsample_strides=strides, out_channels=[128] * 5))
    S.fill_state_dict_(neck, "img_neck.")
    sd0 = {k: v.detach().clone() for k, v in neck.state_dict().items()}
    neck = neck.to(DEV).train()
    hw = [(8, 12), (4, 6), (2, 3), (1, 2), (1, 2)]         # /4 ... /32 of a 32 x 48 image: not all consistent -> use explicit sizes
    hw = [(8, 16), (4, 8), (2, 4), (1, 2), (1, 2)]
    feats = [S.hash_normal(f"neck/f{i}", (2, c, h, w)) for i, (c, (h, w)) in enumerate(zip(in_ch, hw))]
    fg = [cl(f).requires_grad_(True) for f in feats]
    got = neck(fg)[0]
    go = S.hash_normal("neck/go", tuple(got.shape))
    got.backward(cl(go))
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd0.items()}
    fc = [f.clone().requires_grad_(True) for f in feats]
    stats = {}
    want = IB.second_fpn(sd, "", fc, strides, train=True, stats_out=stats)[0]
    want.backward(go)
    assert got.shape == want.shape == (2, 640, 4, 8)
    assert maxdiff(got, want) < 1e-4 * max(1.0, want.abs().max().item())
    for a, b in zip(fg, fc):
        assert maxdiff(a.grad, b.grad) < 1e-4 * max(1.0, b.grad.abs().max().item())
    for name, p in neck.named_parameters():
        assert maxdiff(p.grad, sd[name].grad) < 2e-4 * max(1.0, sd[name].grad.abs().max().item()), name
    new = neck.state_dict()
    for k, v in stats.items():
        assert maxdiff(new[k], v) < 1e-5, k


def test_detector_with_image_branch_trains_end_to_end():
    """Raw stereo images -> EfficientNet + SECONDFPN -> hot path -> losses -> backward (tiny config, arch b0 widths are
    not compatible with the neck, so the config's b7 is used at a 64 x 160 image)."""
    from stereoscene_amd import model_zoo
    from stereoscene_amd.registry import DETECTORS
    cfg = S.CFG_T
    mc = model_zoo.model_cfg(cfg, image_branch=True)
    mc["img_backbone"]["with_cp"] = True
    model = DETECTORS.build(mc)
    S.fill_state_dict_(model)
    model = model.to(DEV).train()
    smp = S.synthetic_sample(cfg, B=1, tag="e2e")
    left, right = model_zoo.img_inputs_from_sample(smp)
    H, W = cfg["input_size"]
    img_l = cl(S.hash_normal("e2e/l", (1, 3, H, W))).unsqueeze(1)
    img_r = cl(S.hash_normal("e2e/r", (1, 3, H, W))).unsqueeze(1)
    inputs = ((img_l,) + tuple(left[1:]), (img_r,) + tuple(right[1:]))
    losses = model.forward_train(img_inputs=inputs, gt_occ=smp["gt_occ"].to(DEV))
    total = sum(v for k, v in losses.items() if k.startswith("loss"))
    total.backward()
    assert torch.isfinite(total)
    gb = [p.grad for n, p in model.named_parameters() if n.startswith("img_backbone.") and p.grad is not None]
    assert len(gb) > 700 and all(torch.isfinite(t).all() for t in gb)
    assert any(t.abs().max().item() > 0 for t in gb)
