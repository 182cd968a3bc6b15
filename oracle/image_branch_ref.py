"""CPU oracle of the image branch -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline); the product path never imports it.

Restates, as plain functions over a state dict (torch fp32 on CPU):
  * ``CustomEfficientNet`` (projects/mmdet3d_plugin/occupancy/backbones/efficientnet.py:231-519): ``model_scaling``
    (:231-271), the layer table (:296-345), ``make_layer`` (:442-509), ``InvertedResidual`` (:112-229), ``forward``
    (:511-519).  Only the 'b' family (InvertedResidual blocks) is restated: the config uses arch='b7'
    (stereoscene.py:60-62); the EdgeTPU 'e' family is out of scope.
  * third-party pieces that are NOT under /root/reference, restated from their published behaviour:
      - mmcv 1.4.0 ``ConvModule`` (conv -> norm -> act; bias only without norm), ``Conv2dAdaptivePadding`` (TF "same":
        out = ceil(in/stride), odd padding at the bottom/right), ``Swish`` (x * sigmoid(x)), ``DropPath``
        (per-sample Bernoulli(keep) / keep);
      - mmdet 2.14 ``SELayer`` (AdaptiveAvgPool2d(1) -> 1x1 conv + act[0] -> 1x1 conv + act[1] -> x * gate, both convs
        with bias) and ``make_divisible`` (round half up to the divisor, never below 90 % of the value);
      - mmdet3d v0.17.1 ``SECONDFPN`` (second_fpn.py: stride >= 1 -> ConvTranspose2d(k = s, stride = s, bias=False),
        stride < 1 -> Conv2d(k = s' = round(1/stride), stride = s', bias=False); BN(eps 1e-3, momentum 0.01) + ReLU;
        channel concat), configured at stereoscene.py:70-74.
Pinned by tests/golden/image_branch.npz: outputs of the REFERENCE's own efficientnet.py executed in the build
container (oracle/make_golden_image_branch.py) -- this pins the scaling / assembly / block logic that lives in the
reference; the third-party pieces above are provided to that run by independent torch.nn stand-ins, so for them parity
is "unpinned" in the sense of the task statement (no upstream golden vectors exist in the reference).
"""
import math

import torch
import torch.nn.functional as F

LAYER_SETTING_B = [[[3, 32, 0, 2, 0, -1]],
                   [[3, 16, 4, 1, 1, 0]],
                   [[3, 24, 4, 2, 6, 0], [3, 24, 4, 1, 6, 0]],
                   [[5, 40, 4, 2, 6, 0], [5, 40, 4, 1, 6, 0]],
                   [[3, 80, 4, 2, 6, 0], [3, 80, 4, 1, 6, 0], [3, 80, 4, 1, 6, 0],
                    [5, 112, 4, 1, 6, 0], [5, 112, 4, 1, 6, 0], [5, 112, 4, 1, 6, 0]],
                   [[5, 192, 4, 2, 6, 0], [5, 192, 4, 1, 6, 0], [5, 192, 4, 1, 6, 0], [5, 192, 4, 1, 6, 0],
                    [3, 320, 4, 1, 6, 0]],
                   [[1, 1280, 0, 1, 0, -1]]]                       # efficientnet.py:311-328
ARCH_SETTINGS = {"b0": (1.0, 1.0), "b1": (1.0, 1.1), "b2": (1.1, 1.2), "b3": (1.2, 1.4), "b4": (1.4, 1.8),
                 "b5": (1.6, 2.2), "b6": (1.8, 2.6), "b7": (2.0, 3.1), "b8": (2.2, 3.6)}     # :352-365 (width, depth)
BN_EPS = 1e-3                                                      # norm_cfg=dict(type='BN', eps=1e-3), :374


def make_divisible(value, divisor, min_value=None, min_ratio=0.9):
    """mmdet.models.utils.make_divisible."""
    if min_value is None:
        min_value = divisor
    new_value = max(min_value, int(value + divisor / 2) // divisor * divisor)
    if new_value < min_ratio * value:
        new_value += divisor
    return new_value


def model_scaling(arch):
    """efficientnet.py:231-271: width scaling, per-stage split at channel changes, depth scaling (ceil), re-merge of
    stride-1 groups into the previous stage.  Returns the list of stages, each a list of
    [kernel, out_channels, se_ratio, stride, expand_ratio, block_type]."""
    width, depth = ARCH_SETTINGS[arch]
    setting = [[list(b) for b in layer] for layer in LAYER_SETTING_B]
    for layer in setting:
        for blk in layer:
            blk[1] = make_divisible(blk[1] * width, 8)
    split = [setting[0]]
    for layer in setting[1:-1]:
        cuts = [0] + [i + 1 for i in range(len(layer) - 1) if layer[i + 1][1] != layer[i][1]] + [len(layer)]
        split += [layer[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
    split.append(setting[-1])
    merged = [split[0]]
    for i, layer in enumerate(split[1:-1]):
        want = int(math.ceil(depth * len(layer)))
        grown = layer[:want] if want <= len(layer) else [list(b) for b in layer] + [layer[-1]] * (want - len(layer))
        if grown[0][3] == 1 and i != 0:
            merged[-1] = merged[-1] + grown
        else:
            merged.append(list(grown))
    merged.append(split[-1])
    return merged


def conv_same(x, w, b, stride, groups=1):
    """Conv2dAdaptivePadding."""
    k = w.shape[-1]
    H, W = x.shape[-2:]
    oh, ow = -(-H // stride), -(-W // stride)
    ph, pw = max((oh - 1) * stride + k - H, 0), max((ow - 1) * stride + k - W, 0)
    if ph or pw:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
    return F.conv2d(x, w, b, stride, 0, 1, groups)


def swish(x):
    return x * torch.sigmoid(x)


def bn(sd, p, x, train, momentum=0.1, eps=BN_EPS, stats_out=None):
    w, b = sd[p + ".weight"], sd[p + ".bias"]
    if train:
        mean = x.mean((0, 2, 3))
        var = x.var((0, 2, 3), unbiased=False)
        if stats_out is not None:
            n = x.numel() // x.shape[1]
            stats_out[p + ".running_mean"] = (1 - momentum) * sd[p + ".running_mean"] + momentum * mean.detach()
            stats_out[p + ".running_var"] = (1 - momentum) * sd[p + ".running_var"] + momentum * var.detach() * n / max(n - 1, 1)
    else:
        mean, var = sd[p + ".running_mean"], sd[p + ".running_var"]
    return (x - mean[None, :, None, None]) * torch.rsqrt(var + eps)[None, :, None, None] * w[None, :, None, None] \
        + b[None, :, None, None]


def conv_module(sd, p, x, stride, groups=1, act=True, train=False, stats_out=None):
    """mmcv ConvModule(conv_cfg=Conv2dAdaptivePadding, norm_cfg=BN(eps=1e-3), act_cfg=Swish | None): no conv bias."""
    y = bn(sd, p + ".bn", conv_same(x, sd[p + ".conv.weight"], None, stride, groups), train, stats_out=stats_out)
    return swish(y) if act else y


def se_layer(sd, p, x):
    """mmdet SELayer with act_cfg=(Swish, Sigmoid); plain Conv2d 1x1 with bias, no norm."""
    g = x.mean((2, 3), keepdim=True)
    g = swish(F.conv2d(g, sd[p + ".conv1.conv.weight"], sd[p + ".conv1.conv.bias"]))
    g = torch.sigmoid(F.conv2d(g, sd[p + ".conv2.conv.weight"], sd[p + ".conv2.conv.bias"]))
    return x * g


def inverted_residual(sd, p, x, cin, cout, k, stride, expand, train=False, drop_mask=None, stats_out=None):
    """efficientnet.py:204-229 (`_inner_forward`); ``drop_mask`` = the per-sample DropPath factor (mask / keep_prob),
    None = identity (eval, or rate 0)."""
    mid = int(cin * expand)
    y = x
    if mid != cin:                                                  # with_expand_conv, :504
        y = conv_module(sd, p + ".expand_conv", y, 1, train=train, stats_out=stats_out)
    y = conv_module(sd, p + ".depthwise_conv", y, stride, groups=mid, train=train, stats_out=stats_out)
    y = se_layer(sd, p + ".se", y)
    y = conv_module(sd, p + ".linear_conv", y, 1, act=False, train=train, stats_out=stats_out)
    if stride == 1 and cin == cout:                                 # with_res_shortcut, :155
        if drop_mask is not None:
            y = y * drop_mask.view(-1, 1, 1, 1)
        return x + y
    return y


def block_table(arch, out_indices):
    """(stage index in ``layers``, block index, cin, cout, k, stride, expand) for every block that is built
    (make_layer stops after max(out_indices), :459-461), plus stem / head channel counts."""
    stages = model_scaling(arch)
    cin = make_divisible(stages[0][0][1], 8)
    stem = cin
    rows = []
    for si, layer in enumerate(stages[1:-1]):
        if si > max(out_indices) - 1:
            break
        for bi, (k, cout, se_ratio, stride, expand, _t) in enumerate(layer):
            cout = make_divisible(cout, 8)
            rows.append((si + 1, bi, cin, cout, k, stride, expand))
            cin = cout
    return stem, rows, cin, stages[-1][0][1]


def efficientnet(sd, p, x, arch="b7", out_indices=(2, 3, 4, 5, 6), train=False, drop_masks=None, stats_out=None):
    """``CustomEfficientNet.forward`` (:511-519): tuple of the feature maps of ``layers[i]``, i in out_indices.
    ``drop_masks``: dict (stage, block) -> per-sample factor, for train-mode DropPath parity."""
    pre = (p + ".") if p else ""
    stem, rows, c_last, c_head = block_table(arch, out_indices)
    x = conv_module(sd, pre + "layers.0", x, 2, train=train, stats_out=stats_out)
    outs = []
    n_stage = max(r[0] for r in rows)
    for si in range(1, n_stage + 1):
        for (s, bi, cin, cout, k, stride, expand) in rows:
            if s == si:
                dm = None if drop_masks is None else drop_masks.get((s, bi))
                x = inverted_residual(sd, f"{pre}layers.{s}.{bi}", x, cin, cout, k, stride, expand, train, dm, stats_out)
        if si in out_indices:
            outs.append(x)
    if n_stage + 1 <= max(out_indices):                             # final 1x1 ConvModule, :427-440
        x = conv_module(sd, f"{pre}layers.{n_stage + 1}", x, 1, train=train, stats_out=stats_out)
        if n_stage + 1 in out_indices:
            outs.append(x)
    return tuple(outs)


def drop_path_rates(arch, rate):
    """torch.linspace(0, rate, total blocks of ALL stages) (:447-451)."""
    total = sum(len(layer) for layer in model_scaling(arch)[1:-1])
    return [v.item() for v in torch.linspace(0, rate, total)]


def second_fpn(sd, p, feats, upsample_strides, train=False, stats_out=None):
    """mmdet3d SECONDFPN.forward: deblocks[i] = (deconv | conv) -> BN(eps 1e-3, momentum 0.01) -> ReLU; concat."""
    pre = (p + ".") if p else ""
    ups = []
    for i, (x, s) in enumerate(zip(feats, upsample_strides)):
        w = sd[f"{pre}deblocks.{i}.0.weight"]
        if s >= 1:
            s = int(s)
            y = F.conv_transpose2d(x, w, None, stride=s)
        else:
            s = int(round(1 / s))
            y = F.conv2d(x, w, None, stride=s)
        y = F.relu(bn(sd, f"{pre}deblocks.{i}.1", y, train, momentum=0.01, stats_out=stats_out))
        ups.append(y)
    return [torch.cat(ups, 1) if len(ups) > 1 else ups[0]]
