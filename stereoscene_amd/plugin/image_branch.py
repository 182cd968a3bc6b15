"""Image branch of the detector: ``CustomEfficientNet`` (backbones/efficientnet.py:275-533, registered
under BACKBONES) and mmdet3d's ``SECONDFPN`` image neck (stereoscene.py:70-74, registered under NECKS), with the
reference's constructor kwargs and state-dict keys (``layers.{i}.{j}.expand_conv.conv.weight`` ...), running on the
HIP kernels: dense 1x1 / stem / neck convolutions on the MFMA conv kernels, depthwise convs, Swish and the
squeeze-excitation pool / rescale on ``csrc/image_ops.hip``, BatchNorm on the two-stage GN/BN kernels.

Only the 'b' family (InvertedResidual blocks) is built -- the config uses arch='b7'; EdgeTPU archs raise.
"""
import math

import torch
import torch.nn as nn
import torch.utils.checkpoint as cp

from .. import functional as F
from ..layers import BatchNorm2d, Conv2d
from ..registry import BACKBONES, NECKS

LAYER_SETTING_B = [[[3, 32, 0, 2, 0, -1]],
                   [[3, 16, 4, 1, 1, 0]],
                   [[3, 24, 4, 2, 6, 0], [3, 24, 4, 1, 6, 0]],
                   [[5, 40, 4, 2, 6, 0], [5, 40, 4, 1, 6, 0]],
                   [[3, 80, 4, 2, 6, 0], [3, 80, 4, 1, 6, 0], [3, 80, 4, 1, 6, 0],
                    [5, 112, 4, 1, 6, 0], [5, 112, 4, 1, 6, 0], [5, 112, 4, 1, 6, 0]],
                   [[5, 192, 4, 2, 6, 0], [5, 192, 4, 1, 6, 0], [5, 192, 4, 1, 6, 0], [5, 192, 4, 1, 6, 0],
                    [3, 320, 4, 1, 6, 0]],
                   [[1, 1280, 0, 1, 0, -1]]]
ARCH_SETTINGS = {"b0": (1.0, 1.0, 224), "b1": (1.0, 1.1, 240), "b2": (1.1, 1.2, 260), "b3": (1.2, 1.4, 300),
                 "b4": (1.4, 1.8, 380), "b5": (1.6, 2.2, 456), "b6": (1.8, 2.6, 528), "b7": (2.0, 3.1, 600),
                 "b8": (2.2, 3.6, 672)}


def make_divisible(value, divisor, min_value=None, min_ratio=0.9):
    floor = divisor if min_value is None else min_value
    v = max(floor, int(value + divisor / 2) // divisor * divisor)
    return v + divisor if v < min_ratio * value else v


def model_scaling(layer_setting, arch_setting):
    """Width / depth scaling of the block table (efficientnet.py:231-271)."""
    width, depth = arch_setting[0], arch_setting[1]
    scaled = [[[b[0], make_divisible(b[1] * width, 8)] + list(b[2:]) for b in layer] for layer in layer_setting]
    groups = []                                   # middle stages cut where the channel count changes
    for layer in scaled[1:-1]:
        start = 0
        for i in range(1, len(layer) + 1):
            if i == len(layer) or layer[i][1] != layer[i - 1][1]:
                groups.append(layer[start:i])
                start = i
    stages = [scaled[0]]
    for gi, grp in enumerate(groups):
        n = int(math.ceil(depth * len(grp)))
        grown = grp[:n] + [grp[-1]] * max(0, n - len(grp))
        if gi != 0 and grown[0][3] == 1:
            stages[-1] = stages[-1] + grown       # a stride-1 group continues the previous stage
        else:
            stages.append(grown)
    stages.append(scaled[-1])
    return stages


class SamePadConv2d(nn.Conv2d):
    """mmcv ``Conv2dAdaptivePadding`` (dense, groups = 1): input zero-padded to ceil(in/stride) outputs, then the MFMA
    conv kernel with padding 0.  Stride-1 odd kernels reduce to symmetric padding inside the kernel (no pad copy)."""

    def forward(self, x):
        k, s = self.kernel_size[0], self.stride[0]
        if k == 1 and s == 1:
            return F.linear_cl(x, self.weight, self.bias)      # pointwise conv = plain GEMM on the channels-last buffer
        if s == 1:
            return F.conv2d(x, self.weight, self.bias, 1, k // 2, 1)
        _, pt, pb = F.same_padding(x.shape[-2], k, s)
        _, pl, pr = F.same_padding(x.shape[-1], k, s)
        if pt or pb or pl or pr:
            x = torch.nn.functional.pad(x, [pl, pr, pt, pb])
        return F.conv2d(x, self.weight, self.bias, s, 0, 1)


class DepthwiseSamePadConv2d(nn.Conv2d):
    """Conv2dAdaptivePadding with groups = channels: the depthwise HIP kernel (padding handled in the kernel)."""

    def forward(self, x):
        return F.depthwise_conv2d_same(x, self.weight, self.stride[0])


class Swish(nn.Module):
    def forward(self, x):
        return F.swish(x)


class ConvModule(nn.Module):
    """mmcv ``ConvModule`` subset used by the branch: conv -> BN -> activation; the conv has a bias only without norm."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, groups=1, conv_cfg=None,
                 norm_cfg=None, act_cfg=None):
        super().__init__()
        bias = norm_cfg is None
        same = conv_cfg is not None and conv_cfg.get("type") == "Conv2dAdaptivePadding"
        if groups == 1:
            pointwise = kernel_size == 1 and stride == 1 and padding == 0
            self.conv = SamePadConv2d(in_channels, out_channels, kernel_size, stride, 0, bias=bias) if (same or pointwise) \
                else Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        else:
            if not (same and groups == in_channels == out_channels):
                raise NotImplementedError("grouped convs other than depthwise 'same' ones are not part of the branch")
            self.conv = DepthwiseSamePadConv2d(in_channels, out_channels, kernel_size, stride, 0, groups=groups, bias=bias)
        if norm_cfg is not None:
            if norm_cfg.get("type", "BN") not in ("BN", "BN2d"):
                raise NotImplementedError(norm_cfg)
            self.bn = BatchNorm2d(out_channels, eps=norm_cfg.get("eps", 1e-5), momentum=norm_cfg.get("momentum", 0.1))
        else:
            self.bn = None
        t = None if act_cfg is None else act_cfg["type"]
        self.act = t
        self.activate = Swish() if t == "Swish" else nn.Sigmoid() if t == "Sigmoid" else nn.ReLU() if t == "ReLU" else None
        if t not in (None, "Swish", "Sigmoid", "ReLU"):
            raise NotImplementedError(act_cfg)

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x, relu=self.act == "ReLU")            # ReLU rides in the BN pass
            if self.act == "ReLU":
                return x
        return x if self.activate is None else self.activate(x)


class SELayer(nn.Module):
    """mmdet ``SELayer``: x * sigmoid(W2 swish(W1 avgpool(x)))."""

    def __init__(self, channels, ratio=16, conv_cfg=None, act_cfg=(dict(type="ReLU"), dict(type="Sigmoid"))):
        super().__init__()
        mid = int(channels / ratio)
        self.conv1 = ConvModule(channels, mid, 1, 1, act_cfg=act_cfg[0])
        self.conv2 = ConvModule(mid, channels, 1, 1, act_cfg=act_cfg[1])

    def forward(self, x):
        return F.chan_scale(x, self.conv2(self.conv1(F.global_avg_pool(x))))


class DropPath(nn.Module):
    """mmcv ``DropPath`` (stochastic depth, per sample)."""

    def __init__(self, drop_prob=0.1):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = (keep + torch.rand((x.shape[0],) + (1,) * (x.dim() - 1), dtype=x.dtype, device=x.device)).floor()
        return x.div(keep) * mask


class InvertedResidual(nn.Module):
    """MBConv block (efficientnet.py:112-229)."""

    def __init__(self, in_channels, out_channels, mid_channels, kernel_size=3, stride=1, se_cfg=None,
                 with_expand_conv=True, conv_cfg=None, norm_cfg=dict(type="BN"), act_cfg=dict(type="ReLU"),
                 drop_path_rate=0.0, with_cp=False, init_cfg=None):
        super().__init__()
        assert stride in (1, 2)
        self.with_res_shortcut = stride == 1 and in_channels == out_channels
        self.with_cp = with_cp
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > 0 else nn.Identity()
        self.with_se = se_cfg is not None
        self.with_expand_conv = with_expand_conv
        if not with_expand_conv:
            assert mid_channels == in_channels
        if with_expand_conv:
            self.expand_conv = ConvModule(in_channels, mid_channels, 1, 1, 0, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                          act_cfg=act_cfg)
        self.depthwise_conv = ConvModule(mid_channels, mid_channels, kernel_size, stride, kernel_size // 2,
                                         groups=mid_channels, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        if self.with_se:
            self.se = SELayer(**se_cfg)
        self.linear_conv = ConvModule(mid_channels, out_channels, 1, 1, 0, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                      act_cfg=None)

    def _inner(self, x):
        out = self.expand_conv(x) if self.with_expand_conv else x
        out = self.depthwise_conv(out)
        if self.with_se:
            out = self.se(out)
        out = self.linear_conv(out)
        return x + self.drop_path(out) if self.with_res_shortcut else out

    def forward(self, x):
        if self.with_cp and x.requires_grad:
            return cp.checkpoint(self._inner, x, use_reentrant=False)
        return self._inner(x)


@BACKBONES.register_module()
class CustomEfficientNet(nn.Module):
    layer_settings = {"b": LAYER_SETTING_B}
    arch_settings = ARCH_SETTINGS

    def __init__(self, arch="b0", drop_path_rate=0.0, out_indices=(6,), frozen_stages=0,
                 conv_cfg=dict(type="Conv2dAdaptivePadding"), norm_cfg=dict(type="BN", eps=1e-3),
                 act_cfg=dict(type="Swish"), norm_eval=False, with_cp=False, init_cfg=None):
        super().__init__()
        if arch not in self.arch_settings:
            raise NotImplementedError(f"arch {arch!r}: only the 'b' family {sorted(self.arch_settings)} is built")
        stages = model_scaling(self.layer_settings["b"], self.arch_settings[arch])
        n_layers = len(stages)
        for index in out_indices:
            if index not in range(n_layers):
                raise ValueError(f"the item in out_indices must in range(0, {n_layers}). But received {index}")
        if frozen_stages not in range(n_layers + 1):
            raise ValueError(f"frozen_stages must be in range(0, {n_layers + 1}). But received {frozen_stages}")
        self.out_indices, self.frozen_stages, self.norm_eval, self.with_cp = tuple(out_indices), frozen_stages, norm_eval, with_cp
        self.drop_path_rate = drop_path_rate
        k0, c0, _, s0 = stages[0][0][:4]
        self.in_channels = make_divisible(c0, 8)
        self.out_channels = stages[-1][0][1]
        self.layers = nn.ModuleList([ConvModule(3, self.in_channels, k0, s0, k0 // 2, conv_cfg=conv_cfg,
                                                norm_cfg=norm_cfg, act_cfg=act_cfg)])
        middle = stages[1:-1]
        total = sum(len(layer) for layer in middle)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, total)]      # stochastic-depth decay rule
        bidx = 0
        for si, layer in enumerate(middle):
            if si > max(self.out_indices) - 1:
                break                                                           # unused stages are never built
            blocks = []
            for (k, cout, se_ratio, stride, expand, _t) in layer:
                mid = int(self.in_channels * expand)
                cout = make_divisible(cout, 8)
                se_cfg = None if se_ratio <= 0 else dict(channels=mid, ratio=expand * se_ratio,
                                                         act_cfg=(act_cfg, dict(type="Sigmoid")))
                blocks.append(InvertedResidual(self.in_channels, cout, mid, k, stride, se_cfg,
                                               with_expand_conv=(mid != self.in_channels), conv_cfg=conv_cfg,
                                               norm_cfg=norm_cfg, act_cfg=act_cfg, drop_path_rate=dpr[bidx],
                                               with_cp=with_cp))
                self.in_channels = cout
                bidx += 1
            self.layers.append(nn.Sequential(*blocks))
        if len(self.layers) < max(self.out_indices) + 1:
            kl, _, _, sl = stages[-1][0][:4]
            self.layers.append(ConvModule(self.in_channels, self.out_channels, kl, sl, kl // 2, conv_cfg=conv_cfg,
                                          norm_cfg=norm_cfg, act_cfg=act_cfg))

    def forward(self, x):
        outs = []
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def _freeze_stages(self):
        for i in range(self.frozen_stages):
            m = self.layers[i]
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self


@NECKS.register_module()
class SECONDFPN(nn.Module):
    """mmdet3d v0.17.1 ``SECONDFPN`` as configured for the image neck (stereoscene.py:70-74): per level a
    ConvTranspose2d (k = s = stride >= 1) or Conv2d (k = s = round(1/stride)) without bias, BN(eps 1e-3,
    momentum 0.01), ReLU; concatenated along channels."""

    def __init__(self, in_channels=(128, 128, 256), out_channels=(256, 256, 256), upsample_strides=(1, 2, 4),
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), upsample_cfg=dict(type="deconv", bias=False),
                 conv_cfg=dict(type="Conv2d", bias=False), use_conv_for_no_stride=False, init_cfg=None):
        super().__init__()
        assert len(out_channels) == len(upsample_strides) == len(in_channels)
        self.in_channels, self.out_channels = list(in_channels), list(out_channels)
        self.upsample_strides = list(upsample_strides)
        blocks = []
        for cin, cout, s in zip(in_channels, out_channels, upsample_strides):
            if s > 1 or (s == 1 and not use_conv_for_no_stride):
                layer = _Deconv2d(cin, cout, int(s))
            else:
                r = int(round(1 / s))
                layer = Conv2d(cin, cout, r, r, 0, bias=False)
            bn = BatchNorm2d(cout, eps=norm_cfg.get("eps", 1e-5), momentum=norm_cfg.get("momentum", 0.1))
            bn.fused_relu = True
            blocks.append(nn.Sequential(layer, bn, nn.Identity()))      # slot 2 = the ReLU, fused into the BN pass
        self.deblocks = nn.ModuleList(blocks)

    def forward(self, x):
        assert len(x) == len(self.in_channels)
        ups = [blk(x[i]) for i, blk in enumerate(self.deblocks)]
        return [torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]]


class _Deconv2d(nn.ConvTranspose2d):
    """ConvTranspose2d(k = s, stride = s, bias=False) through the transposed MFMA conv (depth-1 volume)."""

    def __init__(self, cin, cout, s):
        super().__init__(cin, cout, s, s, bias=False)

    def forward(self, x):
        s = self.stride[0]
        y = F.conv_transpose3d(x.unsqueeze(2), self.weight.unsqueeze(2), None, (1, s, s), 0, 0)
        return y.squeeze(2)
