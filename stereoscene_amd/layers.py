"""Parameter-compatible layer classes whose forward runs on the ssbev HIP kernels.

Each class keeps the parameter names/shapes of the torch layer the reference instantiates
(``weight``/``bias`` of nn.Conv3d etc.), so checkpoints load by key, but the arithmetic goes
through stereoscene_amd.functional (C ABI -> gfx950 kernels).  CPU tensors raise: there is no
fallback path.  ``build_*_layer`` mirror the mmcv.cnn builders the reference calls
(``build_norm_layer``/``build_conv_layer``/``build_upsample_layer``).
"""
import math

import torch
import torch.nn as nn

from . import functional as F


def _ntuple(v, n):
    return (v,) * n if isinstance(v, int) else tuple(v)


class _ConvBase(nn.Module):
    ND = 3
    TRANSPOSED = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, output_padding=0):
        super().__init__()
        if groups != 1:
            raise NotImplementedError("grouped convolution is only provided by DeformConv2dPack")
        n = self.ND
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _ntuple(kernel_size, n)
        self.stride, self.padding = _ntuple(stride, n), _ntuple(padding, n)
        self.dilation, self.output_padding = _ntuple(dilation, n), _ntuple(output_padding, n)
        shape = ((in_channels, out_channels) if self.TRANSPOSED else (out_channels, in_channels)) + self.kernel_size
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.weight[0].numel() if not self.TRANSPOSED else self.weight.shape[0] * self.weight[0, 0].numel()
            bound = 1 / math.sqrt(max(fan_in, 1))
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, hip=mfma_f32_32x32x2")


class Conv3d(_ConvBase):
    def forward(self, x, relu=False):
        return F.conv3d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, relu=relu)


class ConvTranspose3d(_ConvBase):
    TRANSPOSED = True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, 1, 1, bias, output_padding)

    def forward(self, x):
        return F.conv_transpose3d(x, self.weight, self.bias, self.stride, self.padding, self.output_padding)


class Conv2d(_ConvBase):
    ND = 2

    def forward(self, x):
        return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation)


class DeformConv2dPack(nn.Module):
    """mmcv ``DCN`` (DeformConv2dPack, DCNv1) as used at BD:490-498: ``weight`` [Cout, Cin/g, k, k],
    ``conv_offset`` (3x3 conv, bias, zero-initialised).  Bilinear tap sampling (and its adjoint) are HIP kernels;
    the grouped contraction runs as one dense MFMA 1x1 convolution per group over the sampled columns."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1,
                 deform_groups=1, im2col_step=128, bias=False):
        super().__init__()
        assert not bias and deform_groups == 1 and stride == 1
        self.groups, self.k, self.padding, self.dilation = groups, kernel_size, padding, dilation
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, kernel_size, kernel_size))
        self.conv_offset = Conv2d(in_channels, 2 * kernel_size * kernel_size, kernel_size, 1, padding, dilation)
        n = in_channels * kernel_size * kernel_size
        nn.init.uniform_(self.weight, -1 / math.sqrt(n), 1 / math.sqrt(n))
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x):
        return F.deform_conv2d(x, self.conv_offset(x), self.weight, self.groups, self.padding, self.dilation)

    def _forward_reference(self, x, off):
        """The same operator in plain tensor ops (used by the parity test of the HIP sampling kernels)."""
        B, C, H, W = x.shape
        k, K = self.k, self.k * self.k
        dev = x.device
        base_h = (torch.arange(H, device=dev, dtype=x.dtype) - self.padding)[:, None]
        base_w = (torch.arange(W, device=dev, dtype=x.dtype) - self.padding)[None, :]
        ki = torch.arange(K, device=dev)
        off = off.reshape(B, K, 2, H, W)
        hh = base_h + (ki // k * self.dilation).view(1, K, 1, 1).to(x.dtype) + off[:, :, 0]
        ww = base_w + (ki % k * self.dilation).view(1, K, 1, 1).to(x.dtype) + off[:, :, 1]
        inside = (hh > -1) & (ww > -1) & (hh < H) & (ww < W)
        h0, w0 = torch.floor(hh), torch.floor(ww)
        lh, lw = hh - h0, ww - w0
        flat = x.reshape(B, C, H * W)
        cols = 0
        for dh, dw, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            hi, wi = (h0 + dh).long(), (w0 + dw).long()
            ok = inside & (hi >= 0) & (hi < H) & (wi >= 0) & (wi < W)
            lin = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(B, 1, K * H * W).expand(B, C, K * H * W)
            g = torch.gather(flat, 2, lin).view(B, C, K, H * W)
            cols = cols + g * (wt * ok).view(B, 1, K, H * W)
        G = self.groups
        cols = cols.view(B, G, (C // G) * K, H * W)
        wmat = self.weight.view(G, self.weight.shape[0] // G, -1)
        out = torch.einsum("gok,bgkn->bgon", wmat, cols)
        return out.reshape(B, self.weight.shape[0], H, W)


class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm parameters; forward = relu?(GN(x) + residual?) in ONE pass of the HIP kernel
    (``fused_relu`` is switched on by ``fuse_relu_`` when the reference places nn.ReLU right after)."""
    fused_relu = False

    def forward(self, x, residual=None, relu=None):
        relu = self.fused_relu if relu is None else relu
        if x.dim() < 3 or self.num_channels % 4 != 0:       # the [B, 30] camera vector: a tiny ATen op
            y = nn.functional.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
            y = y if residual is None else y + residual
            return torch.relu(y) if relu else y
        return F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps, residual, relu)


class _HipBatchNorm:
    """Mixin: nn.BatchNormNd parameters/buffers; training mode = batch statistics from the two-stage HIP
    reduction (+ momentum update of the running stats), eval mode = running statistics.  Optional fused
    residual add and ReLU: relu?(BN(x) + residual?) in one streaming pass."""
    fused_relu = False

    def _running(self):
        """(running_mean, running_var, momentum) for the operator to update in place (the momentum update of nn.BatchNorm in
        training mode happens in the finalize tail of the statistics kernel), or None; counts the batch."""
        if not self.track_running_stats:
            return None
        with torch.no_grad():
            m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked + 1)
            self.num_batches_tracked += 1
        return (self.running_mean, self.running_var, m)

    def forward(self, x, residual=None, relu=None):
        relu = self.fused_relu if relu is None else relu
        if self.training:
            y, mean, rstd = F.batch_norm_train(x, self.weight, self.bias, self.eps, residual, relu, running=self._running())
            return y
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            # eval-mode BN is a per-channel affine map; keep it differentiable with plain device ops
            shape = (1, -1) + (1,) * (x.dim() - 2)
            scale = self.weight * torch.rsqrt(self.running_var + self.eps)
            y = x * scale.view(shape) + (self.bias - self.running_mean * scale).view(shape)
            y = y if residual is None else y + residual
            return torch.relu(y) if relu else y
        return F.batch_norm_eval(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, residual,
                                 relu)


class BatchNorm3d(_HipBatchNorm, nn.BatchNorm3d):
    pass


class BatchNorm2d(_HipBatchNorm, nn.BatchNorm2d):
    pass


def norm_pair(norm_a, xa, norm_b, xb, relu=True):
    """relu?(norm_a(xa) + norm_b(xb)) for two of the norm layers above.  Training-mode BatchNorm / GroupNorm pairs run as ONE
    operator (functional.dual_norm: the first norm's output never makes the round trip through HBM that the residual form
    ``norm_a(xa, residual=norm_b(xb))`` costs); anything else (eval-mode BatchNorm, CPU tensors) takes that residual form."""
    def fusable(n):
        return isinstance(n, GroupNorm) or (isinstance(n, _HipBatchNorm) and n.training)
    if not (F.DUAL_NORM and fusable(norm_a) and fusable(norm_b) and F.dual_norm_supported(xa, xb)):
        return norm_a(xa, residual=norm_b(xb), relu=relu)

    def side(n):
        if isinstance(n, GroupNorm):
            return n.num_groups, False
        return n.num_features, True
    (ga, a_batch), (gb, b_batch) = side(norm_a), side(norm_b)
    run = [n._running() if isinstance(n, _HipBatchNorm) else None for n in (norm_a, norm_b)]
    y, sa, sb = F.dual_norm(xa, norm_a.weight, norm_a.bias, ga, norm_a.eps, xb, norm_b.weight, norm_b.bias, gb, norm_b.eps,
                            relu=relu, a_batch=a_batch, b_batch=b_batch, running_a=run[0], running_b=run[1])
    return y


def norm_cat(norms, xs, relu=True, extra=None):
    """torch.cat([relu?(n(x)) for n, x in zip(norms, xs)] (+ [extra]), dim=1) for GroupNorm / training-mode BatchNorm layers:
    one operator whose branches write their slice of the concatenated tensor (functional.norm_cat); anything else takes the
    plain form."""
    def fusable(n):
        return isinstance(n, GroupNorm) or (isinstance(n, _HipBatchNorm) and n.training)
    if not (all(fusable(n) for n in norms) and F.norm_cat_supported(list(xs))
            and (extra is None or (extra.is_cuda and extra.shape[1] % 4 == 0))):
        ys = [n(x, relu=relu) for n, x in zip(norms, xs)]
        return torch.cat(ys + ([extra] if extra is not None else []), dim=1)
    spec = [(n.weight, n.bias, n.num_groups, n.eps, False) if isinstance(n, GroupNorm)
            else (n.weight, n.bias, n.num_features, n.eps, True) for n in norms]
    run = tuple(n._running() if isinstance(n, _HipBatchNorm) else None for n in norms)
    y, stats = F.norm_cat(list(xs), spec, relu=relu, extra=extra, running=run)
    return y


def _last_leaf(m):
    while isinstance(m, nn.Sequential) and len(m) > 0:
        m = m[len(m) - 1]
    return m


def fuse_relu_(module):
    """Walk every nn.Sequential below ``module``: a ReLU that directly follows one of our norm layers
    becomes part of that layer's kernel (the nn.ReLU slot is kept as nn.Identity, so Sequential
    indices -- i.e. state-dict keys -- do not move)."""
    for seq in [m for m in module.modules() if isinstance(m, nn.Sequential)]:
        for i in range(1, len(seq)):
            prev = _last_leaf(seq[i - 1])
            if isinstance(seq[i], nn.ReLU) and isinstance(prev, (GroupNorm, BatchNorm3d, BatchNorm2d)):
                prev.fused_relu = True
                seq[i] = nn.Identity()
    return module


# ------------------------------------------------------------------ mmcv.cnn-style builders
def build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    t = cfg.pop("type")
    requires_grad = cfg.pop("requires_grad", True)
    if t == "GN":
        name, layer = "gn", GroupNorm(cfg["num_groups"], num_features, eps=cfg.get("eps", 1e-5))
    elif t in ("BN", "BN2d"):
        name, layer = "bn", BatchNorm2d(num_features)
    elif t == "BN3d":
        name, layer = "bn", BatchNorm3d(num_features)
    else:
        raise KeyError(f"unsupported norm type {t}")
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return name + str(postfix), layer


def build_conv_layer(cfg, *args, **kwargs):
    cfg = dict(cfg) if cfg else dict(type="Conv2d")
    t = cfg.pop("type")
    kwargs = {**cfg, **kwargs}
    table = {"Conv2d": Conv2d, "Conv": Conv2d, "Conv3d": Conv3d, "DCN": DeformConv2dPack}
    if t not in table:
        raise KeyError(f"unsupported conv type {t}")
    return table[t](*args, **kwargs)


def build_upsample_layer(cfg, *args, **kwargs):
    cfg = dict(cfg)
    t = cfg.pop("type")
    kwargs = {**cfg, **kwargs}
    if t != "deconv3d":
        raise KeyError(f"unsupported upsample type {t}")
    return ConvTranspose3d(*args, **kwargs)
