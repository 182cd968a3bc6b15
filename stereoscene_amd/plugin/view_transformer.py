"""Stereo-geometry view transformer: host-side mirror of the reference's
``ViewTransformerLiftSplatShootVoxel`` (VT:273-526) with its ancestors folded in
(``ViewTransformerLSSBEVDepth`` BD:577-767, ``ViewTransformerLiftSplatShoot`` BD:70-309).

Same registry name, constructor kwargs, forward signature and state-dict keys as the reference;
the arithmetic runs on the ssbev HIP kernels:

    a2+a3  build_gwc_volume + warp        -> functional.gwc_warp (one fused kernel)
    a4/a8  conv3d / deconv3d stacks       -> MFMA implicit-GEMM kernels (layers.Conv3d, ...)
    a9+a11 lift + voxel_pooling/bev_pool  -> functional.lift_splat (CSR gather, deterministic)

VT = projects/mmdet3d_plugin/occupancy/image2bev/ViewTransformerLSSVoxel.py
BD = .../image2bev/ViewTransformerLSSBEVDepth.py, ATT = .../image2bev/attention.py
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as TF

from .. import capi
from .. import functional as F
from ..layers import (BatchNorm2d, BatchNorm3d, Conv2d, Conv3d, ConvTranspose3d, GroupNorm, build_conv_layer, build_norm_layer,
                      fuse_relu_, norm_cat, norm_pair)
from ..registry import NECKS

GN2 = dict(type="GN", num_groups=2, requires_grad=True)
# DepthNet on a second HIP stream next to the stereo branch (0: everything on the caller's stream)
VT_STREAMS = os.environ.get("SSBEV_VT_STREAMS", "1") != "0"


# ------------------------------------------------------------------------------- small blocks
class Mlp(nn.Module):
    """fc1 -> ReLU -> fc2 (BD:417-439; both dropouts are p=0)."""

    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        h = torch.relu(F.small_linear(x, self.fc1.weight, self.fc1.bias) if x.dim() == 2 else self.fc1(x))
        return F.small_linear(h, self.fc2.weight, self.fc2.bias) if h.dim() == 2 else self.fc2(h)


class SELayer(nn.Module):
    """Camera-aware squeeze-excite gate (BD:442-454).  The 1x1 convs act on a [B,C,1,1] vector."""

    def __init__(self, channels):
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, channels, 1, bias=True)
        self.conv_expand = nn.Conv2d(channels, channels, 1, bias=True)

    def forward(self, x, x_se):
        w, b = self.conv_reduce.weight.flatten(1), self.conv_reduce.bias
        s = torch.relu(F.small_linear(x_se.flatten(1), w, b))
        s = F.small_linear(s, self.conv_expand.weight.flatten(1), self.conv_expand.bias)
        gate = torch.sigmoid(s)[..., None, None]
        if x.is_cuda and x.dim() == 4 and x.shape[1] % 4 == 0:
            return F.chan_scale(x, gate)          # one streaming pass forward, gate gradient by a two-stage reduction
        return x * gate


class BasicBlock2d(nn.Module):
    """mmdet ResNet BasicBlock (third-party, BD:486-488): conv3x3-BN-ReLU-conv3x3-BN, +x, ReLU."""

    def __init__(self, inplanes, planes):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, 1, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm2d(planes)

    def forward(self, x):
        xa, xb = F.fork(x)
        y = self.bn1(self.conv1(xa), relu=True)
        return self.bn2(self.conv2(y), residual=xb, relu=True)


class _ASPPModule(nn.Module):
    def __init__(self, inplanes, planes, kernel_size, padding, dilation):
        super().__init__()
        self.atrous_conv = Conv2d(inplanes, planes, kernel_size, 1, padding, dilation, bias=False)
        self.bn = BatchNorm2d(planes)
        nn.init.kaiming_normal_(self.atrous_conv.weight)

    def forward(self, x):
        return self.bn(self.atrous_conv(x), relu=True)


class ASPP(nn.Module):
    """Atrous pyramid, dilations 1/6/12/18 + image-level branch (BD:343-414)."""

    def __init__(self, inplanes, mid_channels=256):
        super().__init__()
        self.aspp1 = _ASPPModule(inplanes, mid_channels, 1, 0, 1)
        self.aspp2 = _ASPPModule(inplanes, mid_channels, 3, 6, 6)
        self.aspp3 = _ASPPModule(inplanes, mid_channels, 3, 12, 12)
        self.aspp4 = _ASPPModule(inplanes, mid_channels, 3, 18, 18)
        self.global_avg_pool = nn.Sequential(
            nn.AdaptiveAvgPool2d((1, 1)),
            nn.Conv2d(inplanes, mid_channels, 1, stride=1, bias=False),
            build_norm_layer(GN2, mid_channels)[1],
            nn.ReLU())
        self.conv1 = Conv2d(int(mid_channels * 5), mid_channels, 1, bias=False)
        self.bn1 = BatchNorm2d(mid_channels)
        self.dropout = nn.Dropout(0.5)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, Conv2d)):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        g = x.mean(dim=(2, 3), dtype=torch.float32)                  # AdaptiveAvgPool2d((1,1)); fp32 also for bf16 activations
        g = F.small_linear(g, self.global_avg_pool[1].weight.flatten(1))   # 1x1 conv on a 1x1 map (own GEMM kernels)
        g = torch.relu(self.global_avg_pool[2](g))[..., None, None]
        g = g.expand(-1, -1, x.shape[2], x.shape[3])                 # bilinear(align_corners) of a 1x1 map
        branches = (self.aspp1, self.aspp2, self.aspp3, self.aspp4)
        # the four BatchNorm + ReLU tails write their 640-channel slices of the 3200-channel tensor directly (BD:404-410)
        y = norm_cat([m.bn for m in branches], [m.atrous_conv(x) for m in branches], relu=True, extra=g)
        return self.dropout(self.bn1(self.conv1(y), relu=True))


class DepthNet(nn.Module):
    """Camera-aware monocular depth logits || context features (BD:457-517)."""

    def __init__(self, in_channels, mid_channels, context_channels, depth_channels, cam_channels=27):
        super().__init__()
        self.reduce_conv = nn.Sequential(
            Conv2d(in_channels, mid_channels, 3, 1, 1), build_norm_layer(GN2, mid_channels)[1], nn.ReLU(inplace=True))
        self.context_conv = Conv2d(mid_channels, context_channels, 1)
        self.bn = build_norm_layer(GN2, cam_channels)[1]
        self.depth_mlp = Mlp(cam_channels, mid_channels, mid_channels)
        self.depth_se = SELayer(mid_channels)
        self.context_mlp = Mlp(cam_channels, mid_channels, mid_channels)
        self.context_se = SELayer(mid_channels)
        self.depth_conv = nn.Sequential(
            BasicBlock2d(mid_channels, mid_channels),
            BasicBlock2d(mid_channels, mid_channels),
            BasicBlock2d(mid_channels, mid_channels),
            ASPP(mid_channels, mid_channels),
            build_conv_layer(dict(type="DCN", in_channels=mid_channels, out_channels=mid_channels, kernel_size=3,
                                  padding=1, groups=4, im2col_step=128)),
            Conv2d(mid_channels, depth_channels, 1))
        fuse_relu_(self)

    def forward(self, x, mlp_input):
        cam = self.bn(mlp_input.reshape(-1, mlp_input.shape[-1]))
        with F.wino_f43_2d_scope():          # F(4,3)^2 tiles for this module's 640 -> 640 3x3 layers (functional.py)
            x = self.reduce_conv(x)
            context = self.context_conv(self.context_se(x, self.context_mlp(cam)[..., None, None]))
            depth = self.depth_conv(self.depth_se(x, self.depth_mlp(cam)[..., None, None]))
        return torch.cat([depth, context], dim=1)


# ------------------------------------------------------------------------------- stereo branch
class stereofeature_net(nn.Module):
    """Matching features of one view (VT:32-65): conv3x3+GN+ReLU, camera SE gate, conv1x1."""

    def __init__(self, in_channels, mid_channels, depth_channels, cam_channels):
        super().__init__()
        self.reduce_conv = nn.Sequential(
            Conv2d(in_channels, mid_channels, 3, 1, 1), build_norm_layer(GN2, mid_channels)[1], nn.ReLU())
        self.bn = nn.Identity()
        self.depth_mlp = Mlp(cam_channels, mid_channels, mid_channels)
        self.depth_se = SELayer(mid_channels)
        self.depth_conv = nn.Sequential(Conv2d(mid_channels, depth_channels, 1, 1, 0))
        fuse_relu_(self)

    def forward(self, x, mlp_input):
        cam = mlp_input.reshape(-1, mlp_input.shape[-1])
        x = self.reduce_conv(x)
        return self.depth_conv(self.depth_se(x, self.depth_mlp(cam)[..., None, None]))


def convbn_3d(cin, cout, kernel_size, stride, pad):
    """Bias-free conv3d + GN(2) (VT:66-69)."""
    return nn.Sequential(Conv3d(cin, cout, kernel_size, stride, pad, bias=False), build_norm_layer(GN2, cout)[1])


class hourglass(nn.Module):
    """3-D encoder/decoder with BatchNorm3d after the transposed convs (VT:70-96)."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Sequential(convbn_3d(c, c * 2, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(convbn_3d(c * 2, c * 2, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv3 = nn.Sequential(convbn_3d(c * 2, c * 4, 3, 2, 1), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn_3d(c * 4, c * 4, 3, 1, 1), nn.ReLU(inplace=True))
        self.conv5 = nn.Sequential(ConvTranspose3d(c * 4, c * 2, 3, 2, 1, 1, bias=False), BatchNorm3d(c * 2))
        self.conv6 = nn.Sequential(ConvTranspose3d(c * 2, c, 3, 2, 1, 1, bias=False), BatchNorm3d(c))
        self.redir1 = convbn_3d(c, c, 1, 1, 0)
        self.redir2 = convbn_3d(c * 2, c * 2, 1, 1, 0)
        fuse_relu_(self)

    def forward(self, x):
        # x and c2 have two consumers each (the stride-2 conv and the 1x1 redirect): F.fork makes their data gradients meet
        # in one buffer instead of in an elementwise add
        xa, xb = F.fork(x)
        c1 = self.conv1(xa)
        c2a, c2b = F.fork(self.conv2(c1))
        c4 = self.conv4(self.conv3(c2a))
        # relu(BN(deconv) + GN(1x1 conv)): both normalisations, the add and the ReLU are one operator (layers.norm_pair)
        c5 = norm_pair(self.redir2[1], self.redir2[0](c2b), self.conv5[1], self.conv5[0](c4), relu=True)
        return norm_pair(self.redir1[1], self.redir1[0](xb), self.conv6[1], self.conv6[0](c5), relu=True)


class GwcNet_volume_encoder(nn.Module):
    """Stereo cost-volume branch (VT:158-224)."""

    def __init__(self, maxdisp, out_c, warp_align_corners=True):
        super().__init__()
        self.maxdisp = maxdisp
        self.num_groups = 32
        self.warp_align_corners = warp_align_corners
        self.feature_withcam = stereofeature_net(640, 128, 64, 30)
        self.dres0 = nn.Sequential(convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True),
                                   convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True))
        self.dres1 = nn.Sequential(convbn_3d(32, 32, 3, 1, 1), nn.ReLU(inplace=True), convbn_3d(32, 32, 3, 1, 1))
        self.dres2 = hourglass(32)
        self.dres3 = hourglass(32)
        self.dres4 = hourglass(32)
        self.classif3_1 = nn.Sequential(convbn_3d(32, out_c, 3, 1, 1), nn.ReLU(inplace=True))
        self.classif3_2 = nn.Sequential(Conv3d(out_c, 1, 3, 1, 1, bias=False))
        fuse_relu_(self)
        for m in self.modules():                          # He-normal re-initialisation (VT:189-203)
            if isinstance(m, (Conv2d, Conv3d)):
                n = math.prod(m.kernel_size) * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.Linear):
                m.bias.data.zero_()

    def forward(self, features_left, features_right, mlp_input_left, mlp_input_right, calib):
        B = features_left.shape[0]
        fea = self.feature_withcam(torch.cat([features_left, features_right], 0),
                                   torch.cat([mlp_input_left, mlp_input_right], 0))
        volume = F.gwc_warp(fea[:B], fea[B:], calib, self.maxdisp, self.num_groups, self.warp_align_corners)
        c0a, c0b = F.fork(self.dres0(volume))
        t = self.dres1[2][0](self.dres1[0](c0a))
        cost0 = self.dres1[2][1](t, residual=c0b)              # dres1(cost0) + cost0, add fused into the GN pass
        out3 = self.dres4(self.dres3(self.dres2(cost0)))
        cost3_1 = self.classif3_1(out3)
        pred3 = F.softmax(self.classif3_2(cost3_1).squeeze(1), dim=1)
        return {"multi_channel": cost3_1, "single_channel": pred3}


# ------------------------------------------------------------------------------- MIE
class attention(nn.Module):
    """BRI cross attention over [B,1,D,H,W] volumes (ATT:45-86): tokens = pixels, head dim = D.
    The 1x1x1 single-channel convs are scalar affine maps."""

    def __init__(self, in_dim):
        super().__init__()
        assert in_dim == 1
        self.chanel_in = in_dim
        self.query_conv = nn.Conv3d(in_dim, in_dim, 1)
        self.key_conv = nn.Conv3d(in_dim, in_dim, 1)
        self.value_conv = nn.Conv3d(in_dim, in_dim, 1)
        self.gamma = nn.Parameter(torch.zeros(1))

    @staticmethod
    def _affine(conv, x):
        return x * conv.weight.view(()) + conv.bias.view(())

    def forward(self, q, kv):
        B, C, D, H, W = kv.shape
        hw = H * W
        conf = F.softmax(q, dim=2).amax(dim=2).view(B, hw)                # [B,HW]
        if (BRI_SHELL and kv.is_cuda and C == 1 and q.dtype == kv.dtype == torch.float32 and F.own_gemm_site("bri")
                and hw % 4 == 0 and hw <= 8192):
            qc, kc, vc = self.query_conv, self.key_conv, self.value_conv
            return _BriBlock.apply(q, kv, conf, qc.weight, qc.bias, kc.weight, kc.bias, vc.weight, vc.bias, self.gamma)
        Q = self._affine(self.query_conv, q).view(B, D, hw)                   # [B,D,HW] (tokens contiguous)
        K = self._affine(self.key_conv, kv).view(B, D, hw)
        V = self._affine(self.value_conv, kv).view(B, D, hw)
        # Six plain NN GEMMs around the materialised T x T attention matrix (236 MB per direction at T = 7680).  (Rounds 1-5 also
        # carried hand-written flash-style kernels -- no T x T matrix, ~45 TF/s, 8 ms per step slower at D = 192,
        # profiles/r1k_bri_paths.txt -- removed in round 6.)
        out = _BriCore.apply(Q, K, V, conf).view(B, C, D, H, W)
        return self.gamma * out + kv


BRI_SHELL = os.environ.get("SSBEV_BRI_SHELL", "1") != "0"   # elementwise shell of the block as four kernels (0 = ~40 ATen ops)


class _BriBlock(torch.autograd.Function):
    """One BRI block end to end for the own-GEMM realisation: ``gamma * BRI(q w_q + b_q, kv w_k + b_k, (kv w_v + b_v) conf) + kv``
    (ATT:45-86).  The six products are _BriCore's; everything around them -- three scalar-affine convolutions, the key-side
    re-weight, the residual, and all their gradients incl. the seven scalar parameter gradients -- is csrc/bri_shell.hip
    (two launches forward, two backward) instead of ~9 + ~30 tensor expressions on 5.9 MB operands."""

    @staticmethod
    def forward(ctx, q, kv, conf, wq, bq, wk, bk, wv, bv, gamma):
        lib = capi.load()
        B, C, D, H, W = kv.shape
        T = H * W
        q3, kv3 = q.contiguous().view(B, D, T), kv.contiguous().view(B, D, T)
        cf = conf.contiguous()
        ps = [t.detach().reshape(1).contiguous() for t in (wq, bq, wk, bk, wv, bv, gamma)]
        Q, K, Vc = (torch.empty_like(q3) for _ in range(3))
        capi.check(lib.ssbev_bri_shell_pre_fwd(capi.ptr(q3), capi.ptr(kv3), capi.ptr(cf), *[capi.ptr(p) for p in ps[:6]],
                                               capi.ptr(Q), capi.ptr(K), capi.ptr(Vc), B, D, T, capi.stream()), "ssbev_bri_shell_pre_fwd")
        att = F.softmax_rows_(F.gemm_tn(Q, K, tag="bri energy"))                         # [B,T(i),T(j)]
        out = F.gemm_nt(Vc, att, tag="bri out")                                         # [B,D,T(i)]
        y = torch.empty_like(kv3)
        capi.check(lib.ssbev_bri_shell_post_fwd(capi.ptr(out), capi.ptr(kv3), capi.ptr(ps[6]), capi.ptr(y), B, D, T, capi.stream()),
                   "ssbev_bri_shell_post_fwd")
        ctx.save_for_backward(q3, kv3, cf, Q, K, Vc, att, out, *ps)
        ctx.shapes = (tuple(q.shape), tuple(kv.shape), [tuple(t.shape) for t in (wq, bq, wk, bk, wv, bv, gamma)])
        return y.view(B, C, D, H, W)

    @staticmethod
    def backward(ctx, gy):
        lib = capi.load()
        q3, kv3, cf, Q, K, Vc, att, out, *ps = ctx.saved_tensors
        B, D, T = kv3.shape
        dev = gy.device
        g3 = gy.contiguous().view(B, D, T)
        nch, nblk = lib.ssbev_bri_shell_chunks(), (B * T + 255) // 256
        gout = torch.empty_like(g3)
        delta_part = torch.empty(nch, B * T, dtype=torch.float32, device=dev)
        part_g = torch.empty(nch * nblk, dtype=torch.float64, device=dev)
        capi.check(lib.ssbev_bri_shell_post_bwd(capi.ptr(g3), capi.ptr(out), capi.ptr(ps[6]), capi.ptr(gout), capi.ptr(delta_part),
                                                capi.ptr(part_g), B, D, T, capi.stream()), "ssbev_bri_shell_post_bwd")
        delta = delta_part.sum(0).view(B, T)
        # the products of _BriCore.backward (softmax backward inside the epilogue of the gatt product)
        gE = F.gemm_tn(gout, Vc, tag="bri gatt+softmax bwd", ep_mul=att, ep_rowsub=delta)
        gVc = F.gemm_nn(gout, att, tag="bri gVc")
        gQ = F.gemm_nt(K, gE, tag="bri gQ")
        gK = F.gemm_nn(Q, gE, tag="bri gK")
        gq, gkv = torch.empty_like(q3), torch.empty_like(kv3)
        gconf_part = torch.empty(nch, B * T, dtype=torch.float32, device=dev)
        part = torch.empty(nch * nblk, 6, dtype=torch.float64, device=dev)
        capi.check(lib.ssbev_bri_shell_pre_bwd(capi.ptr(gQ), capi.ptr(gK), capi.ptr(gVc), capi.ptr(q3), capi.ptr(kv3), capi.ptr(cf),
                                               capi.ptr(g3), capi.ptr(ps[0]), capi.ptr(ps[2]), capi.ptr(ps[4]), capi.ptr(ps[5]),
                                               capi.ptr(gq), capi.ptr(gkv), capi.ptr(gconf_part), capi.ptr(part), B, D, T,
                                               capi.stream()), "ssbev_bri_shell_pre_bwd")
        gconf = gconf_part.sum(0).view(B, T)
        gp = torch.cat((part.sum(0), part_g.sum().reshape(1))).to(torch.float32)        # d(wq, bq, wk, bk, wv, bv, gamma)
        qs, ks, pshapes = ctx.shapes
        return (gq.view(qs), gkv.view(ks), gconf, *[gp[i].view(sh) for i, sh in enumerate(pshapes)])


class _BriCore(torch.autograd.Function):
    """out[b,:,i] = sum_j softmax_j(Q[:,i].K[:,j]) * conf[j] * V[:,j]   (ATT:72-81) for [B,D,T] operands.

    The six products run on the path's own MFMA GEMM kernels (csrc/gemm.hip) in the form each operand already has in
    memory -- [D, T] with the token axis contiguous -- so nothing is transposed or copied:
      energy = Q^T K      TN (reduction over the D rows)        out  = Vc att^T   NT
      gatt   = go^T Vc    TN                                    gVc  = go att     NN
      gQ     = K gE^T     NT                                    gK   = Q gE       NN
    """

    @staticmethod
    def forward(ctx, Q, K, V, conf):
        own = F.own_gemm_site("bri")
        Vc = V * conf.unsqueeze(1)                                                      # key-side re-weight
        if own:
            att = F.gemm_tn(Q, K, tag="bri energy")                                     # [B,T(i),T(j)]
            att = F.softmax_rows_(att) if F.softmax_rows_ok(att) else torch.softmax(att, dim=-1)
            out = F.gemm_nt(Vc, att, tag="bri out")                                     # [B,D,T(i)]
        else:       # library realisation: plain NN products around explicit operand transposes
            att = torch.softmax(torch.bmm(Q.transpose(1, 2).contiguous(), K), dim=-1)
            out = torch.bmm(att, Vc.transpose(1, 2).contiguous()).transpose(1, 2).contiguous()
        ctx.save_for_backward(Q, K, V, conf, att, out)
        ctx.own = own
        return out

    @staticmethod
    def backward(ctx, gout):
        Q, K, V, conf, att, out = ctx.saved_tensors
        Vc = V * conf.unsqueeze(1)
        gout = gout.contiguous()
        if ctx.own:
            # softmax backward gE = att (.) (gatt - rowsum(gatt (.) att)) with gatt = go^T Vc.  The row sums need no pass over
            # the T x T matrices: sum_j gatt[i,j] att[i,j] = <go[:, i], sum_j att[i,j] Vc[:, j]> = <go[:, i], out[:, i]>; the
            # rest is the epilogue of the product that forms gatt, which therefore never exists in memory
            delta = (gout * out).sum(1)                                                 # [B,T(i)]
            gE = F.gemm_tn(gout, Vc, tag="bri gatt+softmax bwd", ep_mul=att, ep_rowsub=delta)
            gVc = F.gemm_nn(gout, att, tag="bri gVc")                                   # [B,D,T(j)]
        else:
            gatt = torch.bmm(gout.transpose(1, 2).contiguous(), Vc)
            gVc = torch.bmm(gout, att)
            gE = att * (gatt - (gatt * att).sum(-1, keepdim=True))
        if ctx.own:
            gQ = F.gemm_nt(K, gE, tag="bri gQ")                                         # [B,D,T(i)]
            gK = F.gemm_nn(Q, gE, tag="bri gK")                                         # [B,D,T(j)]
        else:
            gQ = torch.bmm(gE, K.transpose(1, 2).contiguous()).transpose(1, 2)
            gK = torch.bmm(Q, gE)
        return gQ, gK, gVc * conf.unsqueeze(1), (gVc * V).sum(1)


class CA3D(nn.Module):
    """DVE channel recalibration (ATT:90-120)."""

    def __init__(self, channel):
        super().__init__()
        self.conv1 = nn.Sequential(Conv3d(channel, channel, 3, 1, 1), nn.GELU(), GroupNorm(1, channel))
        self.conv2 = nn.Sequential(nn.Conv3d(channel, channel // 8, 1), nn.GELU(),
                                   nn.Conv3d(channel // 8, channel, 1), nn.GELU())
        self.conv = nn.Sequential(Conv3d(channel, channel, 3, 1, 1), nn.GELU(), GroupNorm(1, channel))

    def forward(self, x, residual=None, alpha=None):
        """ATT:113-120.  Three algebraic rearrangements keep the 189 MB activations out of elementwise passes: GELU is the
        pre-activation of the GroupNorm kernel; the per-channel gate sigmoid(s) multiplies the INPUT channels of the last
        conv, i.e. its weight columns (B == 1); ``alpha * CA3D(x) + x`` of the Residual wrapper (VT:236-268) is GroupNorm
        with affine (alpha gamma, alpha beta) and a fused residual add."""
        c1, gn1, c2, gn2 = self.conv1[0], self.conv1[2], self.conv[0], self.conv[2]
        if not x.is_cuda:
            data = self.conv1(x)
        else:
            data = F.group_norm(c1(x), gn1.num_groups, gn1.weight, gn1.bias, gn1.eps, pre_act="gelu")
        if x.is_cuda:
            dpool, data = F.fork(data)          # squeeze and the last conv: their data gradients meet in one buffer
            pool = F.spatial_mean(dpool)
        else:
            pool = data.mean(dim=(2, 3, 4))
        s = TF.gelu(F.small_linear(pool, self.conv2[0].weight.flatten(1), self.conv2[0].bias))
        s = TF.gelu(F.small_linear(s, self.conv2[2].weight.flatten(1), self.conv2[2].bias))
        gate = torch.sigmoid(s)
        if not x.is_cuda:
            y = self.conv(gate[..., None, None, None] * data)
            return y if residual is None else alpha * y + residual
        if data.shape[0] == 1:
            y = F.conv3d(data, c2.weight * gate.view(1, -1, 1, 1, 1), c2.bias, c2.stride, c2.padding, c2.dilation)
        else:
            y = c2(gate.to(data.dtype)[..., None, None, None] * data)      # (bf16 storage: the gate joins the chain's dtype, no fp32 pass)
        gamma, beta = (gn2.weight, gn2.bias) if alpha is None else (alpha * gn2.weight, alpha * gn2.bias)
        return F.group_norm(y, gn2.num_groups, gamma, beta, gn2.eps, residual=residual, pre_act="gelu")


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn
        self.alpha = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        if isinstance(self.fn, CA3D):
            xa, xb = F.fork(x)
            return self.fn(xa, residual=xb, alpha=self.alpha)
        return self.alpha * self.fn(x) + x


class volume_interaction(nn.Module):
    """Mutual Interactive Ensemble: BRI both ways + DVE (VT:236-268)."""

    def __init__(self, out_channels=1):
        super().__init__()
        self.redir1 = Conv3d(2, 32, 3, 1, 1)
        self.dres1 = hourglass(32)
        self.redir2 = Conv3d(32, out_channels, 3, 1, 1)
        self.lss2stereo = attention(in_dim=1)
        self.stereo2lss = attention(in_dim=1)
        self.CA3D = Residual(CA3D(32))

    def forward(self, stereo_volume, lss_volume):
        s, m = stereo_volume.unsqueeze(1), lss_volume.unsqueeze(1)
        a = self.lss2stereo(q=s, kv=m)
        b = self.stereo2lss(q=m, kv=s)
        x = self.redir1(torch.cat((a, b), dim=1), relu=True)       # ReLU in the kernels' epilogue
        x = self.CA3D(self.dres1(x))
        x = self.redir2(x, relu=True).squeeze(1)
        return F.softmax(x, dim=1)


# ------------------------------------------------------------------------------- the transformer
def attach_host_inverses(post_rots, intrins, post_rots_host=None, intrins_host=None):
    """Data-layer hook: attach inverse(post_rots) and inverse(intrins[..., :3, :3]), computed on the host in fp32 exactly as
    ``get_geometry`` would (``*_host`` = CPU copies of the same matrices, e.g. the batch before it was moved to the GPU), to
    the device tensors, so that the forward pass does not have to read the matrices back."""
    for dev_t, host_t in ((post_rots, post_rots_host), (intrins, intrins_host)):
        h = (host_t if host_t is not None else dev_t.cpu()).float()
        # (version, inverse): an in-place edit of the matrix after this call (test-time augmentation, a reused staging
        # buffer) bumps tensor._version and the stale hint is ignored by get_geometry
        dev_t._ssbev_inverse = (dev_t._version, torch.inverse(h[..., :3, :3]).to(dev_t.device, non_blocking=True))
    return post_rots, intrins


def gen_dx_bx(xbound, ybound, zbound):
    rows = (xbound, ybound, zbound)
    dx = torch.tensor([r[2] for r in rows], dtype=torch.float64).float()
    bx = torch.tensor([r[0] + r[2] / 2.0 for r in rows], dtype=torch.float64).float()
    nx = torch.tensor([(r[1] - r[0]) / r[2] for r in rows], dtype=torch.float64).float()
    return dx, bx, nx


@NECKS.register_module()
class ViewTransformerLiftSplatShootVoxel(nn.Module):
    """Registry-compatible stand-in for the reference class of the same name (VT:273-526).

    Extra kwarg ``warp_align_corners`` selects the grid_sample convention of ``warp`` (VT:151-154):
    True = literal behaviour on torch >= 1.3, False = what torch 1.10.1 (README pin) executed.

    Extra kwarg / attribute ``ablation`` (BASELINE configs[4]; the reference has no such switch, its forward is "full"):
    "full" = stereo volume and monocular depth fused by the MIE block (VT:497-505); "bev_only" = the monocular DepthNet
    distribution lifts the features (cost volume and MIE skipped); "stereo_only" = the softmax of the stereo cost volume
    lifts them (MIE skipped).  All sub-modules are built in every mode, so state dicts stay interchangeable.
    """
    ABLATIONS = ("full", "bev_only", "stereo_only")

    def __init__(self, loss_depth_weight, semkitti=False, imgseg=False, imgseg_class=20, lift_with_imgseg=False,
                 point_cloud_range=None, loss_seg_weight=1.0, loss_depth_type="bce", point_xyz_channel=0,
                 point_xyz_mode="cat", cam_channels=27, loss_depth_reg_weight=0.0, use_voxel_net=False,
                 grid_config=None, data_config=None, numC_input=512, numC_Trans=64, downsample=16,
                 accelerate=False, use_bev_pool=True, vp_megvii=False, vp_stero=False,
                 warp_align_corners=True, ablation="full", **kwargs):
        super().__init__()
        if ablation not in self.ABLATIONS:
            raise ValueError(f"ablation must be one of {self.ABLATIONS}, got {ablation!r}")
        self.ablation = ablation
        if imgseg or point_xyz_channel or use_voxel_net or vp_megvii:
            raise NotImplementedError("options unused by projects/configs/.../stereoscene.py are not built")
        self.grid_config, self.data_config = grid_config, data_config
        dx, bx, nx = gen_dx_bx(grid_config["xbound"], grid_config["ybound"], grid_config["zbound"])
        self.dx = nn.Parameter(dx, requires_grad=False)
        self.bx = nn.Parameter(bx, requires_grad=False)
        self.nx = nn.Parameter(nx, requires_grad=False)
        self.downsample = downsample
        self.frustum = self.create_frustum()
        self.D = self.frustum.shape[0]
        self.numC_input, self.numC_Trans, self.cam_channels = numC_input, numC_Trans, cam_channels
        self.loss_depth_weight, self.loss_depth_type = loss_depth_weight, loss_depth_type
        self.imgseg, self.semkitti = False, semkitti
        self.depth_net = DepthNet(numC_input, numC_input, numC_Trans, self.D, cam_channels=cam_channels)
        self.stereo_volume_net = GwcNet_volume_encoder(self.D, 32, warp_align_corners)
        self.volume_interaction = volume_interaction()
        self.cam_depth_range = grid_config["dbound"]

    # geometry -------------------------------------------------------------------------------
    def create_frustum(self):
        H_img, W_img = self.data_config["input_size"]
        fH, fW = H_img // self.downsample, W_img // self.downsample
        ds = torch.arange(*self.grid_config["dbound"], dtype=torch.float)
        fr = torch.empty(ds.shape[0], fH, fW, 3)
        fr[..., 0] = torch.linspace(0, W_img - 1, fW, dtype=torch.float).view(1, 1, fW)
        fr[..., 1] = torch.linspace(0, H_img - 1, fH, dtype=torch.float).view(1, fH, 1)
        fr[..., 2] = ds.view(-1, 1, 1)
        return nn.Parameter(fr, requires_grad=False)

    @staticmethod
    def _apply3x3(m, p):
        """(m @ p) for m [B,N,3,3] (or [B,1,3,3]) and points p [B,N,D,H,W,3] as three broadcast
        multiply-adds: a batched 3x3 BLAS call over 1.5 M points costs ~19 ms, this costs ~0.1 ms."""
        m = m.unsqueeze(2).unsqueeze(2).unsqueeze(2)                      # [B,N,1,1,1,3,3]
        x, y, z = p[..., 0:1], p[..., 1:2], p[..., 2:3]
        return m[..., 0] * x + m[..., 1] * y + m[..., 2] * z

    def get_geometry(self, rots, trans, intrins, post_rots, post_trans, bda):
        """Frustum -> ego frame (BD:123-156).  3x3 inverses are taken on the host in fp32 (the oracle's arithmetic).  For
        device-resident matrices that read-back is a stream synchronisation at the top of every step; the data layer
        therefore attaches the host-computed inverses to the tensors it hands over (``attach_host_inverses``), and they
        are used when present."""
        B, N, _ = trans.shape

        def hint_of(t):
            h = getattr(t, "_ssbev_inverse", None)
            return h[1] if h is not None and h[0] == t._version and h[1].device == t.device else None

        def inv(m, hint):
            if hint is not None:
                return hint
            return torch.inverse(m[..., :3, :3].float().cpu()).to(m.device)      # same 3x3 block the hint inverts

        intr_hint = hint_of(intrins)
        if (F.GEOM_FUSED and trans.is_cuda and self.frustum.is_cuda and all(
                t.dtype == torch.float32 for t in (rots, trans, intrins, post_rots, post_trans, bda))):
            # the per-point chain below as ONE kernel (same separately rounded products and sums in the same order: the
            # points and the voxel indices are bit-identical); only the 3x3 matrix algebra stays in ATen
            m1 = inv(post_rots, hint_of(post_rots))
            t2 = intrins[:, :, :3, 3] if intrins.shape[3] == 4 else None
            m2 = rots.matmul(inv(intrins[:, :, :3, :3] if intrins.shape[3] == 4 else intrins, intr_hint))
            m3, t3 = (bda[:, :3, :3], bda[:, :3, 3]) if bda.shape[-1] == 4 else (bda, None)
            return F.frustum_geometry(self.frustum, m1[..., :3, :3], post_trans, m2, t2, trans, m3, t3)
        pts = self.frustum - post_trans.view(B, N, 1, 1, 1, 3)
        pts = self._apply3x3(inv(post_rots, hint_of(post_rots)), pts)
        pts = torch.cat((pts[..., :2] * pts[..., 2:3], pts[..., 2:3]), -1)
        if intrins.shape[3] == 4:
            pts = pts - intrins[:, :, :3, 3].view(B, N, 1, 1, 1, 3)
            intrins = intrins[:, :, :3, :3]
        pts = self._apply3x3(rots.matmul(inv(intrins, intr_hint)), pts)
        pts = pts + trans.view(B, N, 1, 1, 1, 3)
        if bda.shape[-1] == 4:
            pts = self._apply3x3(bda[:, None, :3, :3], pts) + bda[:, :3, 3].view(B, 1, 1, 1, 1, 3)
        else:
            pts = self._apply3x3(bda[:, None], pts)
        return pts

    def _cached_tables(self, *calib):
        """Frustum geometry -> voxel ids -> CSR of this calibration.  With ``geometry_cache = True`` (opt-in attribute; the
        reference recomputes per forward, BD:123-156 + VT:432-476) the tables of the previous call are reused when the SAME
        six calibration tensors come in again unmodified (identity + version counter; the cache holds references, so their
        storage cannot be recycled under it) and the grid parameters are unchanged -- SemanticKITTI calibration is constant
        per sequence.  Off by default: bench.py's headline always recomputes."""
        grid = self._grid_host()
        if getattr(self, "geometry_cache", False):
            c = getattr(self, "_geo_cache", None)
            if c is not None and c[2] is grid and len(c[0]) == len(calib) and \
                    all(a is b and a._version == v for a, (b, v) in zip(calib, c[0])):
                return c[1]
        geom = self.get_geometry(*calib)
        tables = F.lift_splat_tables(geom, self.bx, self.dx, self.nx, grid_host=grid)
        if getattr(self, "geometry_cache", False):
            self._geo_cache = ([(t, t._version) for t in calib], tables, grid)
        return tables

    def _side_stream(self, x):
        from .. import streams
        return streams.side_stream(x.device)[1]      # the one side stream of the device (shared with the weight gradients)

    def _grid_host(self):
        """Host copies of the (constant) voxel-grid parameters: read back once, not once per step (each read-back of a
        device-resident nn.Parameter is a stream synchronisation in the middle of the forward pass)."""
        key = tuple((p.data_ptr(), p._version) for p in (self.bx, self.dx, self.nx))
        if getattr(self, "_grid_host_key", None) != key:
            self._grid_host_val = (F.grid_origin(self.bx, self.dx).tolist(), self.dx.detach().float().cpu().tolist(),
                                   [int(v) for v in self.nx.tolist()])
            self._grid_host_key = key
        return self._grid_host_val

    def get_mlp_input(self, rot, tran, intrin, post_rot, post_tran, bda=None):
        """30-vector of camera parameters for the SE gates (BD:604-659)."""
        B, N = rot.shape[:2]
        if bda is None:
            bda = torch.eye(3).to(rot).view(1, 3, 3).repeat(B, 1, 1)
        bda = bda.view(B, 1, *bda.shape[-2:]).repeat(1, N, 1, 1)
        kitti = intrin.shape[-1] == 4
        parts = [intrin[:, :, 0, 0], intrin[:, :, 1, 1], intrin[:, :, 0, 2], intrin[:, :, 1, 2]]
        if kitti:
            parts += [intrin[:, :, 0, 3], intrin[:, :, 1, 3], intrin[:, :, 2, 3]]
        parts += [post_rot[:, :, 0, 0], post_rot[:, :, 0, 1], post_tran[:, :, 0],
                  post_rot[:, :, 1, 0], post_rot[:, :, 1, 1], post_tran[:, :, 1],
                  bda[:, :, 0, 0], bda[:, :, 0, 1], bda[:, :, 1, 0], bda[:, :, 1, 1], bda[:, :, 2, 2]]
        v = torch.stack(parts, dim=-1)
        if kitti and bda.shape[-1] == 4:
            v = torch.cat((v, bda[:, :, :3, -1]), dim=2)
        sensor2ego = torch.cat([rot, tran.reshape(B, N, 3, 1)], dim=-1).reshape(B, N, -1)
        return torch.cat([v, sensor2ego], dim=-1)

    def get_depth_dist(self, x):
        # x is the first D channels of DepthNet's channels-last output: a standard-contiguous copy (5.9 MB) puts the softmax on the
        # strided-axis HIP kernel (17 us) instead of ATen's generic path on the sliced view (138 us forward, 112 us backward)
        return F.softmax(x.contiguous(), dim=1)

    # splat ------------------------------------------------------------------------------------
    def voxel_pooling(self, geom_feats, x):
        """Reference-shaped entry (VT:432-476): x [B,N,D,H,W,C] materialised -> [B,C,X,Y,Z] via the
        bev_pool drop-in.  ``forward`` uses the fused functional.lift_splat instead."""
        B, N, D, H, W, C = x.shape
        vox, idx3 = F.voxel_index(geom_feats, self.bx, self.dx, self.nx, return_idx=True)
        kept = vox >= 0
        bcol = torch.arange(B, device=x.device, dtype=torch.int32).repeat_interleave(N * D * H * W)[:, None]
        coords = torch.cat((idx3, bcol), 1)[kept]
        n = [int(v) for v in self.nx.tolist()]
        final = F.bev_pool(x.reshape(-1, C)[kept], coords, B, n[2], n[0], n[1])
        return final.permute(0, 1, 3, 4, 2)

    # losses -----------------------------------------------------------------------------------
    def get_downsampled_gt_depth(self, gt_depths):
        B, N, H, W = gt_depths.shape
        ds = self.downsample
        g = gt_depths.view(B * N, H // ds, ds, W // ds, ds).permute(0, 1, 3, 2, 4).reshape(-1, ds * ds)
        g = torch.where(g == 0.0, torch.full_like(g, 1e5), g).min(dim=-1).values
        db = self.grid_config["dbound"]
        g = (g - (db[0] - db[2] / 2)) / db[2]
        vals = g.clone()
        g = torch.where((g < self.D + 1) & (g >= 0.0), g, torch.zeros_like(g))
        return vals, TF.one_hot(g.long(), num_classes=self.D + 1)[:, 1:].float()

    def get_depth_loss(self, depth_labels, depth_preds):
        """BCE depth loss (VT:375-388,405-416)."""
        if self.loss_depth_type != "bce":
            raise NotImplementedError(self.loss_depth_type)
        if (F.DEPTH_BCE and depth_preds.is_cuda and depth_preds.dtype == torch.float32 and depth_labels.dtype == torch.float32
                and depth_preds.shape[1] == self.D):
            return F.depth_bce_loss(depth_labels, depth_preds, self.downsample, self.grid_config["dbound"], self.loss_depth_weight)
        _, labels = self.get_downsampled_gt_depth(depth_labels)
        preds = depth_preds.float().permute(0, 2, 3, 1).reshape(-1, self.D)
        fg = labels.max(dim=1).values > 0.0
        # rows without a LiDAR return are masked out instead of being gathered away (VT:411-414 index with the boolean mask,
        # which is a nonzero() + host synchronisation in the middle of the step); the background rows' labels are all zero,
        # their predictions are replaced by zero, so their BCE terms vanish exactly
        m = fg.unsqueeze(1)
        bce = TF.binary_cross_entropy(torch.where(m, preds, torch.zeros_like(preds)), labels, reduction="none")
        loss = bce.sum() / torch.clamp(fg.sum(), min=1.0)
        return self.loss_depth_weight * loss

    # forward ----------------------------------------------------------------------------------
    def forward(self, input):
        x, rots, trans, intrins, post_rots, post_trans, bda, mlp_input = input[:8]
        feature_right, mlp_input_right = input[8], input[15]
        calib = input[16]
        # geometry first: its tiny host-side 3x3 inverses must not stall the queued device work
        tables = self._cached_tables(rots, trans, intrins, post_rots, post_trans, bda)
        B, N, C, H, W = x.shape
        # The monocular branch (DepthNet: 2-D layers on the 48 x 160 map, kernels of a few hundred workgroups that cannot
        # fill 256 CUs) and the stereo branch (3-D stack on the cost volume: chip-filling kernels with idle tails) are
        # independent until the MIE block.  DepthNet runs on a second HIP stream; its backward follows on that stream by
        # itself (autograd replays every node on its forward stream and orders the streams with events).
        side = self._side_stream(x) if (VT_STREAMS and self.ablation != "bev_only" and x.is_cuda) else None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                y = self.depth_net(x.view(B * N, C, H, W), mlp_input)
                depth_prob = self.get_depth_dist(y[:, :self.D])
        stereo = None
        if self.ablation != "bev_only":
            stereo = self.stereo_volume_net(x.squeeze(1), feature_right.squeeze(1), mlp_input, mlp_input_right,
                                            calib)["single_channel"]
        if side is not None:
            main.wait_stream(side)
            for t in (y, depth_prob):            # allocated on the side stream, consumed on this one from here on
                t.record_stream(main)
        else:
            y = self.depth_net(x.view(B * N, C, H, W), mlp_input)
            depth_prob = self.get_depth_dist(y[:, :self.D])
        img_feat = y[:, self.D:self.D + self.numC_Trans]
        if self.ablation == "full":
            depth_prob = self.volume_interaction(stereo, depth_prob)
        elif self.ablation == "stereo_only":
            depth_prob = stereo
        bev_feat = F.lift_splat(depth_prob, img_feat, None, self.bx, self.dx, self.nx, grid_host=self._grid_host(), tables=tables)
        return bev_feat, depth_prob
