"""Dense 3-D encoder, neck and occupancy head over the voxel grid: host-side mirrors of
``CustomResNet3D`` (resnet3d.py:106-246), ``SECONDFPN3D`` (second_fpn_3d.py:14-117) and
``OccHead`` (occhead.py:28-426) with identical registry names, kwargs and state-dict keys.
All convolutions run on the MFMA implicit-GEMM kernels (stereoscene_amd.layers)."""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as F
from ..layers import Conv3d, GroupNorm, build_conv_layer, build_norm_layer, build_upsample_layer, fuse_relu_, norm_pair
from ..registry import BACKBONES, HEADS, NECKS
from . import losses as L


class BasicBlock3d(nn.Module):
    """conv3-GN-ReLU-conv3-GN, projected shortcut when the shape changes, ReLU (resnet3d.py:35-65)."""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, downsample=None, norm_cfg=None):
        super().__init__()
        self.conv1 = Conv3d(in_planes, planes, 3, stride, 1, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.conv2 = Conv3d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        xa, xb = F.fork(x)                                   # two consumers: conv1 and the shortcut
        out = self.bn1(self.conv1(xa), relu=True)
        if self.downsample is None:
            return self.bn2(self.conv2(out), residual=xb, relu=True)   # relu(GN(conv2) + shortcut) in one pass
        # projected shortcut: relu(GN(conv2(out)) + GN(conv1x1(x))) as one two-norm operator
        return norm_pair(self.bn2, self.conv2(out), self.downsample[1], self.downsample[0](xb), relu=True)


@BACKBONES.register_module()
class CustomResNet3D(nn.Module):
    def __init__(self, depth, block_inplanes=(64, 128, 256, 512), block_strides=(1, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 num_stage=4, n_input_channels=3, shortcut_type="B", norm_cfg=dict(type="BN3d", requires_grad=True),
                 crp3d=False, crp_level=2, widen_factor=1.0):
        super().__init__()
        if depth not in (10, 18, 34) or crp3d or shortcut_type != "B":
            raise NotImplementedError("only the BasicBlock / shortcut-B / no-CRP variants used by the config are built")
        nblocks = {10: [1, 1, 1, 1], 18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}[depth]
        planes = [int(c * widen_factor) for c in block_inplanes]
        self.in_planes = planes[0]
        self.out_indices, self.num_stage, self.crp3d = out_indices, num_stage, False
        self.input_proj = nn.Sequential(Conv3d(n_input_channels, self.in_planes, 1, 1, 0, bias=False),
                                        build_norm_layer(norm_cfg, self.in_planes)[1], nn.ReLU(inplace=True))
        self.layers = nn.ModuleList()
        for i, c in enumerate(planes[:num_stage]):
            self.layers.append(self._make_layer(c, nblocks[i], block_strides[i], norm_cfg))
        fuse_relu_(self)
        for m in self.modules():
            if isinstance(m, Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride, norm_cfg):
        down = None
        if stride != 1 or self.in_planes != planes:
            down = nn.Sequential(Conv3d(self.in_planes, planes, 1, stride, 0, bias=False),
                                 build_norm_layer(norm_cfg, planes)[1])
        layers = [BasicBlock3d(self.in_planes, planes, stride, down, norm_cfg)]
        self.in_planes = planes
        layers += [BasicBlock3d(planes, planes, norm_cfg=norm_cfg) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.input_proj(x)
        res = []
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i in self.out_indices:
                if i + 1 < len(self.layers):
                    x, out = F.fork(x)                       # next stage + neck
                    res.append(out)
                else:
                    res.append(x)
        return res


@NECKS.register_module()
class SECONDFPN3D(nn.Module):
    def __init__(self, in_channels=(128, 128, 256), out_channels=(256, 256, 256), upsample_strides=(1, 2, 4),
                 norm_cfg=dict(type="GN", num_groups=32, requires_grad=True),
                 upsample_cfg=dict(type="deconv3d", bias=False), conv_cfg=dict(type="Conv3d", bias=False),
                 use_conv_for_no_stride=False, use_output_upsample=False, with_cp=False, init_cfg=None):
        super().__init__()
        assert len(out_channels) == len(upsample_strides) == len(in_channels) and not use_output_upsample
        self.in_channels, self.out_channels = list(in_channels), list(out_channels)
        blocks = []
        for cin, cout, s in zip(in_channels, out_channels, upsample_strides):
            if s >= 1 and not (s == 1 and use_conv_for_no_stride):
                up = build_upsample_layer(upsample_cfg, in_channels=cin, out_channels=cout, kernel_size=s, stride=s)
            else:
                k = int(np.round(1 / s))
                up = build_conv_layer(conv_cfg, in_channels=cin, out_channels=cout, kernel_size=k, stride=k)
            blocks.append(nn.Sequential(up, build_norm_layer(norm_cfg, cout)[1], nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(blocks)
        fuse_relu_(self)

    def forward(self, x):
        assert len(x) == len(self.in_channels)
        norms = [blk[1] for blk in self.deblocks]
        if (len(x) > 1 and all(isinstance(n, GroupNorm) and n.fused_relu for n in norms)
                and all(isinstance(blk[2], nn.Identity) for blk in self.deblocks)):
            raw = [blk[0](f) for blk, f in zip(self.deblocks, x)]
            if F.norm_cat_supported(raw):
                # every branch's GroupNorm + ReLU writes its channel slice of the concatenated tensor (FPN:113-116 without
                # the torch.cat pass and, backward, without the three slice copies)
                return [F.norm_cat(raw, [(n.weight, n.bias, n.num_groups, n.eps, False) for n in norms], relu=True)[0]]
            ups = [blk[2](blk[1](r)) for blk, r in zip(self.deblocks, raw)]
        else:
            ups = [blk(f) for blk, f in zip(self.deblocks, x)]
        return [torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]]


@HEADS.register_module()
class OccHead(nn.Module):
    def __init__(self, in_channels, out_channel, out_point_channel=None, semantic_kitti=False, supervise_voxel=True,
                 num_level=1, num_img_level=1, in_img_channels=512, sampling_img_feats=False, soft_weights=False,
                 supervise_points=False, loss_weight_cfg=None, semkitti_loss_weight_cfg=None,
                 loss_voxel_prototype="cylinder3d", use_ohem_loss=False, use_sc_ohem_loss=False, ohem_topk=0.25,
                 conv_cfg=dict(type="Conv3d", bias=False), norm_cfg=dict(type="GN", num_groups=32, requires_grad=True),
                 point_cloud_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0), with_cp=False, train_cfg=None, test_cfg=None):
        super().__init__()
        if supervise_points or not semantic_kitti or not supervise_voxel:
            raise NotImplementedError("only the voxel-supervised SemanticKITTI head used by the config is built")
        self.in_channels = list(in_channels) if isinstance(in_channels, (list, tuple)) else [in_channels]
        self.out_channel, self.num_level = out_channel, num_level
        self.semantic_kitti, self.supervise_voxel, self.supervise_points = True, True, False
        self.point_cloud_range = torch.tensor(np.array(point_cloud_range))
        self.occ_convs = nn.ModuleList()
        for i in range(num_level):
            mid = self.in_channels[i] // 2
            self.occ_convs.append(nn.Sequential(
                build_conv_layer(conv_cfg, in_channels=self.in_channels[i], out_channels=mid, kernel_size=3, stride=1,
                                 padding=1),
                build_norm_layer(norm_cfg, mid)[1], nn.ReLU(inplace=True),
                build_conv_layer(conv_cfg, in_channels=mid, out_channels=out_channel, kernel_size=1, stride=1,
                                 padding=0)))
        fuse_relu_(self)
        self.class_names = L.KITTI_CLASS_NAMES
        assert out_channel == len(self.class_names)
        # non-persistent buffer: follows the module to the device (a per-step .to(device) of a host tensor is a blocking copy =
        # a stream synchronisation right before the losses) without adding a state-dict key the reference does not have
        self.register_buffer("class_weights", L.semkitti_class_weights(), persistent=False)
        self.semkitti_loss_weight_cfg = semkitti_loss_weight_cfg or {}
        for k in ("voxel_ohem", "voxel_lovasz", "frustum_dist", "voxel_dice", "voxel_lga"):
            if self.semkitti_loss_weight_cfg.get(k, 0.0) > 0:
                raise NotImplementedError(f"loss '{k}' is disabled in the reference config and not built")

    def forward_voxel(self, voxel_feats):
        return [conv(f) for f, conv in zip(voxel_feats, self.occ_convs)]

    def forward(self, voxel_feats, points=None, img_metas=None, img_feats=None, points_uv=None, **kwargs):
        assert type(voxel_feats) is list and len(voxel_feats) == self.num_level
        return {"output_voxels": self.forward_voxel(voxel_feats), "output_points": None}

    def loss_voxel_single_semkitti(self, output_voxels, target_voxels, tag, compute_metric=False, **kwargs):
        w = self.semkitti_loss_weight_cfg
        return L.occ_losses(output_voxels, target_voxels, self.class_weights.to(output_voxels), tag,
                            w.get("voxel_ce", 0.0), w.get("voxel_sem_scal", 0.0), w.get("voxel_geo_scal", 0.0),
                            compute_metric)

    def loss(self, output_voxels=None, target_voxels=None, output_points=None, target_points=None, img_metas=None,
             **kwargs):
        targets = target_voxels[:self.num_level] if type(target_voxels) is list else [target_voxels] * self.num_level
        out = {}
        for i, o in enumerate(output_voxels):
            out.update(self.loss_voxel_single_semkitti(o, targets[i], tag=str(i), compute_metric=(i == 0)))
        return out
