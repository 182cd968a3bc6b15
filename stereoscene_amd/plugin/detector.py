"""``BEVDepthOccupancy`` orchestration (bevdepth_occupancy.py:23-297) from the image-neck output
onwards.  The 2-D image backbone/neck (EfficientNet-B7 + SECONDFPN) is outside the hot path
(SURVEY 8(f1)): when ``img_backbone`` is absent from the registry the detector expects the
image-neck feature maps in place of raw images (``img_inputs[k][0]`` = [B,1,640,fH,fW])."""
import torch
import torch.nn as nn
import torch.nn.functional as TF

from ..registry import BACKBONES, DETECTORS, HEADS, NECKS


@DETECTORS.register_module()
class BEVDepthOccupancy(nn.Module):
    def __init__(self, img_backbone=None, img_neck=None, img_view_transformer=None, img_bev_encoder_backbone=None,
                 img_bev_encoder_neck=None, pts_bbox_head=None, loss_cfg=None, use_grid_mask=False,
                 disable_loss_depth=False, train_cfg=None, test_cfg=None, **kwargs):
        super().__init__()
        self.img_backbone = BACKBONES.build(img_backbone) if img_backbone and img_backbone["type"] in BACKBONES else None
        self.img_neck = NECKS.build(img_neck) if img_neck and img_neck["type"] in NECKS else None
        self.img_view_transformer = NECKS.build(img_view_transformer)
        self.img_bev_encoder_backbone = BACKBONES.build(img_bev_encoder_backbone)
        self.img_bev_encoder_neck = NECKS.build(img_bev_encoder_neck)
        self.pts_bbox_head = HEADS.build(pts_bbox_head)
        self.disable_loss_depth = disable_loss_depth
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    def image_encoder(self, img):
        if self.img_backbone is None:
            return img                      # already image-neck features [B,N,C,fH,fW]
        B, N, C, H, W = img.shape
        x = self.img_backbone(img.view(B * N, C, H, W))
        if self.img_neck is not None:
            x = self.img_neck(x)
            x = x[0] if isinstance(x, (list, tuple)) else x
        return x.view(B, N, *x.shape[1:])

    def bev_encoder(self, x):
        return self.img_bev_encoder_neck(self.img_bev_encoder_backbone(x.float()))

    def extract_img_feat(self, img, img_metas=None):
        left, right = img[0], img[1]
        B = left[0].shape[0]
        feats = self.image_encoder(torch.cat([left[0], right[0]], 0))      # both views in one pass (DET:94)
        x, x2 = feats[:B], feats[B:]
        geo = list(left[1:7])
        geo2 = list(right[1:7])
        vt = self.img_view_transformer
        mlp = vt.get_mlp_input(*geo)
        mlp2 = vt.get_mlp_input(*geo2)
        calib = left[-1]
        bev, depth = vt([x] + geo + [mlp] + [x2] + geo2 + [mlp2] + [calib] + [left, right])   # 19-list, DET:112-116
        out = self.bev_encoder(bev)
        return (out if isinstance(out, list) else [out]), depth, x

    def extract_feat(self, points, img, img_metas=None):
        voxel_feats, depth, img_feats = self.extract_img_feat(img, img_metas)
        return voxel_feats, img_feats, depth

    def forward_pts_train(self, voxel_feats, gt_occ, points_occ=None, img_metas=None, **kwargs):
        outs = self.pts_bbox_head(voxel_feats=voxel_feats, points=points_occ, img_metas=img_metas)
        return self.pts_bbox_head.loss(output_voxels=outs["output_voxels"], target_voxels=gt_occ,
                                       output_points=outs["output_points"], target_points=points_occ,
                                       img_metas=img_metas)

    def forward_train(self, points=None, img_metas=None, img_inputs=None, gt_occ=None, points_occ=None,
                      points_uv=None, **kwargs):
        voxel_feats, img_feats, depth = self.extract_feat(points, img=img_inputs, img_metas=img_metas)
        losses = {}
        if not self.disable_loss_depth:
            losses["loss_depth"] = self.img_view_transformer.get_depth_loss(img_inputs[0][7], depth)   # DET:230
        losses.update(self.forward_pts_train(voxel_feats, gt_occ, points_occ, img_metas))
        return losses

    def simple_test(self, img_metas=None, img=None, rescale=False, points_occ=None, gt_occ=None, points_uv=None):
        voxel_feats, img_feats, depth = self.extract_feat(points=None, img=img, img_metas=img_metas)
        out = self.pts_bbox_head(voxel_feats=voxel_feats, points=points_occ, img_metas=img_metas)
        out["evaluation_semantic"] = 0
        from ..functional import upsample_trilinear
        out["output_voxels"] = upsample_trilinear(out["output_voxels"][0], gt_occ.shape[1:])
        out["target_voxels"] = gt_occ
        return out

    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(**kwargs)
        return self.simple_test(kwargs.get("img_metas"), kwargs.get("img_inputs"), gt_occ=kwargs.get("gt_occ"))
