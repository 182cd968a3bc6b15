"""Model construction helpers for the benchmark / smoke / tests: the reference's model dict
(projects/configs/occupancy/semantickitti/stereoscene.py:57-126) re-derived for a given grid."""
import torch

from . import synthetic as S


def image_branch_cfg(arch="b7"):
    """``img_backbone`` / ``img_neck`` of the reference config (stereoscene.py:59-74), minus the checkpoint init."""
    return dict(
        img_backbone=dict(type="CustomEfficientNet", arch=arch, drop_path_rate=0.2, frozen_stages=0, norm_eval=False,
                          out_indices=(2, 3, 4, 5, 6), with_cp=True),
        img_neck=dict(type="SECONDFPN", in_channels=[48, 80, 224, 640, 2560], upsample_strides=[0.5, 1, 2, 4, 4],
                      out_channels=[128, 128, 128, 128, 128]))


def model_cfg(cfg, numC_Trans=128, warp_align_corners=True, image_branch=False):
    """Same structure and hyper-parameters as the reference config's ``model`` dict for the sizes in ``cfg`` (a
    synthetic.CFG_*).  ``image_branch=False`` (default, the benchmarked hot path a1-a16) leaves the 2-D
    image backbone/neck out: the detector then takes the image-neck features in place of raw images."""
    norm_cfg = dict(type="GN", num_groups=32, requires_grad=True)
    channels = [128, 256, 512]
    return dict(
        type="BEVDepthOccupancy", **(image_branch_cfg() if image_branch else {}),
        img_view_transformer=dict(
            type="ViewTransformerLiftSplatShootVoxel", downsample=cfg["downsample"], numC_input=640,
            cam_channels=30, semkitti=False, loss_depth_weight=1.0, grid_config=S.grid_config(cfg),
            data_config=dict(input_size=tuple(cfg["input_size"])), numC_Trans=numC_Trans, vp_megvii=False,
            warp_align_corners=warp_align_corners),
        img_bev_encoder_backbone=dict(type="CustomResNet3D", depth=18, num_stage=3, n_input_channels=numC_Trans,
                                      block_inplanes=channels, out_indices=(0, 1, 2), norm_cfg=norm_cfg),
        img_bev_encoder_neck=dict(type="SECONDFPN3D", norm_cfg=norm_cfg, in_channels=channels,
                                  upsample_strides=[1, 2, 4], out_channels=[128, 128, 128]),
        pts_bbox_head=dict(type="OccHead", num_level=1, in_channels=[384], out_channel=20, semantic_kitti=True,
                           point_cloud_range=list(cfg["pc_range"]), supervise_points=False,
                           sampling_img_feats=True, in_img_channels=640, soft_weights=True,
                           semkitti_loss_weight_cfg={"voxel_ce": 1.0, "voxel_sem_scal": 1.0, "voxel_geo_scal": 1.0,
                                                     "voxel_ohem": 0.0, "voxel_lovasz": 0.0, "frustum_dist": 0.0}),
        train_cfg=dict(pts=None), test_cfg=dict(pts=None))


def build_detector(cfg, device="cuda", fill=True, **kw):
    from . import plugin  # noqa: F401  (fills the registries)
    from .registry import DETECTORS
    m = DETECTORS.build(model_cfg(cfg, **kw))
    if fill:
        S.fill_state_dict_(m)
    return m.to(device)


def img_inputs_from_sample(smp, device="cuda"):
    """(left10, right10) tuples in the reference's ``img_inputs`` layout; slot 0 carries
    the image-neck features instead of raw images."""
    def side(x, geo, gt_depths, calib):
        rots, trans, intr, post_rots, post_trans, bda = (t.to(device) for t in geo)
        if torch.device(device).type == "cuda":      # what the data layer does (pipelines.collate): see attach_host_inverses
            from .plugin.view_transformer import attach_host_inverses
            attach_host_inverses(post_rots, intr, geo[3], geo[2])
        B = x.shape[0]
        s2s = intr.new_zeros(B, 1, 4, 4)
        return (x.to(device), rots, trans, intr, post_rots, post_trans, bda, gt_depths.to(device), s2s,
                calib.to(device))
    gd = smp["gt_depths"]
    return (side(smp["x_l"], smp["geo_l"], gd, smp["calib"]), side(smp["x_r"], smp["geo_r"], gd, smp["calib"]))
