"""Data side of the hot path: the on-disk formats the path's inputs come from and the pipeline step that
produces the ``gt_depths`` slot of ``img_inputs`` -- ``CreateDepthFromLiDAR``
(projects/mmdet3d_plugin/datasets/pipelines/occ_to_depth.py:189-412), registered under the reference's type string with the
reference's constructor kwargs and ``results`` keys.  The projection + nearest-point-per-pixel scatter run on the MI355X
(``csrc/lidar_depth.hip``, bit-pattern atomicMin); file parsing stays numpy, as upstream.

Formats: velodyne ``.bin`` = float32 x, y, z, intensity; lidarseg ``.label`` = uint32, semantic id in the low 16 bits
(occ_to_depth.py:238-246); KITTI ``calib.txt`` = ``P0..P3`` 3x4, ``Tr`` 3x4 (semantic_kitti_dataset.py:85-114);
preprocessed voxel labels ``*_1_1.npy`` = uint8 [256,256,32] (semantic_kitti_dataset.py:137-139).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import capi
from .registry import Registry

PIPELINES = Registry("pipeline")

# SemanticKITTI `learning_map` (raw semantic id -> training id; semantic-kitti-api config, identical to the reference's
# semantickitti.yaml -- tests/test_pipelines.py compares when the reference checkout is present)
LEARNING_MAP = {0: 0, 1: 0, 10: 1, 11: 2, 13: 5, 15: 3, 16: 5, 18: 4, 20: 5, 30: 6, 31: 7, 32: 8, 40: 9, 44: 10, 48: 11,
                49: 12, 50: 13, 51: 14, 52: 0, 60: 9, 70: 15, 71: 16, 72: 17, 80: 18, 81: 19, 99: 0, 252: 1, 253: 7,
                254: 6, 255: 8, 256: 5, 257: 5, 258: 4, 259: 5}


def read_calib(calib_path):
    """KITTI odometry ``calib.txt`` -> dict of 4x4 float64 matrices P2, P3, Tr (rows beyond the 3x4 block = identity)."""
    raw = {}
    with open(calib_path, "r") as f:
        for line in f:
            if line == "\n":
                break
            key, value = line.split(":", 1)
            raw[key] = np.array([float(v) for v in value.split()])
    out = {}
    for k in ("P2", "P3", "Tr"):
        m = np.identity(4)
        m[:3, :4] = raw[k].reshape(3, 4)
        out[k] = m
    return out


def load_velodyne(path):
    return np.fromfile(path, dtype=np.float32).reshape(-1, 4)


def load_lidarseg(path, learning_map=None):
    """uint32 labels -> training ids (int32): low 16 bits, then ``learning_map``."""
    lm = LEARNING_MAP if learning_map is None else learning_map
    raw = np.fromfile(path, dtype=np.uint32).reshape(-1) & 0xFFFF
    lut = np.zeros(max(max(lm) + 1, int(raw.max(initial=0)) + 1), dtype=np.int32)
    for k, v in lm.items():
        lut[k] = v
    unknown = np.setdiff1d(np.unique(raw), np.fromiter(lm.keys(), dtype=np.int64))
    if unknown.size:
        raise KeyError(int(unknown[0]))                    # upstream: dict.__getitem__ raises on an unmapped id
    return lut[raw]


def load_voxel_labels(path):
    """Preprocessed SemanticKITTI voxel labels ``*_1_1.npy``: uint8 [256,256,32] (255 = ignore)."""
    a = np.load(path)
    if a.dtype != np.uint8:
        raise TypeError(f"{path}: uint8 expected, got {a.dtype}")
    return a


def lidar_depth_map(points, labels, rots, trans, intrins, post_rots, post_trans, H, W):
    """points [N,3] fp32 on the GPU, labels [N] fp32 or None; camera tensors as in img_inputs (rots [1,3,3], trans [1,3],
    intrins [1,4,4], post_rots [1,3,3], post_trans [1,3]).  Returns (uvd [N,3], valid [N] bool, depth [H,W], seg [H,W])."""
    lib = capi.load()
    dev = points.device
    inv = torch.inverse(rots.detach().float().cpu().reshape(3, 3))
    cam = torch.cat([inv.reshape(-1), trans.detach().float().cpu().reshape(-1)[:3],
                     intrins.detach().float().cpu().reshape(4, 4).reshape(-1),
                     post_rots.detach().float().cpu().reshape(3, 3)[:2, :2].reshape(-1),
                     post_trans.detach().float().cpu().reshape(-1)[:2]]).contiguous()
    assert cam.numel() == 34
    pts = points.float().contiguous()
    n = pts.shape[0]
    uvd = torch.empty(n, 3, dtype=torch.float32, device=dev)
    valid = torch.empty(n, dtype=torch.uint8, device=dev)
    depth = torch.empty(H, W, dtype=torch.float32, device=dev)
    seg = torch.empty(H, W, dtype=torch.float32, device=dev) if labels is not None else None
    ws = torch.empty(lib.ssbev_lidar_depth_workspace(H, W), dtype=torch.uint8, device=dev)
    lab = None if labels is None else labels.float().contiguous()
    capi.check(lib.ssbev_lidar_depth_map(capi.ptr(pts) if n else None, n, C.c_void_p(cam.data_ptr()), capi.ptr(lab),
                                         capi.ptr(uvd) if n else None, capi.ptr(valid) if n else None, capi.ptr(depth),
                                         capi.ptr(seg), H, W, capi.ptr(ws), ws.numel(), capi.stream()),
               "ssbev_lidar_depth_map")
    return uvd, valid.bool(), depth, seg


@PIPELINES.register_module()
class CreateDepthFromLiDAR:
    """Pipeline step of stereoscene.py:143.  Reads the frame's velodyne scan and lidarseg labels, and for the left then
    the right camera writes the sparse depth map into slot 7 of ``results['img_inputs'][k]``; ``points_occ``,
    ``points_uv`` and ``img_seg`` are those of the right view (upstream overwrites them the same way)."""

    def __init__(self, point_cloud_range, grid_size, projective_filter=True, label_mapping="semantickitti.yaml",
                 lidar_root="./data/lidar/velodyne/dataset/sequences", lidarseg_root="./data/lidar/lidarseg/dataset/sequences",
                 device="cuda"):
        self.grid_size = np.array(grid_size)
        self.point_cloud_range = torch.tensor(point_cloud_range)
        self.voxel_size = (self.point_cloud_range[3:] - self.point_cloud_range[:3]) / torch.as_tensor(self.grid_size)
        self.projective_filter = projective_filter
        self.lidar_root, self.lidarseg_root, self.device = lidar_root, lidarseg_root, device
        self.learning_map = LEARNING_MAP
        if label_mapping and os.path.exists(label_mapping):
            import yaml
            with open(label_mapping, "r") as stream:
                self.learning_map = yaml.safe_load(stream)["learning_map"]

    def _view(self, results, k, bda_mat):
        img_filename = results["img_filename"][k]
        seq_id, _, filename = img_filename.split("/")[-3:]
        pts = load_velodyne(os.path.join(self.lidar_root, seq_id, "velodyne", filename.replace(".png", ".bin")))[:, :3]
        seg = load_lidarseg(os.path.join(self.lidarseg_root, seq_id, "labels", filename.replace(".png", ".label")),
                            self.learning_map)
        dev = self.device
        points = torch.from_numpy(np.ascontiguousarray(pts)).to(dev)
        labels = torch.from_numpy(seg.astype(np.float32)).to(dev)
        view = results["img_inputs"][k]
        imgs, rots, trans, intrins, post_rots, post_trans = view[:6]
        H, W = imgs[0].shape[-2:]
        uvd, valid, depth, img_seg = lidar_depth_map(points, labels, rots, trans, intrins, post_rots, post_trans, H, W)
        bda = bda_mat.to(dev).float()
        if bda.shape[-1] == 4:
            homo = torch.cat((points, torch.ones(points.shape[0], 1, device=dev)), dim=1) @ bda.t()
            lidar_points = homo[:, :3]
        else:
            lidar_points = points @ bda.t()
        results["points_occ"] = torch.cat((lidar_points, labels[:, None]), dim=1)[valid]
        puv = uvd[valid].clone()
        puv[:, 0] /= W
        puv[:, 1] /= H
        puv[:, :2] = (puv[:, :2] - 0.5) * 2
        results["points_uv"] = puv.unsqueeze(1)
        results["img_seg"] = img_seg
        out = list(view)
        out[7] = depth.unsqueeze(0)
        return out

    def __call__(self, results):
        bda_mat = results["img_inputs"][0][6]
        left = self._view(results, 0, bda_mat)
        right = self._view(results, 1, bda_mat)
        results["img_inputs"] = [left, right]
        return results
