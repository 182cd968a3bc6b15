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
import math
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


# -------------------------------------------------------------------------------------------------
# Image loading: Pillow-exact resize + crop / flip / normalise on the GPU
# -------------------------------------------------------------------------------------------------

_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    x = np.abs(x)
    a = -0.5
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def pil_resample_tables(in_size, out_size):
    """Fixed-point coefficient table of Pillow's antialiased bicubic resize along one axis (libImaging Resample.c,
    `precompute_coeffs` + `normalize_coeffs_8bpc`): returns (kk int32 [out, ksize], bounds int32 [out, 2], ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = _bicubic((np.arange(xmax) + xmin - center + 0.5) * ss)
        tot = w.sum()
        if tot != 0.0:
            w = w / tot
        fixed = np.where(w < 0, -0.5 + w * (1 << _PRECISION_BITS), 0.5 + w * (1 << _PRECISION_BITS))
        kk[xx, :xmax] = np.trunc(fixed).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return kk, bounds, ksize


_TABLE_CACHE = {}


def _tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _TABLE_CACHE:
        kk, bounds, ksize = pil_resample_tables(in_size, out_size)
        _TABLE_CACHE[key] = (torch.from_numpy(kk).to(device), torch.from_numpy(bounds).to(device), ksize)
    return _TABLE_CACHE[key]


def resize_u8(img, size):
    """``PIL.Image.resize(size)`` (default bicubic, antialiased) of a uint8 [H, W, C] GPU tensor; ``size`` = (W, H) as in
    PIL.  Byte-exact with Pillow (tests/test_pipelines.py)."""
    lib = capi.load()
    Hs, Ws, Cc = img.shape
    Wd, Hd = int(size[0]), int(size[1])
    img = img.contiguous()
    dst = torch.empty(Hd, Wd, Cc, dtype=torch.uint8, device=img.device)
    kh = bh = kv = bv = None
    ksh = ksv = 0
    if Wd != Ws:
        kh, bh, ksh = _tables(Ws, Wd, img.device)
    if Hd != Hs:
        kv, bv, ksv = _tables(Hs, Hd, img.device)
    tmp = torch.empty(Hs, Wd, Cc, dtype=torch.uint8, device=img.device) if (Wd != Ws and Hd != Hs) else None
    capi.check(lib.ssbev_resize_pil_u8(capi.ptr(img), Hs, Ws, Cc, capi.ptr(kh), capi.ptr(bh), ksh, capi.ptr(kv), capi.ptr(bv),
                                       ksv, capi.ptr(tmp), capi.ptr(dst), Hd, Wd, capi.stream()), "ssbev_resize_pil_u8")
    return dst


def crop_normalize(img, crop, flip, mean, std, swap_rb=False):
    """``img.crop(crop)`` (+ FLIP_LEFT_RIGHT) + mmcv ``imnormalize`` + ``permute(2, 0, 1)``: uint8 [H, W, 3] -> float32
    [3, h, w].  ``(x - float32(mean)) * float32(1 / float64(std))`` as mmcv/OpenCV compute it."""
    lib = capi.load()
    Hs, Ws, _ = img.shape
    x0, y0, x1, y1 = (int(v) for v in crop)
    w, h = x1 - x0, y1 - y0
    out = torch.empty(3, h, w, dtype=torch.float32, device=img.device)
    m = np.asarray(mean, dtype=np.float32)
    si = (1.0 / np.asarray(std, dtype=np.float32).astype(np.float64)).astype(np.float32)
    capi.check(lib.ssbev_crop_normalize_u8(capi.ptr(img.contiguous()), Hs, Ws, capi.ptr(out), x0, y0, w, h, int(bool(flip)),
                                           m.ctypes.data_as(C.c_void_p), si.ctypes.data_as(C.c_void_p), int(bool(swap_rb)),
                                           capi.stream()), "ssbev_crop_normalize_u8")
    return out


def read_image_rgb(path):
    """PNG/JPEG -> uint8 [H, W, 3] in RGB order (upstream reads BGR with cv2 and swaps inside imnormalize: same pixels)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


@PIPELINES.register_module()
class LoadMultiViewImageFromFiles_SemanticKitti:
    """Pipeline step of stereoscene.py:140 (loading_semkitti.py:76-302): loads the stereo pair, applies the image-view
    augmentation (resize / crop / flip; the same draw for both views), normalises, and assembles
    ``results['img_inputs'] = [left, right]`` with each view = [img, rot, tran, intrin, post_rot, post_tran, depth,
    cam2lidar, calib] (leading axis of 1).  Pixels are produced on the GPU (``resize_u8`` / ``crop_normalize``); the
    augmentation draw and the 3x3 bookkeeping are host code, as upstream.  ``rot != 0`` and ``colorjitter`` are not built
    (the config sets rot = (0, 0), colorjitter = False)."""

    def __init__(self, data_config, is_train=False, colorjitter=False, img_norm_cfg=None, load_depth=False, device="cuda"):
        if colorjitter or load_depth:
            raise NotImplementedError("colorjitter / load_depth are off in stereoscene.py and not built")
        self.is_train, self.data_config, self.img_norm_cfg, self.device = is_train, data_config, img_norm_cfg, device

    # The image-view augmentation of one sample = five numbers (scale, left, top, mirrored, angle).  TRAIN draws them from
    # numpy's global RNG; the fixture tests/golden/image_loading.npz pins the ORDER of the draws and the integer truncations
    # (loading_semkitti.py:138-166), nothing else: each row below is (name, "does it consume the RNG stream?", draw).
    _DRAWS = (
        ("jitter", lambda c, st: True, lambda c, st: np.random.uniform(*c["resize"])),
        ("top_frac", lambda c, st: True, lambda c, st: np.random.uniform(*c["crop_h"])),
        ("left", lambda c, st: True, lambda c, st: np.random.uniform(0, max(0, st["size"][0] - st["target"][0]))),
        ("mirrored", lambda c, st: bool(c["flip"]), lambda c, st: np.random.choice([0, 1])),     # no draw when flips are off
        ("angle", lambda c, st: True, lambda c, st: np.random.uniform(*c["rot"])),
    )

    def sample_augmentation(self, H, W, flip=None, scale=None):
        """-> (resize, (newW, newH), (x0, y0, x1, y1), flip, rotate): one draw per sample, shared by both views."""
        cfg = self.data_config
        fH, fW = cfg["input_size"]
        st = {"target": (fW, fH)}
        base = float(fW) / float(W)
        if self.is_train:
            for name, consumes, draw in self._DRAWS:
                if name == "top_frac":                       # the resized size is known once the scale jitter is drawn
                    st["scale"] = base + st["jitter"]
                    st["size"] = (int(W * st["scale"]), int(H * st["scale"]))
                st[name] = draw(cfg, st) if consumes(cfg, st) else cfg["flip"]
            top = int((1 - st["top_frac"]) * st["size"][1]) - fH
            left, mirrored, angle = int(st["left"]), st["mirrored"], st["angle"]
        else:
            st["scale"] = scale if scale is not None else base + cfg.get("resize_test", 0.0)
            st["size"] = (int(W * st["scale"]), int(H * st["scale"]))
            top = int((1 - np.mean(cfg["crop_h"])) * st["size"][1]) - fH
            left = int(max(0, st["size"][0] - fW) / 2)
            mirrored, angle = (False if flip is None else flip), 0
        return st["scale"], st["size"], (left, top, left + fW, top + fH), mirrored, angle

    @staticmethod
    def _then(step, cur):
        """Affine maps of image coordinates as (M [2,2], t [2]) fp32 pairs: ``step`` applied after ``cur``."""
        return step[0].matmul(cur[0]), step[0].matmul(cur[1]) + step[1]

    def pixel_map(self, post_rot, post_tran, resize, crop, flip):
        """The pixel map raw image -> network input composed onto (post_rot, post_tran): scale, shift by the crop corner,
        mirror about the crop width (what the reference accumulates at loading_semkitti.py:109-125; a rotation about the crop
        centre would be the fourth step).  This is a true composition of affine maps; the reference's in-place update leaves an
        incoming translation unscaled (loading_semkitti.py:113-114), and the two agree exactly when the incoming translation is
        zero -- which is the only way the loader calls it (``_view`` passes eye(2) / zeros(2)).  Anything else is refused."""
        if bool((torch.as_tensor(post_tran) != 0).any()):
            raise ValueError("pixel_map: a non-zero incoming post_tran is not what the reference loader composes onto "
                             "(loading_semkitti.py:113-114 would leave it unscaled)")
        eye = torch.eye(2)
        steps = [(eye * resize, torch.zeros(2)),
                 (eye, -torch.Tensor([crop[0], crop[1]]))]
        if flip:
            steps.append((torch.Tensor([[-1, 0], [0, 1]]), torch.Tensor([crop[2] - crop[0], 0])))
        m = (post_rot, post_tran)
        for step in steps:
            m = self._then(step, m)
        return m

    def img_transform(self, img, post_rot, post_tran, resize, resize_dims, crop, flip, rotate):
        """img: uint8 [H, W, 3] on the GPU -> normalised float [3, fH, fW] + the updated (post_rot, post_tran)."""
        if rotate != 0:
            raise NotImplementedError("image rotation is off in stereoscene.py (rot = (0, 0)) and not built")
        cfg = self.img_norm_cfg or dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
        out = crop_normalize(resize_u8(img, resize_dims), crop, flip, cfg["mean"], cfg["std"], swap_rb=False)
        post_rot, post_tran = self.pixel_map(post_rot, post_tran, resize, crop, flip)
        return out, post_rot, post_tran

    def _view(self, results, k, augs):
        raw = torch.from_numpy(np.array(read_image_rgb(results["img_filename"][k]))).to(self.device)
        resize, resize_dims, crop, flip, rotate = augs
        img, post_rot2, post_tran2 = self.img_transform(raw, torch.eye(2), torch.zeros(2), resize, resize_dims, crop, flip, rotate)
        post_tran, post_rot = torch.zeros(3), torch.eye(3)
        post_tran[:2] = post_tran2
        post_rot[:2, :2] = post_rot2
        intrin = torch.Tensor(results["cam_intrinsic"][k])
        cam2lidar = torch.Tensor(results["lidar2cam"][k]).inverse()
        res = [img, cam2lidar[:3, :3], cam2lidar[:3, 3], intrin, post_rot, post_tran, torch.zeros(1), cam2lidar,
               results["calib"]]
        return [x[None] if torch.is_tensor(x) else torch.as_tensor(x)[None] for x in res]

    def get_inputs(self, results, flip=None, scale=None):
        assert len(results["img_filename"]) == 2
        from PIL import Image
        with Image.open(results["img_filename"][1]) as im:          # the draw uses the right image's size (:189-190)
            W, H = im.size
        augs = self.sample_augmentation(H=H, W=W, flip=flip, scale=scale)
        right = self._view(results, 1, augs)
        left = self._view(results, 0, augs)
        return [left, right]

    def __call__(self, results):
        results["img_inputs"] = self.get_inputs(results)
        return results


def bev_transform(voxel_labels, rotate_angle, scale_ratio, flip_dx, flip_dy, transform_center):
    """BEV augmentation of the voxel labels and its 4x4 matrix about the centre of the point-cloud range
    (loading_semkitti.py:304-356): ``denorm @ flip_x @ flip_y @ rot @ norm``; labels rotated with nearest-neighbour
    resampling (fill 255) and flipped.  ``scale_ratio`` is drawn but unused upstream."""
    trans_norm, trans_denorm = torch.eye(4), torch.eye(4)
    trans_norm[:3, -1] = -transform_center
    trans_denorm[:3, -1] = transform_center
    ang = torch.tensor(rotate_angle / 180 * np.pi)
    s, c = torch.sin(ang), torch.cos(ang)
    rot_mat = torch.Tensor([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    flip_mat = torch.eye(4)
    if flip_dx:
        flip_mat = flip_mat @ torch.diag(torch.Tensor([-1, 1, 1, 1]))
    if flip_dy:
        flip_mat = flip_mat @ torch.diag(torch.Tensor([1, -1, 1, 1]))
    bda_mat = trans_denorm @ flip_mat @ rot_mat @ trans_norm
    lab = voxel_labels.numpy().astype(np.uint8)
    if not np.isclose(rotate_angle, 0):
        import scipy.ndimage
        scipy.ndimage.rotate(lab, rotate_angle, output=lab, mode="constant", order=0, cval=255, axes=(0, 1), reshape=False)
    if flip_dy:
        lab = lab[:, ::-1]
    if flip_dx:
        lab = lab[::-1]
    return torch.from_numpy(lab.copy()).long(), bda_mat


@PIPELINES.register_module()
class LoadSemKittiAnnotation:
    """Pipeline step of stereoscene.py:142 (loading_semkitti.py:358-402): turns ``results['gt_occ']`` into a tensor,
    optionally applies the BEV augmentation, and inserts the BEV matrix as slot 6 of both views' ``img_inputs``."""

    def __init__(self, bda_aug_conf, is_train=True, apply_bda=False, point_cloud_range=(0, -25.6, -2, 51.2, 25.6, 4.4)):
        self.bda_aug_conf, self.is_train, self.apply_bda = bda_aug_conf, is_train, apply_bda
        self.point_cloud_range = torch.tensor(point_cloud_range)
        self.transform_center = (self.point_cloud_range[:3] + self.point_cloud_range[3:]) / 2

    def sample_bda_augmentation(self):
        rotate_bda = np.random.uniform(*self.bda_aug_conf["rot_lim"])
        scale_bda = np.random.uniform(*self.bda_aug_conf["scale_lim"])
        flip_dx = np.random.uniform() < self.bda_aug_conf["flip_dx_ratio"]
        flip_dy = np.random.uniform() < self.bda_aug_conf["flip_dy_ratio"]
        return rotate_bda, scale_bda, flip_dx, flip_dy

    def __call__(self, results):
        g = results["gt_occ"]
        gt_occ = [torch.tensor(x) for x in g] if type(g) is list else torch.tensor(g)
        if self.apply_bda:
            if self.is_train:
                gt_occ, bda_rot = bev_transform(gt_occ, *self.sample_bda_augmentation(), self.transform_center)
            else:
                bda_rot = torch.eye(4)
        else:
            bda_rot = torch.eye(3)
        views = []
        for v in results["img_inputs"]:
            imgs, rots, trans, intrins, post_rots, post_trans, gt_depths, sensor2sensors, calib = v
            views.append([imgs, rots, trans, intrins, post_rots, post_trans, bda_rot, gt_depths, sensor2sensors, calib])
        results["img_inputs"] = tuple(views)
        results["gt_occ"] = gt_occ
        return results


# -------------------------------------------------------------------------------------------------
# dataset: SemanticKITTI stereo + voxel labels (semantic_kitti_dataset.py:85-160, semantic_kitti_lss_dataset.py:153-229)
# -------------------------------------------------------------------------------------------------
DATASETS = Registry("dataset")

SPLITS = {"train": ["00", "01", "02", "03", "04", "05", "06", "07", "09", "10"], "val": ["08"],
          "test": ["11", "12", "13", "14", "15", "16", "17", "18", "19", "20", "21"]}


def read_calib_file(filepath):
    """Raw ``key: numbers`` table of a calib.txt (semantic_kitti_lss_dataset.py:212-222)."""
    data = {}
    with open(filepath, "r") as f:
        for line in f:
            line = line.rstrip()
            if not line:
                continue
            key, value = line.split(":", 1)
            try:
                data[key] = np.array([float(x) for x in value.split()])
            except ValueError:
                pass
    return data


def dynamic_baseline(calib_info):
    """Stereo baseline from the rectified projection matrices (semantic_kitti_lss_dataset.py:223-227)."""
    P3 = np.reshape(calib_info["P3"], [3, 4])
    P2 = np.reshape(calib_info["P2"], [3, 4])
    return P3[0, 3] / (-P3[0, 0]) - P2[0, 3] / (-P2[0, 0])


class Compose:
    def __init__(self, steps):
        self.steps = [PIPELINES.build(s) if isinstance(s, dict) else s for s in steps]

    def __call__(self, results):
        for s in self.steps:
            results = s(results)
            if results is None:
                return None
        return results


@DATASETS.register_module()
class CustomSemanticKITTILssDataset:
    """Index of <data_root>/dataset/sequences/<seq>/{image_2,image_3,voxels,calib.txt} + <ann_file>/<seq>/<id>_1_1.npy and
    the per-sample dict the pipeline starts from (``get_data_info``); ``__getitem__`` runs the pipeline."""

    def __init__(self, data_root, ann_file, pipeline, split="train", camera_used=("left", "right"), occ_size=(256, 256, 32),
                 pc_range=(0, -25.6, -2, 51.2, 25.6, 4.4), test_mode=False, sequences=None, **kwargs):
        self.data_root, self.ann_file, self.split, self.test_mode = data_root, ann_file, split, test_mode
        self.sequences = list(sequences) if sequences is not None else SPLITS[split]
        self.camera_map = {"left": "2", "right": "3"}
        self.camera_used = [self.camera_map[c] for c in camera_used]
        self.occ_size, self.pc_range = list(occ_size), list(pc_range)
        self.pipeline = Compose(pipeline) if pipeline is not None else None
        self.data_infos = self.load_annotations(ann_file)

    def load_annotations(self, ann_file):
        import glob
        scans = []
        for sequence in self.sequences:
            base = os.path.join(self.data_root, "dataset", "sequences", sequence)
            if not os.path.exists(os.path.join(base, "calib.txt")):
                continue
            calib = read_calib(os.path.join(base, "calib.txt"))
            P2, P3, Tr = calib["P2"], calib["P3"], calib["Tr"]
            for id_path in sorted(glob.glob(os.path.join(base, "voxels", "*.bin"))):
                img_id = os.path.basename(id_path).split(".")[0]
                voxel_path = os.path.join(ann_file, sequence, img_id + "_1_1.npy")
                scans.append(dict(img_2_path=os.path.join(base, "image_2", img_id + ".png"),
                                  img_3_path=os.path.join(base, "image_3", img_id + ".png"), sequence=sequence, frame_id=img_id,
                                  P2=P2, P3=P3, T_velo_2_cam=Tr, proj_matrix_2=P2 @ Tr, proj_matrix_3=P3 @ Tr,
                                  voxel_path=voxel_path if os.path.exists(voxel_path) else None,
                                  calib_path=os.path.join(base, "calib.txt")))
        return scans

    def __len__(self):
        return len(self.data_infos)

    def get_ann_info(self, index):
        p = self.data_infos[index]["voxel_path"]
        return None if p is None else np.load(p)

    def get_data_info(self, index):
        info = self.data_infos[index]
        calib_info = read_calib_file(info["calib_path"])
        calib = np.reshape(calib_info["P2"], [3, 4])[0, 0] * dynamic_baseline(calib_info)
        return dict(occ_size=np.array(self.occ_size), pc_range=np.array(self.pc_range),
                    img_filename=[info[f"img_{c}_path"] for c in self.camera_used],
                    lidar2img=[info[f"proj_matrix_{c}"] for c in self.camera_used],
                    cam_intrinsic=[info[f"P{c}"] for c in self.camera_used],
                    lidar2cam=[info["T_velo_2_cam"] for _ in self.camera_used], calib=calib,
                    sequence=info["sequence"], frame_id=info["frame_id"], gt_occ=self.get_ann_info(index))

    def __getitem__(self, index):
        d = self.get_data_info(index)
        return d if self.pipeline is None else self.pipeline(d)


def collate(samples, device="cuda"):
    """Batch of pipeline outputs -> the detector's call signature: ``img_inputs = (left10, right10)`` with a leading
    batch axis on every entry (what mmcv's collate + scatter produce upstream), ``gt_occ`` [B, X, Y, Z] int64."""
    on_gpu = torch.device(device).type == "cuda"

    def put(t):
        # pinned staging + non-blocking copy: a plain .to(device) of pageable memory blocks until the stream has drained,
        # i.e. every tensor of the next batch would wait for the whole previous step
        return t.pin_memory().to(device, non_blocking=True) if (on_gpu and not t.is_cuda) else t.to(device)

    views = []
    for k in range(2):
        cols = []
        for j in range(10):
            items = [torch.as_tensor(s["img_inputs"][k][j]) for s in samples]
            if j == 9:                                              # calib: one scalar per sample
                cols.append(put(torch.stack([t.reshape(()).float() for t in items])))
            else:
                cols.append(put(torch.stack([t.float() for t in items])))
        if torch.device(device).type == "cuda":
            # inverse(post_rots) / inverse(intrins) of get_geometry, taken here on the host copies: the forward pass then has
            # no read-back of device matrices (a stream synchronisation per step)
            from .plugin.view_transformer import attach_host_inverses
            host = [torch.stack([torch.as_tensor(s["img_inputs"][k][j]).float() for s in samples]) for j in (4, 3)]
            attach_host_inverses(cols[4], cols[3], host[0], host[1])
        views.append(tuple(cols))
    gt = put(torch.stack([torch.as_tensor(s["gt_occ"]).long() for s in samples]))
    return dict(img_inputs=tuple(views), gt_occ=gt)


@PIPELINES.register_module()
class OccDefaultFormatBundle3D:
    """Formatting step of stereoscene.py:144 / :153 (datasets/pipelines/formating.py:50-89): ``gt_occ``, ``points_occ`` and
    ``points_uv`` become tensors.  mmcv's DataContainer wrapping (a hint for its collate / scatter) has no counterpart
    here: ``collate`` below stacks plain tensors."""

    def __init__(self, class_names=None, with_gt=True, with_label=True, **kwargs):
        self.class_names, self.with_gt, self.with_label = class_names, with_gt, with_label

    def __call__(self, results):
        if results.get("gt_occ") is not None:
            g = results["gt_occ"]
            results["gt_occ"] = tuple(torch.as_tensor(x) for x in g) if type(g) is list else torch.as_tensor(g)
        for k in ("points_occ", "points_uv"):
            if k in results:
                results[k] = torch.as_tensor(results[k])
        return results


@PIPELINES.register_module()
class Collect3D:
    """mmdet3d ``Collect3D`` as used at stereoscene.py:145-146 / :154-155: keeps ``keys`` and gathers the available
    ``meta_keys`` into ``img_metas`` (a plain dict)."""

    def __init__(self, keys, meta_keys=("pc_range", "occ_size", "sequence", "frame_id", "img_filename")):
        self.keys, self.meta_keys = list(keys), list(meta_keys)

    def __call__(self, results):
        data = {"img_metas": {k: results[k] for k in self.meta_keys if k in results}}
        for k in self.keys:
            if k in results:
                data[k] = results[k]
        return data
