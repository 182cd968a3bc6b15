"""Deterministic synthetic tensors: weights-by-key and SemanticKITTI-shaped inputs.

Everything is generated from integer hashes (no torch RNG), so the same tensors are reproduced
bit-for-bit in the build container (fixture generation against the imported reference), in the
CPU test-suite and on the GPU box, independent of torch version or module construction order.
SURVEY.md section 8(c) "fill-by-key" and section 8(d) "synthetic inputs".
"""
import math
import zlib

import numpy as np
import torch


def hash_uniform(name, shape, lo=-0.5, hi=0.5):
    """float32 tensor u[i] in [lo, hi): ((i * 2654435761 + crc32(name)) mod 2^32) / 2^32."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = zlib.crc32(name.encode())
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    # second mixing round so that consecutive indices are not a plain lattice
    h = (h ^ (h >> np.uint64(15))) * np.uint64(2246822519) & np.uint64(0xFFFFFFFF)
    h = (h ^ (h >> np.uint64(13))) * np.uint64(3266489917) & np.uint64(0xFFFFFFFF)
    h = h ^ (h >> np.uint64(16))
    u = h.astype(np.float64) / 4294967296.0
    out = (lo + (hi - lo) * u).astype(np.float32)
    return torch.from_numpy(out.reshape(shape))


def hash_normal(name, shape, std=1.0):
    """Approximately N(0, std^2): sum of four hashed uniforms (Irwin-Hall), exactly reproducible."""
    acc = sum(hash_uniform(f"{name}#{j}", shape) for j in range(4))
    return (acc * (std * math.sqrt(3.0))).float()


_NORM_TOKENS = (".bn", ".gn", "norm")


def fill_value_for(key, t):
    """Deterministic content for state-dict entry ``key`` with the shape/dtype of ``t``."""
    leaf = key.rsplit(".", 1)[-1]
    if key.endswith(("frustum", ".dx", ".bx", ".nx")) or key in ("dx", "bx", "nx", "frustum"):
        return None  # geometry buffers stay as constructed
    if leaf == "num_batches_tracked":
        return torch.zeros_like(t)
    if leaf in ("gamma", "alpha"):
        return torch.full_like(t, 0.5)
    if leaf == "running_mean":
        return hash_uniform(key, t.shape, -0.1, 0.1)
    if leaf == "running_var":
        return 1.0 + hash_uniform(key, t.shape, 0.0, 0.5)
    if t.dim() <= 1:
        if leaf == "weight":  # norm scale (1-D weights only belong to norms)
            return 1.0 + hash_uniform(key, t.shape, -0.25, 0.25)
        return hash_uniform(key, t.shape, -0.1, 0.1)  # biases
    if t.numel() == 1:  # BRI 1x1x1 single-channel convs
        return hash_uniform(key, t.shape, 0.5, 1.5)
    fan_in = int(np.prod(t.shape[1:]))
    if "conv_offset" in key:
        return hash_uniform(key, t.shape, -0.5, 0.5) * (0.5 / math.sqrt(fan_in))
    return hash_uniform(key, t.shape, -1.0, 1.0) * math.sqrt(3.0 / fan_in)


@torch.no_grad()
def fill_state_dict_(module_or_sd, prefix=""):
    """In-place deterministic fill of every tensor of a module / state dict, keyed by its name."""
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    for k, t in sd.items():
        v = fill_value_for(prefix + k, t)
        if v is not None:
            t.copy_(v.to(t.dtype))
    return module_or_sd


# ---------------------------------------------------------------------------------------------
# SemanticKITTI-shaped synthetic sample
# ---------------------------------------------------------------------------------------------

CFG_K112 = dict(name="kitti_d112", input_size=(384, 1280), downsample=8, occ_size=(256, 256, 32),
                pc_range=(0.0, -25.6, -2.0, 51.2, 25.6, 4.4), dbound=(2.0, 58.0, 0.5))
CFG_K192 = dict(name="kitti_d192", input_size=(384, 1280), downsample=8, occ_size=(256, 256, 32),
                pc_range=(0.0, -25.6, -2.0, 51.2, 25.6, 4.4), dbound=(2.0, 98.0, 0.5))
CFG_S = dict(name="small_d48", input_size=(96, 320), downsample=8, occ_size=(64, 64, 16),
             pc_range=(0.0, -12.8, -2.0, 25.6, 12.8, 4.4), dbound=(2.0, 26.0, 0.5))
CFG_T = dict(name="tiny_d16", input_size=(64, 160), downsample=8, occ_size=(32, 32, 8),
             pc_range=(0.0, -6.4, -2.0, 12.8, 6.4, 4.4), dbound=(2.0, 10.0, 0.5))
CONFIGS = {c["name"]: c for c in (CFG_K112, CFG_K192, CFG_S, CFG_T)}


def grid_config(cfg, lss_downsample=(2, 2, 2)):
    """xbound/ybound/zbound/dbound exactly as the reference config computes them (CFG:23-49)."""
    r, o = cfg["pc_range"], cfg["occ_size"]
    vox = [(r[3 + i] - r[i]) / o[i] for i in range(3)]
    return {
        "xbound": [r[0], r[3], vox[0] * lss_downsample[0]],
        "ybound": [r[1], r[4], vox[1] * lss_downsample[1]],
        "zbound": [r[2], r[5], vox[2] * lss_downsample[2]],
        "dbound": list(cfg["dbound"]),
    }


def kitti_calibration(B, img_w, right=False):
    """KITTI-like camera (scaled to ``img_w`` pixels) -> rots, trans, intrins(4x4), post_rots,
    post_trans, bda, calib  for B samples and N=1 camera."""
    s = img_w / 1241.0
    fx = 707.0912 * s
    cx, cy = 601.8873 * s, 183.1104 * s
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = fx
    K[0, 2], K[1, 2] = cx, cy
    K[0, 3] = -0.54 * fx if right else 0.0
    R = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])  # cam -> ego (inverse of velo->cam)
    t_velo = torch.tensor([0.0, -0.08, -0.27])
    t = -(R @ t_velo)
    rots = R.view(1, 1, 3, 3).repeat(B, 1, 1, 1)
    trans = t.view(1, 1, 3).repeat(B, 1, 1)
    intr = K.view(1, 1, 4, 4).repeat(B, 1, 1, 1)
    post_rots = torch.eye(3).view(1, 1, 3, 3).repeat(B, 1, 1, 1)
    post_trans = torch.zeros(B, 1, 3)
    bda = torch.eye(3).view(1, 3, 3).repeat(B, 1, 1)
    calib = torch.full((B,), fx * 0.54)
    return rots, trans, intr, post_rots, post_trans, bda, calib


def synthetic_sample(cfg, B=1, C_in=640, tag="s0", with_targets=True):
    """Image-neck outputs for both views + geometry + targets, all hash-generated.

    Returns a dict: x_l, x_r [B,1,C,fH,fW]; geo_l, geo_r (6-tuples); calib [B];
    gt_depths [B,1,H,W]; gt_occ [B,X,Y,Z] int64 in {0..19, 255}.
    """
    H, W = cfg["input_size"]
    fH, fW = H // cfg["downsample"], W // cfg["downsample"]
    x_l = hash_normal(f"{tag}/x_l", (B, 1, C_in, fH, fW))
    # right view = left view shifted by a few feature pixels + noise: keeps the correlation volume informative
    shift = 3
    x_r = torch.roll(x_l, -shift, dims=-1) * 0.8 + 0.2 * hash_normal(f"{tag}/x_r", (B, 1, C_in, fH, fW))
    gl = kitti_calibration(B, W, right=False)
    gr = kitti_calibration(B, W, right=True)
    out = dict(x_l=x_l, x_r=x_r, geo_l=gl[:6], geo_r=gr[:6], calib=gl[6])
    if with_targets:
        u = hash_uniform(f"{tag}/gt_depth_mask", (B, 1, H, W), 0.0, 1.0)
        d = hash_uniform(f"{tag}/gt_depth_val", (B, 1, H, W), cfg["dbound"][0], cfg["dbound"][1] - 6.0)
        out["gt_depths"] = torch.where(u < 0.05, d, torch.zeros_like(d))
        X, Y, Z = cfg["occ_size"]
        c = (hash_uniform(f"{tag}/gt_occ", (B, X, Y, Z), 0.0, 20.0)).long().clamp_(0, 19)
        ig = hash_uniform(f"{tag}/gt_occ_ignore", (B, X, Y, Z), 0.0, 1.0) < 0.10
        out["gt_occ"] = torch.where(ig, torch.full_like(c, 255), c)
    return out
