"""CPU oracle of the data-side step CreateDepthFromLiDAR -- TEST INFRASTRUCTURE ONLY.

Restates datasets/pipelines/occ_to_depth.py:216-303 (``project_points`` and the per-view body of ``__call__``) as
functions over arrays; pinned by tests/golden/lidar_depth.npz (outputs of the reference class itself run on synthetic
velodyne / lidarseg files, oracle/make_golden_data.py)."""
import torch


def project_points(points, rots, trans, intrins, post_rots, post_trans):
    """occ_to_depth.py:216-236: lidar -> camera -> raw pixel -> augmented pixel; returns [N, 3] (u, v, depth)."""
    p = points.view(-1, 1, 3) - trans.view(1, -1, 3)
    p = rots.inverse().unsqueeze(0) @ p.unsqueeze(-1)
    p = torch.cat((p, torch.ones((p.shape[0], 1, 1, 1))), dim=2)
    p = (intrins.unsqueeze(0) @ p).squeeze(-1)
    d = p[..., 2:3]
    uv = p[..., :2] / d
    uv = (post_rots[:, :2, :2].unsqueeze(0) @ uv.unsqueeze(-1)).squeeze(-1) + post_trans[..., :2].unsqueeze(0)
    return torch.cat((uv, d), dim=2)[:, 0]


def nearest_scatter(proj, values, H, W):
    """The sorted index_put of :297-303 / :311-318: the NEAREST projected point of a pixel decides its value (what the
    sequential index_put of a single-threaded DataLoader worker produces).  Written as an explicit per-pixel arg-min so
    that the result does not depend on ATen's intra-op threading (index_put with duplicate indices is a race there)."""
    pix = proj[:, 1].round().long() * W + proj[:, 0].round().long()
    order = torch.argsort(proj[:, 2], descending=True, stable=True)        # farthest first
    out = torch.zeros(H * W)
    pix_o, val_o = pix[order].tolist(), values[order].tolist()
    for p, v in zip(pix_o, val_o):                                          # sequential: last (nearest) write wins
        out[p] = v
    return out.view(H, W)


def create_depth_view(points, seg, rots, trans, intrins, post_rots, post_trans, bda, H, W):
    """One view of ``CreateDepthFromLiDAR.__call__``: returns dict(depth, img_seg, points_occ, points_uv, valid)."""
    proj = project_points(points, rots, trans, intrins, post_rots, post_trans)
    valid = (proj[:, 0] >= 0) & (proj[:, 1] >= 0) & (proj[:, 0] <= W - 1) & (proj[:, 1] <= H - 1) & (proj[:, 2] > 0)
    if bda.shape[-1] == 4:
        lp = (torch.cat((points, torch.ones(points.shape[0], 1)), dim=1) @ bda.t())[:, :3]
    else:
        lp = points @ bda.t()
    pv = proj[valid]
    uv = pv.clone()
    uv[:, 0] /= W
    uv[:, 1] /= H
    uv[:, :2] = (uv[:, :2] - 0.5) * 2
    return dict(depth=nearest_scatter(pv, pv[:, 2], H, W), img_seg=nearest_scatter(pv, seg[valid], H, W),
                points_occ=torch.cat((lp, seg[:, None]), dim=1)[valid], points_uv=uv.unsqueeze(1), valid=valid, proj=proj)


# -------------------------------------------------------------------------------------------------
# image loading (loading_semkitti.py:76-302) -- numpy restatement
# -------------------------------------------------------------------------------------------------
import math  # noqa: E402

import numpy as np  # noqa: E402


def _cubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_resize_axis_u8(img, axis, out_size):
    """One pass of Pillow's ``ImagingResample`` (bicubic, 8 bpc) along ``axis`` -- scalar loops, oracle only."""
    in_size = img.shape[axis]
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 2.0 * fs
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.zeros((out_size,) + src.shape[1:], dtype=np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_cubic((x + xmin - center + 0.5) / fs) for x in range(xmax)]
        tot = sum(w)
        w = [v / tot for v in w] if tot != 0.0 else w
        kk = [int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22)) for v in w]
        acc = np.full(src.shape[1:], 1 << 21, dtype=np.int64)
        for j, k in enumerate(kk):
            acc = acc + src[xmin + j] * k
        out[xx] = np.clip(acc >> 22, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def pil_resize_u8(img, size):
    """``Image.fromarray(img).resize(size)`` with size = (W, H): horizontal pass first, then vertical."""
    Wd, Hd = size
    if img.shape[1] != Wd:
        img = pil_resize_axis_u8(img, 1, Wd)
    if img.shape[0] != Hd:
        img = pil_resize_axis_u8(img, 0, Hd)
    return img


def load_view(img_rgb, lidar2cam, cam_intrinsic, data_config, is_train=False, mean=(123.675, 116.28, 103.53),
              std=(58.395, 57.12, 57.375)):
    """Deterministic (test-time / zero-range) path of ``get_inputs`` for one view: returns (img [3,fH,fW] float32,
    rot, tran, post_rot [3,3], post_tran [3])."""
    H, W = img_rgb.shape[:2]
    fH, fW = data_config["input_size"]
    resize = float(fW) / float(W) + (0.0 if is_train else data_config.get("resize_test", 0.0))
    newW, newH = int(W * resize), int(H * resize)
    crop_h = int((1 - np.mean(data_config["crop_h"])) * newH) - fH
    crop_w = int(max(0, newW - fW) / 2)
    img = pil_resize_u8(img_rgb, (newW, newH))
    canvas = np.zeros((fH, fW, 3), dtype=np.uint8)                      # PIL crop: zero fill outside
    ys, xs = max(crop_h, 0), max(crop_w, 0)
    ye, xe = min(crop_h + fH, newH), min(crop_w + fW, newW)
    canvas[ys - crop_h:ye - crop_h, xs - crop_w:xe - crop_w] = img[ys:ye, xs:xe]
    m = np.asarray(mean, dtype=np.float32)
    si = (1.0 / np.asarray(std, dtype=np.float32).astype(np.float64)).astype(np.float32)
    out = (canvas.astype(np.float32) - m) * si
    post_rot = torch.eye(3)
    post_rot[:2, :2] *= resize
    post_tran = torch.zeros(3)
    post_tran[:2] = -torch.Tensor([crop_w, crop_h])
    b = torch.Tensor([fW, fH]) / 2                                       # rotate = 0: A = I, b = -b + b
    post_tran[:2] = post_tran[:2] + ((-b) + b)
    cam2lidar = torch.Tensor(lidar2cam).inverse()
    return torch.from_numpy(out).permute(2, 0, 1).contiguous(), cam2lidar[:3, :3], cam2lidar[:3, 3], post_rot, post_tran
