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
