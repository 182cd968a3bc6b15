"""Evaluation loop of the hot path: ``simple_test`` per sample -> argmax on the upsampled
logits -> SSC counts -> dataset-level scores with the reference's key names and rounding
(semantic_kitti_lss_dataset.py:231-287; apis/test.py:141-224 gathers per-rank results through pickle files,
here the integer counts are summed over ranks with one all-reduce)."""
import numpy as np
import torch
import torch.distributed as dist

from .plugin.losses import KITTI_CLASS_NAMES, ssc_counts

CLASS_NAMES = ["unlabeled"] + KITTI_CLASS_NAMES[1:]


@torch.no_grad()
def evaluate(model, samples, device="cuda"):
    """``samples`` yields dicts with ``img_inputs`` (left10, right10) and ``gt_occ`` [B,X,Y,Z].
    Returns the reference's ``eval_results`` dict (percent, 2 decimals, + 'semkitti_combined_IoU')."""
    model.eval()
    acc = torch.zeros(3 + 3 * len(CLASS_NAMES), dtype=torch.float64, device=device)
    for s in samples:
        gt = s["gt_occ"].to(device)
        out = model.simple_test(None, s["img_inputs"], gt_occ=gt)
        pred = out["output_voxels"].argmax(dim=1)
        tp, fp, fn, tpc, fpc, fnc = ssc_counts(pred, gt, len(CLASS_NAMES), recompute_mask=True)
        acc += torch.cat([torch.stack([tp, fp, fn]).double(), tpc.double(), fpc.double(), fnc.double()])
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(acc)
    return scores_from_counts(acc.cpu().numpy())


def scores_from_counts(acc):
    n = len(CLASS_NAMES)
    tp, fp, fn = acc[0], acc[1], acc[2]
    tps, fps, fns = acc[3:3 + n], acc[3 + n:3 + 2 * n], acc[3 + 2 * n:3 + 3 * n]
    iou_ssc = tps / (tps + fps + fns + 1e-5)
    res = {"SC_Precision": tp / (tp + fp), "SC_Recall": tp / (tp + fn), "SC_IoU": tp / (tp + fp + fn),
           "SSC_mIoU": iou_ssc[1:].mean()}
    for name, v in zip(CLASS_NAMES, iou_ssc):
        res[f"SSC_{name}_IoU"] = v
    out = {f"semkitti_{k}": round(float(v) * 100, 2) for k, v in res.items()}
    out["semkitti_combined_IoU"] = out["semkitti_SC_IoU"] + out["semkitti_SSC_mIoU"]
    return out
