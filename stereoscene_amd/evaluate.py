"""Evaluation loop of the hot path: ``simple_test`` per sample -> argmax on the upsampled
logits -> SSC counts -> dataset-level scores with the reference's key names and rounding
(semantic_kitti_lss_dataset.py:231-287; apis/test.py:141-224 gathers per-rank results through pickle files,
here the integer counts are summed over ranks with one all-reduce)."""
import os

import numpy as np
import torch
import torch.distributed as dist

from .plugin.losses import KITTI_CLASS_NAMES, ssc_counts

CLASS_NAMES = ["unlabeled"] + KITTI_CLASS_NAMES[1:]


def evaluate(model, samples, device="cuda", dataset_len=None, sampler=None):
    """``eval_results`` dict of the reference (see ``evaluate_counts`` for the arguments)."""
    return scores_from_counts(evaluate_counts(model, samples, device, dataset_len, sampler).cpu().numpy())


@torch.no_grad()
def evaluate_counts(model, samples, device="cuda", dataset_len=None, sampler=None, reduce=True):
    """``samples`` yields dicts with ``img_inputs`` (left10, right10) and ``gt_occ`` [B,X,Y,Z].
    Returns the integer SSC counts (float64 vector: tp, fp, fn, then per-class tp / fp / fn), summed over ranks when a
    process group is initialised and ``reduce``; ``evaluate`` turns them into the reference's ``eval_results`` dict (percent,
    2 decimals, + 'semkitti_combined_IoU').

    Distributed evaluation: ``runner.DistributedSampler`` tiles the index list up to a multiple of the world size, so the
    last ranks see duplicates of the first samples.  The reference drops them (``collect_results_cpu`` keeps
    ``ordered_results[:len(dataset)]``, apis/test.py); pass the ``sampler`` (or ``dataset_len`` + the process group's
    rank / world size) and the padded tail is skipped before the counts are accumulated.
    The module's train / eval mode is restored on exit (the runner has no model handle to do it)."""
    was_training = model.training
    model.eval()
    rank, world = 0, 1
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    if sampler is not None:
        dataset_len, rank, world = sampler.n, sampler.rank, sampler.num_replicas
    elif world > 1 and dataset_len is None:
        # the old call signature under an initialised process group: the padded duplicates would be counted silently
        # (the reference always truncates to len(dataset)) -- refuse instead
        raise ValueError("evaluate(): torch.distributed is initialised with world_size > 1; pass sampler= (or dataset_len=) "
                         "so that the samples DistributedSampler pads the index list with are dropped")
    per_rank = -(-dataset_len // world) if dataset_len is not None else None
    acc = torch.zeros(3 + 3 * len(CLASS_NAMES), dtype=torch.float64, device=device)
    try:
        seen = 0                      # samples of this rank's block consumed so far
        for s in samples:
            gt = s["gt_occ"].to(device)
            out = model.simple_test(None, s["img_inputs"], gt_occ=gt)
            pred = out["output_voxels"].argmax(dim=1)
            nb = gt.shape[0]
            if per_rank is not None:
                # global position of sample i of this batch in the tiled index list = rank * per_rank + seen + i
                keep = [i for i in range(nb) if rank * per_rank + seen + i < dataset_len]
                seen += nb
                if not keep:
                    continue
                if len(keep) < nb:
                    pred, gt = pred[keep], gt[keep]
            tp, fp, fn, tpc, fpc, fnc = ssc_counts(pred, gt, len(CLASS_NAMES), recompute_mask=True)
            acc += torch.cat([torch.stack([tp, fp, fn]).double(), tpc.double(), fpc.double(), fnc.double()])
    finally:
        model.train(was_training)
    if reduce and dist.is_available() and dist.is_initialized():
        dist.all_reduce(acc)
    return acc


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


# ---------------------------------------------------------------------------------------------------------------------
# test-set submission writer and checkpoint loading
# ---------------------------------------------------------------------------------------------------------------------
# SemanticKITTI `learning_map_inv` (semantickitti.yaml:144-164 of the reference, the file get_inv_map() opens at
# utils/semkitti_io.py:99-112): training id -> raw lidarseg label id
LEARNING_MAP_INV = np.array([0, 10, 11, 15, 18, 20, 30, 31, 32, 40, 44, 48, 49, 50, 51, 70, 71, 72, 80, 81], dtype=np.int32)


def save_output_semantic_kitti(output_voxels, save_path, sequence_id, frame_id):
    """``save_output_semantic_kitti`` of apis/test.py:49-64: argmax over the class axis of ``output_voxels`` [C,X,Y,Z],
    remap to raw label ids, write ``<save_path>/sequences/<seq>/predictions/<frame>.label`` as uint16.  Returns the path."""
    pred = torch.argmax(output_voxels, dim=0).cpu().numpy().reshape(-1)
    raw = LEARNING_MAP_INV[pred].astype(np.uint16)
    folder = os.path.join(save_path, "sequences", str(sequence_id), "predictions")
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, f"{frame_id}.label")
    with open(path, "wb") as f:
        raw.tofile(f)
    return path


def load_checkpoint(model, path_or_state, strict=False):
    """Load a reference checkpoint (``pretrain_stereoscene.pth``: a dict with 'state_dict', or a bare state dict) into the
    hot-path modules.  Parameter names are the reference's (tests/test_layout.py checks the key set against the
    reference-generated manifest), so this is a filtered ``load_state_dict``: keys of the image backbone / neck (outside the
    path) are reported, not loaded.  Returns (missing, unexpected_outside_path)."""
    state = torch.load(path_or_state, map_location="cpu", weights_only=True) if isinstance(path_or_state, (str, os.PathLike)) else path_or_state
    state = state.get("state_dict", state)
    own = model.state_dict()
    inside = {k: v for k, v in state.items() if k in own}
    outside = sorted(k for k in state if k not in own)
    for k, v in inside.items():
        if tuple(v.shape) != tuple(own[k].shape):
            raise ValueError(f"checkpoint tensor {k}: shape {tuple(v.shape)} != model {tuple(own[k].shape)}")
    missing = sorted(k for k in own if k not in inside)
    if strict and missing:
        raise KeyError(f"{len(missing)} hot-path tensors missing from the checkpoint, e.g. {missing[:3]}")
    model.load_state_dict(inside, strict=False)
    return missing, outside
