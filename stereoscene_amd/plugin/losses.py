"""SemanticKITTI occupancy losses and SSC metric of the hot path (device-side torch ops).

Mirrors utils/semkitti.py:67-149 (CE_ssc_loss / sem_scal_loss / geo_scal_loss), the loss assembly
of occhead.py:291-361 and SSCMetrics (utils/ssc_metric.py:40-169).  The per-class Python loop of
the reference's sem_scal_loss is replaced by one pass of class-indexed reductions (same sums)."""
import numpy as np
import torch
import torch.nn.functional as TF

KITTI_CLASS_NAMES = ["empty", "car", "bicycle", "motorcycle", "truck", "other-vehicle", "person", "bicyclist",
                     "motorcyclist", "road", "parking", "sidewalk", "other-ground", "building", "fence",
                     "vegetation", "trunk", "terrain", "pole", "traffic-sign"]

# SemanticKITTI per-class voxel counts (dataset statistics, utils/semkitti.py:8-31)
CLASS_FREQUENCIES = np.array([
    5.41773033e09, 1.57835390e07, 1.25136000e05, 1.18809000e05, 6.46799000e05, 8.21951000e05, 2.62978000e05,
    2.83696000e05, 2.04750000e05, 6.16887030e07, 4.50296100e06, 4.48836500e07, 2.26992300e06, 5.68402180e07,
    1.57196520e07, 1.58442623e08, 2.06162300e06, 3.69705220e07, 1.15198800e06, 3.34146000e05])


def semkitti_class_weights():
    return torch.from_numpy(1 / np.log(CLASS_FREQUENCIES + 0.001)).float()


def _nll_to_one(v):
    """BCE(v, 1) with torch's log clamp at -100."""
    return -torch.clamp(torch.log(v), min=-100.0)


def ce_ssc_loss(logits, target, class_weights):
    return TF.cross_entropy(logits, target.long(), weight=class_weights, ignore_index=255, reduction="mean")


def scal_losses(logits, target):
    """(sem_scal, geo_scal): all class-wise sums in one sweep over the softmax volume."""
    n_cls = logits.shape[1]
    prob = torch.softmax(logits, dim=1)
    valid = target != 255
    p = prob.permute(0, 2, 3, 4, 1)[valid]                    # [M, C]
    t = target[valid].long()                                   # [M]
    onehot = TF.one_hot(t, n_cls).to(p.dtype)                  # [M, C]
    M = p.shape[0]
    sum_p = p.sum(0)                                           # sum of p_c
    cnt = onehot.sum(0)                                        # |{t == c}|
    nom = (p * onehot).sum(0)                                  # sum p_c [t == c]
    spec_num = (M - cnt) - (sum_p - nom)                       # sum (1-p_c)(1-[t==c])
    present = cnt > 0
    loss_c = torch.zeros_like(sum_p)
    loss_c = loss_c + torch.where(sum_p > 0, _nll_to_one(nom / sum_p.clamp_min(1e-30)), torch.zeros_like(sum_p))
    loss_c = loss_c + _nll_to_one(nom / cnt.clamp_min(1.0))
    neg = M - cnt
    loss_c = loss_c + torch.where(neg > 0, _nll_to_one(spec_num / neg.clamp_min(1.0)), torch.zeros_like(sum_p))
    sem = (loss_c * present).sum() / present.sum()
    # geometric: occupied vs empty
    occ_t = (t != 0).to(p.dtype)
    occ_p = 1 - p[:, 0]
    inter = (occ_t * occ_p).sum()
    geo = (_nll_to_one(inter / occ_p.sum()) + _nll_to_one(inter / occ_t.sum())
           + _nll_to_one(((1 - occ_t) * p[:, 0]).sum() / (1 - occ_t).sum()))
    return sem, geo


def ssc_counts(pred, gt, n_classes=20, recompute_mask=False):
    """Integer tp/fp/fn for completion and per class (SSCMetrics semantics, see oracle docstring)."""
    valid = gt != 255
    p = torch.where(valid, pred, torch.zeros_like(pred)).reshape(-1)
    g = torch.where(valid, gt, torch.zeros_like(gt)).reshape(-1)
    v = valid.reshape(-1)
    po, go = (p > 0) & v, (g > 0) & v
    tp = (po & go).sum()
    fp = (po & ~go & v).sum()
    fn = (~po & go & v).sum()
    sel = torch.ones_like(v) if recompute_mask else v
    conf = torch.bincount((g[sel] * n_classes + p[sel]).long(), minlength=n_classes * n_classes)
    conf = conf.view(n_classes, n_classes)                       # [gt, pred]
    tpc = conf.diagonal()
    fpc = conf.sum(0) - tpc
    fnc = conf.sum(1) - tpc
    return tp, fp, fn, tpc, fpc, fnc


def _nll1(v):
    return -torch.clamp(torch.log(v), min=-100.0)


def occ_losses_fused(logits, gt_occ, class_weights, tag="0", w_ce=1.0, w_sem=1.0, w_geo=1.0, compute_metric=False):
    """Same losses / metric as ``occ_losses`` from the sums of the fused HIP epilogue (one pass over the coarse
    logits; the up-sampled logits, probabilities and one-hot volumes are never materialised)."""
    from .. import functional as F
    from ..functional import occ_loss_sums
    nc = logits.shape[1]
    label = gt_occ.to(torch.uint8)                      # classes 0..19, 255 = ignore
    diff, aux = occ_loss_sums(logits, label, class_weights)
    if F.OCC_TAIL:                                      # the algebra below in one launch (+ its Jacobian for backward)
        ce, sem, geo, iou, miou = F.occ_loss_tail(diff, aux, w_ce, w_sem, w_geo)
        out = {}
        if w_ce > 0:
            out[f"loss_voxel_ce_{tag}"] = ce
        if w_sem > 0:
            out[f"loss_voxel_sem_scal_{tag}"] = sem
        if w_geo > 0:
            out[f"loss_voxel_geo_scal_{tag}"] = geo
        if compute_metric:
            out[f"sc_iou_{tag}"], out[f"ssc_miou_{tag}"] = iou.detach(), miou.detach()
        return out
    ce_num, sum_p, nom = diff[0], diff[1:1 + nc], diff[1 + nc:1 + 2 * nc]
    ce_den, M, cnt, conf = aux[0], aux[1], aux[2:2 + nc], aux[2 + nc:].view(nc, nc)
    out = {}
    if w_ce > 0:
        out[f"loss_voxel_ce_{tag}"] = (ce_num / ce_den).float() * w_ce
    if w_sem > 0:
        present = cnt > 0
        neg = M - cnt
        spec_num = neg - (sum_p - nom)
        one = torch.ones_like(sum_p)
        loss_c = torch.where(sum_p > 0, _nll1(nom / torch.where(sum_p > 0, sum_p, one)), torch.zeros_like(sum_p))
        loss_c = loss_c + _nll1(nom / cnt.clamp_min(1.0))
        loss_c = loss_c + torch.where(neg > 0, _nll1(spec_num / neg.clamp_min(1.0)), torch.zeros_like(sum_p))
        out[f"loss_voxel_sem_scal_{tag}"] = ((loss_c * present).sum() / present.sum()).float() * w_sem
    if w_geo > 0:
        occ_t = M - cnt[0]
        inter = occ_t - (sum_p[0] - nom[0])
        geo = _nll1(inter / (M - sum_p[0])) + _nll1(inter / occ_t) + _nll1(nom[0] / cnt[0])
        out[f"loss_voxel_geo_scal_{tag}"] = geo.float() * w_geo
    if compute_metric:
        with torch.no_grad():
            tp = conf[1:, 1:].sum()
            fp = conf[0, 1:].sum()
            fn = conf[1:, 0].sum()
            tpc = conf.diagonal()
            fpc = conf.sum(0) - tpc
            fnc = conf.sum(1) - tpc
            out[f"sc_iou_{tag}"] = (tp / (tp + fp + fn)).float()
            out[f"ssc_miou_{tag}"] = (tpc / (tpc + fpc + fnc + 1e-5))[1:].mean().float()
    return out


def occ_losses(logits, gt_occ, class_weights, tag="0", w_ce=1.0, w_sem=1.0, w_geo=1.0, compute_metric=False):
    """occhead.py:291-361: trilinear upsample to the label grid, CE + sem_scal + geo_scal (+ metric)."""
    if (logits.is_cuda and logits.shape[1] == 20 and all(o == 2 * i for o, i in zip(gt_occ.shape[-3:], logits.shape[-3:]))):
        return occ_losses_fused(logits, gt_occ, class_weights, tag, w_ce, w_sem, w_geo, compute_metric)
    if logits.shape[-3:] != gt_occ.shape[-3:]:
        from ..functional import upsample_trilinear
        logits = upsample_trilinear(logits, gt_occ.shape[-3:])
    t = gt_occ.long()
    out = {}
    if w_ce > 0:
        out[f"loss_voxel_ce_{tag}"] = ce_ssc_loss(logits, t, class_weights) * w_ce
    if w_sem > 0 or w_geo > 0:
        sem, geo = scal_losses(logits, t)
        if w_sem > 0:
            out[f"loss_voxel_sem_scal_{tag}"] = sem * w_sem
        if w_geo > 0:
            out[f"loss_voxel_geo_scal_{tag}"] = geo * w_geo
    if compute_metric:
        with torch.no_grad():
            tp, fp, fn, tpc, fpc, fnc = ssc_counts(logits.argmax(1), t)
            out[f"sc_iou_{tag}"] = tp / (tp + fp + fn)
            out[f"ssc_miou_{tag}"] = (tpc / (tpc + fpc + fnc + 1e-5))[1:].mean()
    return out


class SSCMetrics(torch.nn.Module):
    """Accumulating SC-IoU / SSC-mIoU metric with the reference's interface
    (utils/ssc_metric.py:14-103): ``compute_single`` -> 6-tuple of numpy counts, ``update``,
    ``compute``.  State lives in non-persistent buffers (torchmetrics' default), reduced over
    ranks by summation."""

    def __init__(self, class_names=None):
        super().__init__()
        self.class_names = class_names or KITTI_CLASS_NAMES
        self.n_classes = len(self.class_names)
        for n in ("tps", "fps", "fns"):
            self.register_buffer(n, torch.zeros(self.n_classes), persistent=False)
        for n in ("completion_tp", "completion_fp", "completion_fn"):
            self.register_buffer(n, torch.zeros(1), persistent=False)

    def compute_single(self, y_pred, y_true):
        c = ssc_counts(y_pred, y_true, self.n_classes, recompute_mask=True)
        return tuple(v.cpu().numpy() for v in c)

    def update(self, y_pred, y_true):
        tp, fp, fn, tpc, fpc, fnc = ssc_counts(y_pred, y_true, self.n_classes, recompute_mask=True)
        self.completion_tp += tp
        self.completion_fp += fp
        self.completion_fn += fn
        self.tps += tpc
        self.fps += fpc
        self.fns += fnc

    def sync(self):
        """Sum the state over ranks (dist_reduce_fx='sum')."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            for b in (self.tps, self.fps, self.fns, self.completion_tp, self.completion_fp, self.completion_fn):
                dist.all_reduce(b)

    def compute(self):
        iou = self.completion_tp / (self.completion_tp + self.completion_fp + self.completion_fn)
        iou_ssc = self.tps / (self.tps + self.fps + self.fns + 1e-5)
        return {"precision": self.completion_tp / (self.completion_tp + self.completion_fp),
                "recall": self.completion_tp / (self.completion_tp + self.completion_fn),
                "iou": iou.item(), "iou_ssc": iou_ssc, "iou_ssc_mean": iou_ssc[1:].mean().item()}
