"""Training-step plumbing around the hot path: flat parameter / gradient buffers, the
RCCL gradient exchange (dp.FlatGradAllReduce) and the fused AdamW + clip kernels.

Reference recipe (projects/configs/occupancy/semantickitti/stereoscene.py:203-218): AdamW lr 1e-4,
weight_decay 0.01, grad_clip max_norm 5, step LR decay x0.1 at epochs 20 and 25.
"""
import ctypes as C

import torch

from . import capi
from .dp import FlatGradAllReduce


class LossScaler:
    """mmcv ``LossScaler`` as ``Fp16OptimizerHook`` drives it (the reference: occupancy/apis/mmdet_train.py:131-134,
    ``fp16 = dict(loss_scale=512.)`` or ``'dynamic'``): the loss is multiplied by ``scale`` before backward; an inf / nan
    gradient norm skips the update and halves a dynamic scale; ``scale_window`` clean steps double it.  Static mode keeps the
    scale (overflow steps are still skipped, as in mmcv)."""

    def __init__(self, init_scale=2.0 ** 32, mode="dynamic", scale_factor=2.0, scale_window=1000):
        if mode not in ("dynamic", "static"):
            raise ValueError("mode must be 'dynamic' or 'static'")
        self.cur_scale, self.mode = float(init_scale), mode
        self.scale_factor, self.scale_window = float(scale_factor), int(scale_window)
        self.cur_iter, self.last_overflow_iter = 0, -1

    @classmethod
    def from_config(cls, fp16_cfg):
        """``fp16_cfg`` = the reference's ``fp16`` dict: ``loss_scale`` = number (static) | 'dynamic' | dict(LossScaler kwargs)."""
        ls = (fp16_cfg or {}).get("loss_scale", 512.0)
        if ls == "dynamic":
            return cls(mode="dynamic")
        if isinstance(ls, dict):
            return cls(**ls)
        return cls(init_scale=float(ls), mode="static")

    @property
    def loss_scale(self):
        return self.cur_scale

    def update_scale(self, overflow):
        if self.mode == "dynamic":
            if overflow:
                self.cur_scale = max(self.cur_scale / self.scale_factor, 1.0)
                self.last_overflow_iter = self.cur_iter
            elif (self.cur_iter - self.last_overflow_iter) % self.scale_window == 0:
                self.cur_scale *= self.scale_factor
        self.cur_iter += 1

    def state_dict(self):
        return dict(cur_scale=self.cur_scale, cur_iter=self.cur_iter, mode=self.mode, scale_factor=self.scale_factor,
                    scale_window=self.scale_window, last_overflow_iter=self.last_overflow_iter)

    def load_state_dict(self, sd):
        for k, v in sd.items():
            setattr(self, k, v)


class FlatAdamW:
    """AdamW over one flat fp32 buffer holding every trainable parameter (``param.data`` become views),
    with clip-by-global-norm folded into the update kernel.  One norm pass + one update pass per step."""

    def __init__(self, module, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=5.0,
                 reducer=None, loss_scaler=None):
        # Mixed-precision policy (BASELINE configs[3]; the reference's Fp16OptimizerHook): parameters, moments and the flat
        # gradient buffer are fp32 -- they ARE the master weights; SSBEV_PRECISION=bf16 stores activations and activation
        # gradients as bf16 between layers and ALSO switches the data-parallel wire format to bf16 (dp.py; override with
        # SSBEV_DP_COMM_DTYPE / comm_dtype) -- so the hook reduces to its loss scaler: scaled loss, unscale before the
        # clip, skip + rescale on overflow.
        self.loss_scaler = loss_scaler
        self.reducer = reducer or FlatGradAllReduce(module)
        params = self.reducer.params                       # same order as the flat gradient buffer
        n = self.reducer.flat.numel()
        dev = self.reducer.flat.device
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_p.zero_()
        with torch.no_grad():
            for p in params:
                off, k = self.reducer._offsets[p], p.numel()     # same offsets as the flat gradient buffer (bucket padding)
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.lr, self.betas, self.eps, self.wd, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        lib = capi.load()
        self._ws = torch.empty(lib.ssbev_grad_norm_workspace(), dtype=torch.uint8, device=dev)

    def zero_grad(self):
        self.reducer.zero_grad()

    @torch.no_grad()
    def step(self):
        """Call after backward(): waits for the gradient exchange, then norm + fused update."""
        lib = capi.load()
        self.reducer.finish()
        g = self.reducer.flat
        n = g.numel()
        norm_ptr = None
        scaler = self.loss_scaler
        if scaler is not None:
            g.mul_(1.0 / scaler.loss_scale)                     # unscale (one pass over the flat buffer; bf16 mode only)
        if (self.max_grad_norm and self.max_grad_norm > 0) or scaler is not None:
            capi.check(lib.ssbev_grad_norm(capi.ptr(g), n, capi.ptr(self.norm), capi.ptr(self._ws), self._ws.numel(),
                                           capi.stream()), "ssbev_grad_norm")
            norm_ptr = capi.ptr(self.norm) if (self.max_grad_norm and self.max_grad_norm > 0) else None
        if scaler is not None:
            overflow = not bool(torch.isfinite(self.norm).item())      # the hook's has_overflow(): one host round trip
            scaler.update_scale(overflow)
            if overflow:
                return self.norm                                        # skipped step: parameters and moments untouched
        self.step_count += 1
        cfg = capi.AdamWCfg(self.lr, self.betas[0], self.betas[1], self.eps, self.wd, float(self.max_grad_norm or 0.0),
                            self.step_count)
        # parameters without a gradient this step (ablation modes) are skipped like torch.optim.AdamW skips grad None:
        # no weight decay, no moment decay.  Normally one range = one launch.
        for s, e in self.reducer.live_ranges():
            at = lambda t: C.c_void_p(t.data_ptr() + 4 * s)      # noqa: E731
            capi.check(lib.ssbev_adamw_step(at(self.flat_p), at(g), at(self.m), at(self.v), e - s,
                                            C.byref(cfg), norm_ptr, capi.stream()), "ssbev_adamw_step")
        return self.norm

    def _compact(self, flat):
        """The padded flat buffer -> a compact vector in parameter order.  Bucket padding depends on the world size and on
        ``bucket_mb``; the checkpointed moments must not (a run trained on 8 GPUs resumes on 1 or 4)."""
        r = self.reducer
        return torch.cat([flat[r._offsets[p]:r._offsets[p] + p.numel()] for p in r.params])

    def _scatter(self, flat, compact):
        r = self.reducer
        total = sum(p.numel() for p in r.params)
        if compact.numel() == flat.numel() and compact.numel() != total:
            flat.copy_(compact)                       # pre-r3 checkpoint written under the SAME padded layout
            return
        if compact.numel() != total:
            raise ValueError(f"optimizer state holds {compact.numel()} elements, the model has {total} trainable parameters "
                             f"(padded layout of this run: {flat.numel()}); it was saved under a different bucket layout")
        flat.zero_()
        pos = 0
        for p in r.params:
            k, o = p.numel(), r._offsets[p]
            flat[o:o + k].copy_(compact[pos:pos + k])
            pos += k

    def state_dict(self):
        """Moments in COMPACT parameter order (reverse registration order, no bucket padding): independent of world size."""
        sd = {"m": self._compact(self.m), "v": self._compact(self.v), "step": self.step_count, "lr": self.lr,
              "layout": "compact"}
        if self.loss_scaler is not None:
            sd["loss_scaler"] = self.loss_scaler.state_dict()
        return sd

    def load_state_dict(self, sd):
        self._scatter(self.m, sd["m"].to(self.m.device))
        self._scatter(self.v, sd["v"].to(self.v.device))
        self.step_count, self.lr = int(sd["step"]), float(sd["lr"])
        if self.loss_scaler is not None and "loss_scaler" in sd:
            self.loss_scaler.load_state_dict(sd["loss_scaler"])


def step_lr(base_lr, epoch, milestones=(20, 25), gamma=0.1):
    """mmcv 'step' LR policy of the reference config (lr_config step=[20, 25])."""
    return base_lr * (gamma ** sum(epoch >= m for m in milestones))


def train_step(model, optimizer, img_inputs, gt_occ):
    """One optimisation step of the hot path: forward, 4 losses, backward (+ overlapped gradient
    exchange), clip + AdamW.  Returns the loss dict (device scalars)."""
    optimizer.zero_grad()
    losses = model.forward_train(img_inputs=img_inputs, gt_occ=gt_occ)
    total = sum(v for k, v in losses.items() if k.startswith("loss"))
    scaler = getattr(optimizer, "loss_scaler", None)
    (total * scaler.loss_scale if scaler is not None else total).backward()
    optimizer.step()
    return losses
