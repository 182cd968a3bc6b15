"""Training-step plumbing around the hot path: flat parameter / gradient buffers, the
RCCL gradient exchange (dp.FlatGradAllReduce) and the fused AdamW + clip kernels.

Reference recipe (projects/configs/occupancy/semantickitti/stereoscene.py:203-218): AdamW lr 1e-4,
weight_decay 0.01, grad_clip max_norm 5, step LR decay x0.1 at epochs 20 and 25.
"""
import ctypes as C

import torch

from . import capi
from .dp import FlatGradAllReduce


class FlatAdamW:
    """AdamW over one flat fp32 buffer holding every trainable parameter (``param.data`` become views),
    with clip-by-global-norm folded into the update kernel.  One norm pass + one update pass per step."""

    def __init__(self, module, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=5.0,
                 reducer=None):
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
        self.step_count += 1
        norm_ptr = None
        if self.max_grad_norm and self.max_grad_norm > 0:
            capi.check(lib.ssbev_grad_norm(capi.ptr(g), n, capi.ptr(self.norm), capi.ptr(self._ws), self._ws.numel(),
                                           capi.stream()), "ssbev_grad_norm")
            norm_ptr = capi.ptr(self.norm)
        cfg = capi.AdamWCfg(self.lr, self.betas[0], self.betas[1], self.eps, self.wd, float(self.max_grad_norm or 0.0),
                            self.step_count)
        # parameters without a gradient this step (ablation modes) are skipped like torch.optim.AdamW skips grad None:
        # no weight decay, no moment decay.  Normally one range = one launch.
        for s, e in self.reducer.live_ranges():
            at = lambda t: C.c_void_p(t.data_ptr() + 4 * s)      # noqa: E731
            capi.check(lib.ssbev_adamw_step(at(self.flat_p), at(g), at(self.m), at(self.v), e - s,
                                            C.byref(cfg), norm_ptr, capi.stream()), "ssbev_adamw_step")
        return self.norm

    def state_dict(self):
        return {"m": self.m, "v": self.v, "step": self.step_count, "lr": self.lr}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count, self.lr = int(sd["step"]), float(sd["lr"])


def step_lr(base_lr, epoch, milestones=(20, 25), gamma=0.1):
    """mmcv 'step' LR policy of the reference config (lr_config step=[20, 25])."""
    return base_lr * (gamma ** sum(epoch >= m for m in milestones))


def train_step(model, optimizer, img_inputs, gt_occ):
    """One optimisation step of the hot path: forward, 4 losses, backward (+ overlapped gradient
    exchange), clip + AdamW.  Returns the loss dict (device scalars)."""
    optimizer.zero_grad()
    losses = model.forward_train(img_inputs=img_inputs, gt_occ=gt_occ)
    sum(v for k, v in losses.items() if k.startswith("loss")).backward()
    optimizer.step()
    return losses
