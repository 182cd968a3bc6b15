"""Training-loop side of SURVEY 8(f4): the samplers that decide which rank sees which sample
(projects/mmdet3d_plugin/datasets/samplers/group_sampler.py:13-110, distributed_sampler.py:9-41 -- the partitioning that
8(e) shards the path by) and an ``EpochBasedRunner`` with the hooks stereoscene.py:203-225 configures: step LR policy
(milestones 20 / 25, x0.1), gradient clipping inside the fused optimizer, ``checkpoint_config(interval=1,
max_keep_ckpts=2)``, ``evaluation(interval=2, save_best='semkitti_combined_IoU', rule='greater')`` and resume.
Host logic only; the compute lives behind ``step_fn`` / ``eval_fn`` (``train.train_step``, ``evaluate.evaluate``)."""
import math
import os
import shutil

import numpy as np
import torch


class DistributedGroupSampler:
    """Per-epoch deterministic shuffle inside aspect-ratio groups (``dataset.flag``), padded so that every rank gets the
    same number of whole ``samples_per_gpu`` chunks, chunks shuffled again, then rank r takes the r-th contiguous block."""

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=1, rank=0, seed=0):
        self.dataset, self.samples_per_gpu, self.num_replicas, self.rank = dataset, samples_per_gpu, num_replicas, rank
        self.epoch, self.seed = 0, (seed if seed is not None else 0)
        self.flag = np.asarray(getattr(dataset, "flag", np.zeros(len(dataset), dtype=np.uint8)))
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = sum(int(math.ceil(s * 1.0 / samples_per_gpu / num_replicas)) * samples_per_gpu
                               for s in self.group_sizes)
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch + self.seed)
        indices = []
        for i, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            members = np.where(self.flag == i)[0]
            order = members[torch.randperm(int(size), generator=g).numpy()].tolist()
            target = int(math.ceil(size * 1.0 / self.samples_per_gpu / self.num_replicas)) * self.samples_per_gpu * self.num_replicas
            padded = list(order)
            while len(padded) < target:                           # whole copies, then a prefix
                padded.extend(order[:target - len(padded)])
            indices.extend(padded)
        assert len(indices) == self.total_size
        spg = self.samples_per_gpu
        chunk_order = torch.randperm(len(indices) // spg, generator=g).tolist()
        indices = [indices[j] for c in chunk_order for j in range(c * spg, (c + 1) * spg)]
        offset = self.num_samples * self.rank
        return iter(indices[offset:offset + self.num_samples])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class DistributedSampler:
    """Evaluation sampler: no shuffle, index list tiled up to a multiple of the world size, rank r takes the r-th
    CONTIGUOUS block (not the strided one of torch's sampler)."""

    def __init__(self, dataset, num_replicas=1, rank=0, shuffle=False, seed=0):
        if shuffle:
            raise AssertionError("the reference's DistributedSampler asserts on shuffle=True")
        self.n, self.num_replicas, self.rank = len(dataset), num_replicas, rank
        self.num_samples = int(math.ceil(self.n / num_replicas))
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        idx = list(range(self.n))
        idx = (idx * math.ceil(self.total_size / len(idx)))[:self.total_size]
        per = self.total_size // self.num_replicas
        return iter(idx[self.rank * per:(self.rank + 1) * per])

    def __len__(self):
        return self.num_samples


def step_lr(base_lr, epoch, step=(20, 25), gamma=0.1):
    """mmcv ``StepLrUpdaterHook`` by epoch: base_lr * gamma ** (number of milestones <= epoch)."""
    return base_lr * gamma ** sum(1 for s in step if epoch >= s)


class EpochBasedRunner:
    """``runner = dict(type='EpochBasedRunner', max_epochs=30)`` with the config's hooks.

    step_fn(batch) -> dict of scalar losses (performs forward / backward / optimizer step);
    eval_fn() -> dict of metrics (called every ``eval_interval`` epochs);  set_lr(lr) applies the schedule;
    state_fn() -> dict to checkpoint, load_fn(state) restores it."""

    def __init__(self, step_fn, set_lr, state_fn, load_fn, work_dir, base_lr=1e-4, lr_step=(20, 25), lr_gamma=0.1,
                 max_epochs=30, ckpt_interval=1, max_keep_ckpts=2, eval_fn=None, eval_interval=2,
                 save_best="semkitti_combined_IoU", rule="greater", rank=0, log=print):
        self.step_fn, self.set_lr, self.state_fn, self.load_fn, self.eval_fn = step_fn, set_lr, state_fn, load_fn, eval_fn
        self.work_dir, self.base_lr, self.lr_step, self.lr_gamma = work_dir, base_lr, tuple(lr_step), lr_gamma
        self.max_epochs, self.ckpt_interval, self.max_keep_ckpts = max_epochs, ckpt_interval, max_keep_ckpts
        self.eval_interval, self.save_best, self.rule, self.rank, self.log = eval_interval, save_best, rule, rank, log
        self.epoch, self.iter, self.best_score, self.best_ckpt, self.history = 0, 0, None, None, []
        if rank == 0:
            os.makedirs(work_dir, exist_ok=True)

    # -- checkpoint hook ---------------------------------------------------------------------------------
    def _ckpt_path(self, epoch):
        return os.path.join(self.work_dir, f"epoch_{epoch}.pth")

    def save_checkpoint(self):
        if self.rank != 0:
            return None
        path = self._ckpt_path(self.epoch)
        torch.save(dict(meta=dict(epoch=self.epoch, iter=self.iter, best_score=self.best_score, best_ckpt=self.best_ckpt),
                        **self.state_fn()), path)
        shutil.copyfile(path, os.path.join(self.work_dir, "latest.pth"))
        if self.max_keep_ckpts > 0:                      # mmcv CheckpointHook: drop epoch_{e - k * interval}, k >= max_keep
            for e in range(self.epoch - self.max_keep_ckpts * self.ckpt_interval, 0, -self.ckpt_interval):
                old = self._ckpt_path(e)
                if os.path.exists(old):
                    os.remove(old)
                else:
                    break
        return path

    def resume(self, path=None):
        path = path or os.path.join(self.work_dir, "latest.pth")
        # weights_only=False: the checkpoint holds the optimizer's python scalars next to tensors, and it is a file this
        # runner wrote itself (never load a checkpoint from an untrusted source this way)
        ck = torch.load(path, map_location="cpu", weights_only=False)
        meta = ck.pop("meta")
        self.epoch, self.iter = meta["epoch"], meta["iter"]
        self.best_score, self.best_ckpt = meta.get("best_score"), meta.get("best_ckpt")
        self.load_fn(ck)
        return meta

    # -- evaluation hook ---------------------------------------------------------------------------------
    def _evaluate(self):
        metrics = self.eval_fn()
        score = metrics.get(self.save_best)
        better = score is not None and (self.best_score is None or
                                        (score > self.best_score if self.rule == "greater" else score < self.best_score))
        if better and self.rank == 0:
            if self.best_ckpt and os.path.exists(self.best_ckpt):
                os.remove(self.best_ckpt)
            self.best_score = score
            self.best_ckpt = os.path.join(self.work_dir, f"best_{self.save_best}_epoch_{self.epoch}.pth")
            torch.save(dict(meta=dict(epoch=self.epoch, iter=self.iter, best_score=score), **self.state_fn()), self.best_ckpt)
        return metrics

    # -- the loop ----------------------------------------------------------------------------------------
    def run(self, loader, sampler=None):
        while self.epoch < self.max_epochs:
            lr = step_lr(self.base_lr, self.epoch, self.lr_step, self.lr_gamma)
            self.set_lr(lr)
            if sampler is not None:
                sampler.set_epoch(self.epoch)
            sums, n = {}, 0
            for batch in loader:
                losses = self.step_fn(batch)
                for k, v in losses.items():        # running sums stay on the device: a float() here is a host sync per step
                    v = v.detach() if torch.is_tensor(v) else v
                    sums[k] = sums[k] + v if k in sums else v
                n += 1
                self.iter += 1
            self.epoch += 1
            rec = dict(epoch=self.epoch, lr=lr, **{k: float(v) / max(n, 1) for k, v in sums.items()})
            # evaluation first: the epoch checkpoint / latest.pth then carry this epoch's best_score / best_ckpt, so a
            # resume() never overwrites a better 'best_*' file with a worse one
            if self.eval_fn is not None and self.eval_interval and self.epoch % self.eval_interval == 0:
                rec["eval"] = self._evaluate()
            if self.ckpt_interval and self.epoch % self.ckpt_interval == 0:
                self.save_checkpoint()
            self.history.append(rec)
            self.log(rec)
        return self.history


def runner_kwargs_from_config(cfg):
    """The hook configuration of a reference config (stereoscene.py:203-225) as ``EpochBasedRunner`` keyword arguments:
    ``optimizer.lr``, ``lr_config(policy='step', step=[...])``, ``checkpoint_config``, ``evaluation``, ``runner.max_epochs``."""
    lr_cfg = dict(cfg.get("lr_config", {}))
    if lr_cfg.get("policy", "step") != "step":
        raise NotImplementedError(f"lr policy {lr_cfg.get('policy')!r}: the hot-path configs use 'step'")
    ck, ev, rn = dict(cfg.get("checkpoint_config", {})), dict(cfg.get("evaluation", {})), dict(cfg.get("runner", {}))
    if rn.get("type", "EpochBasedRunner") != "EpochBasedRunner":
        raise NotImplementedError(rn.get("type"))
    return dict(base_lr=cfg["optimizer"]["lr"], lr_step=tuple(lr_cfg.get("step", ())), lr_gamma=lr_cfg.get("gamma", 0.1),
                max_epochs=rn.get("max_epochs", 1), ckpt_interval=ck.get("interval", 1), max_keep_ckpts=ck.get("max_keep_ckpts", -1),
                eval_interval=ev.get("interval", 1), save_best=ev.get("save_best"), rule=ev.get("rule", "greater"))


def optimizer_from_config(model, cfg, reducer=None):
    """``optimizer = dict(type='AdamW', lr, weight_decay)`` + ``optimizer_config.grad_clip.max_norm`` -> train.FlatAdamW.
    A ``fp16 = dict(loss_scale=...)`` entry (mmdet_train.py:131-134 wraps the optimizer step in ``Fp16OptimizerHook``) selects
    the mixed-precision policy: bf16 conv arithmetic (functional.set_precision) + the hook's loss scaler; fp32 master state."""
    from .train import FlatAdamW, LossScaler
    from . import functional as F
    scaler = None
    if cfg.get("fp16") is not None:
        scaler = LossScaler.from_config(cfg.get("fp16"))
        F.set_precision("bf16")
    opt = dict(cfg["optimizer"])
    if opt.pop("type") != "AdamW":
        raise NotImplementedError("the hot-path configs train with AdamW")
    clip = (cfg.get("optimizer_config") or {}).get("grad_clip") or {}
    if clip and clip.get("norm_type", 2) != 2:
        raise NotImplementedError("only the 2-norm clip is fused")
    return FlatAdamW(model, lr=opt["lr"], weight_decay=opt.get("weight_decay", 0.01), betas=tuple(opt.get("betas", (0.9, 0.999))),
                     eps=opt.get("eps", 1e-8), max_grad_norm=float(clip.get("max_norm", 0.0)), reducer=reducer, loss_scaler=scaler)
