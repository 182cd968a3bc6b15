"""CPU: samplers (bit-exact index lists against the reference's own sampler classes, tests/golden/samplers.npz) and the
epoch runner's hook logic (step LR, checkpoint rotation, save_best, resume) -- SURVEY 8(f4), host logic only."""
import os

import numpy as np
import torch

from conftest import load_golden
from oracle.make_golden_samplers import CASES, _DS
from stereoscene_amd import runner as R


def test_samplers_reproduce_reference_index_assignment():
    g = load_golden("samplers")
    checked = 0
    for (n, spg, world) in CASES:
        for two in (False, True):
            for epoch in (0, 3):
                seen = []
                for rank in range(world):
                    s = R.DistributedGroupSampler(_DS(n, two), samples_per_gpu=spg, num_replicas=world, rank=rank, seed=0)
                    s.set_epoch(epoch)
                    got = np.asarray(list(iter(s)), dtype=np.int64)
                    assert np.array_equal(got, g[f"group:{n}:{spg}:{world}:{int(two)}:{epoch}:{rank}"])
                    assert len(got) == len(s)
                    seen.extend(got.tolist())
                    checked += 1
                assert set(seen) == set(range(n))                       # every sample is visited by some rank
        for rank in range(world):
            got = np.asarray(list(iter(R.DistributedSampler(_DS(n), num_replicas=world, rank=rank))), dtype=np.int64)
            assert np.array_equal(got, g[f"dist:{n}:{world}:{rank}"])
            checked += 1
    assert checked == len(g)


def test_step_lr_policy():
    lrs = [R.step_lr(1e-4, e) for e in range(30)]
    assert lrs[0] == lrs[19] == 1e-4 and abs(lrs[20] - 1e-5) < 1e-18 and abs(lrs[24] - 1e-5) < 1e-18
    assert abs(lrs[25] - 1e-6) < 1e-18 and abs(lrs[29] - 1e-6) < 1e-18


def test_runner_hooks_checkpoint_rotation_best_and_resume(tmp_path):
    state = dict(w=torch.zeros(3), lr=None)
    scores = iter([10.0, 12.0, 11.0, 15.0])                          # evaluations at epochs 2, 4, 6, 8

    def step_fn(batch):
        state["w"] += batch
        return dict(loss=float(state["w"].sum()))

    def make(max_epochs):
        return R.EpochBasedRunner(step_fn, lambda lr: state.__setitem__("lr", lr), lambda: dict(state_dict=dict(w=state["w"].clone())),
                                  lambda ck: state.__setitem__("w", ck["state_dict"]["w"].clone()), str(tmp_path), base_lr=1e-4,
                                  lr_step=(3, 5), max_epochs=max_epochs, ckpt_interval=1, max_keep_ckpts=2,
                                  eval_fn=lambda: {"semkitti_combined_IoU": next(scores)}, eval_interval=2, log=lambda r: None)
    loader = [torch.ones(3), torch.ones(3)]
    run = make(5)
    hist = run.run(loader)
    assert [h["epoch"] for h in hist] == [1, 2, 3, 4, 5] and run.iter == 10
    assert [round(h["lr"] / 1e-4, 6) for h in hist] == [1, 1, 1, 0.1, 0.1]
    assert "eval" in hist[1] and "eval" in hist[3] and "eval" not in hist[2]
    files = sorted(os.listdir(tmp_path))
    assert "epoch_5.pth" in files and "epoch_4.pth" in files and "epoch_3.pth" not in files and "latest.pth" in files
    assert run.best_score == 12.0 and os.path.basename(run.best_ckpt) == "best_semkitti_combined_IoU_epoch_4.pth"
    assert not os.path.exists(tmp_path / "best_semkitti_combined_IoU_epoch_2.pth")
    # resume from latest.pth in a fresh runner and continue to epoch 8
    state["w"] = torch.full((3,), -1.0)
    run2 = make(8)
    meta = run2.resume()
    assert meta["epoch"] == 5 and run2.iter == 10 and float(state["w"][0]) == 10.0 and run2.best_score == 12.0
    hist2 = run2.run(loader)
    assert [h["epoch"] for h in hist2] == [6, 7, 8] and abs(hist2[0]["lr"] - 1e-6) < 1e-18
    assert run2.best_score == 15.0 and os.path.basename(run2.best_ckpt).endswith("epoch_8.pth")
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("epoch_")) == ["epoch_7.pth", "epoch_8.pth"]


def test_runner_kwargs_from_the_reference_config():
    import pytest
    from stereoscene_amd.registry import Config
    path = "/root/reference/projects/configs/occupancy/semantickitti/stereoscene.py"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present (GPU box)")
    cfg = Config.fromfile(path)
    kw = R.runner_kwargs_from_config(cfg)
    assert kw == dict(base_lr=1e-4, lr_step=(20, 25), lr_gamma=0.1, max_epochs=30, ckpt_interval=1, max_keep_ckpts=2,
                      eval_interval=2, save_best="semkitti_combined_IoU", rule="greater")
    assert cfg.optimizer["type"] == "AdamW" and cfg.optimizer_config["grad_clip"]["max_norm"] == 5


def test_loss_scaler_policy_matches_mmcv_semantics():
    """The reference's Fp16OptimizerHook drives mmcv's LossScaler (mmdet_train.py:131-134): static scale from
    ``fp16 = dict(loss_scale=512.)``, dynamic = halve on overflow / double after ``scale_window`` clean steps."""
    from stereoscene_amd.train import LossScaler
    s = LossScaler.from_config(dict(loss_scale=512.0))
    assert s.mode == "static" and s.loss_scale == 512.0
    s.update_scale(True)
    s.update_scale(False)
    assert s.loss_scale == 512.0                               # static: never changes
    d = LossScaler(init_scale=1024.0, mode="dynamic", scale_window=3)
    d.update_scale(True)                                       # iter 0 overflows -> 512
    assert d.loss_scale == 512.0 and d.last_overflow_iter == 0
    for _ in range(2):
        d.update_scale(False)                                  # iters 1, 2: (i - 0) % 3 != 0
    assert d.loss_scale == 512.0
    d.update_scale(False)                                      # iter 3: window complete -> doubled
    assert d.loss_scale == 1024.0
    assert LossScaler.from_config(dict(loss_scale="dynamic")).mode == "dynamic"
    sd = d.state_dict()
    e = LossScaler()
    e.load_state_dict(sd)
    assert e.loss_scale == d.loss_scale and e.cur_iter == d.cur_iter
