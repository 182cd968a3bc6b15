"""Submission writer and checkpoint loader of the evaluation side: pure host logic."""
import numpy as np
import torch
import torch.nn as nn

from stereoscene_amd.evaluate import LEARNING_MAP_INV, load_checkpoint, save_output_semantic_kitti


def test_label_file_layout_and_inverse_map(tmp_path):
    logits = torch.zeros(20, 4, 3, 2)
    cls = torch.arange(24).reshape(4, 3, 2) % 20
    logits.scatter_(0, cls[None], 1.0)
    path = save_output_semantic_kitti(logits, str(tmp_path), "08", "000123")
    assert path.endswith("sequences/08/predictions/000123.label")
    raw = np.fromfile(path, dtype=np.uint16)
    assert raw.shape == (24,) and np.array_equal(raw, LEARNING_MAP_INV[cls.reshape(-1).numpy()].astype(np.uint16))
    assert LEARNING_MAP_INV[0] == 0 and LEARNING_MAP_INV[1] == 10 and LEARNING_MAP_INV[19] == 81 and len(LEARNING_MAP_INV) == 20


def test_checkpoint_loader_filters_keys_outside_the_path():
    model = nn.Sequential(nn.Linear(3, 2), nn.Linear(2, 1))
    ref = {"state_dict": {"0.weight": torch.ones(2, 3), "0.bias": torch.zeros(2), "img_backbone.conv.weight": torch.zeros(1)}}
    missing, outside = load_checkpoint(model, ref)
    assert missing == ["1.bias", "1.weight"] and outside == ["img_backbone.conv.weight"]
    assert torch.equal(model[0].weight, torch.ones(2, 3))
    try:
        load_checkpoint(model, {"0.weight": torch.ones(5, 3)})
        raise AssertionError("shape mismatch not detected")
    except ValueError:
        pass


class _FakeOcc(nn.Module):
    """simple_test stand-in: logits that depend only on the sample (its gt shifted by the sample id)."""

    def simple_test(self, img_metas, img_inputs, gt_occ=None):
        sid = img_inputs
        pred = (gt_occ.clamp(max=19) + sid.view(-1, 1, 1, 1)) % 20
        return {"output_voxels": nn.functional.one_hot(pred.long(), 20).permute(0, 4, 1, 2, 3).float()}


def _fake_batches(ids, batch):
    g = torch.Generator().manual_seed(0)
    vox = torch.randint(0, 21, (64, 4, 4, 2), generator=g)
    vox[vox == 20] = 255
    for i in range(0, len(ids), batch):
        sel = torch.tensor(ids[i:i + batch])
        yield {"img_inputs": sel, "gt_occ": vox[sel]}


def test_distributed_evaluation_drops_the_padded_duplicates():
    """ADVICE r2 (evaluate.py:44): the per-rank blocks of runner.DistributedSampler are padded with the first samples; the
    counts summed over ranks must equal a single-process evaluation for dataset lengths that do not divide by the world."""
    from stereoscene_amd.evaluate import evaluate_counts, scores_from_counts
    from stereoscene_amd.runner import DistributedSampler
    model = _FakeOcc()
    for n, world, batch in ((7, 2, 2), (10, 3, 4), (5, 4, 1), (9, 3, 2)):
        want = evaluate_counts(model, _fake_batches(list(range(n)), batch), device="cpu")
        acc = 0
        for rank in range(world):
            smp = DistributedSampler(range(n), num_replicas=world, rank=rank)
            acc = acc + evaluate_counts(model, _fake_batches(list(smp), batch), device="cpu", sampler=smp)
        assert torch.equal(acc, want), (n, world, batch)
        # without the sampler the duplicates WOULD be counted (what the guard below protects against)
        naive = sum(evaluate_counts(model, _fake_batches(list(DistributedSampler(range(n), world, r)), batch), device="cpu")
                    for r in range(world))
        assert (n % world == 0) == bool(torch.equal(naive, want))
    assert "semkitti_combined_IoU" in scores_from_counts(want.numpy())


def _guard_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stereoscene_amd.evaluate import evaluate_counts
    from stereoscene_amd.runner import DistributedSampler
    try:
        evaluate_counts(_FakeOcc(), _fake_batches([0, 1], 1), device="cpu")
        out[rank] = "no error"
    except ValueError as e:
        smp = DistributedSampler(range(5), num_replicas=world, rank=rank)
        acc = evaluate_counts(_FakeOcc(), _fake_batches(list(smp), 2), device="cpu", sampler=smp)     # all-reduced over gloo
        out[rank] = ("raised", acc.tolist())
    dist.destroy_process_group()


def test_evaluate_refuses_the_old_signature_under_a_process_group():
    import socket
    import torch.multiprocessing as mp
    from stereoscene_amd.evaluate import evaluate_counts
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(_guard_worker, args=(2, port, out), nprocs=2, join=True)
    want = evaluate_counts(_FakeOcc(), _fake_batches(list(range(5)), 2), device="cpu").tolist()
    assert out[0][0] == "raised" and out[1][0] == "raised"
    assert out[0][1] == want and out[1][1] == want
