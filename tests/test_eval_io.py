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
