import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _fresh_library_switches(request):
    """The HIP library caches its SSBEV_* environment switches (ssbev_env_refresh in include/ssbev.h); GPU tests flip some with
    monkeypatch.setenv inside one process, so every GPU test starts from -- and leaves behind -- an empty cache."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from stereoscene_amd import capi
    capi.load().ssbev_env_refresh()
    yield
    capi.load().ssbev_env_refresh()


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def state_dict_from_manifest(g, prefix_filter=""):
    """Rebuild a reference state dict from 'shape:<key>' manifest entries with the fill-by-key rule."""
    from stereoscene_amd import synthetic as S
    sd = {}
    for k, shp in g.items():
        if not k.startswith("shape:"):
            continue
        key = k[len("shape:"):]
        if not key.startswith(prefix_filter):
            continue
        leaf = key.rsplit(".", 1)[-1]
        dtype = torch.int64 if leaf == "num_batches_tracked" else torch.float32
        t = torch.zeros(tuple(int(v) for v in shp), dtype=dtype)
        v = S.fill_value_for(key, t)
        sd[key] = t if v is None else v.to(dtype)
    return sd


@pytest.fixture(scope="session")
def golden():
    return load_golden
