"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/ssbev.h declares."""
import os
import re

import pytest

from conftest import ROOT
from stereoscene_amd import capi


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ssbev.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssbev_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_and_exports_header_symbols():
    import __graft_entry__ as ge
    ge.build()
    lib = capi.load()
    names = header_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ssbev.h but not exported"
    assert set(names) == set(capi.SIGNATURES), "ctypes table and header disagree"
    assert lib.ssbev_version() >= 100
    assert lib.ssbev_build_arch() == b"gfx950"


def test_operators_fail_loudly_without_gpu():
    import torch
    from stereoscene_amd import functional as F
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    x = torch.zeros(1, 4, 2, 4, 4)
    w = torch.zeros(4, 4, 3, 3, 3)
    with pytest.raises(capi.SsbevError):
        F.conv3d(x, w, None, 1, 1)
    with pytest.raises(capi.SsbevError):
        F.gwc_warp(torch.zeros(1, 64, 2, 8), torch.zeros(1, 64, 2, 8), torch.ones(1), 4)


def test_workspace_queries_and_einval_on_host():
    """Entry points that do no device work can be exercised on the CPU box."""
    import ctypes as C
    lib = capi.load()
    d = capi.PoolDims()
    d.B, d.P, d.C, d.nx, d.ny, d.nz = 1, 1000, 128, 32, 32, 8
    assert lib.ssbev_pool_prepare_workspace(1000, C.byref(d)) >= 2 * 32 * 32 * 8 * 4 + 4000
    d.nx = 0
    assert lib.ssbev_pool_prepare_workspace(1000, C.byref(d)) == 0
    assert lib.ssbev_voxel_index(None, None, None, C.byref(d), None) == capi.EINVAL
    g = capi.GwcDims(1, 64, 32, 8, 2, 8, 2.0, 1)   # down != 1 is not supported
    assert lib.ssbev_gwc_warp_fwd(None, None, None, None, C.byref(g), None) == capi.EINVAL
    c = capi.ConvDims(1, 32, 32, 8, 8, 8, 8, 8, 8, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0)
    assert lib.ssbev_conv_packed_weight_elems(C.byref(c)) >= 27 * 32 * 32      # (the tap-split layout pads to 28 taps)
    assert lib.ssbev_conv_bwd_weight_workspace(C.byref(c)) >= 27 * 32 * 32 * 4
