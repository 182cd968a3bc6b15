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
    assert lib.ssbev_pool_prepare_workspace(1000, C.byref(d)) >= 2 * 1000 * 4 + 2049 * 4     # (key, id) pairs + digit bases
    d.nx = 0
    assert lib.ssbev_pool_prepare_workspace(1000, C.byref(d)) == 0
    assert lib.ssbev_voxel_index(None, None, None, C.byref(d), None) == capi.EINVAL
    g = capi.GwcDims(1, 64, 32, 8, 2, 8, 2.0, 1)   # down != 1 is not supported
    assert lib.ssbev_gwc_warp_fwd(None, None, None, None, C.byref(g), None) == capi.EINVAL
    c = capi.ConvDims(1, 32, 32, 8, 8, 8, 8, 8, 8, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0)
    assert lib.ssbev_conv_packed_weight_elems(C.byref(c)) >= 27 * 32 * 32      # (the tap-split layout pads to 28 taps)
    assert lib.ssbev_conv_bwd_weight_workspace(C.byref(c)) >= 27 * 32 * 32 * 4


def test_round3_entry_points_validate_their_arguments_on_host():
    """Workspace queries and argument checks of the round-3 entry points (no device work): the fused depth loss, the loss
    tail, the BRI shell, the frustum geometry, the strided GroupNorm and the cost-volume backward workspace."""
    import ctypes as C
    lib = capi.load()
    npix = 2 * 48 * 160
    ws = lib.ssbev_depth_bce_workspace(2, 48, 160)
    assert ws >= npix * 4 + (npix // 256) * lib.ssbev_bri_shell_chunks() * 16          # labels + (bce, count) partials
    assert lib.ssbev_depth_bce_workspace(0, 48, 160) == 0
    assert lib.ssbev_depth_bce_fwd(None, None, None, 1, 192, 48, 160, 8, 1.75, 0.5, 1.0, None, 0, None) == capi.EINVAL
    assert lib.ssbev_depth_bce_bwd(None, None, None, None, 1, 192, 48, 160, 8, 1.75, 0.5, 1.0, None, None) == capi.EINVAL
    assert lib.ssbev_occ_loss_tail(None, 1.0, 1.0, 1.0, None, None, None) == capi.EINVAL
    assert lib.ssbev_bri_shell_chunks() >= 1
    nul = [None] * 12
    assert lib.ssbev_bri_shell_pre_fwd(*nul, 1, 192, 7680, None) == capi.EINVAL
    assert lib.ssbev_bri_shell_post_fwd(None, None, None, None, 1, 192, 7680, None) == capi.EINVAL
    assert lib.ssbev_bri_shell_post_bwd(*[None] * 6, 1, 192, 7680, None) == capi.EINVAL
    assert lib.ssbev_bri_shell_pre_bwd(*[None] * 15, 0, 192, 7680, None) == capi.EINVAL
    gd = capi.GeomDims(1, 1, 192, 48, 160)
    assert lib.ssbev_frustum_geometry(*[None] * 9, C.byref(gd), None) == capi.EINVAL
    # strided GroupNorm output / gradient: the row stride must hold the slice and keep 16-byte alignment
    nd = capi.NormDims(1, 128, 32, 1000, 1e-5, 1, 0, 0, 384, 0)
    assert lib.ssbev_groupnorm_workspace(C.byref(nd)) > 0
    for bad in (64, 130):
        nd.ld_y = bad
        assert lib.ssbev_groupnorm_workspace(C.byref(nd)) == 0
    g = capi.GwcDims(1, 64, 32, 192, 48, 160, 1.0, 1)
    assert lib.ssbev_gwc_warp_bwd_workspace(C.byref(g)) >= 2 * 2 * 48 * 160 * 64 * 4   # >= two chunks of both partial gradients


def test_bf16_storage_entry_points_refuse_a_mismatched_precision_on_host():
    """Round 5 ("a wrong precision value is silent garbage"): bf16-storage problems (ssbev_conv_dims.precision 2 / 3) are
    served by the typed ``*_bf16`` entry points only, fp32 / bf16-operand problems (0 / 1) by the ``float*`` ones; either family
    answers SSBEV_EINVAL for the other's modes before touching a pointer."""
    import ctypes as C
    lib = capi.load()
    fake = C.c_void_p(256)                # never dereferenced: the calls are refused on their arguments
    for prec in (0, 1, 2, 3):
        c = capi.ConvDims(1, 32, 32, 8, 8, 8, 8, 8, 8, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, prec)
        storage = prec in (2, 3)
        if storage:
            assert lib.ssbev_conv_fwd(fake, fake, None, fake, C.byref(c), None) == capi.EINVAL
            assert lib.ssbev_conv_bwd_data(fake, fake, fake, C.byref(c), None) == capi.EINVAL
            assert lib.ssbev_conv_bwd_weight(fake, fake, fake, C.byref(c), fake, 1 << 30, None) == capi.EINVAL
        else:
            assert lib.ssbev_conv_fwd_bf16(fake, fake, None, fake, C.byref(c), None) == capi.EINVAL
            assert lib.ssbev_conv_bwd_data_bf16(fake, fake, fake, C.byref(c), None) == capi.EINVAL
            assert lib.ssbev_conv_bwd_weight_bf16(fake, fake, fake, C.byref(c), fake, 1 << 30, None) == capi.EINVAL
    c3 = capi.ConvDims(1, 32, 32, 8, 8, 8, 8, 8, 8, 3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 3)
    assert lib.ssbev_conv_bwd_weight_bf16(fake, fake, fake, C.byref(c3), fake, 1 << 30, None) == capi.EINVAL   # fp32-result mode has no wgrad
    # normalisation: the BatchNorm running statistics ride in ssbev_norm_ext; NULL ext is allowed, bad dims are not
    n = capi.NormDims(1, 30, 2, 100, 1e-5, 0, 0, 0, 0, 0, 0)          # C % 4 != 0
    assert lib.ssbev_groupnorm_fwd_ext(fake, fake, fake, None, fake, fake, fake, None, C.byref(n), None, fake, 1 << 20, None) == capi.EINVAL


def test_only_the_switch_table_reads_the_environment():
    """no getenv on the launch path.  Every SSBEV_* switch of csrc/ goes through ssbev_env (capi.hip: one lookup
    per name per process, a table afterwards; ssbev_env_refresh in the C ABI) or, for tuning hooks, through ssbev_tune, which the
    product build compiles to a null constant -- so `getenv` may be an undefined symbol of capi.o only, and no source but capi.hip
    may name it."""
    import glob
    import re
    import subprocess
    csrc = os.path.join(ROOT, "stereoscene_amd", "csrc")
    for src in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        text = open(src).read()
        hits = re.findall(r"(?<![A-Za-z_:])getenv\(", text)
        assert not hits or os.path.basename(src) == "capi.hip", (src, len(hits))
    objs = glob.glob(os.path.join(ROOT, "stereoscene_amd", "_lib", "*.o"))
    assert objs, "in-tree build expected (python -m stereoscene_amd.build)"
    for o in objs:
        und = subprocess.run(["nm", "-u", o], capture_output=True, text=True).stdout
        if re.search(r"\bgetenv\b", und):
            assert os.path.basename(o) == "capi.o", o
    lib = capi.load()
    lib.ssbev_env_refresh()          # callable without a GPU; returns nothing
