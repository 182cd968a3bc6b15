"""torch.autograd wrappers around the C ABI (stereoscene_amd.capi).  Host-side mirror of the
operator boundary the reference sits behind:

    voxel_index / pool_prepare / bev_pool  <-  VT:432-476 + mmdet3d.ops.bev_pool (VT:473)
    lift_splat                             <-  VT:517-523 fused
    gwc_warp                               <-  VT:104-114 + VT:128-156 fused
    conv_nd / conv_transpose_nd            <-  ATen conv3d / conv_transpose3d / conv2d call sites

Tensors are logical NC[D]HW (what the reference modules exchange) held in channels-last memory,
which is the layout the kernels read and write; nothing is permuted on the way in or out.
"""
import ctypes as C

import os

import torch

from . import capi, streams

# -------------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py's roofline leg): HIP events on the launch stream
# -------------------------------------------------------------------------------------------------


class KernelTimer:
    """Collects (kernel family, algorithmic flops, algorithmic bytes, executed flops) per C-ABI launch.  Families in
    ``families`` (None = all) are additionally bracketed by HIP events on the launch stream; the others are only
    COUNTED (no event traffic), which is what bench.py's whole-step roofline floor needs."""

    def __init__(self, families=None):
        self.records = []
        self.counts = {}              # family -> dict(launches, flops, bytes, executed): every span, timed or not
        self.families = families      # None = every instrumented family; a set restricts the event traffic

    def span(self, family, flops, nbytes=0.0, tag=None, executed=None):
        c = self.counts.setdefault(family, dict(launches=0, flops=0.0, bytes=0.0, executed=0.0))
        c["launches"] += 1
        c["flops"] += flops
        c["bytes"] += nbytes
        c["executed"] += flops if executed is None else executed
        if self.families is not None and family.split(":")[0] not in self.families:      # "family:instance" = one kernel symbol
            return _NOSPAN
        return _Span(self, family, flops, nbytes, tag, flops if executed is None else executed)

    def by_tag(self):
        """{(family, tag): dict(launches, flops, ms)} -- the per-layer table of tools/layer_table.py."""
        torch.cuda.synchronize()
        out = {}
        for fam, flops, nbytes, a, b, tag, ex in self.records:
            d = out.setdefault((fam, tag), dict(launches=0, flops=0.0, bytes=0.0, ms=0.0, executed=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
            d["executed"] += ex
            d["ms"] += a.elapsed_time(b)
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for fam, flops, nbytes, a, b, _tag, ex in self.records:
            d = out.setdefault(fam, dict(launches=0, flops=0.0, bytes=0.0, ms=0.0, executed=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
            d["executed"] += ex
            d["ms"] += a.elapsed_time(b)
        return out


class _Span:
    def __init__(self, timer, family, flops, nbytes, tag=None, executed=0.0):
        self.t, self.family, self.flops, self.nbytes, self.tag, self.executed = timer, family, flops, nbytes, tag, executed

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()          # torch's current stream == the stream handed to the C ABI

    def __exit__(self, *exc):
        self.b.record()
        self.t.records.append((self.family, self.flops, self.nbytes, self.a, self.b, self.tag, self.executed))


class _NoSpan:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


KERNEL_TIMER = None
_NOSPAN = _NoSpan()


def _span(family, flops, nbytes=0.0, tag=None, executed=None):
    return KERNEL_TIMER.span(family, flops, nbytes, tag, executed) if KERNEL_TIMER is not None else _NOSPAN


def _conv_tag(d, role):
    return (f"{role:5s} {'T' if d.transposed else 'C'} {d.Cin:3d}->{d.Cout:3d} in {d.Di}x{d.Hi}x{d.Wi} "
            f"k{d.kd}{d.kh}{d.kw} s{d.sd}{d.sh}{d.sw} d{d.dd}{d.dh}{d.dw}")


def _conv_family(lib, d, mode):
    """Span family of a direct conv launch: the <= 32-channel layers on conv_taph_kernel execute 2/3 of the operator's
    multiply-adds (F(2,3) along h), on conv_tapdh_kernel 4/9 (F(2,3) along d and h), which bench.py's executed-FLOP figure
    accounts for."""
    if KERNEL_TIMER is None:
        return "conv_gather"
    return {2: "conv_tap_h", 9: "conv_tap_dh"}.get(lib.ssbev_conv_kernel_class(C.byref(d), mode), "conv_gather")


# conv_taph_kernel runs F(2,3) along h inside the direct kernel: 2/3 of the operator's multiply-adds are executed;
# conv_tapdh_kernel (round 4) F(2,3) along d and h: 4/9
_EXEC_DIV = {"conv_gather": 1.0, "conv_tap_h": 1.5, "conv_tap_dh": 2.25}


def conv_bytes(d):
    """Algorithmic bytes of one conv problem: every input, weight and output element touched once."""
    taps = d.kd * d.kh * d.kw
    return 4.0 * (d.B * d.Di * d.Hi * d.Wi * d.Cin + d.B * d.Do * d.Ho * d.Wo * d.Cout + taps * d.Cin * d.Cout)


def conv_flops(d):
    """Algorithmic flops (2*MAC) of one conv / deconv problem (any of fwd, dgrad, wgrad)."""
    taps = d.kd * d.kh * d.kw
    vox = d.B * (d.Di * d.Hi * d.Wi if d.transposed else d.Do * d.Ho * d.Wo)
    return 2.0 * vox * d.Cin * d.Cout * taps


# -------------------------------------------------------------------------------------------------
# layout helpers
# -------------------------------------------------------------------------------------------------


def to_cl(x):
    """Logical [B,C,*spatial] -> contiguous [B,*spatial,C] buffer (no copy if already channels-last)."""
    perm = (0,) + tuple(range(2, x.dim())) + (1,)
    return x.permute(perm).contiguous()


def from_cl(buf):
    """Contiguous [B,*spatial,C] buffer -> logical [B,C,*spatial] view (channels-last strides)."""
    n = buf.dim()
    return buf.permute((0, n - 1) + tuple(range(1, n - 1)))


def _f32(t, what):
    """Input of an fp32 operator.  In the bf16 storage mode a bf16 activation is widened here (an fp32 island: the autograd
    engine narrows the gradient the operator returns back to the input's dtype); anything else is an error."""
    if t.dtype == torch.bfloat16 and PRECISION == "bf16":
        return t.float()
    if t.dtype != torch.float32:
        raise capi.SsbevError(f"{what}: fp32 expected, got {t.dtype}")
    return t


def _act(t, what):
    """Input of an operator with fp32 AND bf16 kernels: returns (tensor, io_dtype code of the C ABI)."""
    if t.dtype == torch.bfloat16:
        return t, 1
    if t.dtype != torch.float32:
        raise capi.SsbevError(f"{what}: fp32 or bf16 expected, got {t.dtype}")
    return t, 0


_WS_NONE = {}


def _ws(nbytes, device):
    """Caller-owned workspace of an entry point.  Most calls need none (size 0: the kernels never touch the pointer); those share
    one 16-byte buffer per device instead of an allocator round trip each (~600 per step: tools/host_profile.py)."""
    n = int(nbytes)
    if n == 0:
        key = (device.type, device.index) if isinstance(device, torch.device) else device
        t = _WS_NONE.get(key)
        if t is None:
            t = _WS_NONE[key] = torch.empty(16, dtype=torch.uint8, device=device)
        return t
    return torch.empty(max(n, 16), dtype=torch.uint8, device=device)


# -------------------------------------------------------------------------------------------------
# frustum -> voxel scatter
# -------------------------------------------------------------------------------------------------


def _pool_dims(B, P, Cch, nx, ny, nz, origin=(0, 0, 0), dx=(1, 1, 1)):
    d = capi.PoolDims()
    d.B, d.P, d.C, d.nx, d.ny, d.nz = int(B), int(P), int(Cch), int(nx), int(ny), int(nz)
    for i in range(3):
        d.origin[i] = float(origin[i])
        d.dx[i] = float(dx[i])
    return d


def grid_origin(bx, dx):
    """fp32(bx - dx/2) computed in fp32 exactly as VT:441 does."""
    return (bx.detach().float().cpu() - dx.detach().float().cpu() / 2.0)


GEOM_FUSED = os.environ.get("SSBEV_GEOM_FUSED", "1") != "0"     # per-point chain of get_geometry in one kernel (0 = ATen passes)


def frustum_geometry(frustum, m1, t0, m2, t2, tr, m3, t3):
    """Per-point part of get_geometry (BD:123-156): frustum [D,H,W,3]; m1, m2 [B,N,3,3]; t0, tr (, t2) [B,N,3]; m3 [B,3,3]
    (, t3 [B,3]) -> geom [B,N,D,H,W,3], bit-identical to the broadcast tensor expression (ssbev_frustum_geometry)."""
    lib = capi.load()
    B, N = m1.shape[:2]
    D, H, W = frustum.shape[:3]
    dev = m1.device

    def c(t):
        return None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()
    fr, m1, t0, m2, t2, tr, m3, t3 = (c(t) for t in (frustum, m1, t0, m2, t2, tr, m3, t3))
    out = torch.empty(B, N, D, H, W, 3, dtype=torch.float32, device=dev)
    d = capi.GeomDims(B, N, D, H, W)
    capi.check(lib.ssbev_frustum_geometry(capi.ptr(fr), capi.ptr(m1), capi.ptr(t0), capi.ptr(m2), capi.ptr(t2), capi.ptr(tr),
                                          capi.ptr(m3), capi.ptr(t3), capi.ptr(out), C.byref(d), capi.stream()),
               "ssbev_frustum_geometry")
    return out


def voxel_index(geom, bx, dx, nx, return_idx=False, grid_host=None):
    """geom [B, ..., 3] fp32 (cuda) -> vox int32 [B*P] (linear voxel or -1) [, idx3 int32 [B*P,3]].
    ``grid_host`` = (origin, dx, n) python lists, if the caller already holds host copies of the grid parameters."""
    lib = capi.load()
    B = geom.shape[0]
    g = _f32(geom, "voxel_index").reshape(B, -1, 3).contiguous()
    P = g.shape[1]
    if grid_host is None:
        grid_host = (grid_origin(bx, dx).tolist(), dx.detach().float().cpu().tolist(), [int(v) for v in nx.tolist()])
    origin, dxl, n = grid_host
    d = _pool_dims(B, P, 1, n[0], n[1], n[2], origin, dxl)
    vox = torch.empty(B * P, dtype=torch.int32, device=geom.device)
    idx3 = torch.empty(B * P, 3, dtype=torch.int32, device=geom.device) if return_idx else None
    capi.check(lib.ssbev_voxel_index(capi.ptr(g), capi.ptr(vox), capi.ptr(idx3), C.byref(d), capi.stream()),
               "ssbev_voxel_index")
    return (vox, idx3) if return_idx else vox


class _Order(torch.Tensor):
    """The `order` tensor of a CSR table carrying the compacted long-voxel list of ssbev_pool_prepare2 (`.long_list`), so that the
    (starts, order) pairs handed around by callers keep their shape."""
    long_list = None


def pool_prepare(vox, B, nx, ny, nz):
    """CSR table (starts int32 [NV+1], order int32 [n]) of the voxel -> ascending point lists; `order.long_list` (int32: count,
    then the ids of the voxels with more than 32 points) is the work list of the fused gather's long-list waves."""
    lib = capi.load()
    n = vox.numel()
    d = _pool_dims(B, max(n // max(B, 1), 0), 1, nx, ny, nz)
    nv = B * nx * ny * nz
    starts = torch.empty(nv + 1, dtype=torch.int32, device=vox.device)
    order = torch.empty(max(n, 1), dtype=torch.int32, device=vox.device).as_subclass(_Order)
    order.long_list = torch.empty(lib.ssbev_pool_long_list_elems(n), dtype=torch.int32, device=vox.device)
    ws = _ws(lib.ssbev_pool_prepare_workspace(n, C.byref(d)), vox.device)
    capi.check(lib.ssbev_pool_prepare2(capi.ptr(vox), n, capi.ptr(starts), capi.ptr(order), capi.ptr(order.long_list), C.byref(d),
                                       capi.ptr(ws), ws.numel(), capi.stream()), "ssbev_pool_prepare2")
    return starts, order


class _BevPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, coords, B, nz, nx, ny):
        lib = capi.load()
        feats = _f32(feats, "bev_pool").contiguous()
        n, Cch = feats.shape
        d = _pool_dims(B, 0, Cch, nx, ny, nz)
        c32 = coords.to(torch.int32).contiguous()
        vox = torch.empty(max(n, 1), dtype=torch.int32, device=feats.device)
        capi.check(lib.ssbev_coords_to_vox(capi.ptr(c32), n, capi.ptr(vox), C.byref(d), capi.stream()),
                   "ssbev_coords_to_vox")
        vox = vox[:n]
        starts, order = pool_prepare(vox, B, nx, ny, nz)
        out = torch.empty(B, nx, ny, nz, Cch, dtype=torch.float32, device=feats.device)
        capi.check(lib.ssbev_bev_pool_fwd(capi.ptr(feats), capi.ptr(starts), capi.ptr(order), capi.ptr(out),
                                          C.byref(d), capi.stream()), "ssbev_bev_pool_fwd")
        ctx.save_for_backward(vox)
        ctx.dims = (B, Cch, nx, ny, nz, n)
        # upstream returns [B, C, nz, nx, ny]
        return out.permute(0, 4, 3, 1, 2)

    @staticmethod
    def backward(ctx, gout):
        lib = capi.load()
        (vox,) = ctx.saved_tensors
        B, Cch, nx, ny, nz, n = ctx.dims
        g = gout.permute(0, 3, 4, 2, 1).contiguous()     # -> [B, nx, ny, nz, C]
        d = _pool_dims(B, 0, Cch, nx, ny, nz)
        gf = torch.empty(n, Cch, dtype=torch.float32, device=gout.device)
        capi.check(lib.ssbev_bev_pool_bwd(capi.ptr(g), capi.ptr(vox), n, capi.ptr(gf), C.byref(d), capi.stream()),
                   "ssbev_bev_pool_bwd")
        return gf, None, None, None, None, None


def bev_pool(feats, coords, B, D, H, W):
    """Drop-in for ``mmdet3d.ops.bev_pool.bev_pool(x, geom_feats, B, D, H, W)`` (VT:473):
    feats [n,C], coords [n,4]=(ix,iy,iz,b), D=nz, H=nx, W=ny -> [B, C, D, H, W]."""
    if feats.shape[0] != coords.shape[0]:
        raise ValueError("bev_pool: feats and coords disagree on the number of points")
    return _BevPool.apply(feats, coords, int(B), int(D), int(H), int(W))


# role-split gather over the compacted long-voxel list (round 4); SSBEV_GATHER_SPLIT=0: the one-kernel gather5
GATHER_SPLIT = os.environ.get("SSBEV_GATHER_SPLIT", "1") != "0"


class _LiftSplat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, feat, vox, starts, order, B, N, grid):
        lib = capi.load()
        nx, ny, nz = grid
        BN, D, H, W = depth.shape
        Cch = feat.shape[1]
        depth = _f32(depth, "lift_splat").contiguous()
        feat_cl = to_cl(_f32(feat, "lift_splat"))                 # [BN, H, W, C]
        d = _pool_dims(B, N * D * H * W, Cch, nx, ny, nz)
        l = capi.LiftDims(N, D, H * W)
        out = torch.empty(B, nx, ny, nz, Cch, dtype=torch.float32, device=depth.device)
        # algorithmic bytes: depth + features + the int32 voxel table in, the BEV volume out
        nby = 4.0 * (depth.numel() + feat_cl.numel() + order.numel() + starts.numel() + out.numel())
        with _span("lift_splat", 2.0 * depth.numel() * Cch, nby, f"fwd   lift_splat D={D} HW={H * W} C={Cch} grid={nx}x{ny}x{nz}"):
            capi.check(lib.ssbev_lift_splat_fwd2(capi.ptr(depth), capi.ptr(feat_cl), capi.ptr(starts), capi.ptr(order),
                                                 capi.ptr(getattr(order, "long_list", None) if GATHER_SPLIT else None),
                                                 capi.ptr(out), C.byref(d), C.byref(l), capi.stream()),
                       "ssbev_lift_splat_fwd2")
        ctx.save_for_backward(depth, feat_cl, vox)
        ctx.meta = (B, N, grid)
        return from_cl(out)                                       # logical [B, C, X, Y, Z]

    @staticmethod
    def backward(ctx, gout):
        lib = capi.load()
        depth, feat_cl, vox = ctx.saved_tensors
        B, N, (nx, ny, nz) = ctx.meta
        BN, D, H, W = depth.shape
        Cch = feat_cl.shape[-1]
        g = to_cl(gout)
        d = _pool_dims(B, N * D * H * W, Cch, nx, ny, nz)
        l = capi.LiftDims(N, D, H * W)
        gd = torch.empty_like(depth)
        gf = torch.empty_like(feat_cl)
        nby = 4.0 * (g.numel() + 2 * depth.numel() + 2 * feat_cl.numel() + vox.numel())
        with _span("lift_splat", 4.0 * depth.numel() * Cch, nby, f"bwd   lift_splat D={D} HW={H * W} C={Cch}"):
            capi.check(lib.ssbev_lift_splat_bwd(capi.ptr(g), capi.ptr(depth), capi.ptr(feat_cl), capi.ptr(vox),
                                                capi.ptr(gd), capi.ptr(gf), C.byref(d), C.byref(l), capi.stream()),
                       "ssbev_lift_splat_bwd")
        return gd, from_cl(gf), None, None, None, None, None, None


def lift_splat_tables(geom, bx, dx, nx, grid_host=None):
    """The frustum -> voxel tables of one geometry: (vox, starts, order).  They depend on the calibration only (SURVEY 8 row
    a10: "static -> cacheable"): a caller that sees the same calibration again can keep them (``lift_splat(tables=...)``)."""
    B = geom.shape[0]
    n = grid_host[2] if grid_host is not None else [int(v) for v in nx.tolist()]
    with torch.no_grad():
        vox = voxel_index(geom, bx, dx, nx, grid_host=grid_host)
        starts, order = pool_prepare(vox, B, n[0], n[1], n[2])
    return vox, starts, order


def lift_splat(depth_prob, img_feat, geom, bx, dx, nx, grid_host=None, tables=None):
    """Fused Lift+Splat: depth_prob [B*N,D,H,W], img_feat [B*N,C,H,W], geom [B,N,D,H,W,3]
    -> bev [B,C,X,Y,Z] (channels-last memory).  Equivalent of VT:517-523.  ``tables`` = lift_splat_tables(geom, ...) of a
    previous call with the same geometry (``geom`` may then be None)."""
    n = grid_host[2] if grid_host is not None else [int(v) for v in nx.tolist()]
    vox, starts, order = tables if tables is not None else lift_splat_tables(geom, bx, dx, nx, grid_host)
    B = starts.numel() // (n[0] * n[1] * n[2])
    N = depth_prob.shape[0] // B
    return _LiftSplat.apply(depth_prob, img_feat, vox, starts, order, B, N, tuple(n))


# -------------------------------------------------------------------------------------------------
# stereo cost volume
# -------------------------------------------------------------------------------------------------


class _GwcWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, left, right, calib, ndisp, groups, align_corners):
        lib = capi.load()
        B, Cch, H, W = left.shape
        l = to_cl(_f32(left, "gwc_warp"))
        r = to_cl(_f32(right, "gwc_warp"))
        cal = calib.to(device=left.device, dtype=torch.float32).contiguous()
        d = capi.GwcDims(B, Cch, groups, ndisp, H, W, 1.0, int(bool(align_corners)))
        vol = torch.empty(B, ndisp, H, W, groups, dtype=torch.float32, device=left.device)
        nby = 4.0 * (l.numel() + r.numel() + vol.numel())        # read both feature maps, write the volume
        with _span("gwc_warp_fwd", 8.0 * vol.numel() * (Cch // groups), nby, f"fwd   gwc_warp D={ndisp} {H}x{W} G={groups}"):
            capi.check(lib.ssbev_gwc_warp_fwd(capi.ptr(l), capi.ptr(r), capi.ptr(cal), capi.ptr(vol), C.byref(d),
                                              capi.stream()), "ssbev_gwc_warp_fwd")
        ctx.save_for_backward(l, r, cal)
        ctx.meta = (ndisp, groups, int(bool(align_corners)))
        return from_cl(vol)                                       # logical [B, G, D, H, W]

    @staticmethod
    def backward(ctx, gvol):
        lib = capi.load()
        l, r, cal = ctx.saved_tensors
        ndisp, groups, ac = ctx.meta
        B, H, W, Cch = l.shape
        g = to_cl(gvol)
        d = capi.GwcDims(B, Cch, groups, ndisp, H, W, 1.0, ac)
        gl = torch.empty_like(l)
        gr = torch.empty_like(r)
        nby = 4.0 * (g.numel() + 2 * l.numel() + 2 * r.numel())  # the gradient volume ONCE + both maps in, both gradients out
        with _span("gwc_warp_bwd", 16.0 * g.numel() * (Cch // groups), nby, f"bwd   gwc_warp D={ndisp} {H}x{W} G={groups}"):
            ws = _ws(lib.ssbev_gwc_warp_bwd_workspace(C.byref(d)), g.device)
            capi.check(lib.ssbev_gwc_warp_bwd_fused(capi.ptr(g), capi.ptr(l), capi.ptr(r), capi.ptr(cal), capi.ptr(gl),
                                                    capi.ptr(gr), C.byref(d), capi.ptr(ws), ws.numel(), capi.stream()),
                       "ssbev_gwc_warp_bwd_fused")
        return from_cl(gl), from_cl(gr), None, None, None, None


def gwc_warp(left, right, calib, ndisp, groups=32, align_corners=True):
    """Fused ``build_gwc_volume`` + ``warp`` (VT:104-156): left/right [B,C,H,W], calib [B]
    -> volume [B, G, ndisp, H, W] sampled at metric depths 1..ndisp."""
    return _GwcWarp.apply(left, right, calib, int(ndisp), int(groups), align_corners)


# -------------------------------------------------------------------------------------------------
# gradient slots: one gradient buffer for all consumers of a multi-consumer activation
# -------------------------------------------------------------------------------------------------
# An activation with two consumers (the input of an hourglass feeds its stride-2 conv and its 1x1 redirect; a residual
# block's input feeds the first conv and the closing add) gets its gradient from autograd as the SUM of two tensors: one
# elementwise pass over three 189 MB tensors per such activation on the cost volume (7 per step, 1.3 ms with the narrower
# levels).  ``fork`` hands every consumer an alias of the activation carrying a shared ``GradSlot``; in backward the first
# consumer leaves its gradient tensor in the slot, the next ones run their data-gradient kernel with ``accumulate = 1`` INTO
# that buffer (the kernels add in their epilogue) and return the same tensor, and ``_Fork.backward`` counts a buffer once.
# Consumers that do not know about slots still work: their gradient is a different tensor and is added as before.
GRAD_SLOTS = os.environ.get("SSBEV_GRAD_SLOTS", "1") != "0"


class GradSlot:
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


class _Fork(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot, n):
        ctx.slot = slot
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        ctx.slot.buf = None
        uniq = []
        for g in gs:
            # same buffer = same gradient (strides are not compared: those of size-1 axes are arbitrary)
            if g is not None and not any(g.data_ptr() == u.data_ptr() and g.shape == u.shape for u in uniq):
                uniq.append(g)
        total = None
        for g in uniq:
            total = g if total is None else total + g
        return total, None, None


def fork(x, n=2):
    """``n`` aliases of ``x`` for ``n`` consumers whose data gradients should meet in one buffer (see above)."""
    if not (GRAD_SLOTS and x.is_cuda and x.requires_grad and torch.is_grad_enabled()):
        return (x,) * n
    slot = GradSlot()
    outs = _Fork.apply(x, slot, n)
    for o in outs:
        o._ssbev_grad_slot = slot
    return outs


class _SpatialMean(torch.autograd.Function):
    """x.mean over the spatial axes of a channels-last volume, [B, C].  Its data gradient is a per-channel constant: with a
    gradient slot it is ADDED into the buffer another consumer has already filled (one in-place pass) instead of being
    materialised as a full tensor and added by autograd (CA3D's squeeze step on the 189 MB activation, ATT:100-104)."""

    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot, ctx.shape, ctx.dtype = slot, tuple(x.shape), x.dtype
        return to_cl(x).mean(dim=tuple(range(1, x.dim() - 1)), dtype=torch.float32)     # fp32 vector also for bf16 activations

    @staticmethod
    def backward(ctx, g):
        B, Cch = ctx.shape[0], ctx.shape[1]
        sp = ctx.shape[2:]
        n = 1
        for v in sp:
            n *= v
        gb = (g / float(n)).view((B,) + (1,) * len(sp) + (Cch,))
        buf = ctx.slot.buf if ctx.slot is not None else None
        if buf is not None and buf.is_contiguous() and buf.numel() == B * n * Cch and buf.shape[-1] == Cch and buf.shape[0] == B:
            tgt = buf.view((B,) + sp + (Cch,))
            tgt.add_(gb)
            return from_cl(tgt), None
        full = gb.to(ctx.dtype).expand((B,) + sp + (Cch,)).contiguous()
        if ctx.slot is not None:
            ctx.slot.buf = full
        return from_cl(full), None


def spatial_mean(x):
    """[B, C, *spatial] -> [B, C] mean; slot-aware (see _SpatialMean)."""
    if not (x.is_cuda and x.requires_grad and torch.is_grad_enabled()):
        return x.mean(dim=tuple(range(2, x.dim())), dtype=torch.float32 if x.dtype in (torch.bfloat16, torch.float16) else None)
    return _SpatialMean.apply(x, _slot_of(x))


def _slot_of(t):
    """The gradient slot an alias from ``fork`` carries -- handed out ONCE: the consumer that asks takes it off the alias.  A
    second slot-aware consumer of the same alias gets None and returns a fresh gradient tensor, which autograd adds as usual
    (with the slot still attached it would have accumulated into the shared buffer AND returned it: autograd's own sum of the
    alias' two gradients would then count that buffer twice without any error)."""
    if t is None:
        return None
    slot = getattr(t, "_ssbev_grad_slot", None)
    if slot is not None:
        t._ssbev_grad_slot = None
    return slot


def _slot_target(slot, like):
    """The tensor a data-gradient kernel should accumulate into, or None (then it writes a fresh tensor)."""
    if slot is None or slot.buf is None:
        return None
    b = slot.buf
    # same channels-last volume (a 2-D layer sees [B, 1, H, W, C] where the norm before it saw [B, H, W, C])
    ok = b.is_contiguous() and b.dtype == like.dtype and b.numel() == like.numel() and b.shape[-1] == like.shape[-1] and \
        b.shape[0] == like.shape[0]
    return b.view(tuple(like.shape)) if ok else None


# -------------------------------------------------------------------------------------------------
# convolution family
# -------------------------------------------------------------------------------------------------


def _triple(v, n):
    if isinstance(v, int):
        return (v,) * n
    v = tuple(int(a) for a in v)
    return v if len(v) == n else (v[0],) * n


TILE_HINT = int(os.environ.get("SSBEV_TILE_HINT", "0"))   # tuning hook (tools/probe_conv.py, A/B runs): forwarded to ssbev_conv_dims.tile_hint


def _conv_dims(xshape_cl, wshape, stride, padding, dilation, transposed, output_padding, relu=0, accumulate=0):
    """xshape_cl = (B, Di, Hi, Wi, Cin); weights in the torch layout."""
    B, Di, Hi, Wi, Cin = xshape_cl
    kd, kh, kw = wshape[2:]
    if transposed:
        Cout = wshape[1]
        outs = [(i - 1) * s - 2 * p + dl * (k - 1) + op + 1
                for i, s, p, dl, k, op in zip((Di, Hi, Wi), stride, padding, dilation, (kd, kh, kw), output_padding)]
    else:
        Cout = wshape[0]
        outs = [(i + 2 * p - dl * (k - 1) - 1) // s + 1
                for i, s, p, dl, k in zip((Di, Hi, Wi), stride, padding, dilation, (kd, kh, kw))]
    d = capi.ConvDims(B, Cin, Cout, Di, Hi, Wi, outs[0], outs[1], outs[2], kd, kh, kw, *stride, *padding, *dilation,
                      int(transposed), int(relu), int(accumulate), int(TILE_HINT), 1 if PRECISION == "bf16_operands" else 0)
    return d


def _packed(weight5, d, mode):
    """MFMA operand layout of a weight tensor (packed on the device; a few microseconds)."""
    lib = capi.load()
    wp = torch.empty(lib.ssbev_conv_packed_weight_elems(C.byref(d)), dtype=torch.float32, device=weight5.device)
    capi.check(lib.ssbev_conv_pack_weight(capi.ptr(weight5.contiguous()), capi.ptr(wp), C.byref(d), mode,
                                          capi.stream()), "ssbev_conv_pack_weight")
    return wp


def _thin_out_run(lib, src, weight5, bias, out, d, mode):
    """32 -> 1 / 2 / 4 channel 3x3x3 layers on the two-pass MFMA kernels (ssbev_conv_thin_*, kernel class 5)."""
    wp = torch.empty(lib.ssbev_conv_thin_packed_elems(C.byref(d), mode), dtype=torch.float32, device=src.device)
    capi.check(lib.ssbev_conv_thin_pack(capi.ptr(weight5.contiguous()), capi.ptr(wp), C.byref(d), mode, capi.stream()),
               "ssbev_conv_thin_pack")
    ws = _ws(lib.ssbev_conv_thin_workspace(C.byref(d), mode), src.device)
    capi.check(lib.ssbev_conv_thin_run(capi.ptr(src), capi.ptr(wp), capi.ptr(bias), capi.ptr(out), C.byref(d), mode,
                                       capi.ptr(ws), ws.numel(), capi.stream()), "ssbev_conv_thin_run")


class _ConvNd(torch.autograd.Function):
    """x logical [B,Cin,D,H,W] (channels-last memory), weight in the torch layout (5-D)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, transposed, output_padding, slot=None, relu=False):
        lib = capi.load()
        ctx.slot = slot
        if ctx.needs_input_grad[1]:
            streams.note_use(weight)
        xcl = to_cl(_f32(x, "conv"))
        kpad = (-xcl.shape[-1]) % 4
        w5 = weight
        thin_in = False
        if kpad and not transposed:   # 1..2 -> 32 channel 3x3x3 layers: conv_thinin_kernel gathers the thin side unpadded
            d0 = _conv_dims(tuple(xcl.shape), tuple(weight.shape), stride, padding, dilation, transposed, output_padding)
            thin_in = lib.ssbev_conv_kernel_class(C.byref(d0), 0) == 4
        if kpad and not thin_in:   # the K-role channel count must be a multiple of 4 (float4 operand loads)
            xcl = torch.nn.functional.pad(xcl, (0, kpad))
            w5 = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, 0) + ((0, 0, 0, kpad) if transposed else (0, kpad)))
        d = _conv_dims(tuple(xcl.shape), tuple(w5.shape), stride, padding, dilation, transposed, output_padding, relu=int(relu))
        y = torch.empty(d.B, d.Do, d.Ho, d.Wo, d.Cout, dtype=torch.float32, device=x.device)
        b = bias.detach().contiguous() if bias is not None else None
        fam = _conv_family(lib, d, 0)
        with _span(fam, conv_flops(d), conv_bytes(d), _conv_tag(d, "fwd"), conv_flops(d) / _EXEC_DIV[fam]):
            if lib.ssbev_conv_kernel_class(C.byref(d), 0) == 5:
                _thin_out_run(lib, xcl, w5.detach(), b, y, d, 0)
            else:
                wp = _packed(w5.detach(), d, 0)
                capi.check(lib.ssbev_conv_fwd(capi.ptr(xcl), capi.ptr(wp), capi.ptr(b), capi.ptr(y), C.byref(d),
                                              capi.stream()), "ssbev_conv_fwd")
        # fused ReLU (the kernels' epilogue): backward masks the incoming gradient with the saved output's sign
        ctx.relu = bool(relu)
        ctx.save_for_backward(xcl, weight, *((y,) if relu else ()))
        ctx.cfg = (stride, padding, dilation, transposed, output_padding, 0 if thin_in else kpad, bias is not None)
        ctx.thin_in = thin_in
        ctx.bias_leaf = bias is not None and bias.is_leaf and bias.grad is None
        return from_cl(y)

    @staticmethod
    def backward(ctx, gy):
        lib = capi.load()
        xcl, weight = ctx.saved_tensors[:2]
        stride, padding, dilation, transposed, output_padding, kpad, has_bias = ctx.cfg
        args = (stride, padding, dilation, transposed, output_padding)
        if ctx.relu:
            ycl = ctx.saved_tensors[2]
            gy = from_cl(torch.ops.aten.threshold_backward(to_cl(gy).contiguous(), ycl, 0.0))     # gy * [y > 0], one pass
        gcl0 = to_cl(gy)                                    # gradient as it arrives: Cout channels
        Cout_g = gcl0.shape[-1]
        cpad = (-Cout_g) % 4                                # the data gradient's K role is the forward Cout: multiple of 4 ...
        tpad = (-xcl.shape[-1]) % 4 if ctx.thin_in else 0   # ... and the weight gradient wants Cin % 4 == 0 (forward ran unpadded)
        want_gx, want_gw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        # ... except on the thin-side kernels (conv_thin_mfma.hip), which take both tensors as they are
        thin_d = thin_w = False
        d0 = None
        if not transposed and not kpad and (cpad or tpad):
            d0 = _conv_dims(tuple(xcl.shape), tuple(weight.shape), *args)
            thin_d = bool(cpad) and lib.ssbev_conv_kernel_class(C.byref(d0), 1) == 4
            thin_w = lib.ssbev_conv_kernel_class(C.byref(d0), 2) == 6
        w5 = weight.detach()
        if kpad:
            w5 = torch.nn.functional.pad(w5, (0, 0, 0, 0, 0, 0) + ((0, 0, 0, kpad) if transposed else (0, kpad)))
        gcl = gcl0
        if cpad and ((want_gx and not thin_d) or (want_gw and not thin_w)):
            gcl = torch.nn.functional.pad(gcl0, (0, cpad))
            w5 = torch.nn.functional.pad(w5, (0, 0, 0, 0, 0, 0) + ((0, cpad) if transposed else (0, 0, 0, cpad)))
        elif cpad:
            w5 = None                                       # nobody needs the padded operands
        d = _conv_dims(tuple(xcl.shape), tuple(w5.shape), *args) if w5 is not None else d0
        gx = gw = gb = None

        def weight_gradient():
            pointwise = (not transposed and tuple(weight.shape[2:]) == (1, 1, 1) and stride == (1, 1, 1) and padding == (0, 0, 0)
                         and not kpad and not cpad and not tpad and weight.shape[0] <= 128 and weight.shape[1] <= 128
                         and gcl0.numel() // Cout_g >= 32768)
            if pointwise and OWN_GEMM and PRECISION == "fp32":
                # 1x1x1 layers on the cost volume: gw[co][ci] = sum_rows gy[row][co] x[row][ci], an HBM-streaming skinny TN product
                with _span("conv_wgrad", conv_flops(d), conv_bytes(d), _conv_tag(d, "wgrad")):
                    gw = gemm_tn(gcl0.reshape(-1, Cout_g), xcl.reshape(-1, xcl.shape[-1])).view_as(weight)
            else:
                if thin_w:
                    xw, gw_src, dw, wshape = xcl, gcl0, d0, tuple(weight.shape)
                else:
                    xw, gw_src, dw, wshape = xcl, gcl, d, tuple(w5.shape)
                    if tpad:      # forward ran on the unpadded thin input (conv_thinin_kernel)
                        xw = torch.nn.functional.pad(xcl, (0, tpad))
                        wshape = (wshape[0], wshape[1] + tpad) + wshape[2:]
                        dw = _conv_dims(tuple(xw.shape), wshape, *args)
                gwp = torch.empty(wshape, dtype=torch.float32, device=gy.device)
                ws = _ws(lib.ssbev_conv_bwd_weight_workspace(C.byref(dw)), gy.device)
                with _span("conv_wgrad", conv_flops(dw), conv_bytes(dw), _conv_tag(dw, "wgrad")):
                    capi.check(lib.ssbev_conv_bwd_weight(capi.ptr(xw), capi.ptr(gw_src), capi.ptr(gwp), C.byref(dw),
                                                         capi.ptr(ws), ws.numel(), capi.stream()),
                               "ssbev_conv_bwd_weight")
                gw = gwp if tuple(wshape) == tuple(weight.shape) else gwp[: weight.shape[0], : weight.shape[1]].contiguous()
            return gw

        # the weight gradient is a leaf of the backward chain: on the side stream it runs NEXT to the data gradient and the
        # normalisation passes that follow it instead of in front of them (streams.py)
        gw_side = want_gw and streams.wgrad_on_side(weight)
        want_gb = has_bias and ctx.needs_input_grad[2]
        gb_side = gw_side and want_gb and ctx.bias_leaf
        if gw_side:
            with streams.on_side(gy.device, xcl, gcl0, gcl) as side:
                gw = weight_gradient()
                if gb_side:                                   # the bias gradient (a column sum of gy) is a leaf as well
                    gb = gcl0.reshape(-1, Cout_g).sum(0)
                side.publish(gw, gb)
        if want_gx:
            dd, gdl, wd = (d0, gcl0, weight.detach()) if thin_d else (d, gcl, w5)
            slot = ctx.slot if (not kpad and not thin_d and lib.ssbev_conv_kernel_class(C.byref(dd), 1) not in (4, 5)) else None
            into = _slot_target(slot, xcl)
            if into is not None:          # another consumer's gradient is already there: add to it in the kernel's epilogue
                dd.accumulate = 1
            gxcl = into if into is not None else torch.empty_like(xcl)
            fam = _conv_family(lib, dd, 1)
            with _span(fam, conv_flops(dd), conv_bytes(dd), _conv_tag(dd, "dgrad"), conv_flops(dd) / _EXEC_DIV[fam]):
                if lib.ssbev_conv_kernel_class(C.byref(dd), 1) == 5:
                    _thin_out_run(lib, gdl, wd, None, gxcl, dd, 1)
                else:
                    wpt = _packed(wd, dd, 1)
                    capi.check(lib.ssbev_conv_bwd_data(capi.ptr(gdl), capi.ptr(wpt), capi.ptr(gxcl), C.byref(dd),
                                                       capi.stream()), "ssbev_conv_bwd_data")
            dd.accumulate = 0
            if slot is not None:
                slot.buf = gxcl
            if kpad:
                gxcl = gxcl[..., : xcl.shape[-1] - kpad]
            gx = from_cl(gxcl)
        if want_gw and not gw_side:
            gw = weight_gradient()
        if want_gb and not gb_side:
            gb = gcl0.reshape(-1, Cout_g).sum(0)
        return gx, gw, gb, None, None, None, None, None, None, None


class _ConvNd16(torch.autograd.Function):
    """Convolution / transposed convolution in the bf16 STORAGE mode (csrc/conv_bf16.hip, ssbev_conv_dims.precision = 2 / 3):
    x and y (and gy, gx) are bf16 channels-last tensors, the weight stays an fp32 master (packed to bf16 operands per launch),
    its gradient comes back in fp32.  ``out_fp32``: fp32 result (the layer in front of an fp32 island)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, transposed, output_padding, slot=None, relu=False, out_fp32=False):
        lib = capi.load()
        ctx.slot = slot
        if ctx.needs_input_grad[1]:
            streams.note_use(weight)
        xcl = to_cl(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16))
        d = _conv_dims(tuple(xcl.shape), tuple(weight.shape), stride, padding, dilation, transposed, output_padding, relu=int(relu))
        d.precision = 3 if out_fp32 else 2
        y = torch.empty(d.B, d.Do, d.Ho, d.Wo, d.Cout, dtype=torch.float32 if out_fp32 else torch.bfloat16, device=x.device)
        b = bias.detach().float().contiguous() if bias is not None else None
        fam = {17: "conv_tap16", 19: "conv_wide16"}.get(lib.ssbev_conv_kernel_class(C.byref(d), 0), "conv_gather16") if KERNEL_TIMER is not None else "conv_gather16"
        with _span(fam, conv_flops(d), conv_bytes(d) / 2.0, _conv_tag(d, "fwd")):
            wp = _packed(weight.detach(), d, 0)
            capi.check(lib.ssbev_conv_fwd_bf16(capi.ptr(xcl), capi.ptr(wp), capi.ptr(b), capi.ptr(y), C.byref(d), capi.stream()),
                       "ssbev_conv_fwd_bf16")
        ctx.relu = bool(relu)
        ctx.save_for_backward(xcl, weight, *((y,) if relu else ()))
        ctx.cfg = (stride, padding, dilation, transposed, output_padding, bias is not None)
        ctx.bias_leaf = bias is not None and bias.is_leaf and bias.grad is None
        return from_cl(y)

    @staticmethod
    def backward(ctx, gy):
        lib = capi.load()
        xcl, weight = ctx.saved_tensors[:2]
        stride, padding, dilation, transposed, output_padding, has_bias = ctx.cfg
        gcl = to_cl(gy)
        if ctx.relu:
            gcl = torch.ops.aten.threshold_backward(gcl.contiguous(), ctx.saved_tensors[2], 0.0)
        if gcl.dtype != torch.bfloat16:
            gcl = gcl.to(torch.bfloat16)
        Cout_g = gcl.shape[-1]
        d = _conv_dims(tuple(xcl.shape), tuple(weight.shape), stride, padding, dilation, transposed, output_padding)
        d.precision = 2
        want_gx, want_gw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gx = gw = gb = None

        def weight_gradient():
            if (WIDE16_WGRAD == "wino" and not transposed and lib.ssbev_conv_kernel_class(C.byref(d), 0) == 19
                    and all(int(n) % 2 == 0 for n in xcl.shape[1:4])):
                with _span("conv_winograd_wgrad", conv_flops(d), conv_bytes(d) / 2.0, _conv_tag(d, "wgrad(wino)"), conv_flops(d) / 3.375):
                    return _wino_wgrad_bf16(xcl, gcl, weight)
            gwp = torch.empty(tuple(weight.shape), dtype=torch.float32, device=gy.device)
            ws = _ws(lib.ssbev_conv_bwd_weight_workspace(C.byref(d)), gy.device)
            fam = "wgrad_ring16" if (KERNEL_TIMER is not None and lib.ssbev_conv_kernel_class(C.byref(d), 2) == 20) else "wgrad16"
            with _span(fam, conv_flops(d), conv_bytes(d) / 2.0, _conv_tag(d, "wgrad")):
                capi.check(lib.ssbev_conv_bwd_weight_bf16(capi.ptr(xcl), capi.ptr(gcl), capi.ptr(gwp), C.byref(d), capi.ptr(ws),
                                                          ws.numel(), capi.stream()), "ssbev_conv_bwd_weight_bf16")
            return gwp

        gw_side = want_gw and streams.wgrad_on_side(weight)
        want_gb = has_bias and ctx.needs_input_grad[2]
        gb_side = gw_side and want_gb and ctx.bias_leaf
        if gw_side:
            with streams.on_side(gy.device, xcl, gcl) as side:
                gw = weight_gradient()
                if gb_side:
                    gb = gcl.reshape(-1, Cout_g).sum(0, dtype=torch.float32)
                side.publish(gw, gb)
        if want_gx:
            into = _slot_target(ctx.slot, xcl)
            if into is not None:
                d.accumulate = 1
            gxcl = into if into is not None else torch.empty_like(xcl)
            fam = {17: "conv_tap16", 19: "conv_wide16"}.get(lib.ssbev_conv_kernel_class(C.byref(d), 1), "conv_gather16") if KERNEL_TIMER is not None else "conv_gather16"
            with _span(fam, conv_flops(d), conv_bytes(d) / 2.0, _conv_tag(d, "dgrad")):
                wpt = _packed(weight.detach(), d, 1)
                capi.check(lib.ssbev_conv_bwd_data_bf16(capi.ptr(gcl), capi.ptr(wpt), capi.ptr(gxcl), C.byref(d), capi.stream()),
                           "ssbev_conv_bwd_data_bf16")
            d.accumulate = 0
            if ctx.slot is not None:
                ctx.slot.buf = gxcl
            gx = from_cl(gxcl)
        if want_gw and not gw_side:
            gw = weight_gradient()
        if want_gb and not gb_side:
            gb = gcl.reshape(-1, Cout_g).sum(0, dtype=torch.float32)
        return gx, gw, gb, None, None, None, None, None, None, None, None


def _conv16_route(x, weight5, bias, st, pd, dl, transposed, op, relu=False):
    """bf16 storage mode: the convolution on csrc/conv_bf16.hip when its channel counts allow 16-byte bf16 voxel-line pieces
    (Cin % 8 == 0 and Cout % 8 == 0), else None (the caller keeps its fp32 island: the 1- / 2- / 20-channel layers)."""
    if not (storage_bf16() and x.is_cuda and TILE_HINT in (0, 7, 9)):
        return None
    cin = weight5.shape[0] if transposed else weight5.shape[1]
    cout = weight5.shape[1] if transposed else weight5.shape[0]
    if cin % 8 != 0 or cout % 8 != 0:        # (the Cout-channel gradient is a bf16 SOURCE in backward)
        return None
    return _ConvNd16.apply(x, weight5, bias, st, pd, dl, transposed, op, _slot_of(x), bool(relu), False)


# bf16 storage mode: the wide stride-1 3x3x3 layers on grids with W % 16 == 0 (voxel encoder level 0, head conv, the 64 -> 64
# hourglass layers) run on conv_wide16_kernel (LDS-ring implicit GEMM) instead of the F(2,3) Winograd pipeline (0 = Winograd)
WIDE16 = os.environ.get("SSBEV_WIDE16", "1") != "0"
# ... and their weight gradient on the LDS-ring kernel wgrad_ring16_kernel ("direct", default) or on the Winograd-domain product
# (input transform + adjoint + batched bf16 library GEMM: "wino", kept for A/B runs)
WIDE16_WGRAD = os.environ.get("SSBEV_WIDE16_WGRAD", "direct")


def _wide16_applicable(x, weight, st, pd, dl):
    if x.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3) or st != (1, 1, 1) or pd != (1, 1, 1) or dl != (1, 1, 1):
        return False
    if weight.shape[0] % 8 or weight.shape[1] % 8:
        return False
    d = _conv_dims((x.shape[0], x.shape[2], x.shape[3], x.shape[4], x.shape[1]), tuple(weight.shape), st, pd, dl, False, (0, 0, 0))
    d.precision = 2
    return capi.load().ssbev_conv_kernel_class(C.byref(d), 0) == 19


def _wino_wgrad_bf16(xcl, gcl, weight):
    """Weight gradient of a 3x3x3 / stride 1 / pad 1 layer in the F(2,3)^3 Winograd domain, bf16 tensors on both sides:
    gU[xi] = V[xi]^T Z[xi] (batched bf16 GEMM, fp32 result), gw = G^T gU G."""
    lib = capi.load()
    B, D, H, W, Cin = xcl.shape
    Cout = gcl.shape[-1]
    T = B * (D // 2) * (H // 2) * (W // 2)
    V = _wino_call("ssbev_wino_input_transform_bf16a", xcl, capi.WinoDims(B, D, H, W, Cin), (64, T, Cin), torch.bfloat16)
    Z = _wino_call("ssbev_wino_output_adjoint_bf16a", gcl, capi.WinoDims(B, D, H, W, Cout), (64, T, Cout), torch.bfloat16)
    gU = _bmm16_tn(V, Z)
    gwt = torch.empty(tuple(weight.shape), dtype=torch.float32, device=xcl.device)
    capi.check(lib.ssbev_wino_weight_grad(capi.ptr(gU), capi.ptr(gwt), Cout, Cin, 3, capi.stream()), "ssbev_wino_weight_grad")
    return gwt


def _to_act(y):
    """Result of an fp32-island convolution on its way back into the bf16 chain (channel counts that cannot be bf16 sources --
    the 1-channel volumes -- stay fp32)."""
    if storage_bf16() and y.dtype == torch.float32 and y.shape[1] % 8 == 0:
        return y.to(torch.bfloat16)
    return y


WINOGRAD = os.environ.get("SSBEV_WINOGRAD", "1") != "0"   # wide 3x3x3 stride-1 layers via F(2,3)^3 (0 = direct MFMA conv)
# bf16 storage mode: wide stride-1 3x3(x3) layers on the F(2,3) Winograd pipeline with bf16 tensors on both sides (1), or on the
# direct bf16 MFMA kernels (0): 3.4x / 2.25x fewer multiply-adds against 8x / 4x larger transformed tensors
WINO_BF16S = os.environ.get("SSBEV_WINO_BF16S", "1") != "0"


def conv3d(x, weight, bias=None, stride=1, padding=0, dilation=1, relu=False):
    """F.conv3d replacement (groups=1): the MFMA implicit-GEMM kernels, or Winograd F(2x2x2,3x3x3) for the wide
    stride-1 3x3x3 layers."""
    st, pd, dl = _triple(stride, 3), _triple(padding, 3), _triple(dilation, 3)
    if storage_bf16() and x.is_cuda and TILE_HINT in (0, 7, 9):
        if WIDE16 and _wide16_applicable(x, weight, st, pd, dl):
            return _ConvNd16.apply(x, weight, bias, st, pd, dl, False, (0, 0, 0), _slot_of(x), bool(relu), False)
        if WINOGRAD and WINO_BF16S and wino_conv3d_applicable(x, weight, st, pd, dl):
            y = _WinoConv.apply(x, weight, _slot_of(x))
            y = y if bias is None else y + _like_act(bias, y).view(1, -1, 1, 1, 1)
            return torch.relu(y) if relu else y
        y = _conv16_route(x, weight, bias, st, pd, dl, False, (0, 0, 0), relu)
        if y is not None:
            return y
        return _to_act(_ConvNd.apply(x, weight, bias, st, pd, dl, False, (0, 0, 0), None, bool(relu)))
    if relu:    # epilogue ReLU of the direct kernels (the MIE block's two conv -> ReLU pairs, VT:250-258); elsewhere a norm follows
        gemm = (GEMM_LAYERS and TILE_HINT == 0 and PRECISION == "fp32" and tuple(weight.shape[2:]) == (1, 1, 1)
                and st == (1, 1, 1) and pd == (0, 0, 0) and weight.shape[1] >= GEMM_MIN_CIN)
        if not x.is_cuda or gemm or (WINOGRAD and TILE_HINT == 0 and wino_conv3d_applicable(x, weight, st, pd, dl)):
            return torch.relu(conv3d(x, weight, bias, stride, padding, dilation))
        return _ConvNd.apply(x, weight, bias, st, pd, dl, False, (0, 0, 0), _slot_of(x), True)
    if (GEMM_LAYERS and TILE_HINT == 0 and PRECISION == "fp32" and x.is_cuda and tuple(weight.shape[2:]) == (1, 1, 1)
            and st == (1, 1, 1) and pd == (0, 0, 0) and weight.shape[1] >= GEMM_MIN_CIN):
        return linear_cl(x, weight, bias)
    if WINOGRAD and TILE_HINT == 0 and wino_conv3d_applicable(x, weight, st, pd, dl):
        B, Cin, D, H, W = x.shape
        if tuple(weight.shape[2:]) == (3, 3, 3) and H % 4 == 0 and W % 4 == 0 and \
                _wino_df_applicable(B, D, H, W, Cin, weight.shape[0]):
            y = _WinoConvDF.apply(x, weight, _slot_of(x))
        else:
            y = _WinoConv.apply(x, weight, _slot_of(x))
        return y if bias is None else y + bias.view(1, -1, 1, 1, 1)
    return _ConvNd.apply(x, weight, bias, st, pd, dl, False, (0, 0, 0), _slot_of(x))


# -------------------------------------------------------------------------------------------------
# plain fp32 GEMMs on the matrix cores (csrc/gemm.hip): no library GEMM on the hot path
# -------------------------------------------------------------------------------------------------
# SSBEV_OWN_GEMM=0 sends these products back to rocBLAS (torch.mm / torch.bmm), for A/B runs.
OWN_GEMM = os.environ.get("SSBEV_OWN_GEMM", "1") != "0"
# Per call site: comma list out of deconv, linear, wino, bri; "all" = every site.  Default = where the own kernels win or tie
# in the step (profiles/r2l_own_gemm_sites.txt, ms/step against 91.3 with none): deconv -0.75 (the depth-to-space permute
# copies disappear), bri +0.2 (no operand transposes; the north-star block on own kernels).  The wide pointwise layers
# (+1.5) and the batched frequency products of the 2-D / weight-streaming Winograd layers (+1.0) are plain library GEMMs
# where rocBLAS's tuned kernels run at 90 % of the fp32 matrix peak against ~75 % here: they stay on rocBLAS by default.
# r3 (one side stream, DepthNet off the critical chain; profiles/r3m_own_gemm_sites.txt, ms/step): deconv,bri 78.85 |
# + linear 78.92 (now free: default) | + wino 79.48 | all 79.85.  The batched frequency products of the 2-D / weight-streaming
# Winograd layers stay on the library: 0.6 ms per step is what its tuned kernels are still worth there.
# r5: "wino" joins the default -- no Cijk_* frequency product is left on a1-a16.  The price, measured with the
# r5 tile configurations (128 x 160 tiles on the 640-column products, 128 x 64 on the 512-channel layer; profiles/
# r5_gemm_cfg_probe.txt): 72.2-72.4 ms per step against 71.4-71.6 with rocBLAS on those 33 launches (own NN 113 vs 122 TF/s, own
# TN 85 vs 99 TF/s on 16 x [1920 x 640 x 640]).  SSBEV_OWN_GEMM_SITES=deconv,bri,linear restores the library there.
OWN_GEMM_SITES = set(os.environ.get("SSBEV_OWN_GEMM_SITES", "deconv,bri,linear,wino").split(","))


def own_gemm_site(name):
    return OWN_GEMM and ("all" in OWN_GEMM_SITES or name in OWN_GEMM_SITES)


def _rows(t):
    """[.., R, C] tensor with a contiguous last axis and uniform row / batch strides -> (tensor, ld, batch_stride)."""
    if t.stride(-1) != 1 or (t.dim() == 3 and t.shape[0] > 1 and t.stride(0) % 4 != 0) or t.stride(-2) % 4 != 0 or t.stride(-2) < t.shape[-1]:
        t = t.contiguous()
    if t.data_ptr() % 16:            # the kernels issue 16-byte global_load_lds: a last-axis slice view may start misaligned
        t = t.clone(memory_format=torch.contiguous_format)
    return t, t.stride(-2), (t.stride(0) if t.dim() == 3 else 0)


def _rawptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _gemm_supported(*ts):
    return OWN_GEMM and all(t.is_cuda and t.dtype == torch.float32 for t in ts)


def _gdims(M, N, K, batch, lda, ldb, ldc, sa, sb, sc, relu=0, d2s=None):
    g = capi.GemmDims(int(M), int(N), int(K), int(batch), int(lda), int(ldb), int(ldc), int(sa), int(sb), int(sc), int(relu),
                      0, 0, 0, 0, 0, 0, 0, None, None, None)
    if d2s is not None:
        (g.d2s_D, g.d2s_H, g.d2s_W, g.d2s_kd, g.d2s_kh, g.d2s_kw, g.d2s_Co), table = d2s
        g.d2s_rowoff = table.data_ptr()
    return g


def gemm_nn(a, b, bias=None, relu=False, tag=None):
    """C = A @ B (+ bias, ReLU): a [M,K] or [Bt,M,K]; b [K,N] or [Bt,K,N] (a 2-D b is shared by the batch)."""
    if not _gemm_supported(a, b) or a.shape[-1] % 4 or b.shape[-1] % 4:
        y = torch.matmul(a, b)
        y = y if bias is None else y + bias
        return torch.relu(y) if relu else y
    a, lda, sa = _rows(a)
    b, ldb, sb = _rows(b)
    batch = a.shape[0] if a.dim() == 3 else 1
    M, K, N = a.shape[-2], a.shape[-1], b.shape[-1]
    out = torch.empty(*a.shape[:-1], N, dtype=torch.float32, device=a.device)
    g = _gdims(M, N, K, batch, lda, ldb, N, sa, sb if b.dim() == 3 else 0, M * N, relu)
    fl = 2.0 * batch * M * N * K
    lib = capi.load()
    ws = _ws(lib.ssbev_gemm_nn_workspace(C.byref(g)), a.device)
    with _span("gemm_own", fl, 4.0 * (a.numel() + b.numel() + out.numel()), tag or f"nn {batch}x{M}x{K}x{N}"):
        capi.check(lib.ssbev_gemm_nn(_rawptr(a), _rawptr(b), _rawptr(bias.contiguous() if bias is not None else None),
                                     capi.ptr(out), C.byref(g), capi.ptr(ws), ws.numel(), capi.stream()), "ssbev_gemm_nn")
    return out


def gemm_nt(a, w, bias=None, relu=False, tag=None):
    """C = A @ W^T (+ bias, ReLU): a [M,K] / [Bt,M,K], w [N,K] / [Bt,N,K] -- the nn.Linear layout, nothing transposed."""
    if not _gemm_supported(a, w) or a.shape[-1] % 4 or w.shape[-2] % 4:
        y = torch.matmul(a, w.transpose(-1, -2))
        y = y if bias is None else y + bias
        return torch.relu(y) if relu else y
    a, lda, sa = _rows(a)
    w, ldb, sb = _rows(w)
    batch = a.shape[0] if a.dim() == 3 else 1
    M, K, N = a.shape[-2], a.shape[-1], w.shape[-2]
    out = torch.empty(*a.shape[:-1], N, dtype=torch.float32, device=a.device)
    g = _gdims(M, N, K, batch, lda, ldb, N, sa, sb if w.dim() == 3 else 0, M * N, relu)
    fl = 2.0 * batch * M * N * K
    lib = capi.load()
    ws = _ws(lib.ssbev_gemm_nt_workspace(C.byref(g)), a.device)
    with _span("gemm_own", fl, 4.0 * (a.numel() + w.numel() + out.numel()), tag or f"nt {batch}x{M}x{K}x{N}"):
        capi.check(lib.ssbev_gemm_nt(_rawptr(a), _rawptr(w), _rawptr(bias.contiguous() if bias is not None else None),
                                     capi.ptr(out), C.byref(g), capi.ptr(ws), ws.numel(), capi.stream()), "ssbev_gemm_nt")
    return out


def gemm_tn(a, b, tag=None, ep_mul=None, ep_rowsub=None):
    """C = A^T @ B over the ROW axis: a [R,K] / [Bt,R,K], b [R,N] / [Bt,R,N] -> [K,N] / [Bt,K,N] (weight gradients).
    ``ep_mul`` [.., K, N] (contiguous) and ``ep_rowsub`` [.., K]: fused epilogue C = ep_mul * (A^T B - ep_rowsub[..., None])."""
    if not _gemm_supported(a, b) or a.shape[-1] % 4 or b.shape[-1] % 4:
        c = torch.matmul(a.transpose(-1, -2), b)
        return c if ep_mul is None else ep_mul * (c - ep_rowsub.unsqueeze(-1))
    a, lda, sa = _rows(a)
    b, ldb, sb = _rows(b)
    batch = a.shape[0] if a.dim() == 3 else 1
    R, K, N = a.shape[-2], a.shape[-1], b.shape[-1]
    out = torch.empty(*a.shape[:-2], K, N, dtype=torch.float32, device=a.device)
    g = _gdims(R, N, K, batch, lda, ldb, N, sa, sb, K * N)
    if ep_mul is not None:
        ep_mul, ep_rowsub = ep_mul.contiguous(), ep_rowsub.contiguous().float()
        if tuple(ep_mul.shape) != tuple(out.shape) or ep_rowsub.numel() != batch * K:
            raise capi.SsbevError("gemm_tn: epilogue operand shapes")
        g.ep_mul, g.ep_rowsub = ep_mul.data_ptr(), ep_rowsub.data_ptr()
    lib = capi.load()
    ws = _ws(lib.ssbev_gemm_tn_workspace(C.byref(g)), a.device)
    fl = 2.0 * batch * R * N * K
    with _span("gemm_own", fl, 4.0 * (a.numel() + b.numel() + out.numel() * (2 if ep_mul is not None else 1)),
               tag or f"tn {batch}x{R}x{K}x{N}"):
        capi.check(lib.ssbev_gemm_tn(_rawptr(a), _rawptr(b), capi.ptr(out), C.byref(g), capi.ptr(ws), ws.numel(), capi.stream()),
                   "ssbev_gemm_tn")
    return out


class _LinearCL(torch.autograd.Function):
    """Pointwise (1x1 / stride 1) convolution of a channels-last map = y[rows, Cout] = x[rows, Cin] @ W[Cout, Cin]^T + b:
    NT forward, NN data gradient, TN weight gradient on the gemm.hip kernels, no transposes anywhere."""

    @staticmethod
    def forward(ctx, x2, w2, bias):
        ctx.save_for_backward(x2, w2)
        ctx.has_bias = bias is not None
        return gemm_nt(x2, w2, bias, tag=f"linear fwd {w2.shape[1]}->{w2.shape[0]} rows={x2.shape[0]}")

    @staticmethod
    def backward(ctx, gy):
        x2, w2 = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm_nn(gy, w2, tag=f"linear dgrad {w2.shape[1]}->{w2.shape[0]} rows={x2.shape[0]}")
        if ctx.needs_input_grad[1]:
            gw = gemm_tn(gy, x2, tag=f"linear wgrad {w2.shape[1]}->{w2.shape[0]} rows={x2.shape[0]}")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        return gx, gw, gb


_D2S_TABLES = {}


def _d2s_table(B, D, H, W, k, Co, device):
    key = (B, D, H, W, k, Co, str(device))
    t = _D2S_TABLES.get(key)
    if t is None:
        t = torch.empty(B * D * H * W, dtype=torch.int64, device=device)
        capi.check(capi.load().ssbev_gemm_d2s_rowoff(capi.ptr(t), B * D * H * W, D, H, W, k[0], k[1], k[2], Co, capi.stream()),
                   "ssbev_gemm_d2s_rowoff")
        _D2S_TABLES[key] = t
    return t


class _DeconvKS(torch.autograd.Function):
    """kernel == stride ConvTranspose3d (SECONDFPN3D levels, second_fpn_3d.py:50-69) as the GEMM it is, with the
    depth-to-space index map inside the kernels: forward = NN with a scattering epilogue (the [B,D,H,W,kd,kh,kw,Co] ->
    [B,D kd,H kh,W kw,Co] permute copy of the library realisation never happens), data gradient = NT with gathered rows,
    weight gradient = TN with gathered rows."""

    @staticmethod
    def forward(ctx, xcl, weight, bias, k):
        B, D, H, W, Ci = xcl.shape
        Co = weight.shape[1]
        kd, kh, kw = k
        taps = kd * kh * kw
        wk = weight.detach().permute(0, 2, 3, 4, 1).reshape(Ci, taps * Co).contiguous()       # [Ci][tap][co]: a few MB
        table = _d2s_table(B, D, H, W, k, Co, xcl.device)
        y = torch.empty(B, D * kd, H * kh, W * kw, Co, dtype=torch.float32, device=xcl.device)
        x2 = xcl.reshape(-1, Ci)
        R = x2.shape[0]
        g = _gdims(R, taps * Co, Ci, 1, Ci, taps * Co, 0, 0, 0, 0, 0, ((D, H, W, kd, kh, kw, Co), table))
        fl = 2.0 * R * Ci * taps * Co
        with _span("gemm_own", fl, 4.0 * (x2.numel() + wk.numel() + y.numel()), f"deconv k=s fwd {Ci}->{Co} k{kd}{kh}{kw}"):
            capi.check(capi.load().ssbev_gemm_nn(capi.ptr(x2), capi.ptr(wk), capi.ptr(bias.detach().contiguous() if bias is not None else None),
                                                 capi.ptr(y), C.byref(g), None, 0, capi.stream()), "ssbev_gemm_nn")
        ctx.save_for_backward(x2, wk)
        ctx.meta = (B, D, H, W, Ci, Co, k, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x2, wk = ctx.saved_tensors
        B, D, H, W, Ci, Co, k, has_bias = ctx.meta
        kd, kh, kw = k
        taps = kd * kh * kw
        gy = gy.contiguous()
        table = _d2s_table(B, D, H, W, k, Co, gy.device)
        d2s = ((D, H, W, kd, kh, kw, Co), table)
        R = x2.shape[0]
        lib = capi.load()
        gx = gw = gb = None
        fl = 2.0 * R * Ci * taps * Co
        if ctx.needs_input_grad[0]:      # gx[r][ci] = sum_(tap,co) gy_fine[r; tap, co] wk[ci][(tap,co)]  (NT, gathered A)
            gx2 = torch.empty(R, Ci, dtype=torch.float32, device=gy.device)
            g = _gdims(R, Ci, taps * Co, 1, 0, taps * Co, Ci, 0, 0, 0, 0, d2s)
            ws = _ws(lib.ssbev_gemm_nt_workspace(C.byref(g)), gy.device)
            with _span("gemm_own", fl, 4.0 * (gy.numel() + wk.numel() + gx2.numel()), f"deconv k=s dgrad {Ci}->{Co} k{kd}{kh}{kw}"):
                capi.check(lib.ssbev_gemm_nt(capi.ptr(gy), capi.ptr(wk), None, capi.ptr(gx2), C.byref(g), capi.ptr(ws), ws.numel(),
                                             capi.stream()), "ssbev_gemm_nt")
            gx = gx2.view(B, D, H, W, Ci)
        if ctx.needs_input_grad[1]:      # gwk[ci][(tap,co)] = sum_r x[r][ci] gy_fine[r; tap, co]               (TN, gathered B)
            gwk = torch.empty(Ci, taps * Co, dtype=torch.float32, device=gy.device)
            g = _gdims(R, taps * Co, Ci, 1, Ci, 0, taps * Co, 0, 0, Ci * taps * Co, 0, d2s)
            ws = _ws(lib.ssbev_gemm_tn_workspace(C.byref(g)), gy.device)
            with _span("gemm_own", fl, 4.0 * (gy.numel() + x2.numel() + gwk.numel()), f"deconv k=s wgrad {Ci}->{Co} k{kd}{kh}{kw}"):
                capi.check(lib.ssbev_gemm_tn(capi.ptr(x2), capi.ptr(gy), capi.ptr(gwk), C.byref(g), capi.ptr(ws), ws.numel(),
                                             capi.stream()), "ssbev_gemm_tn")
            gw = gwk.view(Ci, kd, kh, kw, Co).permute(0, 4, 1, 2, 3).contiguous()
        if has_bias and ctx.needs_input_grad[2]:
            gb = gy.reshape(-1, Co).sum(0)
        return gx, gw, gb, None


# Layers that are plain GEMMs in the channels-last layout go to rocBLAS (forward, data and weight gradient through
# autograd's mm): wide pointwise convolutions ([pixels, Cin] x [Cin, Cout]) and the kernel == stride transposed
# convolutions of the FPN ([voxels, Cin] x [Cin, k^3 * Cout] followed by a depth-to-space copy).  The hand-written
# kernels stay on everything with spatial taps.  SSBEV_GEMM_LAYERS=0 keeps these layers on the MFMA conv kernels.
GEMM_LAYERS = os.environ.get("SSBEV_GEMM_LAYERS", "1") != "0"
GEMM_MIN_CIN = int(os.environ.get("SSBEV_GEMM_MIN_CIN", "512"))       # pointwise convs narrower than this stay on the HIP kernels


def _deconv_k_eq_s_gemm(x, weight, bias, k):
    if tuple(k) == (1, 1, 1) and own_gemm_site("deconv") and own_gemm_site("linear") and weight.shape[1] % 4 == 0:
        # k = s = 1 (SECONDFPN3D's first branch, FPN:53-69) is a pointwise layer: the plain NT / NN / skinny-TN products of linear_cl.
        # Through the depth-to-space table of the general k == s path its weight gradient ran at 26 TF/s (0.33 ms for 8.6 GF:
        # profiles/r6c layer table) where the skinny TN kernel needs 0.12 ms for the same product.
        return linear_cl(x, weight.reshape(weight.shape[0], weight.shape[1]).t(), bias)
    xcl = to_cl(_f32(x, "deconv_gemm"))
    B, D, H, W, Ci = xcl.shape
    Co = weight.shape[1]
    kd, kh, kw = k
    if own_gemm_site("deconv") and Co % 64 == 0 and Ci % 4 == 0:
        return from_cl(_DeconvKS.apply(xcl, weight, bias, (int(kd), int(kh), int(kw))))
    w2 = weight.permute(0, 2, 3, 4, 1).reshape(Ci, kd * kh * kw * Co)
    fl = 2.0 * B * D * H * W * Ci * kd * kh * kw * Co
    nby = 4.0 * (xcl.numel() + w2.numel() + B * D * H * W * kd * kh * kw * Co)
    mult = 3.0 if (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)) else 1.0
    with _span("gemm_lib", fl * mult, nby * mult, f"fwd   deconv k=s {Ci}->{Co} k{kd}{kh}{kw}"):
        g = torch.mm(xcl.reshape(-1, Ci), w2).view(B, D, H, W, kd, kh, kw, Co)
    y = g.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D * kd, H * kh, W * kw, Co)
    if bias is not None:
        y = y + bias
    return from_cl(y)


def conv_transpose3d(x, weight, bias=None, stride=1, padding=0, output_padding=0):
    """F.conv_transpose3d replacement (weight [Cin, Cout, kd, kh, kw])."""
    st, pd, op = _triple(stride, 3), _triple(padding, 3), _triple(output_padding, 3)
    if storage_bf16() and x.is_cuda and TILE_HINT in (0, 9):
        y = _conv16_route(x, weight, bias, st, pd, (1, 1, 1), True, op)
        if y is not None:
            return y
        return _to_act(_ConvNd.apply(x, weight, bias, st, pd, (1, 1, 1), True, op))
    if (GEMM_LAYERS and TILE_HINT == 0 and PRECISION == "fp32" and x.is_cuda and tuple(weight.shape[2:]) == tuple(st)
            and pd == (0, 0, 0) and op == (0, 0, 0) and weight.shape[0] >= 128):
        return _deconv_k_eq_s_gemm(x, weight, bias, tuple(st))
    return _ConvNd.apply(x, weight, bias, _triple(stride, 3), _triple(padding, 3), (1, 1, 1), True,
                         _triple(output_padding, 3))


# A 3x3 convolution with dilation d and padding d only couples pixels of the same residue class modulo d: it is d*d
# independent ordinary 3x3 / pad 1 convolutions on the (H/d) x (W/d) sub-grids.  The ASPP branches of DepthNet (640 -> 640,
# dilation 6 / 12 / 18 on the 48 x 160 map) are therefore sent through the Winograd path as a batch of d*d small images
# (space-to-batch / batch-to-space are two tensor copies of 20 MB): 2.25x fewer multiply-adds than the direct dilated
# kernel.  Only when zero-padding the map to multiples of 2d costs < 35 % extra pixels (not for d = 18 at 48 x 160).
DILATED_POLYPHASE = os.environ.get("SSBEV_DILATED_POLYPHASE", "1") != "0"
POLYPHASE_MAX_PAD = float(os.environ.get("SSBEV_POLYPHASE_MAX_PAD", "1.35"))


def _dilated_polyphase(x, weight, d):
    B, Cc, H, W = x.shape
    Hp, Wp = -(-H // (2 * d)) * 2 * d, -(-W // (2 * d)) * 2 * d
    if Hp * Wp > POLYPHASE_MAX_PAD * H * W:
        return None
    xp = torch.nn.functional.pad(x, [0, Wp - W, 0, Hp - H]) if (Hp != H or Wp != W) else x
    xs = xp.reshape(B, Cc, Hp // d, d, Wp // d, d).permute(0, 3, 5, 1, 2, 4).reshape(B * d * d, Cc, Hp // d, Wp // d)
    ys = conv2d(xs, weight, None, 1, 1, 1)
    Co = weight.shape[0]
    y = ys.reshape(B, d, d, Co, Hp // d, Wp // d).permute(0, 3, 4, 1, 5, 2).reshape(B, Co, Hp, Wp)
    return y[:, :, :H, :W]


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1):
    """F.conv2d replacement: a depth-1 volume through the same kernels (groups=1); wide 3x3 stride-1 layers via
    Winograd F(2x2,3x3)."""
    s, p, dl = _triple(stride, 2), _triple(padding, 2), _triple(dilation, 2)
    if storage_bf16() and x.is_cuda and TILE_HINT in (0, 9):
        x5, w5 = x.unsqueeze(2), weight.unsqueeze(2)
        if (DILATED_POLYPHASE and WINOGRAD and WINO_BF16S and tuple(weight.shape[2:]) == (3, 3) and s == (1, 1) and dl[0] == dl[1]
                and dl[0] > 1 and p == dl and min(weight.shape[0], weight.shape[1]) >= 64):
            y = _dilated_polyphase(x, weight, dl[0])
            if y is not None:
                return y if bias is None else y + _like_act(bias, y).view(1, -1, 1, 1)
        if WINOGRAD and WINO_BF16S and wino_conv3d_applicable(x5, w5, (1,) + s, (0,) + p, (1,) + dl):
            y = _WinoConv.apply(x5, w5, _slot_of(x)).squeeze(2)
            return y if bias is None else y + _like_act(bias, y).view(1, -1, 1, 1)
        y = _conv16_route(x5, w5, bias, (1,) + s, (0,) + p, (1,) + dl, False, (0, 0, 0))
        if y is None:
            y = _to_act(_ConvNd.apply(x5, w5, bias, (1,) + s, (0,) + p, (1,) + dl, False, (0, 0, 0)))
        return y.squeeze(2)
    if (GEMM_LAYERS and TILE_HINT == 0 and PRECISION == "fp32" and x.is_cuda and tuple(weight.shape[2:]) == (1, 1)
            and s == (1, 1) and p == (0, 0) and weight.shape[1] >= GEMM_MIN_CIN):
        return linear_cl(x, weight, bias)           # wide pointwise conv = plain GEMM on the channels-last buffer
    if (DILATED_POLYPHASE and WINOGRAD and TILE_HINT == 0 and x.is_cuda and tuple(weight.shape[2:]) == (3, 3) and s == (1, 1)
            and dl[0] == dl[1] and dl[0] > 1 and p == dl and min(weight.shape[0], weight.shape[1]) >= 64):
        y = _dilated_polyphase(x, weight, dl[0])
        if y is not None:
            return y if bias is None else y + bias.view(1, -1, 1, 1)
    x5, w5 = x.unsqueeze(2), weight.unsqueeze(2)
    if WINOGRAD and TILE_HINT == 0 and wino_conv3d_applicable(x5, w5, (1,) + s, (0,) + p, (1,) + dl):
        y = _WinoConv.apply(x5, w5, _slot_of(x)).squeeze(2)
        return y if bias is None else y + bias.view(1, -1, 1, 1)
    y = _ConvNd.apply(x5, w5, bias, (1,) + s, (0,) + p, (1,) + dl, False, (0, 0, 0), _slot_of(x))
    return y.squeeze(2)


# -------------------------------------------------------------------------------------------------
# Winograd F(2x2x2, 3x3x3) convolution for the wide stride-1 3x3x3 layers
# -------------------------------------------------------------------------------------------------

_WINO_G = None


def _wino_g(device):
    global _WINO_G
    if _WINO_G is None or _WINO_G.device != device:
        _WINO_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], device=device)
    return _WINO_G


# BASELINE configs[3] ("bf16 mixed precision, MFMA 3D-conv path").  Three modes, never switched implicitly:
#   "fp32"           the parity contract of the headline metric (default);
#   "bf16"           (round 4) bf16 STORAGE: activations and activation gradients are bf16 channels-last tensors between the
#                    layers (convolutions on csrc/conv_bf16.hip, normalisations with bf16 I/O, the F(2,3) Winograd pipeline with
#                    bf16 on both sides); fp32 islands: parameters and their gradients (masters), normalisation statistics,
#                    softmaxes, the BRI attention, the scatter, the losses -- what mmcv's Fp16OptimizerHook / auto_fp16 keep
#                    in fp32 (reference mmdet_train.py:131-134), with bf16 in place of fp16;
#   "bf16_operands"  (rounds 1-3) tensors stay fp32 in HBM, the MFMA operands are rounded to bf16 in registers and the
#                    Winograd-domain tensors are stored as bf16: kept for A/B runs.
PRECISION = os.environ.get("SSBEV_PRECISION", "fp32")
_MODES = ("fp32", "bf16", "bf16_operands")


def set_precision(mode):
    global PRECISION
    if mode not in _MODES:
        raise ValueError(f"precision must be one of {_MODES}, got {mode!r}")
    PRECISION = mode


def storage_bf16():
    """Are activations stored as bf16 between layers (precision mode "bf16")?"""
    return PRECISION == "bf16"


def act_dtype():
    return torch.bfloat16 if PRECISION == "bf16" else torch.float32


def _like_act(y, ref):
    """A broadcast parameter (bias) in the dtype of the activation it is added to (bf16 + fp32 would promote the sum)."""
    return y if y.dtype == ref.dtype else y.to(ref.dtype)


def _wino_call(name, src, dims, out_shape, dtype=torch.float32):
    lib = capi.load()
    dst = torch.empty(out_shape, dtype=dtype, device=src.device)
    capi.check(getattr(lib, name)(capi.ptr(src), capi.ptr(dst), C.byref(dims), capi.stream()), name)
    return dst


# Opt-in: (h,w)-only transforms + the depth axis of F(2,3) inside a hand-written MFMA GEMM kernel (half the HBM traffic of
# the batched-GEMM pipeline, but this first version of the kernel runs at 60-75 TF/s and does not beat it yet:
# 128->128 1.34 vs 1.41 ms, 384->192 4.6 vs 3.6 ms; tools/wino_gemm_probe.py)
WINO_DEPTH_FUSED = os.environ.get("SSBEV_WINO_DEPTH_FUSED", "0") != "0"


# Opt-in: hand-written LDS-streaming MFMA kernel for the 64 frequency GEMMs (default: one rocBLAS batched GEMM, which is
# still 20-40 % faster on these shapes: tools/wino_gemm_probe.py)
WINO_OWN_GEMM = os.environ.get("SSBEV_WINO_OWN_GEMM", "0") != "0"


def gemm16_nn(a16, b32, out_fp32=False, tag=None):
    """C[b] = A[b] @ B[b]: a16 [Bt, M, K] bf16, b32 [Bt, K, N] fp32 (rounded to bf16 while it is packed) -> [Bt, M, N] bf16 / fp32,
    fp32 accumulation on conv_igemm16_kernel (csrc/conv_bf16.hip).  None when the shapes do not fit (K % 32, N % 8)."""
    lib = capi.load()
    Bt, M, K = a16.shape
    N = b32.shape[2]
    d = capi.Gemm16Dims(M, N, K, Bt, int(out_fp32))
    n = lib.ssbev_gemm16_packed_elems(C.byref(d))
    if n == 0 or a16.dtype != torch.bfloat16 or b32.dtype != torch.float32 or tuple(b32.shape[:2]) != (Bt, K):
        return None
    a16, b32 = a16.contiguous(), b32.contiguous()
    packed = torch.empty(n, dtype=torch.int16, device=a16.device)
    out = torch.empty(Bt, M, N, dtype=torch.float32 if out_fp32 else torch.bfloat16, device=a16.device)
    with _span("gemm16", 2.0 * Bt * M * K * N, 2.0 * (a16.numel() + out.numel()) + 4.0 * b32.numel(), tag or f"gemm16 {Bt}x[{M}x{K}x{N}]"):
        capi.check(lib.ssbev_gemm16_pack(capi.ptr(b32), capi.ptr(packed), C.byref(d), capi.stream()), "ssbev_gemm16_pack")
        capi.check(lib.ssbev_gemm16_nn(capi.ptr(a16), capi.ptr(packed), capi.ptr(out), C.byref(d), capi.stream()), "ssbev_gemm16_nn")
    return out


# bf16 storage mode: the Winograd frequency products on the own bf16 kernel instead of torch.bmm / rocBLAS (0 = library)
GEMM16_OWN = os.environ.get("SSBEV_GEMM16_OWN", "1") != "0"


def _bmm16(V, U):
    """V [nf, T, K] bf16 x U [nf, K, N] fp32 -> bf16 [nf, T, N]"""
    if GEMM16_OWN and V.is_cuda:
        M = gemm16_nn(V, U, tag=f"wino16 gemm {V.shape[0]}x[{V.shape[1]}x{V.shape[2]}x{U.shape[2]}]")
        if M is not None:
            return M
    return torch.bmm(V, U.to(torch.bfloat16))


def gemm16_tn(a16, b16, tag=None):
    """C[b] = A[b]^T @ B[b] over the row axis: a16 [Bt, M, K], b16 [Bt, M, N] bf16 -> [Bt, K, N] fp32 on gemm16_tn_kernel
    (csrc/conv_bf16.hip: LDS-DMA staging + ds_read_b64_tr_b16 operands).  None when the shapes do not fit (K % 8, N % 8)."""
    lib = capi.load()
    Bt, M, K = a16.shape
    N = b16.shape[2]
    d = capi.Gemm16Dims(M, N, K, Bt, 1)
    if (K % 8 or N % 8 or a16.dtype != torch.bfloat16 or b16.dtype != torch.bfloat16 or tuple(b16.shape[:2]) != (Bt, M)
            or Bt >= 65536):
        return None
    a16, b16 = a16.contiguous(), b16.contiguous()
    out = torch.empty(Bt, K, N, dtype=torch.float32, device=a16.device)
    ws = torch.empty(max(int(lib.ssbev_gemm16_tn_workspace(C.byref(d))), 4), dtype=torch.float32, device=a16.device)
    with _span("gemm16", 2.0 * Bt * M * K * N, 2.0 * (a16.numel() + b16.numel()) + 4.0 * out.numel(), tag or f"gemm16 tn {Bt}x[{K}x{M}x{N}]"):
        capi.check(lib.ssbev_gemm16_tn(capi.ptr(a16), capi.ptr(b16), capi.ptr(out), C.byref(d), capi.ptr(ws), ws.numel(),
                                       capi.stream()), "ssbev_gemm16_tn")
    return out


def _bmm16_tn(V, Z):
    """V [nf, T, K]^T x Z [nf, T, N] (bf16) -> fp32 [nf, K, N]: the weight-gradient frequency products"""
    if GEMM16_OWN and V.is_cuda:
        g = gemm16_tn(V, Z, tag=f"wino16 wgrad gemm {V.shape[0]}x[{V.shape[2]}x{V.shape[1]}x{Z.shape[2]}]")
        if g is not None:
            return g
    return torch.bmm(V.transpose(1, 2), Z, out_dtype=torch.float32)


def _wino_bgemm(V, w, Cout, Cin, mode):
    """M[xi] = V[xi] @ U[xi] for the 64 frequencies on the LDS-streaming MFMA kernel (mode 0 forward, 1 data gradient)."""
    lib = capi.load()
    T, K = V.shape[1], V.shape[2]
    N = Cout if mode == 0 else Cin
    Wp = torch.empty(lib.ssbev_wino_dgemm_packed_elems(Cout, Cin), dtype=torch.float32, device=V.device)
    capi.check(lib.ssbev_wino_dgemm_pack(capi.ptr(w), capi.ptr(Wp), Cout, Cin, mode, capi.stream()), "ssbev_wino_dgemm_pack")
    M = torch.empty(64, T, N, dtype=torch.float32, device=V.device)
    capi.check(lib.ssbev_wino_bgemm(capi.ptr(V), capi.ptr(Wp), capi.ptr(M), T, K, N, capi.stream()), "ssbev_wino_bgemm")
    return M


def _wino_depth_fused(xcl, w, B, D, H, W, K, N, mode):
    """3-D Winograd conv of a channels-last volume with K input / N output channels: (h,w) transform -> depth-fused
    frequency GEMM (HIP MFMA kernel, the depth axis of F(2,3) done in registers) -> (h,w) output transform.
    mode 0: forward (w [N,K,3,3,3]); mode 1: data gradient (w [K,N,3,3,3], mirrored taps)."""
    lib = capi.load()
    Cout, Cin = (N, K) if mode == 0 else (K, N)
    R = B * D * (H // 2) * (W // 2)
    P = _wino_call("ssbev_wino2d_input_transform", xcl, capi.WinoDims(B, D, H, W, K), (16, R, K))
    Wp = torch.empty(lib.ssbev_wino_dgemm_packed_elems(Cout, Cin), dtype=torch.float32, device=xcl.device)
    capi.check(lib.ssbev_wino_dgemm_pack(capi.ptr(w), capi.ptr(Wp), Cout, Cin, mode, capi.stream()), "ssbev_wino_dgemm_pack")
    Mo = torch.empty(16, R, N, dtype=torch.float32, device=xcl.device)
    dims = capi.WinoDims(B, D, H, W, K)
    capi.check(lib.ssbev_wino_dgemm(capi.ptr(P), capi.ptr(Wp), capi.ptr(Mo), C.byref(dims), N, capi.stream()),
               "ssbev_wino_dgemm")
    del P
    return _wino_call("ssbev_wino2d_output_transform", Mo, capi.WinoDims(B, D, H, W, N), (B, D, H, W, N))


# F(4,3) tiles along h and w (F(2,3) along d): 4.5x / 2.25x transformed domain instead of 8x / 4x and 6x / 4x fewer
# multiply-adds instead of 3.375x / 2.25x; used whenever H and W are multiples of 4 (SSBEV_WINO_F43=0: F(2,3) everywhere)
WINO_F43 = os.environ.get("SSBEV_WINO_F43", "1") != "0"
# ... and for the 2-D layers (DepthNet, 640 channels on the 48x160 map) only on request: there F(4,3)^2 saves ~1 ms per step
# but its rounding error (reduction over 640 channels of values scaled by the 4 / 5 / 8 entries) more than doubles the
# gradient noise of the DepthNet parameters in the full-step parity test (L2 1.1 % vs 0.45 %), for no accuracy budget left
WINO_F43_2D = os.environ.get("SSBEV_WINO_F43_2D", "0") != "0"
# ... or, opt-in, inside a `wino_f43_2d_scope()` (r3: DepthNet wraps its own 640 -> 640 layers; SSBEV_WINO_F43_2D_DEPTHNET=1).
# Measured in r3 (profiles/r3p_f43_2d.txt): F(4,3)^2 on every 2-D layer = -1.5 ms per step, forward logits unchanged
# (1.6e-4), but the stereo feature net's 640 -> 128 conv sits in front of the cost volume and its 1e-5 forward change flips
# ReLU decisions in the 20-layer 3-D stack behind it: gradient distance of that stack to the oracle 0.87 % -> 1.85-2.1 %,
# over the 2e-2 gate.  Limited to DepthNet the step is still 1.5 ms faster and every gate passes (worst gradient 1.3e-2),
# but DepthNet's own gradients then sit at 0.9 % where the oracle moves by 0.2 % under a one-ulp input change
# (tests/test_gpu_fullsize.py::test_gradient_gate_vs_oracle_noise_floor fails its 3x criterion): visible rounding, not
# noise floor.  Parity first: off by default.
WINO_F43_2D_DEPTHNET = os.environ.get("SSBEV_WINO_F43_2D_DEPTHNET", "0") != "0"
_F43_2D_SCOPE = 0


class wino_f43_2d_scope:
    """2-D 3x3 layers built inside this scope use F(4,3)^2 tiles (the choice is recorded per call for its backward)."""

    def __enter__(self):
        global _F43_2D_SCOPE
        _F43_2D_SCOPE += 1 if WINO_F43_2D_DEPTHNET else 0
        return self

    def __exit__(self, *exc):
        global _F43_2D_SCOPE
        _F43_2D_SCOPE -= 1 if WINO_F43_2D_DEPTHNET else 0
        return False
# F(4,3) along d as well (F(4x4x4): 216 GEMMs, 8x fewer multiply-adds, 3.375x transformed domain) when D % 4 == 0: opt-in.
# Measured 106.3 vs 108.8 ms/step, but the third non-+-1 axis brings the gradient noise of the full-step parity test back
# to 1.1 % (tools/grad_l2_probe.py), so the default keeps F(2,3) along d.
WINO_F444 = os.environ.get("SSBEV_WINO_F444", "0") != "0"
WINO_F444_MIN_CIN = int(os.environ.get("SSBEV_WINO_F444_MIN_CIN", "0"))     # with F444 on: only layers at least this wide


# Default realisation of the wide 3-D layers (r2): F(4,3)^2 over (h, w) in memory + F(2,3) along d in registers inside the
# hand-written MFMA contraction csrc/winograd_fused.hip (forward, data and weight gradient): no library GEMM, transformed
# tensors 2.25x instead of 4.5x.  SSBEV_WINO_DF=0 restores the F(2x4x4) transforms + 144 batched rocBLAS GEMMs.
WINO_DF = os.environ.get("SSBEV_WINO_DF", "1") != "0"
WINO_DF_MIN_ROWS = int(os.environ.get("SSBEV_WINO_DF_MIN_ROWS", "1024"))


def _wino_df_applicable(B, D, H, W, Cin, Cout):
    if not (WINO_DF and WINO_F43 and PRECISION == "fp32" and not (WINO_DEPTH_FUSED or WINO_OWN_GEMM or WINO_F444)):
        return False
    # Layers with few rows per frequency (512 channels on the 32 x 32 x 4 grid: 256 rows against 4 x 512 x 512 weights per
    # frequency) are weight-streaming problems: 151 MB of transformed weights for 10 GF of MFMA work.  The row-blocked fused
    # kernel re-reads that slab per 32-row block; they stay on the 144-GEMM pipeline (SSBEV_WINO_DF_MIN_ROWS=0: all fused).
    if B * D * (H // 4) * (W // 4) < WINO_DF_MIN_ROWS:
        return False
    lib = capi.load()
    return bool(lib.ssbev_wino43_df_supported(C.byref(capi.WinoDims(B, D, H, W, Cin)), Cout)) and \
        bool(lib.ssbev_wino43_df_supported(C.byref(capi.WinoDims(B, D, H, W, Cout)), Cin)) and Cout % 4 == 0


def _wino_df_gemm(xcl, w, B, D, H, W, K, N, mode, tag, fl, into=None):
    """(h,w) input transform -> depth-fused MFMA contraction -> (h,w) output transform of a channels-last volume with K
    input / N output channels.  mode 0: forward (w [N,K,3,3,3]); mode 1: data gradient (w [K,N,3,3,3], mirrored taps).
    Returns (y_cl, P) -- P [36, B*D*H/4*W/4, K] is what the weight gradient needs."""
    lib = capi.load()
    Cout, Cin = (N, K) if mode == 0 else (K, N)
    R = B * D * (H // 4) * (W // 4)
    with _span("wino_transform", 0.0, 4.0 * xcl.numel() * 3.25, tag + " in"):
        P = _wino_call("ssbev_wino43_2d_input_transform", xcl, capi.WinoDims(B, D, H, W, K), (36, R, K))
    Wp = torch.empty(lib.ssbev_wino43_df_packed_elems(Cout, Cin), dtype=torch.float32, device=xcl.device)
    capi.check(lib.ssbev_wino43_df_pack(capi.ptr(w), capi.ptr(Wp), Cout, Cin, mode, capi.stream()), "ssbev_wino43_df_pack")
    Mo = torch.empty(36, R, N, dtype=torch.float32, device=xcl.device)
    dims = capi.WinoDims(B, D, H, W, K)
    nby = 4.0 * (P.numel() + Mo.numel() + Wp.numel())          # the kernel's own operands: P and Mo are 2.25x the activations
    # span family "conv_wino_fused:<MT><NW>": one entry per template instance of wino_df_kernel (= per kernel symbol)
    with _span(f"conv_wino_fused:{lib.ssbev_wino43_df_instance(C.byref(dims), N)}" if KERNEL_TIMER is not None else "conv_wino_fused",
               fl, nby, tag, fl / 6.0):
        capi.check(lib.ssbev_wino43_df_gemm(capi.ptr(P), capi.ptr(Wp), capi.ptr(Mo), C.byref(dims), N, capi.stream()),
                   "ssbev_wino43_df_gemm")
    with _span("wino_transform", 0.0, 4.0 * B * D * H * W * N * (3.25 + (into is not None)), tag + " out"):
        if into is not None:          # gradient slot: add to the gradient another consumer already left there
            odims = capi.WinoDims(B, D, H, W, N)
            capi.check(lib.ssbev_wino43_2d_output_transform_acc(capi.ptr(Mo), capi.ptr(into), C.byref(odims), capi.stream()),
                       "ssbev_wino43_2d_output_transform_acc")
            y = into
        else:
            y = _wino_call("ssbev_wino43_2d_output_transform", Mo, capi.WinoDims(B, D, H, W, N), (B, D, H, W, N))
    return y, P


class _WinoConvDF(torch.autograd.Function):
    """3x3x3 / stride 1 / pad 1 convolution on the depth-fused Winograd kernels (see WINO_DF)."""

    @staticmethod
    def forward(ctx, x, weight, slot=None):
        ctx.slot = slot
        if ctx.needs_input_grad[1]:
            streams.note_use(weight)
        xcl = to_cl(_f32(x, "wino_conv"))
        B, D, H, W, Cin = xcl.shape
        Cout = weight.shape[0]
        fl = 2.0 * B * D * H * W * Cin * Cout * 27
        w = weight.detach().contiguous()
        y, P = _wino_df_gemm(xcl, w, B, D, H, W, Cin, Cout, 0, f"winoDF fwd {Cin}->{Cout} {D}x{H}x{W}", fl)
        ctx.save_for_backward(P if weight.requires_grad else None, weight)
        ctx.geom = (B, D, H, W, Cin, Cout, fl)
        return from_cl(y)

    @staticmethod
    def backward(ctx, gy):
        P, weight = ctx.saved_tensors
        B, D, H, W, Cin, Cout, fl = ctx.geom
        gcl = to_cl(gy)
        lib = capi.load()
        w = weight.detach().contiguous()
        gx = gw = None

        def weight_gradient():
            R = B * D * (H // 4) * (W // 4)
            with _span("wino_transform", 0.0, 4.0 * gcl.numel() * 3.25, "winoDF wgrad adjoint"):
                Z = _wino_call("ssbev_wino43_2d_output_adjoint", gcl, capi.WinoDims(B, D, H, W, Cout), (36, R, Cout))
            dims = capi.WinoDims(B, D, H, W, Cin)
            ws = _ws(lib.ssbev_wino43_df_wgrad_workspace(C.byref(dims), Cout), gy.device)
            gwt = torch.empty_like(w)
            nby = 4.0 * (P.numel() + Z.numel() + 144 * Cin * Cout)
            with _span("conv_wino_fused_wgrad", fl, nby, f"winoDF wgrad {Cin}->{Cout} {D}x{H}x{W}", fl / 6.0):
                capi.check(lib.ssbev_wino43_df_wgrad(capi.ptr(P), capi.ptr(Z), capi.ptr(gwt), C.byref(dims), Cout, capi.ptr(ws),
                                                     ws.numel(), capi.stream()), "ssbev_wino43_df_wgrad")
            return gwt

        gw_side = ctx.needs_input_grad[1] and streams.wgrad_on_side(weight)
        if gw_side:                          # leaf of the backward chain: next to the data gradient, not in front of it
            with streams.on_side(gy.device, gcl, P) as side:
                gw = weight_gradient()
                side.publish(gw)
        if ctx.needs_input_grad[0]:
            into = _slot_target(ctx.slot, torch.empty((B, D, H, W, Cin), device="meta"))
            gxcl, _ = _wino_df_gemm(gcl, w, B, D, H, W, Cout, Cin, 1, f"winoDF dgrad {Cin}->{Cout} {D}x{H}x{W}", fl, into=into)
            if ctx.slot is not None:
                ctx.slot.buf = gxcl
            gx = from_cl(gxcl)
        if ctx.needs_input_grad[1] and not gw_side:
            gw = weight_gradient()
        return gx, gw, None


class _WinoConv(torch.autograd.Function):
    """3x3(x3) / stride 1 / pad 1 convolution in the Winograd domain: HIP transforms + NF plain GEMMs.  x logical
    [B,Cin,D,H,W] channels-last, weight [Cout,Cin,kd,3,3] with kd = 3 (3-D, even D) or kd = 1 (2-D over (H,W); D is a
    batch axis).  Tiles: F(4,3) along h and w when both are multiples of 4 (NF = 144 / 36), else F(2,3) (NF = 64 / 16)."""

    @staticmethod
    def _plan(three_d, D, H, W, bf, cin=1 << 30):
        # bf16 mode stays on F(2,3): its +-1 transforms add no error of their own, while the F(4,3) matrices amplify the
        # bf16 rounding of V / M by their 4 / 5 / 8 entries (measured: 11 % max error against 1 % for F(2,3))
        f43 = WINO_F43 and (three_d or WINO_F43_2D or _F43_2D_SCOPE > 0) and not bf and H % 4 == 0 and W % 4 == 0 and \
            not (three_d and (WINO_DEPTH_FUSED or WINO_OWN_GEMM))
        if f43 and three_d and WINO_F444 and D % 4 == 0 and cin >= WINO_F444_MIN_CIN:
            return 4, "ssbev_wino444_", 216, 4, 8.0
        if f43:
            pre = "ssbev_wino43_" if three_d else "ssbev_wino43_2d_"
            nf, th, reduction = (144, 4, 6.0) if three_d else (36, 4, 4.0)
        else:
            pre = "ssbev_wino_" if three_d else "ssbev_wino2d_"
            nf, th, reduction = (64, 2, 3.375) if three_d else (16, 2, 2.25)
        return (1 if f43 else 0), pre, nf, th, reduction

    @staticmethod
    def forward(ctx, x, weight, slot=None):
        ctx.slot = slot
        if ctx.needs_input_grad[1]:
            streams.note_use(weight)
        bf = PRECISION != "fp32"
        a16 = storage_bf16()              # bf16 tensors on the activation side too (`_bf16a` transforms)
        xcl = to_cl(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)) if a16 else to_cl(_f32(x, "wino_conv"))
        B, D, H, W, Cin = xcl.shape
        Cout, three_d = weight.shape[0], weight.shape[2] == 3
        f43, pre, nf, th, red = _WinoConv._plan(three_d, D, H, W, bf, Cin)
        T = B * (D // (4 if f43 == 4 else 2) if three_d else D) * (H // th) * (W // th)
        lib = capi.load()
        w = weight.detach().contiguous()
        fl = 2.0 * B * D * H * W * Cin * Cout * (27 if three_d else 9)
        fused = three_d and WINO_DEPTH_FUSED and not bf and not f43
        wt = lib.ssbev_wino43_weight_transform if f43 else lib.ssbev_wino_weight_transform
        wdim = 4 if f43 == 4 else (3 if three_d else 2)
        if not fused:
            U = torch.empty(nf, Cin, Cout, dtype=torch.float32, device=x.device)
            capi.check(wt(capi.ptr(w), capi.ptr(U), Cout, Cin, wdim, 0, capi.stream()), "ssbev_wino_weight_transform")
        tag = f"wino{ {0: '', 1: '43', 4: '444'}[f43]} fwd {Cin}->{Cout} {D}x{H}x{W}"
        nby = 4.0 * (B * D * H * W * (Cin + Cout) + (27 if three_d else 9) * Cin * Cout)
        with _span("conv_winograd", fl, nby, tag, fl / red):
            if fused:      # (h,w)-transformed tensors only (4x), the depth axis of F(2,3) inside the GEMM kernel
                y = _wino_depth_fused(xcl, w, B, D, H, W, Cin, Cout, 0)
                V = None
            elif bf:
                sfx = "_bf16a" if a16 else "_bf16"
                V = _wino_call(pre + "input_transform" + sfx, xcl, capi.WinoDims(B, D, H, W, Cin), (nf, T, Cin), torch.bfloat16)
                M = _bmm16(V, U)
                y = _wino_call(pre + "output_transform" + sfx, M, capi.WinoDims(B, D, H, W, Cout), (B, D, H, W, Cout),
                               torch.bfloat16 if a16 else torch.float32)
            else:
                V = _wino_call(pre + "input_transform", xcl, capi.WinoDims(B, D, H, W, Cin), (nf, T, Cin))
                M = _wino_bgemm(V, w, Cout, Cin, 0) if (three_d and WINO_OWN_GEMM and not f43) else (gemm_nn(V, U, tag=tag + " gemm") if own_gemm_site("wino") else torch.bmm(V, U))
                y = _wino_call(pre + "output_transform", M, capi.WinoDims(B, D, H, W, Cout), (B, D, H, W, Cout))
        ctx.save_for_backward(xcl if fused else V, weight)
        ctx.geom = (B, D, H, W, Cin, Cout, T, three_d, fl, fused, bf, f43, pre, nf, red, a16)
        return from_cl(y)

    @staticmethod
    def backward(ctx, gy):
        V, weight = ctx.saved_tensors            # (depth-fused path: V is the channels-last input, transformed below)
        B, D, H, W, Cin, Cout, T, three_d, fl, fused, bf, f43, pre, nf, red, a16 = ctx.geom
        sfx, fdt = (("_bf16a" if a16 else "_bf16"), torch.bfloat16) if bf else ("", torch.float32)
        gcl = to_cl(gy)
        if a16 and gcl.dtype != torch.bfloat16:
            gcl = gcl.to(torch.bfloat16)
        adt = torch.bfloat16 if a16 else torch.float32          # dtype of the activation-side tensors
        lib = capi.load()
        w = weight.detach().contiguous()
        nd = 4 if f43 == 4 else (3 if three_d else 2)
        gx = gw = None
        vtag = {0: "", 1: "43", 4: "444"}[f43]
        nby = 4.0 * (B * D * H * W * (Cin + Cout) + (27 if three_d else 9) * Cin * Cout)
        if ctx.needs_input_grad[0] and fused:
            with _span("conv_winograd", fl, nby, f"wino dgrad {Cin}->{Cout} {D}x{H}x{W}", fl / red):
                gx = from_cl(_wino_depth_fused(gcl, w, B, D, H, W, Cout, Cin, 1))
        elif ctx.needs_input_grad[0]:
            Ut = torch.empty(nf, Cout, Cin, dtype=torch.float32, device=gy.device)
            wt = lib.ssbev_wino43_weight_transform if f43 else lib.ssbev_wino_weight_transform
            capi.check(wt(capi.ptr(w), capi.ptr(Ut), Cout, Cin, nd, 1, capi.stream()), "ssbev_wino_weight_transform")
            with _span("conv_winograd", fl, nby, f"wino{vtag} dgrad {Cin}->{Cout} {D}x{H}x{W}", fl / red):
                Vg = _wino_call(pre + "input_transform" + sfx, gcl, capi.WinoDims(B, D, H, W, Cout), (nf, T, Cout), fdt)
                Mx = _wino_bgemm(Vg, w, Cout, Cin, 1) if (three_d and WINO_OWN_GEMM and not bf and not f43) \
                    else ((_bmm16(Vg, Ut) if fdt == torch.bfloat16 else torch.bmm(Vg, Ut.to(fdt))) if (bf or not own_gemm_site("wino")) else gemm_nn(Vg, Ut, tag="wino dgrad gemm"))
                del Vg
                acc_fn = {"ssbev_wino2d_": "ssbev_wino2d_output_transform_acc",
                          "ssbev_wino43_2d_": "ssbev_wino43_2d_output_transform_acc"}.get(pre) if not bf else None
                into = _slot_target(ctx.slot, torch.empty((B, D, H, W, Cin), device="meta")) if acc_fn else None
                if into is not None:
                    odims = capi.WinoDims(B, D, H, W, Cin)
                    capi.check(getattr(lib, acc_fn)(capi.ptr(Mx), capi.ptr(into), C.byref(odims), capi.stream()), acc_fn)
                    gxcl = into
                else:
                    gxcl = _wino_call(pre + "output_transform" + sfx, Mx, capi.WinoDims(B, D, H, W, Cin), (B, D, H, W, Cin), adt)
                if ctx.slot is not None:
                    ctx.slot.buf = gxcl
            gx = from_cl(gxcl)
        if ctx.needs_input_grad[1]:
            def weight_gradient(V=V):
                with _span("conv_winograd_wgrad", fl, nby, f"wino{vtag} wgrad {Cin}->{Cout} {D}x{H}x{W}", fl / red):
                    if fused:
                        V = _wino_call(pre + "input_transform", V, capi.WinoDims(B, D, H, W, Cin), (nf, T, Cin))
                    Z = _wino_call(pre + "output_adjoint" + sfx, gcl, capi.WinoDims(B, D, H, W, Cout), (nf, T, Cout), fdt)
                    gU = _bmm16_tn(V, Z) if bf else (gemm_tn(V, Z, tag="wino wgrad gemm") if own_gemm_site("wino") else torch.bmm(V.transpose(1, 2), Z))
                gwt = torch.empty_like(w)
                wg = lib.ssbev_wino43_weight_grad if f43 else lib.ssbev_wino_weight_grad
                capi.check(wg(capi.ptr(gU), capi.ptr(gwt), Cout, Cin, nd, capi.stream()), "ssbev_wino_weight_grad")
                return gwt

            if streams.wgrad_on_side(weight):    # after the data gradient here (rocBLAS workspaces are per stream: no reordering)
                with streams.on_side(gy.device, gcl, V) as side:
                    gw = weight_gradient()
                    side.publish(gw)
            else:
                gw = weight_gradient()
        return gx, gw, None


def conv_flops_3x3(B, D, H, W, Cin, Cout):
    return 2.0 * B * D * H * W * Cin * Cout * 27


def wino_conv3d_applicable(x, weight, stride, padding, dilation):
    """Wide stride-1 3x3x3 (or 1x3x3) 'same' layers on even grids; narrow ones are memory-bound in the 8x (4x) larger
    transformed domain."""
    k = tuple(weight.shape[2:])
    if not (x.is_cuda and tuple(stride) == (1, 1, 1) and tuple(dilation) == (1, 1, 1) and weight.shape[0] >= 64
            and weight.shape[1] >= 64 and weight.shape[1] % 4 == 0):
        return False
    if k == (3, 3, 3):
        return tuple(padding) == (1, 1, 1) and all(int(n) % 2 == 0 for n in x.shape[2:])
    if k == (1, 3, 3):
        return tuple(padding) == (0, 1, 1) and all(int(n) % 2 == 0 for n in x.shape[3:])
    return False


# -------------------------------------------------------------------------------------------------
# deformable convolution v1 (DepthNet's DCN)
# -------------------------------------------------------------------------------------------------


class _DcnIm2col(torch.autograd.Function):
    """x [B,C,H,W], offset [B,2*k*k,H,W] -> cols [G, B*H*W, k*k*C/G] (bilinear taps, group-major)."""

    @staticmethod
    def forward(ctx, x, offset, groups, k, pad, dil):
        lib = capi.load()
        xcl, ocl = to_cl(_f32(x, "dcn")), to_cl(_f32(offset, "dcn"))
        B, H, W, Cch = xcl.shape
        d = capi.DcnDims(B, Cch, H, W, groups, k, pad, dil)
        cols = torch.empty(groups, B * H * W, k * k * (Cch // groups), dtype=torch.float32, device=x.device)
        with _span("dcn_sample", 0.0, 4.0 * (xcl.numel() + ocl.numel() + cols.numel()), "fwd   dcn_im2col"):
            capi.check(lib.ssbev_dcn_im2col(capi.ptr(xcl), capi.ptr(ocl), capi.ptr(cols), C.byref(d), capi.stream()),
                       "ssbev_dcn_im2col")
        ctx.save_for_backward(xcl, ocl)
        ctx.d = d
        return cols

    @staticmethod
    def backward(ctx, gcols):
        lib = capi.load()
        xcl, ocl = ctx.saved_tensors
        gx, goff = torch.empty_like(xcl), torch.empty_like(ocl)
        ws = _ws(lib.ssbev_dcn_col2im_workspace(C.byref(ctx.d)), xcl.device)
        with _span("dcn_sample", 0.0, 4.0 * (2 * xcl.numel() + 2 * ocl.numel() + gcols.numel()), "bwd   dcn_col2im"):
            capi.check(lib.ssbev_dcn_col2im(capi.ptr(xcl), capi.ptr(ocl), capi.ptr(gcols.contiguous()), capi.ptr(gx),
                                            capi.ptr(goff), C.byref(ctx.d), capi.ptr(ws), ws.numel(), capi.stream()),
                       "ssbev_dcn_col2im")
        return from_cl(gx), from_cl(goff), None, None, None, None


def deform_conv2d(x, offset, weight, groups=1, padding=1, dilation=1):
    """mmcv deform_conv2d (v1, stride 1, deform_groups 1, no bias): HIP sampling + one dense MFMA 1x1 conv per group."""
    Cout, Cg, k, _ = weight.shape
    B, _, H, W = x.shape
    cols = _DcnIm2col.apply(x, offset, int(groups), int(k), int(padding), int(dilation))
    Cog = Cout // groups
    Kc = k * k * Cg
    if (x.is_cuda and own_gemm_site("linear") and _gemm_supported(cols) and PRECISION == "fp32" and Cog % 4 == 0 and Kc % 4 == 0):
        # the grouped contraction as ONE batched product per direction, written straight into / read straight out of the
        # channels-last [pixels, Cout] map through the kernels' row and batch strides: no concatenation forward, no 177 MB
        # stack of the column gradients backward (round 6; rounds 2-5: four products + cat / unbind per direction)
        w3 = weight.view(groups, Cog, Cg, k, k).permute(0, 1, 3, 4, 2).reshape(groups, Cog, Kc)
        y2 = _GroupedLinearCL.apply(cols, w3)                              # [B*H*W, Cout]
        return from_cl(y2.view(B, H, W, Cout))
    outs = []
    # unbind, not cols[g]: the backward of four selects is four zero-filled [groups, ...] tensors plus three accumulation adds
    # (0.4 ms per step on the 177 MB column tensor); the backward of unbind is one stack
    for g, cg in enumerate(cols.unbind(0)):
        colg = cg.view(B, H, W, Kc).permute(0, 3, 1, 2)                 # channels-last [B, K*Cg, H, W]
        wg = weight[g * Cog:(g + 1) * Cog].permute(0, 2, 3, 1).reshape(Cog, Kc, 1, 1)
        outs.append(conv2d(colg, wg, None, 1, 0, 1))
    return torch.cat(outs, dim=1) if groups > 1 else outs[0]


class _GroupedLinearCL(torch.autograd.Function):
    """y[r, g * N + n] = sum_k cols[g, r, k] * w[g, n, k]: the group contraction of a grouped pointwise layer (DepthNet's DCN,
    BD:490-498, groups = 4) on the own GEMM kernels, batch element = group.  The group's [R, N] block is a column slice of the
    channels-last [R, G * N] map: leading dimension G * N, batch stride N -- the operand strides of ssbev_gemm_* do the
    concatenation (forward, C side) and the split (backward, A side)."""

    @staticmethod
    def forward(ctx, cols, w3):
        lib = capi.load()
        cols, w3 = cols.contiguous(), w3.contiguous()
        G, R, K = cols.shape
        N = w3.shape[1]
        y = torch.empty(R, G * N, dtype=torch.float32, device=cols.device)
        g = _gdims(R, N, K, G, K, K, G * N, R * K, N * K, N)
        ws = _ws(lib.ssbev_gemm_nt_workspace(C.byref(g)), cols.device)
        with _span("gemm_own", 2.0 * G * R * N * K, 4.0 * (cols.numel() + w3.numel() + y.numel()), f"linear fwd grouped {G}x[{K}->{N}] rows={R}"):
            capi.check(lib.ssbev_gemm_nt(capi.ptr(cols), capi.ptr(w3), None, capi.ptr(y), C.byref(g), capi.ptr(ws), ws.numel(),
                                         capi.stream()), "ssbev_gemm_nt")
        ctx.save_for_backward(cols, w3)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = capi.load()
        cols, w3 = ctx.saved_tensors
        G, R, K = cols.shape
        N = w3.shape[1]
        gy = gy.contiguous()
        gcols = gw = None
        fl = 2.0 * G * R * N * K
        if ctx.needs_input_grad[0]:      # gcols[g] [R, K] = gy[:, g N : (g + 1) N] [R, N] @ w3[g] [N, K]            (NN)
            gcols = torch.empty_like(cols)
            g = _gdims(R, K, N, G, G * N, K, K, N, N * K, R * K)
            ws = _ws(lib.ssbev_gemm_nn_workspace(C.byref(g)), gy.device)
            with _span("gemm_own", fl, 4.0 * (gy.numel() + w3.numel() + gcols.numel()), f"linear dgrad grouped {G}x[{K}->{N}] rows={R}"):
                capi.check(lib.ssbev_gemm_nn(capi.ptr(gy), capi.ptr(w3), None, capi.ptr(gcols), C.byref(g), capi.ptr(ws), ws.numel(),
                                             capi.stream()), "ssbev_gemm_nn")
        if ctx.needs_input_grad[1]:      # gw[g] [N, K] = gy[:, g N : (g + 1) N]^T [N, R] @ cols[g] [R, K]               (TN)
            gw = torch.empty_like(w3)
            g = _gdims(R, K, N, G, G * N, K, K, N, R * K, N * K)
            ws = _ws(lib.ssbev_gemm_tn_workspace(C.byref(g)), gy.device)
            with _span("gemm_own", fl, 4.0 * (gy.numel() + cols.numel() + gw.numel()), f"linear wgrad grouped {G}x[{K}->{N}] rows={R}"):
                capi.check(lib.ssbev_gemm_tn(capi.ptr(gy), capi.ptr(cols), capi.ptr(gw), C.byref(g), capi.ptr(ws), ws.numel(),
                                             capi.stream()), "ssbev_gemm_tn")
        return gcols, gw


# -------------------------------------------------------------------------------------------------
# normalisation
# -------------------------------------------------------------------------------------------------


# The fused ReLU's sign pattern travels to backward as a bit mask (1/32 of the tensor) instead of y being re-read twice
GN_RELU_MASK = os.environ.get("SSBEV_GN_RELU_MASK", "1") != "0"


def _norm_ext(device, running=None, n=0):
    """ssbev_norm_ext: (running_mean, running_var, momentum) of a training-mode BatchNorm, updated inside the statistics finalize."""
    e = capi.NormExt(None, None, 0.0, int(n))
    if running is not None:
        e.running_mean, e.running_var, e.momentum = running[0].data_ptr(), running[1].data_ptr(), float(running[2])
    return e


class _GroupNorm(torch.autograd.Function):
    """GroupNorm on a channels-last volume with optional fused residual add and ReLU."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, groups, eps, relu, as_batch, given_mean, given_rstd, pre_act=0, res_slot=None,
                running=None):
        lib = capi.load()
        ctx.res_slot = res_slot
        ctx.set_materialize_grads(False)           # no zero tensors for the (non-differentiable) statistics outputs
        xcl, io = _act(to_cl(x), "group_norm")
        Cch = xcl.shape[-1]
        B = 1 if as_batch else xcl.shape[0]
        S = xcl.numel() // (B * Cch)
        rcl = to_cl(residual if residual.dtype == xcl.dtype else residual.to(xcl.dtype)) if residual is not None else None
        given = given_mean is not None
        d = capi.NormDims(B, Cch, groups, S, float(eps), int(relu), int(given), int(pre_act), 0, 0, io)
        y = torch.empty_like(xcl)
        mean = given_mean.contiguous() if given else torch.empty(B * groups, dtype=torch.float32, device=x.device)
        rstd = given_rstd.contiguous() if given else torch.empty(B * groups, dtype=torch.float32, device=x.device)
        ws = _ws(lib.ssbev_groupnorm_workspace(C.byref(d)), x.device)
        w, b = weight.detach().contiguous(), bias.detach().contiguous()
        nb = float(xcl.element_size()) * xcl.numel() * (3 + (residual is not None) - 2 * given)      # stats read + apply read/write
        use_mask = bool(relu) and GN_RELU_MASK and not given and any(ctx.needs_input_grad[:4])
        mask = torch.empty(lib.ssbev_groupnorm_mask_words(C.byref(d)), dtype=torch.int64, device=x.device) if use_mask else None
        ext = _norm_ext(x.device, running, B * S)
        with _span("groupnorm", 0.0, nb, f"fwd   N C={Cch} G={groups} S={S} res={int(residual is not None)}"):
            capi.check(lib.ssbev_groupnorm_fwd_ext(capi.ptr(xcl), capi.ptr(w), capi.ptr(b), capi.ptr(rcl), capi.ptr(y),
                                                   capi.ptr(mean), capi.ptr(rstd), capi.ptr(mask) if use_mask else None,
                                                   C.byref(d), C.byref(ext), capi.ptr(ws), ws.numel(), capi.stream()),
                       "ssbev_groupnorm_fwd_ext")
        ctx.save_for_backward(xcl, (mask if use_mask else y) if relu else None, w, mean, rstd)
        ctx.use_mask = use_mask
        ctx.meta = (B, S, Cch, groups, float(eps), int(relu), residual is not None, given, int(pre_act), io)
        ctx.mark_non_differentiable(mean, rstd)
        return from_cl(y), mean, rstd

    @staticmethod
    def backward(ctx, gy, _gm, _gr):
        lib = capi.load()
        xcl, y, w, mean, rstd = ctx.saved_tensors
        B, S, Cch, groups, eps, relu, has_res, given, pre_act, io = ctx.meta
        if given:
            raise capi.SsbevError("eval-mode (given statistics) normalisation has no HIP backward; use torch for it")
        gcl = to_cl(gy if gy.dtype == xcl.dtype else gy.to(xcl.dtype))
        d = capi.NormDims(B, Cch, groups, S, eps, relu, 0, pre_act, 0, 0, io)
        gx = torch.empty_like(xcl)
        gres = torch.empty_like(xcl) if has_res else None
        gg = torch.empty(Cch, dtype=torch.float32, device=gy.device)
        gb = torch.empty(Cch, dtype=torch.float32, device=gy.device)
        ws = _ws(lib.ssbev_groupnorm_workspace(C.byref(d)), gy.device)
        # stats: x, gy (, y); apply: x, gy (, y) -> gx (, gres); the ReLU bit mask replaces both y reads
        nb = float(xcl.element_size()) * xcl.numel() * (5 + (relu and not ctx.use_mask) + has_res)
        ext = _norm_ext(gy.device)
        with _span("groupnorm", 0.0, nb, f"bwd   N C={Cch} G={groups} S={S} res={int(has_res)}"):
            capi.check(lib.ssbev_groupnorm_bwd_ext(capi.ptr(gcl), capi.ptr(xcl), None if ctx.use_mask else capi.ptr(y),
                                                   capi.ptr(y) if ctx.use_mask else None, capi.ptr(w), capi.ptr(mean),
                                                   capi.ptr(rstd), capi.ptr(gx), capi.ptr(gres), capi.ptr(gg), capi.ptr(gb),
                                                   C.byref(d), C.byref(ext), capi.ptr(ws), ws.numel(), capi.stream()),
                       "ssbev_groupnorm_bwd_ext")
        if has_res and ctx.res_slot is not None and ctx.res_slot.buf is None:
            ctx.res_slot.buf = gres        # first gradient of a forked activation: later consumers accumulate into it
        return from_cl(gx), gg, gb, (from_cl(gres) if has_res else None), None, None, None, None, None, None, None, None, None


NORM_CAT = os.environ.get("SSBEV_NORM_CAT", "1") != "0"     # concatenations of normalised branches without torch.cat (0 = cat)


class _NormCat(torch.autograd.Function):
    """``torch.cat([relu?(norm_i(x_i)) for i], dim=1)`` with every normalisation writing straight into its channel slice of
    the concatenated channels-last tensor (``ssbev_norm_dims.ld_y``) and reading its slice of the incoming gradient in place
    (``ld_gy``): no concatenation pass forward, no slice copies backward (SECONDFPN3D, second_fpn3d.py:113-116: 0.4 GB written
    and re-read per step; ASPP, BD:404-410).  ``parts``: (groups, eps, as_batch) per branch; tensors: x_i, weight_i, bias_i."""

    @staticmethod
    def forward(ctx, parts, relu, running, *tensors):
        lib = capi.load()
        ctx.set_materialize_grads(False)
        n = len(parts)
        extra = tensors[3 * n] if len(tensors) > 3 * n else None     # an un-normalised last branch (ASPP's image-level one)
        xs = [to_cl(tensors[3 * i]) for i in range(n)]
        adt = xs[0].dtype if all(x.dtype == xs[0].dtype for x in xs) else torch.float32      # one storage type for the whole cat
        if adt == torch.bfloat16 and (any(x.shape[-1] % 8 for x in xs) or (extra is not None and extra.shape[1] % 8)):
            adt = torch.float32            # 16-byte bf16 lanes need channel slices that start and end on multiples of 8
        xs = [_act(x if x.dtype == adt else x.to(adt), "norm_cat")[0] for x in xs]
        io, esz = (1, 2) if adt == torch.bfloat16 else (0, 4)
        ws_ = [tensors[3 * i + 1].detach().contiguous() for i in range(n)]
        bs_ = [tensors[3 * i + 2].detach().contiguous() for i in range(n)]
        Cs = [x.shape[-1] for x in xs]
        Ctot = sum(Cs) + (extra.shape[1] if extra is not None else 0)
        out = torch.empty(xs[0].shape[:-1] + (Ctot,), dtype=adt, device=xs[0].device)
        if extra is not None:
            out[..., sum(Cs):].copy_(extra.detach().movedim(1, -1))
        saved, stats, metas = [], [], []
        c0 = 0
        for i, (groups, eps, as_batch) in enumerate(parts):
            x = xs[i]
            B = 1 if as_batch else x.shape[0]
            S = x.numel() // (B * Cs[i])
            d = capi.NormDims(B, Cs[i], groups, S, float(eps), int(relu), 0, 0, Ctot, 0, io)
            mean = torch.empty(B * groups, dtype=torch.float32, device=x.device)
            rstd = torch.empty(B * groups, dtype=torch.float32, device=x.device)
            ws = _ws(lib.ssbev_groupnorm_workspace(C.byref(d)), x.device)
            mask = torch.empty(lib.ssbev_groupnorm_mask_words(C.byref(d)), dtype=torch.int64, device=x.device) if relu else None
            run = running[i] if running is not None else None
            ext = _norm_ext(x.device, run, B * S)
            with _span("groupnorm", 0.0, 3.0 * esz * x.numel(), f"fwd   N C={Cs[i]} G={groups} S={S} cat@{c0}/{Ctot}"):
                capi.check(lib.ssbev_groupnorm_fwd_ext(capi.ptr(x), capi.ptr(ws_[i]), capi.ptr(bs_[i]), None,
                                                       C.c_void_p(out.data_ptr() + esz * c0), capi.ptr(mean), capi.ptr(rstd), capi.ptr(mask),
                                                       C.byref(d), C.byref(ext), capi.ptr(ws), ws.numel(), capi.stream()),
                           "ssbev_groupnorm_fwd_ext")
            saved += [x, mask, ws_[i], mean, rstd]
            stats += [mean, rstd]
            metas.append((B, S, Cs[i], groups, float(eps), c0))
            c0 += Cs[i]
        ctx.save_for_backward(*[t for t in saved if t is not None])
        ctx.has_mask = bool(relu)
        ctx.metas, ctx.relu, ctx.Ctot, ctx.io = metas, int(relu), Ctot, io
        ctx.extra_at = sum(Cs) if extra is not None else None
        ctx.mark_non_differentiable(*stats)
        return (from_cl(out), *stats)

    @staticmethod
    def backward(ctx, gy, *_):
        lib = capi.load()
        per = 5 if ctx.has_mask else 4
        sv = ctx.saved_tensors
        adt, esz = (torch.bfloat16, 2) if ctx.io else (torch.float32, 4)
        gcl = to_cl(gy if gy.dtype == adt else gy.to(adt))    # the whole concatenated gradient, channels-last
        grads = []
        for i, (B, S, Cch, groups, eps, c0) in enumerate(ctx.metas):
            if ctx.has_mask:
                x, mask, w, mean, rstd = sv[per * i: per * i + 5]
            else:
                (x, w, mean, rstd), mask = sv[per * i: per * i + 4], None
            d = capi.NormDims(B, Cch, groups, S, eps, ctx.relu, 0, 0, 0, ctx.Ctot, ctx.io)
            gx = torch.empty_like(x)
            gg = torch.empty(Cch, dtype=torch.float32, device=gy.device)
            gb = torch.empty(Cch, dtype=torch.float32, device=gy.device)
            ws = _ws(lib.ssbev_groupnorm_workspace(C.byref(d)), gy.device)
            ext = _norm_ext(gy.device)
            with _span("groupnorm", 0.0, 5.0 * esz * x.numel(), f"bwd   N C={Cch} G={groups} S={S} cat@{c0}/{ctx.Ctot}"):
                capi.check(lib.ssbev_groupnorm_bwd_ext(C.c_void_p(gcl.data_ptr() + esz * c0), capi.ptr(x), None, capi.ptr(mask),
                                                       capi.ptr(w), capi.ptr(mean), capi.ptr(rstd), capi.ptr(gx), None, capi.ptr(gg),
                                                       capi.ptr(gb), C.byref(d), C.byref(ext), capi.ptr(ws), ws.numel(), capi.stream()),
                           "ssbev_groupnorm_bwd_ext")
            grads += [from_cl(gx), gg, gb]
        if ctx.extra_at is not None:
            grads.append(gcl[..., ctx.extra_at:].movedim(-1, 1))
        return (None, None, None, *grads)


def norm_cat(xs, norms, relu=True, extra=None, running=None):
    """Concatenation along the channel axis of relu?(norm_i(x_i)) (+ ``extra`` as it is, behind them): ``norms`` = (weight, bias,
    groups, eps, as_batch) per branch; ``running``: per branch None or (running_mean, running_var, momentum) of a training-mode
    BatchNorm, updated inside the operator.  Returns (y, [(mean_i, rstd_i)])."""
    parts = tuple((int(g), float(e), bool(ab)) for (_, _, g, e, ab) in norms)
    flat = []
    for x, (w, b, _, _, _) in zip(xs, norms):
        flat += [x, w, b]
    if extra is not None:
        flat.append(extra)
    out = _NormCat.apply(parts, bool(relu), running, *flat)
    return out[0], [(out[1 + 2 * i], out[2 + 2 * i]) for i in range(len(xs))]


def norm_cat_supported(xs):
    return (NORM_CAT and GN_RELU_MASK and all(x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.shape[1] % 4 == 0 for x in xs)
            and all(x.shape[0] == xs[0].shape[0] and x.shape[2:] == xs[0].shape[2:] for x in xs))


def group_norm(x, groups, weight, bias, eps=1e-5, residual=None, relu=False, pre_act=None):
    """relu?(GroupNorm(act?(x)) + residual?) on a channels-last volume (any spatial rank); ``pre_act="gelu"`` normalises
    gelu(x) (exact form) without materialising it."""
    if pre_act not in (None, "gelu"):
        raise ValueError(f"group_norm: unknown pre_act {pre_act!r}")
    return _GroupNorm.apply(x, weight, bias, residual, int(groups), eps, relu, False, None, None, 1 if pre_act else 0,
                            _slot_of(residual))[0]


def batch_norm_train(x, weight, bias, eps=1e-5, residual=None, relu=False, running=None):
    """Training-mode BatchNorm (batch statistics): returns (y, mean[C], rstd[C]).  ``running`` = (running_mean, running_var,
    momentum): nn.BatchNorm's momentum update of the running statistics, done inside the operator (in the finalize tail of
    the statistics kernel; no launch of its own)."""
    return _GroupNorm.apply(x, weight, bias, residual, x.shape[1], eps, relu, True, None, None, 0, _slot_of(residual), running)


def bn_update_running_(running_mean, running_var, mean, rstd, momentum, eps, n):
    """In-place momentum update of BatchNorm running statistics from the batch (mean, rstd): one launch."""
    lib = capi.load()
    capi.check(lib.ssbev_bn_update_running(capi.ptr(mean), capi.ptr(rstd), capi.ptr(running_mean), capi.ptr(running_var),
                                           running_mean.numel(), float(momentum), float(eps), int(n), capi.stream()),
               "ssbev_bn_update_running")


def batch_norm_eval(x, weight, bias, running_mean, running_var, eps=1e-5, residual=None, relu=False):
    """Inference-mode BatchNorm with the running statistics (forward only on the HIP path)."""
    rstd = torch.rsqrt(running_var + eps)
    return _GroupNorm.apply(x, weight, bias, residual, x.shape[1], eps, relu, True, running_mean, rstd)[0]


class _DualNorm(torch.autograd.Function):
    """relu?(N_a(xa) + N_b(xb)) in one operator (csrc/groupnorm.hip, ssbev_groupnorm2_*): each side a per-sample GroupNorm or a
    batch normalisation; returns (y, mean_a, rstd_a, mean_b, rstd_b)."""

    @staticmethod
    def forward(ctx, xa, wa, ba, xb, wb, bb, ga, gb, eps_a, eps_b, relu, a_batch, b_batch, run_a=None, run_b=None):
        lib = capi.load()
        ctx.set_materialize_grads(False)
        acl, io = _act(to_cl(xa), "dual_norm")
        bcl = to_cl(xb if xb.dtype == acl.dtype else xb.to(acl.dtype))
        if acl.shape != bcl.shape:
            raise capi.SsbevError(f"dual_norm: shapes differ, {tuple(acl.shape)} vs {tuple(bcl.shape)}")
        B, Cch = acl.shape[0], acl.shape[-1]
        S = acl.numel() // (B * Cch)
        d = capi.Norm2Dims(B, Cch, int(ga), int(gb), S, float(eps_a), float(eps_b), int(relu), int(a_batch), int(b_batch), io)
        dev = xa.device
        y = torch.empty_like(acl)
        mean_a = torch.empty((1 if a_batch else B) * ga, dtype=torch.float32, device=dev)
        rstd_a = torch.empty_like(mean_a)
        mean_b = torch.empty((1 if b_batch else B) * gb, dtype=torch.float32, device=dev)
        rstd_b = torch.empty_like(mean_b)
        nbytes = lib.ssbev_groupnorm2_workspace(C.byref(d))
        if nbytes == 0:
            raise capi.SsbevError("dual_norm: unsupported dims (C % 4, C <= 1024, C % G)")
        ws = _ws(nbytes, dev)
        nd = capi.NormDims(B, Cch, ga, S, float(eps_a), int(relu), 0, 0, 0, 0, io)
        mask = torch.empty(lib.ssbev_groupnorm_mask_words(C.byref(nd)), dtype=torch.int64, device=dev) if relu else None
        wa_, ba_, wb_, bb_ = (t.detach().contiguous() for t in (wa, ba, wb, bb))
        ext = capi.Norm2Ext(None, None, 0.0, None, None, 0.0)
        if run_a is not None and a_batch:
            ext.running_mean_a, ext.running_var_a, ext.momentum_a = run_a[0].data_ptr(), run_a[1].data_ptr(), float(run_a[2])
        if run_b is not None and b_batch:
            ext.running_mean_b, ext.running_var_b, ext.momentum_b = run_b[0].data_ptr(), run_b[1].data_ptr(), float(run_b[2])
        with _span("groupnorm", 0.0, float(acl.element_size()) * acl.numel() * 5, f"fwd   N2 C={Cch} Ga={ga} Gb={gb} S={S}"):
            capi.check(lib.ssbev_groupnorm2_fwd_ext(capi.ptr(acl), capi.ptr(wa_), capi.ptr(ba_), capi.ptr(mean_a), capi.ptr(rstd_a),
                                                    capi.ptr(bcl), capi.ptr(wb_), capi.ptr(bb_), capi.ptr(mean_b), capi.ptr(rstd_b),
                                                    capi.ptr(y), capi.ptr(mask), C.byref(d), C.byref(ext), capi.ptr(ws), ws.numel(),
                                                    capi.stream()),
                       "ssbev_groupnorm2_fwd_ext")
        ctx.save_for_backward(acl, bcl, mask, wa_, wb_, mean_a, rstd_a, mean_b, rstd_b)
        ctx.meta = (B, Cch, int(ga), int(gb), S, float(eps_a), float(eps_b), int(relu), int(a_batch), int(b_batch), io)
        ctx.mark_non_differentiable(mean_a, rstd_a, mean_b, rstd_b)
        return from_cl(y), mean_a, rstd_a, mean_b, rstd_b

    @staticmethod
    def backward(ctx, gy, *_unused):
        lib = capi.load()
        acl, bcl, mask, wa_, wb_, mean_a, rstd_a, mean_b, rstd_b = ctx.saved_tensors
        d = capi.Norm2Dims(*ctx.meta)
        Cch, dev = ctx.meta[1], gy.device
        gcl = to_cl(gy if gy.dtype == acl.dtype else gy.to(acl.dtype))
        gxa, gxb = torch.empty_like(acl), torch.empty_like(bcl)
        gga, gba, ggb, gbb = (torch.empty(Cch, dtype=torch.float32, device=dev) for _ in range(4))
        ws = _ws(lib.ssbev_groupnorm2_workspace(C.byref(d)), dev)
        with _span("groupnorm", 0.0, float(acl.element_size()) * acl.numel() * 8, f"bwd   N2 C={Cch} Ga={ctx.meta[2]} Gb={ctx.meta[3]} S={ctx.meta[4]}"):
            ext = capi.Norm2Ext(None, None, 0.0, None, None, 0.0)
            capi.check(lib.ssbev_groupnorm2_bwd_ext(capi.ptr(gcl), capi.ptr(mask), capi.ptr(acl), capi.ptr(wa_), capi.ptr(mean_a),
                                                    capi.ptr(rstd_a), capi.ptr(bcl), capi.ptr(wb_), capi.ptr(mean_b), capi.ptr(rstd_b),
                                                    capi.ptr(gxa), capi.ptr(gxb), capi.ptr(gga), capi.ptr(gba), capi.ptr(ggb),
                                                    capi.ptr(gbb), C.byref(d), C.byref(ext), capi.ptr(ws), ws.numel(), capi.stream()),
                       "ssbev_groupnorm2_bwd_ext")
        return from_cl(gxa), gga, gba, from_cl(gxb), ggb, gbb, None, None, None, None, None, None, None, None, None


DUAL_NORM = os.environ.get("SSBEV_DUAL_NORM", "1") != "0"      # 0: the two-operator residual form (A/B timing)


def dual_norm_supported(xa, xb):
    return xa.is_cuda and xa.shape == xb.shape and xa.dim() >= 3 and xa.shape[1] % 4 == 0 and xa.shape[1] <= 1024


def dual_norm(xa, wa, ba, groups_a, eps_a, xb, wb, bb, groups_b, eps_b, relu=False, a_batch=False, b_batch=False,
              running_a=None, running_b=None):
    """relu?(N_a(xa) + N_b(xb)); ``x_batch`` = statistics over the batch axis too (BatchNorm when groups == channels);
    ``running_x`` = (running_mean, running_var, momentum) of a training-mode BatchNorm side, updated inside the operator.
    Returns (y, (mean_a, rstd_a), (mean_b, rstd_b))."""
    y, ma, ra, mb, rb = _DualNorm.apply(xa, wa, ba, xb, wb, bb, int(groups_a), int(groups_b), eps_a, eps_b, relu,
                                        bool(a_batch), bool(b_batch), running_a, running_b)
    return y, (ma, ra), (mb, rb)


# -------------------------------------------------------------------------------------------------
# softmax along a strided axis
# -------------------------------------------------------------------------------------------------


class _SoftmaxAxis(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dim):
        lib = capi.load()
        x = _f32(x, "softmax").contiguous()
        outer = 1
        for n in x.shape[:dim]:
            outer *= n
        Cn = x.shape[dim]
        inner = x.numel() // (outer * Cn)
        y = torch.empty_like(x)
        with _span("softmax", 0.0, 8.0 * x.numel(), "fwd   softmax_axis"):
            capi.check(lib.ssbev_softmax_axis_fwd(capi.ptr(x), capi.ptr(y), outer, Cn, inner, capi.stream()),
                       "ssbev_softmax_axis_fwd")
        ctx.save_for_backward(y)
        ctx.meta = (outer, Cn, inner)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = capi.load()
        (y,) = ctx.saved_tensors
        outer, Cn, inner = ctx.meta
        gy = gy.contiguous()
        gx = torch.empty_like(y)
        with _span("softmax", 0.0, 12.0 * y.numel(), "bwd   softmax_axis"):
            capi.check(lib.ssbev_softmax_axis_bwd(capi.ptr(y), capi.ptr(gy), capi.ptr(gx), outer, Cn, inner, capi.stream()),
                       "ssbev_softmax_axis_bwd")
        return gx, None


def softmax_rows_(x):
    """In-place softmax over the innermost axis of a contiguous fp32 tensor with rows of <= 8192 floats (the BRI
    attention matrix, attention.py:66-68): ssbev_softmax_rows_fwd, one read + one write.  No autograd."""
    n = x.shape[-1]
    capi.check(capi.load().ssbev_softmax_rows_fwd(capi.ptr(x), capi.ptr(x), x.numel() // n, n, capi.stream()),
               "ssbev_softmax_rows_fwd")
    return x


def softmax_rows_bwd_(y, gy):
    """gy <- y * (gy - sum(y * gy, -1)) in place (softmax backward along the innermost axis)."""
    n = y.shape[-1]
    capi.check(capi.load().ssbev_softmax_rows_bwd(capi.ptr(y), capi.ptr(gy), capi.ptr(gy), y.numel() // n, n,
                                                  capi.stream()), "ssbev_softmax_rows_bwd")
    return gy


def softmax_rows_ok(x):
    return x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] % 4 == 0 and x.shape[-1] <= 8192


def softmax(x, dim):
    """softmax over ``dim``.  A non-innermost axis of a standard-contiguous tensor runs on the strided-axis HIP kernel;
    an axis that is innermost in memory (channels-last 2-D maps) is already a single fast row kernel in ATen."""
    dim = dim % x.dim()
    if x.is_cuda and x.is_contiguous() and dim != x.dim() - 1:
        return _SoftmaxAxis.apply(x, dim)
    return torch.softmax(x, dim)


# -------------------------------------------------------------------------------------------------
# trilinear x2 upsample of the logits
# -------------------------------------------------------------------------------------------------


class _Trilinear2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = capi.load()
        xcl = to_cl(_f32(x, "trilinear2x"))
        B, D, H, W, Cch = xcl.shape
        d = capi.UpsampleDims(B, D, H, W, Cch)
        y = torch.empty(B, 2 * D, 2 * H, 2 * W, Cch, dtype=torch.float32, device=x.device)
        with _span("trilinear", 0.0, 4.0 * (xcl.numel() + y.numel()), "fwd   trilinear2x"):
            capi.check(lib.ssbev_trilinear2x_fwd(capi.ptr(xcl), capi.ptr(y), C.byref(d), capi.stream()),
                       "ssbev_trilinear2x_fwd")
        ctx.dims = (B, D, H, W, Cch)
        return from_cl(y)

    @staticmethod
    def backward(ctx, gy):
        lib = capi.load()
        B, D, H, W, Cch = ctx.dims
        gcl = to_cl(gy)
        d = capi.UpsampleDims(B, D, H, W, Cch)
        gx = torch.empty(B, D, H, W, Cch, dtype=torch.float32, device=gy.device)
        with _span("trilinear", 0.0, 4.0 * (gcl.numel() + gx.numel()), "bwd   trilinear2x"):
            capi.check(lib.ssbev_trilinear2x_bwd(capi.ptr(gcl), capi.ptr(gx), C.byref(d), capi.stream()),
                       "ssbev_trilinear2x_bwd")
        return from_cl(gx)


def upsample_trilinear(x, size):
    """F.interpolate(x, size, mode='trilinear', align_corners=False); the exact-x2 case the path uses
    (logits -> label grid) runs on the HIP kernel, any other ratio on ATen's device kernel."""
    size = tuple(int(v) for v in size)
    if tuple(x.shape[-3:]) == size:
        return x
    if all(o == 2 * i for o, i in zip(size, x.shape[-3:])) and x.shape[1] % 4 == 0:
        return _Trilinear2x.apply(x)
    return torch.nn.functional.interpolate(x.contiguous(), size=size, mode="trilinear", align_corners=False)


# -------------------------------------------------------------------------------------------------
# fused occupancy-head epilogue
# -------------------------------------------------------------------------------------------------


class _OccLossSums(torch.autograd.Function):
    """logits [B,20,D,H,W] (coarse), label uint8 [B,2D,2H,2W] -> (diff [41] f64: ce_num, sum_p[20], nom[20];
    aux [3+20+400] f64: ce_den, M, (unused), cnt[20], conf[20][20]).  Only ``diff`` carries gradient."""

    @staticmethod
    def forward(ctx, logits, label_u8, class_weight):
        lib = capi.load()
        xcl = to_cl(_f32(logits, "occ_loss"))
        B, D, H, W, Cch = xcl.shape
        d = capi.UpsampleDims(B, D, H, W, Cch)
        ns = lib.ssbev_occ_loss_num_sums()
        sums = torch.empty(ns, dtype=torch.float64, device=logits.device)
        cw = class_weight.to(device=logits.device, dtype=torch.float32).contiguous()
        lab = label_u8.contiguous()
        ws = _ws(lib.ssbev_occ_loss_workspace(C.byref(d)), logits.device)
        with _span("occ_loss", 0.0, 4.0 * xcl.numel() + lab.numel(), "fwd   occ_loss"):
            capi.check(lib.ssbev_occ_loss_fwd(capi.ptr(xcl), capi.ptr(lab), capi.ptr(cw), capi.ptr(sums), C.byref(d),
                                              capi.ptr(ws), ws.numel(), capi.stream()), "ssbev_occ_loss_fwd")
        ctx.save_for_backward(xcl, lab, cw)
        nc = Cch
        diff = torch.cat((sums[0:1], sums[3:3 + 2 * nc]))
        aux = torch.cat((sums[1:3], sums[3 + 2 * nc:]))
        ctx.mark_non_differentiable(aux)
        return diff, aux

    @staticmethod
    def backward(ctx, gdiff, _gaux):
        lib = capi.load()
        xcl, lab, cw = ctx.saved_tensors
        B, D, H, W, Cch = xcl.shape
        d = capi.UpsampleDims(B, D, H, W, Cch)
        coef = gdiff.to(torch.float32).contiguous()
        gx = torch.empty_like(xcl)
        ws = _ws(lib.ssbev_occ_loss_bwd_workspace(C.byref(d)), xcl.device)
        with _span("occ_loss", 0.0, 8.0 * xcl.numel() + lab.numel(), "bwd   occ_loss"):
            capi.check(lib.ssbev_occ_loss_bwd(capi.ptr(xcl), capi.ptr(lab), capi.ptr(cw), capi.ptr(coef), capi.ptr(gx),
                                              C.byref(d), capi.ptr(ws), ws.numel(), capi.stream()), "ssbev_occ_loss_bwd")
        return from_cl(gx), None, None


def occ_loss_sums(logits, label_u8, class_weight):
    return _OccLossSums.apply(logits, label_u8, class_weight)


DEPTH_BCE = os.environ.get("SSBEV_DEPTH_BCE", "1") != "0"    # fused depth loss (0 = the ~35 ATen ops of the tensor expression)


class _DepthBce(torch.autograd.Function):
    """weight * BCE(depth_pred, one-hot of the down-sampled LiDAR depth) over the pixels with a return (VT:349-416),
    ``ssbev_depth_bce_fwd / _bwd``: gt_depths [B, N, H, W], depth_pred [B*N, D, H/ds, W/ds] -> 0-dim loss."""

    @staticmethod
    def forward(ctx, gt_depths, depth_pred, ds, dbound, weight):
        lib = capi.load()
        B, N, H, W = gt_depths.shape
        BN, D, fH, fW = depth_pred.shape
        assert BN == B * N and fH * ds == H and fW * ds == W
        gt = _f32(gt_depths, "depth_bce").contiguous()
        pred = _f32(depth_pred, "depth_bce").contiguous()
        c0 = float(dbound[0] - dbound[2] / 2)                      # in double, like the reference's Python expression
        args = (BN, D, fH, fW, int(ds), c0, float(dbound[2]), float(weight))
        ws = torch.empty(lib.ssbev_depth_bce_workspace(BN, fH, fW), dtype=torch.uint8, device=pred.device)   # kept for backward
        out = torch.empty(2, dtype=torch.float32, device=pred.device)
        capi.check(lib.ssbev_depth_bce_fwd(capi.ptr(gt), capi.ptr(pred), capi.ptr(out), *args, capi.ptr(ws), ws.numel(),
                                           capi.stream()), "ssbev_depth_bce_fwd")
        ctx.save_for_backward(pred, out, ws)
        ctx.args = args
        return out[0]

    @staticmethod
    def backward(ctx, g):
        lib = capi.load()
        pred, out, ws = ctx.saved_tensors
        gp = torch.empty_like(pred)
        gl = g.to(torch.float32).reshape(1).contiguous()
        capi.check(lib.ssbev_depth_bce_bwd(capi.ptr(pred), capi.ptr(gl), capi.ptr(out), capi.ptr(gp), *ctx.args, capi.ptr(ws),
                                           capi.stream()), "ssbev_depth_bce_bwd")
        return None, gp, None, None, None


def depth_bce_loss(gt_depths, depth_pred, ds, dbound, weight):
    return _DepthBce.apply(gt_depths, depth_pred, ds, dbound, weight)


OCC_TAIL = os.environ.get("SSBEV_OCC_TAIL", "1") != "0"      # scalar loss algebra in one launch (0 = ~110 tiny ATen ops)


class _OccLossTail(torch.autograd.Function):
    """(diff [41] f64, aux f64) from _OccLossSums -> the three weighted losses + the two metric scalars (five 0-dim float
    tensors), ``ssbev_occ_loss_tail``: forward also returns the Jacobian, backward scales its three rows."""

    @staticmethod
    def forward(ctx, diff, aux, w_ce, w_sem, w_geo):
        lib = capi.load()
        ctx.set_materialize_grads(False)
        nd = diff.numel()
        nc = (nd - 1) // 2
        # back to the kernel's layout: ce_num, ce_den, M, sum_p, nom, cnt, conf
        sums = torch.cat((diff[0:1], aux[0:2], diff[1:], aux[2:])).contiguous()
        assert sums.numel() == lib.ssbev_occ_loss_num_sums() and nc * 2 + 1 == nd
        out = torch.empty(5, dtype=torch.float32, device=diff.device)
        jac = torch.empty(3, nd, dtype=torch.float64, device=diff.device)
        capi.check(lib.ssbev_occ_loss_tail(capi.ptr(sums), float(w_ce), float(w_sem), float(w_geo), capi.ptr(out), capi.ptr(jac),
                                           capi.stream()), "ssbev_occ_loss_tail")
        ctx.save_for_backward(jac)
        outs = tuple(out[i] for i in range(5))
        ctx.mark_non_differentiable(outs[3], outs[4])
        return outs

    @staticmethod
    def backward(ctx, g_ce, g_sem, g_geo, _g3, _g4):
        (jac,) = ctx.saved_tensors
        gd = None
        for k, g in enumerate((g_ce, g_sem, g_geo)):
            if g is not None:
                term = jac[k] * g.to(torch.float64)
                gd = term if gd is None else gd + term
        return gd, None, None, None, None


def occ_loss_tail(diff, aux, w_ce, w_sem, w_geo):
    return _OccLossTail.apply(diff, aux, w_ce, w_sem, w_geo)


# -------------------------------------------------------------------------------------------------
# Image branch: depthwise conv with "same" padding, Swish, squeeze-excitation pieces
# -------------------------------------------------------------------------------------------------

def same_padding(size, k, stride):
    """mmcv Conv2dAdaptivePadding (the conv_cfg of CustomEfficientNet, efficientnet.py:373): output = ceil(size/stride),
    total padding max((out-1)*stride + k - size, 0), the odd element goes to the bottom/right.  Returns (out, before, after)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2, total - total // 2


def linear_cl(x, weight, bias=None):
    """1x1 / stride-1 convolution of a channels-last map as the plain GEMM it is -- [pixels, Cin] x [Cin, Cout] on the
    channels-last buffer, no layout change -- on the own GEMM kernels (csrc/gemm.hip: forward, data and weight gradient) whenever
    Cout % 4 == 0, else on the library through autograd's mm.
    Used by the image branch, whose pointwise convs range from 2 rows (SE gates) to 245 k rows x 32..3840 channels."""
    xcl = to_cl(_f32(x, "linear_cl"))
    shp = xcl.shape
    w2 = weight.reshape(weight.shape[0], -1)
    x2 = xcl.reshape(-1, shp[-1])
    if own_gemm_site("linear") and x2.is_cuda and w2.shape[0] % 4 == 0:
        y = small_linear(x2, w2, bias)                 # (any row count; Cin zero-padded to a multiple of 4 when it is not one)
        return from_cl(y.view(*shp[:-1], w2.shape[0]))
    fl = 2.0 * x2.shape[0] * x2.shape[1] * w2.shape[0]
    nby = 4.0 * (x2.numel() + w2.numel() + x2.shape[0] * w2.shape[0])
    mult = 3.0 if (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)) else 1.0
    with _span("gemm_lib", fl * mult, nby * mult, f"fwd   linear_cl {x2.shape[1]}->{w2.shape[0]} rows={x2.shape[0]}"):
        y = torch.addmm(bias, x2, w2.t()) if bias is not None else torch.mm(x2, w2.t())
    return from_cl(y.view(*shp[:-1], w2.shape[0]))


def small_linear(x2, weight2, bias=None):
    """nn.functional.linear for the few-row products of the path (camera MLPs, squeeze-excite gates, CA3D's channel MLP:
    [B, K] x [N, K]^T): on the own GEMM kernels (N a multiple of 4; K is zero-padded to one), so that no rocBLAS kernel is launched
    for them either."""
    if x2.is_cuda and x2.dtype != torch.float32:
        x2 = x2.float()              # (a bf16 pooled vector in the storage mode: the few-row products stay fp32, like their weights)
    if (x2.is_cuda and x2.dim() == 2 and own_gemm_site("linear") and weight2.shape[0] % 4 == 0
            and weight2.shape[1] == x2.shape[1]):
        pad = -x2.shape[1] % 4
        if pad:                      # (the 27-feature camera vector: zero columns on both operands, 16-byte rows)
            x2, weight2 = torch.nn.functional.pad(x2, (0, pad)), torch.nn.functional.pad(weight2, (0, pad))
        return _LinearCL.apply(x2.contiguous(), weight2.contiguous(), bias)
    fl = 2.0 * x2.shape[0] * x2.shape[-1] * weight2.shape[0]
    with _span("gemm_lib", fl, 4.0 * (x2.numel() + weight2.numel()), f"fwd   linear (library) {x2.shape[-1]}->{weight2.shape[0]} rows={x2.shape[0]}"):
        return torch.nn.functional.linear(x2, weight2, bias)


class _DwConv2d(torch.autograd.Function):
    """Depthwise k x k conv (groups = channels), TF-"same" padding; x logical [B,C,H,W], weight [C,1,k,k]."""

    @staticmethod
    def forward(ctx, x, weight, stride):
        xcl = to_cl(_f32(x, "dwconv2d"))
        B, H, W, Cc = xcl.shape
        k = weight.shape[-1]
        Ho, pt, _ = same_padding(H, k, stride)
        Wo, pl, _ = same_padding(W, k, stride)
        d = capi.DwDims(B, Cc, H, W, Ho, Wo, k, stride, pt, pl)
        wt = weight.detach().reshape(Cc, k * k).t().contiguous()
        y = torch.empty(B, Ho, Wo, Cc, dtype=torch.float32, device=x.device)
        lib = capi.load()
        with _span("dwconv", 2.0 * y.numel() * k * k, 4.0 * (xcl.numel() + y.numel()), f"dw fwd C={Cc} {H}x{W} k{k} s{stride}"):
            capi.check(lib.ssbev_dwconv2d_fwd(capi.ptr(xcl), capi.ptr(wt), capi.ptr(y), C.byref(d), capi.stream()),
                       "ssbev_dwconv2d_fwd")
        ctx.save_for_backward(xcl, wt)
        ctx.d = d
        return from_cl(y)

    @staticmethod
    def backward(ctx, gy):
        xcl, wt = ctx.saved_tensors
        d = ctx.d
        gcl = to_cl(gy)
        lib = capi.load()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gxcl = torch.empty_like(xcl)
            with _span("dwconv", 2.0 * gcl.numel() * d.k * d.k, 4.0 * (xcl.numel() + gcl.numel()), f"dw dgrad C={d.C} k{d.k} s{d.stride}"):
                capi.check(lib.ssbev_dwconv2d_bwd_data(capi.ptr(gcl), capi.ptr(wt), capi.ptr(gxcl), C.byref(d), capi.stream()),
                           "ssbev_dwconv2d_bwd_data")
            gx = from_cl(gxcl)
        if ctx.needs_input_grad[1]:
            n = lib.ssbev_dwconv2d_bwd_weight_workspace(C.byref(d))
            ws = torch.empty(max(n, 4), dtype=torch.float32, device=gy.device)
            gwt = torch.empty_like(wt)
            with _span("dwconv", 2.0 * gcl.numel() * d.k * d.k, 4.0 * (xcl.numel() + gcl.numel()), f"dw wgrad C={d.C} k{d.k} s{d.stride}"):
                capi.check(lib.ssbev_dwconv2d_bwd_weight(capi.ptr(xcl), capi.ptr(gcl), capi.ptr(gwt), C.byref(d), capi.ptr(ws),
                                                         ws.numel(), capi.stream()), "ssbev_dwconv2d_bwd_weight")
            gw = gwt.t().reshape(d.C, 1, d.k, d.k).contiguous()
        return gx, gw, None


def depthwise_conv2d_same(x, weight, stride=1):
    return _DwConv2d.apply(x, weight, int(stride))


class _Swish(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x, "swish")
        ctx.cl = x.dim() >= 3
        xc = to_cl(x) if ctx.cl else x.contiguous()       # no copy for the channels-last activations of the branch
        n = xc.numel()
        ctx.torch_path = n % 4 != 0
        ctx.save_for_backward(xc)
        if ctx.torch_path:
            y = xc * torch.sigmoid(xc)
        else:
            y = torch.empty_like(xc)
            capi.check(capi.load().ssbev_swish_fwd(capi.ptr(xc), capi.ptr(y), n, capi.stream()), "ssbev_swish_fwd")
        return from_cl(y) if ctx.cl else y

    @staticmethod
    def backward(ctx, gy):
        (xc,) = ctx.saved_tensors
        g = to_cl(gy) if ctx.cl else gy.contiguous()
        if ctx.torch_path:
            s = torch.sigmoid(xc)
            gx = g * (s + xc * s * (1 - s))
        else:
            gx = torch.empty_like(xc)
            capi.check(capi.load().ssbev_swish_bwd(capi.ptr(xc), capi.ptr(g), capi.ptr(gx), xc.numel(), capi.stream()),
                       "ssbev_swish_bwd")
        return from_cl(gx) if ctx.cl else gx


def swish(x):
    """mmcv ``Swish``: x * sigmoid(x)."""
    return _Swish.apply(x)


def _chan_sum(a, bmul, B, S, Cc, scale):
    lib = capi.load()
    out = torch.empty(B, Cc, dtype=torch.float32, device=a.device)
    ws = torch.empty(max(lib.ssbev_chan_sum_workspace(B, S, Cc), 4), dtype=torch.float32, device=a.device)
    capi.check(lib.ssbev_chan_sum(capi.ptr(a), capi.ptr(bmul), capi.ptr(out), B, S, Cc, float(scale), capi.ptr(ws),
                                  ws.numel(), capi.stream()), "ssbev_chan_sum")
    return out


class _GlobalAvgPool(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d(1) of mmdet's SELayer: logical [B,C,H,W] -> [B,C,1,1]."""

    @staticmethod
    def forward(ctx, x):
        xcl = to_cl(_f32(x, "global_avg_pool"))
        B, H, W, Cc = xcl.shape
        ctx.shape = (B, H, W, Cc)
        return _chan_sum(xcl, None, B, H * W, Cc, 1.0 / (H * W)).view(B, Cc, 1, 1)

    @staticmethod
    def backward(ctx, g):
        B, H, W, Cc = ctx.shape
        gx = torch.empty(B, H, W, Cc, dtype=torch.float32, device=g.device)
        capi.check(capi.load().ssbev_chan_scale(None, capi.ptr(g.reshape(B, Cc).contiguous()), capi.ptr(gx), B, H * W, Cc,
                                                1.0 / (H * W), capi.stream()), "ssbev_chan_scale")
        return from_cl(gx)


class _ChanScale(torch.autograd.Function):
    """x * gate with gate [B,C,1,1] (the SE rescale, mmdet SELayer.forward)."""

    @staticmethod
    def forward(ctx, x, gate):
        xcl = to_cl(_f32(x, "chan_scale"))
        B, H, W, Cc = xcl.shape
        gt = gate.reshape(B, Cc).contiguous()
        y = torch.empty_like(xcl)
        capi.check(capi.load().ssbev_chan_scale(capi.ptr(xcl), capi.ptr(gt), capi.ptr(y), B, H * W, Cc, 1.0, capi.stream()),
                   "ssbev_chan_scale")
        ctx.save_for_backward(xcl, gt)
        return from_cl(y)

    @staticmethod
    def backward(ctx, gy):
        xcl, gt = ctx.saved_tensors
        B, H, W, Cc = xcl.shape
        gcl = to_cl(gy)
        gx = ggate = None
        if ctx.needs_input_grad[0]:
            gxcl = torch.empty_like(xcl)
            capi.check(capi.load().ssbev_chan_scale(capi.ptr(gcl), capi.ptr(gt), capi.ptr(gxcl), B, H * W, Cc, 1.0,
                                                    capi.stream()), "ssbev_chan_scale")
            gx = from_cl(gxcl)
        if ctx.needs_input_grad[1]:
            ggate = _chan_sum(gcl, xcl, B, H * W, Cc, 1.0).view(B, Cc, 1, 1)
        return gx, ggate


def global_avg_pool(x):
    return _GlobalAvgPool.apply(x)


def chan_scale(x, gate):
    return _ChanScale.apply(x, gate)
