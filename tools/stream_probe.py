"""The north-star streaming kernels in isolation at the KITTI size (D=192, 48x160 features, 128x128x16 grid): HIP-event time
per launch and algorithmic bytes / time against the 8 TB/s HBM peak.  Kernel variants are chosen by the SSBEV_* environment
switches read at the first launch (one process per variant).  usage: python tools/stream_probe.py [iters]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stereoscene_amd import capi, functional as F, model_zoo, synthetic as S  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def timed(fn, iters=ITERS):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]


def report(name, us, nbytes):
    print(f"{name:28s} {us:8.1f} us   {nbytes / 1e6:7.1f} MB algorithmic   {nbytes / us / 1e6:5.2f} TB/s = {nbytes / us / 1e6 / 8 * 100:4.1f} % of 8 TB/s")


def main():
    lib = capi.load()
    dev = "cuda"
    # ---- cost volume
    B, Cc, H, W, D, G = 1, 64, 48, 160, 192, 32
    l = torch.randn(B, H, W, Cc, device=dev)
    r = torch.randn(B, H, W, Cc, device=dev)
    cal = torch.tensor([393.8], device=dev)
    vol = torch.empty(B, D, H, W, G, device=dev)
    d = capi.GwcDims(B, Cc, G, D, H, W, 1.0, 1)
    st = capi.stream()
    report("reference: torch fill_", timed(lambda: vol.fill_(1.0)), 4.0 * vol.numel())
    src = torch.randn_like(vol)
    report("reference: torch sum (read)", timed(lambda: src.sum()), 4.0 * vol.numel())
    report("reference: torch copy_", timed(lambda: vol.copy_(src)), 8.0 * vol.numel())
    del src
    report("gwc_warp_fwd", timed(lambda: capi.check(lib.ssbev_gwc_warp_fwd(capi.ptr(l), capi.ptr(r), capi.ptr(cal), capi.ptr(vol),
                                                                        C.byref(d), st), "fwd")),
           4.0 * (l.numel() + r.numel() + vol.numel()))
    g = torch.randn_like(vol)
    gl, gr = torch.empty_like(l), torch.empty_like(r)
    ws = torch.empty(max(lib.ssbev_gwc_warp_bwd_workspace(C.byref(d)), 16), dtype=torch.uint8, device=dev)
    report("gwc_warp_bwd (fused, L+R)", timed(lambda: capi.check(lib.ssbev_gwc_warp_bwd_fused(
        capi.ptr(g), capi.ptr(l), capi.ptr(r), capi.ptr(cal), capi.ptr(gl), capi.ptr(gr), C.byref(d), capi.ptr(ws), ws.numel(), st),
        "bwd")), 4.0 * (g.numel() + 2 * l.numel() + 2 * r.numel()))
    report("gwc_warp_bwd (r1, 2 launches)", timed(lambda: capi.check(lib.ssbev_gwc_warp_bwd(
        capi.ptr(g), capi.ptr(l), capi.ptr(r), capi.ptr(cal), capi.ptr(gl), capi.ptr(gr), C.byref(d), st), "bwd1")),
        4.0 * (2 * g.numel() + 2 * l.numel() + 2 * r.numel()))
    del vol, g
    # ---- lift / splat
    cfg = S.CONFIGS["kitti_d192"]
    vt = model_zoo.build_detector(cfg).eval().img_view_transformer
    smp = S.synthetic_sample(cfg, B=1, tag="bench0")
    geom = vt.get_geometry(*[t.cuda() for t in smp["geo_l"]])
    depth = torch.softmax(torch.randn(1, vt.D, 48, 160, device=dev), 1).contiguous()
    feat = torch.randn(1, 48, 160, 128, device=dev)
    n = [int(v) for v in vt.nx.tolist()]
    vox = F.voxel_index(geom, vt.bx, vt.dx, vt.nx)
    starts, order = F.pool_prepare(vox, 1, *n)
    pd = F._pool_dims(1, vt.D * 48 * 160, 128, *n)
    ld = capi.LiftDims(1, vt.D, 48 * 160)
    out = torch.empty(1, *n, 128, device=dev)
    kept = int((vox >= 0).sum())
    print(f"frustum points kept: {kept} of {vox.numel()}, non-empty voxels {int((starts[1:] > starts[:-1]).sum())} of {starts.numel() - 1}, "
          f"longest list {int((starts[1:] - starts[:-1]).max())}")
    report("lift_splat_fwd (gather)", timed(lambda: capi.check(lib.ssbev_lift_splat_fwd(
        capi.ptr(depth), capi.ptr(feat), capi.ptr(starts), capi.ptr(order), capi.ptr(out), C.byref(pd), C.byref(ld), st), "ls")),
        4.0 * (depth.numel() + feat.numel() + order.numel() + starts.numel() + out.numel()))
    report("pool_prepare (CSR build)", timed(lambda: F.pool_prepare(vox, 1, *n)), 4.0 * 6 * vox.numel())
    go = torch.randn_like(out)
    gd, gf = torch.empty_like(depth), torch.empty_like(feat)
    report("lift_splat_bwd", timed(lambda: capi.check(lib.ssbev_lift_splat_bwd(
        capi.ptr(go), capi.ptr(depth), capi.ptr(feat), capi.ptr(vox), capi.ptr(gd), capi.ptr(gf), C.byref(pd), C.byref(ld), st),
        "lsb")), 4.0 * (go.numel() + 2 * depth.numel() + 2 * feat.numel() + vox.numel()))


if __name__ == "__main__":
    main()
