"""Where the fused lift + splat gather spends its time: list-length histogram of the KITTI frustum and the kernel timed on
(a) the real CSR, (b) an empty CSR (pure 134 MB zero store), (c) the real CSR with every list cut to <= L points."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F, capi, model_zoo, synthetic as S
dev = "cuda"
lib = capi.load()
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
lens = (starts[1:] - starts[:-1]).cpu()
edges = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
print("list length histogram:", {f"<{b}": int(((lens >= a) & (lens < b)).sum()) for a, b in zip(edges, edges[1:])})
print("points in lists of length >=", {t: int(lens[lens >= t].sum()) for t in (16, 32, 64, 128)})


def timed(st_, n_it=10):
    fn = lambda: capi.check(lib.ssbev_lift_splat_fwd(capi.ptr(depth), capi.ptr(feat), capi.ptr(st_), capi.ptr(order), capi.ptr(out),
                                                      C.byref(pd), C.byref(ld), capi.stream()), "fwd")
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_it)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3


print(f"real CSR                 {timed(starts):7.1f} us")
print(f"empty CSR (zero store)   {timed(torch.zeros_like(starts)):7.1f} us")
for L in (1, 4, 16, 64):
    # same voxels, lists truncated to L points: starts'[v] = cumulative min(len, L) (order is re-packed accordingly)
    l2 = torch.clamp(lens, max=L)
    st2 = torch.zeros_like(starts)
    st2[1:] = torch.cumsum(l2, 0).to(starts.dtype).to(dev)
    print(f"lists cut to <= {L:3d} points {timed(st2):7.1f} us   ({int(l2.sum())} points)")
