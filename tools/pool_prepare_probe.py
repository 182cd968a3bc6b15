"""ssbev_pool_prepare (CSR build of the frustum -> voxel table) timed on the KITTI frustum: whole call (event pair around the five
launches) at D = 112 / 192, B = 1 / 2, and the bucket-size distribution the level-2 waves see."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F, model_zoo, synthetic as S

CASES = (("kitti_d192", 1), ("kitti_d112", 1), ("kitti_d192", 2))
for name, B in (CASES[:1] if len(sys.argv) > 1 and sys.argv[1] == "one" else CASES):
    cfg = S.CONFIGS[name]
    vt = model_zoo.build_detector(cfg).eval().img_view_transformer
    smp = S.synthetic_sample(cfg, B=B, tag="bench0")
    geom = vt.get_geometry(*[t.cuda() for t in smp["geo_l"]])
    n = [int(v) for v in vt.nx.tolist()]
    vox = F.voxel_index(geom, vt.bx, vt.dx, vt.nx)
    starts, order = F.pool_prepare(vox, B, *n)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); F.pool_prepare(vox, B, *n); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    nvalid = int(starts[-1])
    nv = B * n[0] * n[1] * n[2]
    bits = max(1, (nv - 1).bit_length())
    hi = min(11, max(bits - 9, (bits + 1) // 2)); lo = bits - hi
    bsz = (starts[:-1].view(-1, 1 << lo)[:, 0]).cpu()
    bsz = torch.diff(torch.cat([bsz, starts[-1:].cpu()]))
    print(f"{name} B={B}: {vox.numel()} points ({nvalid} kept), {nv} voxels: median {ts[len(ts) // 2] * 1e3:6.1f} us, min {ts[0] * 1e3:6.1f} us;"
          f" level-2 buckets: {1 << hi} x {1 << lo} voxels, points per bucket max {int(bsz.max())} mean {float(bsz.float().mean()):.0f}")
