"""Fused lift + splat gather on the KITTI frustum (kitti_d192, C = 128): role-split kernels over the compacted long-voxel list
(ssbev_lift_splat_fwd2, default) against the one-kernel pool_gather5 (SSBEV_GATHER_SPLIT=0).  Run under rocprofv3 --kernel-trace
--stats for the kernel times; prints event-pair medians of the whole call and checks that both give the same bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F, model_zoo, synthetic as S

cfg = S.CONFIGS["kitti_d192"]
vt = model_zoo.build_detector(cfg).eval().img_view_transformer
smp = S.synthetic_sample(cfg, B=1, tag="bench0")
geom = vt.get_geometry(*[t.cuda() for t in smp["geo_l"]])
depth = torch.softmax(torch.randn(1, vt.D, 48, 160, device="cuda"), 1).contiguous()
feat = torch.randn(1, 128, 48, 160, device="cuda").contiguous(memory_format=torch.channels_last)
tables = F.lift_splat_tables(geom, vt.bx, vt.dx, vt.nx)
print("long voxels:", int(tables[2].long_list[0]))
outs = {}
for split in (True, False, True, False):
    F.GATHER_SPLIT = split
    with torch.no_grad():
        y = F.lift_splat(depth, feat, None, vt.bx, vt.dx, vt.nx, tables=tables)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); y = F.lift_splat(depth, feat, None, vt.bx, vt.dx, vt.nx, tables=tables); b.record()
        torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    outs[split] = y
    print(f"split={split}: whole call median {ts[10] * 1e3:.1f} us, min {ts[0] * 1e3:.1f} us")
print("bit-identical:", torch.equal(outs[True], outs[False]))
