"""When does each gradient bucket become ready during backward?  (evidence, on ONE GPU, of how much of the
data-parallel exchange can run under backward.)  One event at the start of backward, one per bucket when its last gradient
has arrived (dp.FlatGradAllReduce(timeline=True)), one at the end; prints per bucket: bytes, ready time, and the time an
xGMI exchange of that bucket would take at a given bus bandwidth -- a bucket whose exchange ends before backward does is
fully hidden.  Usage: python tools/bucket_timeline.py [bucket_mb] [bus_GBps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import model_zoo, synthetic as S
from stereoscene_amd.dp import FlatGradAllReduce

bucket_mb = float(sys.argv[1]) if len(sys.argv) > 1 else 64
bus = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0        # GB/s of bus bandwidth assumed for an 8-GPU ring-equivalent
cfg = S.CONFIGS["kitti_d192"]
model = model_zoo.build_detector(cfg).train()
red = FlatGradAllReduce(model, bucket_mb=bucket_mb, timeline=True)
smp = S.synthetic_sample(cfg, B=1, tag="bench0")
inputs = model_zoo.img_inputs_from_sample(smp)
gt = smp["gt_occ"].cuda()
rows = None
for it in range(4):
    red.zero_grad()
    losses = model.forward_train(img_inputs=inputs, gt_occ=gt)
    total = sum(v for k, v in losses.items() if k.startswith("loss"))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    total.backward()
    e1.record()
    tl = list(red.timeline)
    red.finish()
    torch.cuda.synchronize()
    rows = [(b, e0.elapsed_time(ev)) for b, ev in tl]
    bwd = e0.elapsed_time(e1)
print(f"backward {bwd:.1f} ms, {len(red.buckets)} buckets of <= {bucket_mb:g} MB, flat gradient buffer {red.flat.numel() * 4 / 1e6:.1f} MB; "
      f"assumed bus bandwidth {bus:g} GB/s (2 (N-1)/N x bytes / time)")
print("bucket   MB    ready at   exchange   done at    hidden under backward")
link_free, hidden_bytes, total_bytes = 0.0, 0.0, 0.0
for b, t in rows:
    s, e, n = red.buckets[b]
    mb = (e - s) * 4 / 1e6
    ex = 2 * 7 / 8 * mb / 1e3 / bus * 1e3           # ms for an 8-rank exchange of this bucket
    start = max(t, link_free)
    link_free = start + ex
    frac = min(1.0, max(0.0, (bwd - start) / ex)) if ex > 0 else 1.0
    hidden_bytes += frac * mb
    total_bytes += mb
    print(f"{b:4d} {mb:7.1f} {t:9.1f} ms {ex:7.2f} ms {link_free:8.1f} ms   {100 * frac:5.1f} %   ({n} tensors)")
print(f"{100 * hidden_bytes / total_bytes:.1f} % of the {total_bytes:.1f} MB exchange overlaps backward; exposed tail {max(0.0, link_free - bwd):.2f} ms")
