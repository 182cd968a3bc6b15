"""Which ATen ops (not ssbev kernels) cost device time in one fwd+bwd step: torch.profiler, grouped by op + input shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from stereoscene_amd import model_zoo, synthetic as S
cfg = S.CONFIGS["kitti_d192"]
model = model_zoo.build_detector(cfg).train()
smp = S.synthetic_sample(cfg, B=1, tag="bench0")
inputs = model_zoo.img_inputs_from_sample(smp)
gt = smp["gt_occ"].cuda()
def step():
    model.zero_grad(set_to_none=True)
    losses = model.forward_train(img_inputs=inputs, gt_occ=gt)
    sum(v for k, v in losses.items() if k.startswith("loss")).backward()
step(); step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
def site(e):
    for fr in (e.stack or []):
        if "stereoscene_amd" in fr and "torch/" not in fr:
            return fr.split("stereoscene_amd/")[-1][:70]
    return ""
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    dt = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if dt > 20 and e.key.startswith("aten::"):
        rows.append((dt, e.count, e.key, site(e) + ' ' + str(e.input_shapes)[:90]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten self device time: {tot / 1e3:.2f} ms")
for dt, n, k, shp in rows[:110]:
    print(f"{dt / 1e3:7.3f} ms {n:4d}x  {k:28s} {shp}")
