"""How close is the step to being host-bound?  Wall time to ENQUEUE one fwd+bwd step (Python + autograd + launch calls, no
synchronisation inside) against the wall time of the step including the final synchronisation.  usage: python tools/host_time_probe.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F, model_zoo, synthetic as S
F.set_precision(os.environ.get("PREC", "fp32"))
cfg = S.CONFIGS["kitti_d192"]
model = model_zoo.build_detector(cfg).train()
smp = S.synthetic_sample(cfg, B=int(os.environ.get("BATCH", "1")), tag="bench0")
inputs = model_zoo.img_inputs_from_sample(smp)
gt = smp["gt_occ"].cuda()


def step():
    model.zero_grad(set_to_none=True)
    losses = model.forward_train(img_inputs=inputs, gt_occ=gt)
    sum(v for k, v in losses.items() if k.startswith("loss")).backward()


for _ in range(4):
    step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3)
    tot.append((t2 - t0) * 1e3)
enq.sort(); tot.sort()
print(f"enqueue (host) median {enq[5]:.1f} ms  min {enq[0]:.1f}   step incl. sync median {tot[5]:.1f} ms  min {tot[0]:.1f}")
# back-to-back steps (the bench's regime): the host runs ahead of the device by up to a step
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"10 back-to-back steps: host done after {(t1 - t0) * 100:.1f} ms/step, device after {(t2 - t0) * 100:.1f} ms/step")
