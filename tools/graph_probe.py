"""Whole-step HIP graph: capture one fwd+bwd step (two streams, custom kernels through the C ABI) with torch.cuda.graph and replay it.
usage: python tools/graph_probe.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F, model_zoo, synthetic as S
F.set_precision(os.environ.get("PREC", "fp32"))
cfg = S.CONFIGS[os.environ.get("CFG", "kitti_d192")]
model = model_zoo.build_detector(cfg).train()
smp = S.synthetic_sample(cfg, B=int(os.environ.get("BATCH", "1")), tag="bench0")
inputs = model_zoo.img_inputs_from_sample(smp)
gt = smp["gt_occ"].cuda()
params = [p for p in model.parameters() if p.requires_grad]


def step():
    for p in params:
        p.grad = None
    losses = model.forward_train(img_inputs=inputs, gt_occ=gt)
    total = sum(v for k, v in losses.items() if k.startswith("loss"))
    total.backward()
    return total


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        tot = step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
ref_loss = float(tot)
ref_grads = [p.grad.detach().clone() for p in params]
print("eager loss", ref_loss, flush=True)

g = torch.cuda.CUDAGraph()
for p in params:
    p.grad = None
try:
    with torch.cuda.graph(g):
        static_total = step()
except Exception as e:
    print("CAPTURE FAILED:", type(e).__name__, str(e)[:600])
    sys.exit(1)
torch.cuda.synchronize()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("graph loss", float(static_total))
bad = 0
for p, r in zip(params, ref_grads):
    if p.grad is None or not torch.equal(p.grad, r):
        bad += 1
print("gradients differing from eager (bitwise):", bad, "of", len(params))
worst = max(((p.grad - r).norm() / (r.norm() + 1e-30)).item() for p, r in zip(params, ref_grads) if p.grad is not None)
print("worst relative L2 difference", worst)
for name, fn in (("graph replay", g.replay), ("eager", step)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) * 100:.2f} ms/step")
