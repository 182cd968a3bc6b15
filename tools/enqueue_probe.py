"""How long the host needs to ENQUEUE one fwd+bwd step (no waiting) against how long the GPU needs to run it: if the first is
well below the second the step is GPU-bound and host-side launch overhead is hidden."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import model_zoo, synthetic as S
cfg = S.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "kitti_d192"]
model = model_zoo.build_detector(cfg).train()
smp = S.synthetic_sample(cfg, B=1, tag="bench0")
inputs = model_zoo.img_inputs_from_sample(smp)
gt = smp["gt_occ"].cuda()
def step():
    model.zero_grad(set_to_none=True)
    losses = model.forward_train(img_inputs=inputs, gt_occ=gt)
    sum(v for k, v in losses.items() if k.startswith("loss")).backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
t = []
t0 = time.perf_counter()
for _ in range(4):
    a = time.perf_counter(); step(); t.append(time.perf_counter() - a)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("host enqueue time per step (ms):", [round(x * 1e3, 1) for x in t], " wall per step incl. GPU:", round(tot / 4 * 1e3, 1))
