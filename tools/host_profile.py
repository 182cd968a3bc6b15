"""cProfile of the host side of fwd+bwd steps (main thread; the autograd thread's Python frames are sampled separately with
sys.setprofile off) -- where the ~30 ms of enqueue time per step go.  usage: python tools/host_profile.py"""
import os, sys, cProfile, pstats, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F, model_zoo, synthetic as S
cfg = S.CONFIGS["kitti_d192"]
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
prof = cProfile.Profile()
prof.enable()
with torch.autograd.set_multithreading_enabled(False):      # backward on this thread, so that its Python frames are profiled too
    for _ in range(5):
        step()
prof.disable()
torch.cuda.synchronize()
st = pstats.Stats(prof)
st.sort_stats(os.environ.get("SORT", "tottime")).print_stats(os.environ.get("FILTER", ""), int(os.environ.get("ROWS", "40")))
