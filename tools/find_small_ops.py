"""Which Python lines issue the small ATen kernels (fill / add / copy) of a step?  TorchDispatchMode + stack walk."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from stereoscene_amd import model_zoo, synthetic as S
cfg = S.CONFIGS["kitti_d192"]
model = model_zoo.build_detector(cfg).train()
smp = S.synthetic_sample(cfg, B=1, tag="bench0")
inputs = model_zoo.img_inputs_from_sample(smp); gt = smp["gt_occ"].cuda()
def step():
    model.zero_grad(set_to_none=True)
    losses = model.forward_train(img_inputs=inputs, gt_occ=gt)
    sum(v for k, v in losses.items() if k.startswith("loss")).backward()
step(); step()
want = set(sys.argv[1:] or ["fill_", "zero_", "zeros", "zeros_like", "add", "add_", "copy_", "mul", "clone", "full", "sum"])
cnt = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in want:
            fr = [f for f in traceback.extract_stack() if "stereoscene_amd" in f.filename and "find_small" not in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line[:70]}" if fr else "(autograd engine)"
            if name == "clone" and fr:
                mods = [f for f in fr if "plugin" in f.filename or "layers.py" in f.filename]
                where += "  <=  " + " <= ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in mods[-3:])
                where += f" shape={tuple(args[0].shape)}"
            numel = next((a.numel() for a in args if isinstance(a, torch.Tensor)), 0)
            cnt[(name, where, "big" if numel > 1 << 18 else "small")] += 1
        return func(*args, **(kwargs or {}))
with Log():
    step()
for (n, f, sz), c in cnt.most_common(50):
    print(f"{c:5d}  {n:10s} {sz:5s} {f}")
