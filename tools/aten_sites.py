"""Which source lines of the package issue the large ATen ops of one fwd+bwd step (TorchDispatchMode log; backward ops
are attributed to the autograd node name).  Usage: python tools/aten_sites.py [min_numel]"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from stereoscene_amd import functional as F, model_zoo, synthetic as S
F.set_precision(os.environ.get("PREC", "fp32"))

MIN = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
cfg = S.CONFIGS["kitti_d192"]
model = model_zoo.build_detector(cfg).train()
smp = S.synthetic_sample(cfg, B=int(os.environ.get("BATCH", "1")), tag="bench0")
inputs = model_zoo.img_inputs_from_sample(smp)
gt = smp["gt_occ"].cuda()
log = collections.Counter()
nbytes = collections.Counter()


def numel(o):
    if isinstance(o, torch.Tensor):
        return o.numel()
    if isinstance(o, (list, tuple)):
        return sum(numel(a) for a in o)
    return 0


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        n = max(numel(out), numel(args))
        name = str(func)
        if n >= MIN and not any(k in name for k in ("view", "empty", "as_strided", "detach", "alias", "reshape", "permute",
                                                     "transpose", "slice.Tensor", "select.int", "unsqueeze", "squeeze", "expand", "aten.t.default",
                                                     "record_stream", "_unsafe_view", "split", "unbind", "chunk", "narrow", "is_", "lift_fresh", "_local_scalar")):
            site = "(autograd engine)"
            frames = [fr for fr in reversed(traceback.extract_stack())
                      if "stereoscene_amd" in fr.filename and "tools" not in fr.filename]
            if frames:
                depth = int(os.environ.get("ATEN_DEPTH", "1"))
                site = " < ".join(f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}" for fr in frames[:depth])
            shp = tuple(a.shape for a in args if isinstance(a, torch.Tensor))[:2]
            log[(name, site, str(shp))] += 1
            nbytes[(name, site, str(shp))] += 4 * n
        return out


def step():
    model.zero_grad(set_to_none=True)
    losses = model.forward_train(img_inputs=inputs, gt_occ=gt)
    sum(v for k, v in losses.items() if k.startswith("loss")).backward()


step()
with Log():
    step()
torch.cuda.synchronize()
rows = sorted(log.items(), key=lambda kv: -nbytes[kv[0]])
print(f"{sum(log.values())} large ATen calls")
for (name, site, shp), c in rows[:int(os.environ.get("ATEN_ROWS", "120"))]:
    print(f"{nbytes[(name, site, shp)] / 1e6:9.1f} MB {c:4d}x  {name:34s} {site:48s} | {shp[:100]}")
