"""GroupNorm passes on the 189 MB cost-volume activation: fp32 at B = 1 against bf16 at B = 2 (the same bytes).  Run under
rocprofv3 --kernel-trace --stats for per-kernel durations.  usage: python tools/gn_dtype_probe.py [C G D H W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F

C, G, D, H, W = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (32, 2, 192, 48, 160)
for mode, B in (("fp32", 1), ("bf16", 2)):
    F.set_precision(mode)
    dt = torch.float32 if mode == "fp32" else torch.bfloat16
    x = torch.randn((B, C, D, H, W), device="cuda").to(dt).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    w = (torch.rand(C, device="cuda") + 0.5).requires_grad_(True)
    b = torch.randn(C, device="cuda").requires_grad_(True)
    go = torch.randn((B, C, D, H, W), device="cuda").to(dt).contiguous(memory_format=torch.channels_last_3d)
    for _ in range(12):
        y = F.group_norm(x, G, w, b, 1e-5, relu=os.environ.get("RELU", "1") != "0")
        y.backward(go)
        x.grad = w.grad = b.grad = None
    torch.cuda.synchronize()
F.set_precision("fp32")
