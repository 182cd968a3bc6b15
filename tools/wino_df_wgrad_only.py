"""Runs fwd+bwd of the wide layers on the depth-fused path a few times (rocprofv3 --kernel-trace --stats target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F
which = sys.argv[1] if len(sys.argv) > 1 else "128"
LAYERS = {"128": (128, 128, 16, 128, 128), "384": (384, 192, 16, 128, 128), "256": (256, 256, 8, 64, 64), "512": (512, 512, 4, 32, 32)}
cin, cout, D, H, W = LAYERS[which]
x = torch.randn(1, cin, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
w = (torch.randn(cout, cin, 3, 3, 3, device="cuda") * (1.0 / (27 * cin)) ** 0.5).requires_grad_(True)
for _ in range(6):
    y = F.conv3d(x, w, None, 1, 1)
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
