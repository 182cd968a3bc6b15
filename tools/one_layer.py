"""Run one conv layer (fwd, dgrad, wgrad) a few times: target for rocprofv3 --pmc runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F
ci, co, D, H, W = [int(v) for v in sys.argv[1:6]]
k = int(sys.argv[6]) if len(sys.argv) > 6 else 3
x = torch.randn(1, ci, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
w = (torch.randn(co, ci, k, k, k, device="cuda") * 0.05).requires_grad_(True)
for _ in range(3):
    y = F.conv3d(x, w, None, 1, k // 2)
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
