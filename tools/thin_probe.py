import sys, os, torch
sys.path.insert(0, os.getcwd())
from stereoscene_amd import functional as F
dev = "cuda"
def run(ci, co):
    x = torch.randn(1, ci, 192, 48, 160, device=dev, requires_grad=True)
    w = torch.randn(co, ci, 3, 3, 3, device=dev, requires_grad=True)
    y = F.conv3d(x, w, None, 1, 1)
    g = torch.randn_like(y)
    y.backward(g)
for _ in range(3):
    run(32, 1); run(2, 32)
torch.cuda.synchronize()
