"""DepthNet's deformable conv (640 channels, 4 groups, 48x160) forward + backward in isolation (timing / PMC target)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F, synthetic as S
from stereoscene_amd.layers import DeformConv2dPack
layer = DeformConv2dPack(640, 640, 3, 1, 1, 1, groups=4).cuda()
with torch.no_grad():
    layer.conv_offset.weight.copy_(S.hash_uniform("dcnp/ow", tuple(layer.conv_offset.weight.shape), -0.5, 0.5) * 0.01)
    layer.conv_offset.bias.copy_(S.hash_uniform("dcnp/ob", tuple(layer.conv_offset.bias.shape), -0.1, 0.1))
x = torch.randn(1, 640, 48, 160, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
def step():
    y = layer(x)
    y.square().mean().backward()
for _ in range(3): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("dcn fwd+bwd %.3f ms" % ((time.perf_counter() - t) * 100))
