"""GPU A/B of conv_tapdh_kernel / wgrad_tapdh_kernel between two builds of the library: seeded inputs, median time of 20 launches (forward and
data gradient of the 32 -> 32 cost-volume layer, 192 x 48 x 160, plus an odd-sized ragged case) and a CRC of the output bytes --
a schedule change of the kernel must leave every bit of the result alone.
    python tools/tapdh_ab.py [path/to/libssbev_hip.so]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import capi
if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
from stereoscene_amd import functional as F


def crc(t):
    return zlib.crc32(t.detach().contiguous().cpu().numpy().tobytes())


def run(B, C, D, H, W, relu, tag):
    torch.manual_seed(1234)
    x = torch.randn(B, C, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(C, C, 3, 3, 3, device="cuda") * 0.03
    bias = torch.randn(C, device="cuda")
    go = torch.randn(B, C, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    with torch.no_grad():
        y = F.conv3d(x, w, bias, 1, 1)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); y = F.conv3d(x, w, bias, 1, 1); b.record()
        torch.cuda.synchronize()
        dt = sorted(a.elapsed_time(b) for a, b in ev)[10]
    xq = x.clone().requires_grad_(True)
    wq = w.clone().requires_grad_(True)
    yq = F.conv3d(xq, wq, bias, 1, 1)
    if relu:
        yq = torch.relu(yq)
    yq.backward(go)
    torch.cuda.synchronize()
    yw = F.conv3d(x, wq, None, 1, 1)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); torch.autograd.grad(yw, wq, go, retain_graph=True); b.record()
    torch.cuda.synchronize()
    dtw = sorted(a.elapsed_time(b) for a, b in ev)[10]
    print(f"{tag}: fwd {dt * 1e3:.1f} us  wgrad {dtw * 1e3:.1f} us  crc(y) {crc(y):08x}  crc(gx) {crc(xq.grad):08x}  crc(gw) {crc(wq.grad):08x}",
          flush=True)


def run_fork(C, D, H, W, tag):
    """two 3x3x3 consumers of one activation: the second data gradient ACCUMULATES into the first one's buffer (gradient slot)"""
    torch.manual_seed(99)
    x = torch.randn(1, C, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    w1 = torch.randn(C, C, 3, 3, 3, device="cuda") * 0.03
    w2 = torch.randn(C, C, 3, 3, 3, device="cuda") * 0.03
    xa, xb = F.fork(x * 1.0, 2)
    y = F.conv3d(xa, w1, None, 1, 1) + 2.0 * F.conv3d(xb, w2, None, 1, 1)
    y.backward(torch.randn_like(y))
    torch.cuda.synchronize()
    print(f"{tag}: crc(gx) {crc(x.grad):08x}", flush=True)


print("library:", capi.LIB_PATH)
run(1, 32, 192, 48, 160, False, "32->32 192x48x160")
run(1, 32, 192, 48, 160, False, "32->32 192x48x160 (again)")
run(2, 32, 24, 12, 40, True, "B=2 32->32 24x12x40")
run(1, 24, 10, 6, 37, False, "24->24 10x6x37 (ragged W, C < 32)")
run(1, 32, 6, 2, 33, False, "32->32 6x2x33 (one block per plane)")
run_fork(32, 24, 12, 40, "fork 32 24x12x40 (accumulate)")
run_fork(24, 10, 6, 37, "fork 24 10x6x37 (accumulate, ragged)")
run_fork(30, 10, 6, 37, "fork 30 10x6x37 (accumulate, N % 4 != 0)")
