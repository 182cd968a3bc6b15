"""GPU probe: time the MFMA conv kernels on the hot-path layer shapes (and torch/MIOpen beside them)."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereoscene_amd import functional as F

dev = "cuda"
LAYERS = [
    # name, Cin, Cout, (D,H,W), k, s, p, transposed, outpad
    ("cost 32->32 k3 @192x48x160", 32, 32, (192, 48, 160), 3, 1, 1, False, 0),
    ("cost 32->64 k3s2", 32, 64, (192, 48, 160), 3, 2, 1, False, 0),
    ("cost 64->64 k3 @96x24x80", 64, 64, (96, 24, 80), 3, 1, 1, False, 0),
    ("cost 64->128 k3s2", 64, 128, (96, 24, 80), 3, 2, 1, False, 0),
    ("cost 128->128 k3 @48x12x40", 128, 128, (48, 12, 40), 3, 1, 1, False, 0),
    ("cost deconv 128->64", 128, 64, (48, 12, 40), 3, 2, 1, True, 1),
    ("cost deconv 64->32", 64, 32, (96, 24, 80), 3, 2, 1, True, 1),
    ("vox 128->128 k3 @128x128x16", 128, 128, (128, 128, 16), 3, 1, 1, False, 0),
    ("vox 128->256 k3s2", 128, 256, (128, 128, 16), 3, 2, 1, False, 0),
    ("vox 256->512 k3s2", 256, 512, (64, 64, 8), 3, 2, 1, False, 0),
    ("vox 256->256 k3 @64x64x8", 256, 256, (64, 64, 8), 3, 1, 1, False, 0),
    ("vox 512->512 k3 @32x32x4", 512, 512, (32, 32, 4), 3, 1, 1, False, 0),
    ("head 384->192 k3", 384, 192, (128, 128, 16), 3, 1, 1, False, 0),
    ("fpn deconv 512->128 k4s4", 512, 128, (32, 32, 4), 4, 4, 0, True, 0),
]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    use_torch = "--torch" in sys.argv
    rows = []
    for name, ci, co, (D, H, W), k, s, p, tr, op in LAYERS:
        x = torch.randn(1, ci, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        wshape = (ci, co, k, k, k) if tr else (co, ci, k, k, k)
        w = (torch.randn(wshape, device=dev) * 0.05).requires_grad_(True)
        f = (lambda: F.conv_transpose3d(x, w, None, s, p, op)) if tr else (lambda: F.conv3d(x, w, None, s, p))
        y = f()
        vox_out = y.numel() // co
        flops = 2.0 * vox_out * co * ci * k ** 3 if not tr else 2.0 * (x.numel() // ci) * ci * co * k ** 3
        t_f = timeit(f)
        go = torch.randn_like(y)

        def fb():
            x.grad = None; w.grad = None
            f().backward(go)
        t_fb = timeit(fb, 3)
        xd = x.detach()
        fw = (lambda: F.conv_transpose3d(xd, w, None, s, p, op)) if tr else (lambda: F.conv3d(xd, w, None, s, p))

        def wg():
            w.grad = None
            fw().backward(go)
        t_w = timeit(wg, 3) - t_f
        row = dict(layer=name, gflop=flops / 1e9, fwd_ms=t_f * 1e3, fwd_tflops=flops / t_f / 1e12,
                   fwdbwd_ms=t_fb * 1e3, fwdbwd_tflops=3 * flops / t_fb / 1e12, wgrad_ms=t_w * 1e3,
                   wgrad_tflops=flops / t_w / 1e12)
        if use_torch:
            import torch.nn.functional as TF
            ft = (lambda: TF.conv_transpose3d(x, w, None, s, p, op)) if tr else (lambda: TF.conv3d(x, w, None, s, p))
            t0 = time.perf_counter(); ft(); torch.cuda.synchronize(); row["torch_first_call_s"] = time.perf_counter() - t0
            row["torch_fwd_ms"] = timeit(ft) * 1e3
        rows.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/probe_conv.json", "w"), indent=1)


if __name__ == "__main__":
    main()
