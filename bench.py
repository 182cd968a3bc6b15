#!/usr/bin/env python
"""Benchmark of the StereoScene hot path on MI355X: output voxels / second, forward + backward.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Either the driver launches this file under ``torch.distributed.run``
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or -- when RANK is unset -- ``--gpus N`` re-executes
itself under ``torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`` (the reference's launcher:
tools/dist_train.sh:9-19 spawns ``--nproc_per_node=$GPUS``).

One "step" = forward of a1-a16 (stereo cost volume -> MIE -> lift/splat -> 3-D encoder/neck/head ->
4 losses) + backward, on one seeded synthetic SemanticKITTI-shaped batch per GPU, inputs resident
in HBM.  N > 1 shards the batch (B per GPU fixed = weak scaling) and adds the gradient exchange
over RCCL/xGMI (flat buckets overlapped with backward, stereoscene_amd/dp.py).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import platform
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The step runs on two streams (streams.py); a RCCL
# process group adds its own, and with four queues the step's two streams end up sharing one: measured on MI355X, the mere
# presence of a one-rank process group cost +4.5 ms per step (76.5 vs 72.4 ms; serial schedule unaffected) and 8 queues give
# it back (72.45 ms, profiles/r5_rccl_hw_queues.txt).  Must be in the environment before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 (no sparsity)
PEAK_FP32_MFMA_TFLOPS = 157.3       # "Peak FP32 (matrix)"
PEAK_HBM_TBS = 8.0                  # HBM3E
XGMI_LINKS, XGMI_LINK_GBS = 7, 153.0
VOXELS_PER_SAMPLE = 256 * 256 * 32

# CPU baseline thread count: profiles/r2_cpu_thread_sweep.txt (oracle step time vs torch threads on the GPU box's host)
CPU_BASELINE_THREADS = 16

# Families whose launches are ONE kernel each: candidates for the `roofline` object (the dominant single kernel of
# the step by summed duration).  value = (kernel name, bound)
EXECUTED_NOTE = {
    "conv_tap_h": "in-kernel Winograd F(2,3) along h: 2/3 of the direct convolution's",
    "conv_tap_dh": "in-kernel Winograd F(2,3) along d and h: 4/9 of the direct convolution's",
    "conv_wino_fused": "F(4,3)^2 over (h, w) in memory and F(2,3) along d in registers: 1/6 of the direct convolution's; "
                       "bytes = the kernel's own operands P / Mo (2.25x the activations) and packed weights",
    "conv_tap16": "direct convolution on v_mfma_f32_32x32x16_bf16, bf16 tensors: all of the direct convolution's",
    "conv_wide16": "direct convolution on v_mfma_f32_32x32x16_bf16 (LDS-ring implicit GEMM, all template instances): all of the "
                   "direct convolution's",
    "wgrad16": "direct weight gradient on v_mfma_f32_32x32x16_bf16 (all template instances of wgrad16_kernel): all of the operator's",
    "wgrad_ring16": "direct weight gradient on v_mfma_f32_32x32x16_bf16, operands by transposing LDS reads out of a shared ring "
                    "(both template instances): all of the operator's",
}
SINGLE_KERNEL_FAMILIES = {
    "conv_tap_h": ("conv_taph_kernel", "mfma"),
    "conv_tap_dh": ("conv_tapdh_kernel", "mfma"),
    "conv_wino_fused": ("wino_df_kernel", "mfma"),
    # bf16 storage mode (--precision bf16, BASELINE configs[3]): priced against the dense bf16 MFMA peak
    "conv_tap16": ("conv_tap16_kernel", "mfma"),
    "conv_wide16": ("conv_wide16_kernel", "mfma"),
    "wgrad16": ("wgrad16_kernel", "mfma"),
    "wgrad_ring16": ("wgrad_ring16_kernel", "mfma"),
}

# ---- the stdout record --------------------------------------------------------------------------------------------------
# The driver captures ONE JSON line from stdout; r5's line had grown to 20 kB and was not parsed (BENCH_r05.json: parsed = null).
# stdout carries the compact record below (< 6000 characters, tests/test_bench_record.py); everything else -- the other timed
# kernels, the serial replay, the per-group floor table, the prose notes -- goes to `bench_detail.json` (SSBEV_BENCH_DETAIL
# overrides the path).
FLOP_CONVENTIONS = {
    "executed": "achieved/frac = multiply-adds the kernel EXECUTES (frac <= 1 = matrix-pipe utilisation); operator_* = "
                "direct-convolution FLOPs 2*voxels*Cin*Cout*27 (SURVEY 8(d))",
}
RECORD_LIMIT = 6000
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "operator_tflops", "operator_frac", "launches_per_step",
              "avg_launch_us", "executed_gflop_per_launch", "algorithmic_gflop_per_launch", "algorithmic_bytes_per_launch",
              "ms_per_step_in_kernel", "traffic", "traffic_source", "traffic_matches_tree")
_TIE_KEYS = ("kernel", "achieved", "frac", "operator_frac", "avg_launch_us", "launches_per_step", "ms_per_step_in_kernel", "traffic")


def _round(v, nd=4):
    """Floats to `nd` significant digits after the leading one (JSON stays readable and short); containers recursively."""
    if isinstance(v, float):
        return float(f"{v:.{nd + 2}g}")
    if isinstance(v, dict):
        return {k: _round(x, nd) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_round(x, nd) for x in v]
    return v


def compact_record(full):
    """The stdout line: the contract keys + `roofline` (dominant kernel, ties) + `step_roofline` (no groups) + `forward_only`
    + `gradient_exchange` (no prose) + `cpu_baseline`.  `full` is the complete record (written to bench_detail.json)."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data", "config") if k in full}
    roof = full.get("roofline")
    if roof:
        r = {k: roof[k] for k in _ROOF_KEYS if k in roof}
        r["flop_convention"] = "executed"
        r["timed"] = "hip_events_in_timed_region" + ("_two_streams" if full.get("streams", {}).get("side_stream_weight_gradients") else "")
        sel = roof.get("selection") or {}
        r["selection"] = {"rule": "largest_summed_event_time",
                          "ms_per_step_by_symbol": sel.get("ms_per_step_by_symbol", {}),
                          "tied_within_5_percent": [{k: t[k] for k in _TIE_KEYS if k in t}
                                                    for t in sel.get("tied_within_5_percent", [])]}
        rep = (full.get("roofline_serial_replay") or {})
        alone = {}
        for name in [roof["kernel"]] + [t["kernel"] for t in sel.get("tied_within_5_percent", [])]:
            if name in rep:
                alone[name] = {k: rep[name][k] for k in ("avg_launch_us", "achieved", "frac", "operator_frac") if k in rep[name]}
        if alone:
            r["alone_on_device"] = alone         # the same kernels in the serial replay (side streams off)
            r["serial_replay_ms_per_step"] = next(iter(rep.values())).get("replay_ms_per_step")
        out["roofline"] = r
    else:
        out["roofline"] = None
    sr = full.get("step_roofline")
    if sr:
        out["step_roofline"] = {k: sr[k] for k in ("floor_ms", "frac", "operator_floor_ms", "operator_frac") if k in sr}
    if full.get("streams"):
        out["streams"] = full["streams"]
    if full.get("losses"):
        out["losses"] = full["losses"]
    fo = full.get("forward_only")
    if fo:
        out["forward_only"] = {k: fo[k] for k in ("ms_per_step", "value", "unit") if k in fo}
    ex = full.get("gradient_exchange")
    if ex:
        out["gradient_exchange"] = {k: v for k, v in ex.items() if k != "note"}
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "workload", "step_seconds_min_median_max",
                                                  "spread", "cpu") if k in cb}
    else:
        out["cpu_baseline"] = cb
    out["detail"] = full.get("detail_file", "bench_detail.json")
    out = _round(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > RECORD_LIMIT:                 # never let the line outgrow the driver's capture again: shed optional keys
        for k in ("losses", "streams", "gradient_exchange", "forward_only"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) <= RECORD_LIMIT:
                break
    assert len(line) <= RECORD_LIMIT, f"stdout record is {len(line)} characters"
    return line


def emit_record(full, json_fd):
    """Full record -> bench_detail.json; compact record -> the saved stdout descriptor (ONE line)."""
    path = os.environ.get("SSBEV_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")
    full["detail_file"] = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
    except OSError as exc:
        full["detail_file"] = f"not written: {exc}"
    sys.stderr.write(f"bench detail: {full['detail_file']}\n")      # (not the record itself: stderr may share the driver's capture)
    sys.stderr.flush()
    os.write(json_fd, (compact_record(full) + "\n").encode())


def csrc_sha16():
    """Hash of the kernel sources (csrc/*.hip, *.h, include/ssbev.h): ties a PMC traffic file to the tree it was collected on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "stereoscene_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/ssbev.h"]:
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="kitti_d192", help="kitti_d192 (BASELINE metric) | kitti_d112 | small_d48")
    ap.add_argument("--batch", type=int, default=1, help="samples per GPU")
    ap.add_argument("--cpu-sample", default="auto", choices=["auto", "small", "full", "none"])
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--skip-forward-extra", action="store_true",
                    help="do not append the secondary forward-only measurement (profiling runs: keeps kernel totals per step clean)")
    ap.add_argument("--skip-serial-replay", action="store_true",
                    help="do not append the serial replay (side streams off) that measures the dominant kernel without co-scheduled "
                         "kernels (profiling runs: keeps the per-kernel averages of the timed schedule clean)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16_operands"],
                    help="fp32 = the BASELINE metric (default); bf16 = BASELINE configs[3] (never the headline)")
    ap.add_argument("--ablation", default="full", choices=["full", "bev_only", "stereo_only"],
                    help="BASELINE configs[4]: depth distribution from the MIE fusion | monocular DepthNet only | stereo volume only")
    ap.add_argument("--no-rccl-world1", action="store_true",
                    help="N = 1 without a launcher: do NOT create the one-rank RCCL process group (default: create it, so that the "
                         "gradient exchange calls of the N > 1 step execute at N = 1 too)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU / gloo dry run of the N-rank launcher, timing protocol and gradient exchange on a toy "
                         "parameter set (tests/test_bench_launcher.py); measures nothing about the HIP path")
    return ap.parse_args()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn_under_launcher(args):
    """--gpus N without a launcher: re-execute this file as N ranks of one node (returns the exit code)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    return subprocess.call(cmd, env=env)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()


def cpu_baseline(mode, cfg_full):
    """The CPU oracle (torch fp32 restatement of the reference, `kind: port`) timed on this box's host cores:
    SURVEY 8(d) protocol = 1 warm-up + 3 timed fwd+bwd steps, median.  Baseline only.
    'auto': the protocol on BASELINE configs[0] (64x64x16 grid, D=48); if the full-size protocol is predicted to fit
    ~2 minutes, it is run on the bench's own workload and reported instead."""
    if mode == "none":
        return None
    from oracle import path_ref as O
    from stereoscene_amd import model_zoo, synthetic as S
    ncores = min(CPU_BASELINE_THREADS, os.cpu_count() or 1)
    torch.set_num_threads(ncores)
    # pinned (the unpinned baseline moved by +-40 % between boxes and runs): the process -- and with it ATen's
    # OpenMP team -- is confined to `ncores` logical CPUs of ONE L3 / NUMA neighbourhood (the first ncores ids of the allowed set)
    pinned = None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        pinned = allowed[:ncores]
        os.sched_setaffinity(0, set(pinned))
    except (AttributeError, OSError):
        allowed = None

    def protocol(cfg, timed=5):
        m = model_zoo.build_detector(cfg, device="cpu")       # parameter container only; never run on CPU
        sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and v.dim() > 0 and "running" not in k
                  and not k.endswith(("frustum", ".dx", ".bx", ".nx")) else v) for k, v in m.state_dict().items()}
        smp = S.synthetic_sample(cfg, B=1, tag="bench0")
        oin = [smp["x_l"], *smp["geo_l"], O.get_mlp_input(*smp["geo_l"]), smp["x_r"], *smp["geo_r"],
               O.get_mlp_input(*smp["geo_r"]), smp["calib"]]
        D = int(round((cfg["dbound"][1] - cfg["dbound"][0]) / cfg["dbound"][2]))
        ocfg = dict(D=D, numC_Trans=128, warp_align_corners=True, downsample=cfg["downsample"], dbound=cfg["dbound"])
        times = []
        for it in range(1 + timed):
            for v in sd.values():
                if v.requires_grad:
                    v.grad = None
            t0 = time.perf_counter()
            losses, _ = O.forward_train(sd, oin, smp["gt_depths"], smp["gt_occ"], ocfg, train=True)
            sum(losses.values()).backward()
            if it:                                    # iteration 0 = warm-up
                times.append(time.perf_counter() - t0)
        vox = cfg["occ_size"][0] * cfg["occ_size"][1] * cfg["occ_size"][2]
        med = statistics.median(times)
        return vox / med, med, times

    try:
        v, med, times = protocol(S.CFG_S)
        sample = f"1 warm-up + 5 timed fwd+bwd steps of configs[0] (64x64x16 grid, D=48, B=1), median {med:.2f} s"
        cfg_name = "configs[0]"
        if mode == "full" or (mode == "auto" and med * 25 * 6 < 210):      # full size ~25x the small step
            v, med, times = protocol(cfg_full)
            sample = f"1 warm-up + 5 timed fwd+bwd steps of {cfg_full['name']} (256x256x32 grid, B=1), median {med:.1f} s"
            cfg_name = cfg_full["name"]
    finally:
        if allowed is not None:
            os.sched_setaffinity(0, set(allowed))
    return {"value": v, "unit": "voxels/s", "cores": ncores, "kind": "port", "sample": sample, "workload": cfg_name,
            "step_seconds": [round(t, 3) for t in times], "step_seconds_min_median_max": [round(min(times), 3), round(med, 3), round(max(times), 3)],
            "spread": round((max(times) - min(times)) / med, 3), "pinned_cpus": pinned,
            "cpu": _cpu_model(), "host_threads_available": os.cpu_count(), "threads_source": "profiles/r2_cpu_thread_sweep.txt"}


def exchange_microbench(reducer, dist, iters=5):
    """Bus bandwidth of the gradient exchange alone (all buckets back to back, nothing to overlap with), for both
    realisations: bus GB/s = 2 (N-1)/N x bytes / time, against the 7 x 153 GB/s of xGMI out of one GPU."""
    world = reducer.world
    out = {}
    scratch = torch.zeros_like(reducer.flat)
    for mode in ("rs_ag", "all_reduce"):
        def run():
            for s, e, _ in reducer.buckets:
                buf = scratch[s:e]
                if mode == "rs_ag":
                    n = (e - s) // world
                    mine = buf[reducer.rank * n:(reducer.rank + 1) * n]
                    dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.AVG)
                    dist.all_gather_into_tensor(buf, mine)
                else:
                    dist.all_reduce(buf, op=dist.ReduceOp.AVG)
        run()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run()
        torch.cuda.synchronize()
        dt = torch.tensor([(time.perf_counter() - t0) / iters], device="cuda")
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        nbytes = scratch.numel() * 4
        bus = 2.0 * (world - 1) / world * nbytes / float(dt) / 1e9
        out[mode] = {"ms": float(dt) * 1e3, "bus_GBps": bus, "frac_of_xgmi_peak": bus / (XGMI_LINKS * XGMI_LINK_GBS)}
    return out


def selftest_launcher(args):
    """CPU / gloo dry run of everything around the HIP path that N > 1 adds: rendezvous, the flat-bucket exchange
    overlapped with backward, barrier + max-over-ranks timing, the one JSON line."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from stereoscene_amd.dp import FlatGradAllReduce
    torch.manual_seed(0)
    toy = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 64))
    red = FlatGradAllReduce(toy, bucket_mb=0.02)
    g = torch.Generator().manual_seed(1 + rank)
    x = torch.randn(32, 64, generator=g)

    def step():
        red.zero_grad()
        toy(x).square().mean().backward()
        return red.finish()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nbytes = step()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0])
    gsum = torch.tensor([float(red.flat.double().sum())])
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        ref = gsum.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(ref, gsum), "ranks disagree on the exchanged gradient"
    if rank == 0:
        print(json.dumps({"metric": "selftest-launcher (CPU, gloo): no HIP path measured", "value": 0.0, "unit": "voxels/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(dt) / args.steps * 1e3,
                          "exchange": {"mode": red.exchange, "buckets": len(red.buckets), "bytes_per_step": nbytes}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(respawn_under_launcher(args))
    if args.selftest_launcher:
        return selftest_launcher(args)
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a GPU is required"
    assert local < torch.cuda.device_count(), f"rank {rank}: local rank {local} but {torch.cuda.device_count()} GPUs visible"
    torch.cuda.set_device(local)
    import torch.distributed as dist
    # stdout carries the ONE JSON record and nothing else: RCCL prints a version banner through C stdio when a process group
    # comes up (block-buffered on a pipe, i.e. flushed at exit, AFTER the record).  File descriptor 1 is pointed at stderr for the
    # life of the process; the record is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    distributed = "RANK" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run
    rccl_world1 = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,       # RCCL on ROCm
                                device_id=torch.device("cuda", local))
    elif world == 1 and not args.forward_only and not args.no_rccl_world1:
        # N = 1 as the driver runs it (no launcher): a one-rank RCCL process group, so that the step measured here is the step
        # of the N > 1 runs call for call -- bucket pack, reduce_scatter_tensor(AVG) + all_gather_into_tensor per bucket from
        # the autograd hooks, wait in finish() -- and the barrier of the timing protocol is RCCL's
        try:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ["MASTER_PORT"] = str(_free_port())
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
            distributed, rccl_world1 = True, "ok"
        except Exception as exc:             # an RCCL that cannot initialise must not cost the measurement: say so in the line
            rccl_world1 = f"failed: {type(exc).__name__}: {exc}"

    from stereoscene_amd import functional as F, model_zoo, synthetic as S
    from stereoscene_amd.dp import FlatGradAllReduce
    cfg = S.CONFIGS[args.config]
    F.set_precision(args.precision)
    torch.manual_seed(rank)
    model = model_zoo.build_detector(cfg)          # deterministic fill-by-key weights, gamma = alpha = 0.5
    model.img_view_transformer.ablation = args.ablation
    model.train()
    reducer = FlatGradAllReduce(model, bucket_mb=64) if not args.forward_only else None
    smp = S.synthetic_sample(cfg, B=args.batch, tag=f"bench{rank}")
    inputs = model_zoo.img_inputs_from_sample(smp)
    gt_occ = smp["gt_occ"].cuda()

    def step():
        if args.forward_only:
            with torch.no_grad():
                return model.forward_train(img_inputs=inputs, gt_occ=gt_occ)
        reducer.zero_grad()
        losses = model.forward_train(img_inputs=inputs, gt_occ=gt_occ)
        sum(v for k, v in losses.items() if k.startswith("loss")).backward()
        reducer.finish()
        return losses

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # HIP events (on the launch stream) around every launch of the candidate dominant kernels only: two event records
    # per launch cost ~10 us of queue time.  Every other instrumented operator is COUNTED (flops / bytes), not timed.
    fams = set(SINGLE_KERNEL_FAMILIES)
    if os.environ.get("SSBEV_TIME_FAMILIES"):
        fams |= set(os.environ["SSBEV_TIME_FAMILIES"].split(","))
    timer = F.KernelTimer(families=fams)
    F.KERNEL_TIMER = None if os.environ.get('SSBEV_NO_TIMER') else timer
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step()
    fence()
    dt = time.perf_counter() - t0
    F.KERNEL_TIMER = None
    tmax = torch.tensor([dt], device="cuda")
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    ms = dt / args.steps * 1e3
    scale = VOXELS_PER_SAMPLE if cfg["occ_size"] == (256, 256, 32) else cfg["occ_size"][0] * cfg["occ_size"][1] * cfg["occ_size"][2]
    value = world * args.batch * scale / (dt / args.steps)

    # secondary figure (north_star states its >= 10x-over-CPU target for the FORWARD pass): same model and inputs, no_grad
    fo_ms = None
    if not args.forward_only and not args.skip_forward_extra:
        with torch.no_grad():
            for _ in range(min(args.warmup, 2)):
                model.forward_train(img_inputs=inputs, gt_occ=gt_occ)
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                model.forward_train(img_inputs=inputs, gt_occ=gt_occ)
            fence()
        tf = torch.tensor([time.perf_counter() - t1], device="cuda")
        if distributed:
            dist.all_reduce(tf, op=dist.ReduceOp.MAX)
        fo_ms = float(tf) / args.steps * 1e3

    # Serial replay: the same step with the side streams off (every kernel alone on the device), dominant kernels bracketed by
    # HIP events.  In the timed region a conv_taph launch shares the CUs with weight-gradient / DepthNet kernels of the second
    # stream, so its event-to-event time there measures the co-schedule, not the kernel; both are reported.
    replay = None
    from stereoscene_amd import streams as _streams
    from stereoscene_amd.plugin import view_transformer as _vtm
    concurrent = bool(_streams.WGRAD_STREAM or _vtm.VT_STREAMS)
    if concurrent and not args.forward_only and not args.skip_serial_replay:
        saved = (_streams.WGRAD_STREAM, _vtm.VT_STREAMS)
        _streams.WGRAD_STREAM = _vtm.VT_STREAMS = False
        try:
            step()
            rt = F.KernelTimer(families=fams)
            F.KERNEL_TIMER = rt
            fence()
            t2 = time.perf_counter()
            nrep = min(args.steps, 5)
            for _ in range(nrep):
                step()
            fence()
            rdt = (time.perf_counter() - t2) / nrep
            F.KERNEL_TIMER = None
            replay = (rt.summary(), nrep, rdt * 1e3)
        finally:
            _streams.WGRAD_STREAM, _vtm.VT_STREAMS = saved
            F.KERNEL_TIMER = None

    exch = None
    if distributed and reducer is not None and (reducer.active or world == 1):
        exch = exchange_microbench(reducer, dist)
        exch["mode_in_step"] = reducer.exchange if reducer.active else "none (one rank)"
        exch["bytes_per_step_per_rank"] = reducer.flat.numel() * 4
        exch["buckets"] = len(reducer.buckets)
        exch["world"] = world
        exch["backend"] = dist.get_backend()
        exch["wire_dtype"] = reducer.comm_dtype
        if world == 1:
            exch["note"] = ("one-rank RCCL process group: the timing protocol's barrier is RCCL's and the exchange calls of the "
                            "N > 1 step (reduce_scatter_tensor AVG + all_gather_into_tensor per bucket, in place) are executed and "
                            "timed HERE, after the timed region -- inside it a one-rank group exchanges nothing (dp.py: the identity "
                            "copies of the 353 MB buffer on RCCL's stream cost +6 ms per step next to backward); bus bandwidth is "
                            "undefined at N = 1 (no peer), `ms` is the cost of the calls alone")
    elif rccl_world1 and rccl_world1 != "ok":
        exch = {"world": 1, "rccl_process_group": rccl_world1}

    if rank == 0:
        ks = timer.summary()
        peak = PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
        # ---- roofline: the dominant SINGLE kernel of the step (largest summed duration among the timed kernels)
        roof = None
        # conv_taph_kernel is ONE symbol; wino_df_kernel<MT,NW> is a family of template instances whose largest member
        # (<2,4>: 6.7 ms/step in profiles/r2z_summary.txt) is below conv_taph_kernel's 7.7 ms/step, so the single dominant
        # kernel is conv_taph_kernel whenever it ran; the other timed kernel family is reported next to it.
        def kernel_roofline(fam, ks=ks, nsteps=args.steps, timing=None):
            k = ks[fam]
            base, _, inst = fam.partition(":")
            kname, bound = SINGLE_KERNEL_FAMILIES[base]
            tname = kname                                 # the PMC traffic file lists wino_df_kernel over all its instances
            if inst:
                kname = f"{kname}<{inst[0]}, {inst[1]}>"
            n = k["launches"]
            avg_s = k["ms"] * 1e-3 / n
            exec_tf = k["executed"] / n / avg_s / 1e12
            oper_tf = k["flops"] / n / avg_s / 1e12
            traffic, traffic_src, traffic_fresh = None, None, None
            cands = (["r6_pmc_traffic_bf16.json", "r5_pmc_traffic_bf16.json", "r4_pmc_traffic_bf16.json"] if args.precision != "fp32"
                     else ["r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json"])
            tfile = next((os.path.join(ROOT, "profiles", c) for c in cands if os.path.exists(os.path.join(ROOT, "profiles", c))), None)
            if tfile and args.config == "kitti_d192" and args.batch == 1:
                # HBM bytes per launch from separate rocprofv3 --pmc passes over this same command (PMC collection cannot run
                # inside the timed process).  `traffic_matches_tree`: the file records a hash of csrc/ at collection time
                # (tools/pmc_traffic.py); it is compared with the tree this process runs ("a HEAD-hash check")
                tj = json.load(open(tfile))
                t = tj["kernels"].get(tname)
                if t:
                    traffic, traffic_src = t["hbm_bytes_per_launch"], os.path.relpath(tfile, ROOT)
                    traffic_fresh = (tj.get("csrc_sha16") == csrc_sha16()) if tj.get("csrc_sha16") else None
            return {"bound": bound, "kernel": kname,
                    "achieved": exec_tf, "peak": peak, "unit": "TFLOP/s", "frac": exec_tf / peak,
                    "flop_convention": "achieved / frac count the multiply-adds the kernel EXECUTES (" + EXECUTED_NOTE[base] +
                                       "), so frac <= 1 is matrix-pipe utilisation; operator_* count direct-convolution "
                                       "FLOPs (2*voxels*Cin*Cout*27, SURVEY 8(d))",
                    "operator_tflops": oper_tf, "operator_frac": oper_tf / peak,
                    "peak_source": "fp32 matrix (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md" if args.precision == "fp32"
                                   else "dense bf16 MFMA, MI355X_MICROARCH.md",
                    "launches_per_step": n / nsteps, "avg_launch_us": avg_s * 1e6,
                    "executed_gflop_per_launch": k["executed"] / n / 1e9,
                    "algorithmic_gflop_per_launch": k["flops"] / n / 1e9,
                    "algorithmic_bytes_per_launch": k["bytes"] / n,
                    "ms_per_step_in_kernel": k["ms"] / nsteps,
                    "traffic": traffic, "traffic_unit": "HBM bytes/launch", "traffic_source": traffic_src,
                    "traffic_matches_tree": traffic_fresh,
                    "timing": timing or ("HIP events on the launch stream around every launch inside the timed region"
                                         + ("; the step runs on TWO streams (weight gradients / DepthNet on the second), so a "
                                            "launch's event-to-event time includes the CU share of co-scheduled kernels -- see "
                                            "roofline_serial_replay for the kernel alone on the device" if concurrent else ""))}

        # every timed entry is ONE kernel symbol (conv_taph_kernel; wino_df_kernel<MT, NW> per template instance): the dominant
        # kernel is the symbol with the largest summed duration in the timed region
        timed = [f for f in ks if f.split(":")[0] in SINGLE_KERNEL_FAMILIES and ks[f]["launches"]]
        roof, roof_other = None, {}
        if timed:
            dom = max(timed, key=lambda f: ks[f]["ms"])
            roof = kernel_roofline(dom)
            roof_other = {}
            ties = []
            for f in sorted(timed, key=lambda f: -ks[f]["ms"]):
                if f != dom:
                    r = kernel_roofline(f)
                    roof_other[r["kernel"]] = r
                    if ks[f]["ms"] >= 0.95 * ks[dom]["ms"]:
                        ties.append(r)
            # summed HIP-event time decides; symbols within 5 % of the largest are reported INSIDE `roofline` too (VERDICT r4
            # item 9: conv_tapdh_kernel and wino_df_kernel<2, 4> are tied as "dominant")
            roof["selection"] = {"rule": "largest summed HIP-event duration over the timed region among the instrumented kernel symbols",
                                 "ms_per_step_by_symbol": {kernel_roofline(f)["kernel"]: ks[f]["ms"] / args.steps
                                                           for f in sorted(timed, key=lambda f: -ks[f]["ms"])[:6]},
                                 "tied_within_5_percent": [{k: r[k] for k in ("kernel", "achieved", "frac", "operator_frac",
                                                                               "avg_launch_us", "launches_per_step",
                                                                               "ms_per_step_in_kernel", "traffic")} for r in ties]}
        roof_replay = None
        if replay is not None and timed:
            rks, nrep, rms = replay
            note = ("HIP events around every launch in a serial replay of the same step after the timed region (side streams "
                    f"off, {nrep} steps, {rms:.2f} ms/step): the kernel alone on the device")
            roof_replay = {}
            for f in sorted(timed, key=lambda f: -ks[f]["ms"]):
                if f in rks and rks[f]["launches"]:
                    r = kernel_roofline(f, rks, nrep, note)
                    r["replay_ms_per_step"] = rms
                    roof_replay[r["kernel"]] = r
        # ---- whole step against its own floor: sum over operator groups of max(flops / MFMA peak, bytes / HBM peak)
        floor_op = floor_ex = 0.0
        groups = {}
        merged = {}
        for fam, c in timer.counts.items():
            m = merged.setdefault(fam.split(":")[0], dict(launches=0, flops=0.0, bytes=0.0, executed=0.0))
            for kk in m:
                m[kk] += c[kk]
        ks_fam = {}
        for fam, v in ks.items():
            m = ks_fam.setdefault(fam.split(":")[0], dict(launches=0, ms=0.0))
            m["launches"] += v["launches"]
            m["ms"] += v["ms"]
        for fam, c in merged.items():
            t_b = c["bytes"] / (PEAK_HBM_TBS * 1e12)
            t_op = max(c["flops"] / (peak * 1e12), t_b)
            t_ex = max(c["executed"] / (peak * 1e12), t_b)
            floor_op += t_op
            floor_ex += t_ex
            groups[fam] = {"launches_per_step": c["launches"] / args.steps, "gflop_per_step": c["flops"] / 1e9 / args.steps,
                           "executed_gflop_per_step": c["executed"] / 1e9 / args.steps,
                           "mbytes_per_step": c["bytes"] / 1e6 / args.steps,
                           "floor_ms_per_step": t_ex * 1e3 / args.steps}
            if fam in ks_fam and ks_fam[fam]["launches"]:
                groups[fam]["measured_ms_per_step"] = ks_fam[fam]["ms"] / args.steps
        step_roof = {"definition": "sum over instrumented operator groups of max(flops / MFMA peak, algorithmic bytes / 8 TB/s) "
                                   "divided by the measured step time; `frac` uses executed FLOPs (Winograd layers execute "
                                   "fewer multiply-adds than the operator defines) and is <= 1 by construction, "
                                   "`operator_frac` uses direct-convolution FLOPs (SURVEY 8(d)) and may exceed 1",
                     "floor_ms": floor_ex * 1e3 / args.steps, "frac": floor_ex * 1e3 / args.steps / ms,
                     "operator_floor_ms": floor_op * 1e3 / args.steps, "operator_frac": floor_op * 1e3 / args.steps / ms,
                     "groups": groups}
        out = {"metric": "voxels/sec fwd+bwd, 256x256x32 grid D=192" if args.config == "kitti_d192" else
               f"voxels/sec fwd+bwd ({args.config})",
               "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.precision == "fp32" else ("bf16 (bf16 activations and activation gradients in HBM, bf16 MFMA with fp32 "
                                                                "accumulation; fp32 master weights, weight gradients, norm statistics, "
                                                                "softmaxes, scatter and losses)" if args.precision == "bf16" else
                                                                "bf16 operands (fp32 tensors in HBM; the rounds 1-3 mode)"),
               "data": "synthetic",
               "config": {"workload": f"{args.config}: stereo pair features 2x[B,640,48,160] -> 256x256x32 occupancy, "
                                      f"D={model.img_view_transformer.D}, fwd+bwd incl. 4 losses"
                                      + (" (forward only)" if args.forward_only else "")
                                      + (f" (ablation: {args.ablation})" if args.ablation != "full" else "")
                                      + (" (precision: bf16 mixed, configs[3])" if args.precision != "fp32" else ""),
                          "batch_per_gpu": args.batch, "global_batch": world * args.batch,
                          "parallelism": f"dp{world}", "train_mode": True},
               "roofline": roof, "roofline_other_timed_kernels": roof_other, "roofline_serial_replay": roof_replay,
               "streams": {"side_stream_weight_gradients": bool(_streams.WGRAD_STREAM), "side_stream_depthnet": bool(_vtm.VT_STREAMS),
                           "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
               "step_roofline": step_roof,
               "losses": {k: float(v.detach()) for k, v in losses.items()}}
        if fo_ms is not None:
            out["forward_only"] = {"ms_per_step": fo_ms, "value": world * args.batch * scale / (fo_ms * 1e-3), "unit": "voxels/s",
                                   "note": "same model / inputs under no_grad, timed after the fwd+bwd region; not the metric"}
        if exch is not None:
            out["gradient_exchange"] = exch
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample if world == 1 else "none", cfg)
        sys.stdout.flush()
        emit_record(out, json_fd)
    os.close(json_fd)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
