"""First RCCL calls: the data-parallel gradient exchange of dp.FlatGradAllReduce under a REAL
``torch.distributed`` process group on the ``nccl`` backend (= RCCL on ROCm), world size 1 on the one GPU a test box has.

With one rank the reduce-scatter(AVG) + all-gather pair is the identity on the bucket, so the packed flat gradient buffer of a
KITTI-size step must be bit-identical to the run without any process group -- what this pins is everything around the
arithmetic that the gloo emulation (tests/test_dp_gloo.py) cannot: the in-place ``reduce_scatter_tensor(mine < buf, AVG)``
and ``all_gather_into_tensor(buf, mine)`` calls RCCL actually accepts, asynchronous work handles issued from autograd hooks
on the home stream while backward is still running on two streams, the bf16 wire format, and the wait in ``finish()``.
The reference exchange: MMDistributedDataParallel, mmdet_train.py:70-79 (launched by tools/dist_train.sh:9-19).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from stereoscene_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture
def rccl_world1():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert not dist.is_initialized()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        yield
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def _setup(tag):
    from stereoscene_amd import model_zoo
    cfg = S.CFG_K112
    model = model_zoo.build_detector(cfg).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    smp = S.synthetic_sample(cfg, B=1, tag=tag)
    inputs = model_zoo.img_inputs_from_sample(smp)
    gt = smp["gt_occ"].to(DEV)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model, inputs, gt, sd0


def _step(model, red, inputs, gt, sd0):
    model.load_state_dict(sd0)
    red.zero_grad()
    losses = model.forward_train(img_inputs=inputs, gt_occ=gt)
    sum(v for k, v in losses.items() if k.startswith("loss")).backward()
    nbytes = red.finish()
    torch.cuda.synchronize()
    return red.flat.detach().clone(), nbytes


def test_rccl_world1_exchange_is_the_identity_on_the_flat_buffer(rccl_world1):
    from stereoscene_amd import dp
    model, inputs, gt, sd0 = _setup("rccl")
    # --- no exchange at all (a reducer that believes there is no process group)
    red0 = dp.FlatGradAllReduce(model, bucket_mb=4)
    assert not red0.active                          # default: a process group of one rank exchanges nothing
    want, n0 = _step(model, red0, inputs, gt, sd0)
    red0.remove()
    assert n0 == 0
    for exchange in ("rs_ag", "all_reduce"):
        red = dp.FlatGradAllReduce(model, bucket_mb=4, exchange=exchange, exchange_at_world1=True)
        assert red.active and red.world == 1 and red.native_avg and red.exchange == exchange and len(red.buckets) > 20
        for _ in range(2):
            got, nbytes = _step(model, red, inputs, gt, sd0)
            assert nbytes == red.flat.numel() * 4                       # every bucket went through RCCL
            assert torch.equal(got, want), exchange          # every kernel of the step is deterministic (round 6: incl. the DCN gradient)
        red.remove()


def test_rccl_world1_bf16_wire_rounds_once(rccl_world1):
    """bf16 wire format (dp.py, default in the bf16 storage mode): with one rank the exchanged gradient is the fp32 gradient
    rounded to bf16 exactly once (AVG over one rank adds no arithmetic)."""
    from stereoscene_amd import dp
    model, inputs, gt, sd0 = _setup("rccl16")
    red0 = dp.FlatGradAllReduce(model, bucket_mb=16)
    want, _ = _step(model, red0, inputs, gt, sd0)
    red0.remove()
    red = dp.FlatGradAllReduce(model, bucket_mb=16, comm_dtype="bf16", exchange_at_world1=True)
    got, nbytes = _step(model, red, inputs, gt, sd0)
    assert nbytes == red.flat.numel() * 2
    assert torch.equal(got, want.to(torch.bfloat16).float())
    red.remove()


def test_rccl_world1_collectives_on_a_bucket_sized_buffer(rccl_world1):
    """The two raw calls of the "rs_ag" exchange on a 64 MB bucket, in place, with AVG: what dp._exchange issues per bucket."""
    buf = torch.randn(16 << 20, device=DEV)
    ref = buf.clone()
    mine = buf[0:buf.numel()]
    h1 = dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.AVG, async_op=True)
    h2 = dist.all_gather_into_tensor(buf, mine, async_op=True)
    h1.wait()
    h2.wait()
    torch.cuda.synchronize()
    assert torch.equal(buf, ref)
    t = torch.tensor([3.0], device=DEV)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    assert float(t) == 3.0


# ------------------------------------------------------------------------------------------------ two ranks, two GPUs
def _launch_two_ranks(script_args, timeout=900):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), *script_args]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)


needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on one node (xGMI peers)")


@needs_two_gpus
def test_rccl_world2_exchange_averages_the_two_ranks_gradients():
    """the first exchange WITH a peer.  Self-skips on a one-GPU box; see tests/rccl_world2_worker.py."""
    out = _launch_two_ranks([os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_world2_worker.py")])
    assert out.returncode == 0 and "RCCL_WORLD2_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


@needs_two_gpus
def test_bench_two_ranks_prints_one_record_with_bus_bandwidth():
    """bench.py exactly as the driver launches it at N = 2 (torch.distributed.run, one rank per GPU over RCCL), small config."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = _launch_two_ranks([os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "small_d48",
                             "--cpu-sample", "none", "--skip-serial-replay"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 2 and rec["scaling"] == "weak"
    ex = rec["gradient_exchange"]
    assert ex["world"] == 2 and ex["backend"] == "nccl" and ex["rs_ag"]["bus_GBps"] > 0
