"""world_size-2 data-parallel gradient exchange on CPU (gloo): the flat-bucket all-reduce gives every
rank the mean gradient, identical to a single process seeing both shards."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8),
                               torch.nn.ReLU(), torch.nn.Linear(8, 4))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stereoscene_amd.dp import FlatGradAllReduce
    m = _model()
    red = FlatGradAllReduce(m, bucket_mb=0.001)          # tiny buckets -> several async all-reduces
    assert len(red.buckets) >= 2
    g = torch.Generator().manual_seed(100 + rank)
    for step in range(2):
        red.zero_grad()
        x = torch.randn(5, 16, generator=g)
        m(x).square().mean().backward()
        red.finish()
    out[rank] = [p.grad.clone() for p in m.parameters()]
    dist.destroy_process_group()


def test_flat_bucket_allreduce_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    # single-process reference: mean of the two ranks' step-2 gradients
    ref = []
    for rank in range(world):
        m = _model()
        g = torch.Generator().manual_seed(100 + rank)
        for step in range(2):
            m.zero_grad()
            x = torch.randn(5, 16, generator=g)
            m(x).square().mean().backward()
        ref.append([p.grad.clone() for p in m.parameters()])
    want = [(a + b) / 2 for a, b in zip(*ref)]
    for r in range(world):
        for got, w in zip(out[r], want):
            assert torch.allclose(got, w, atol=1e-6)


class _Done:
    def wait(self):
        return True


def _worker_rs_ag(rank, world, port, out):
    """exchange="rs_ag" (the RCCL default) with reduce_scatter_tensor / all_gather_into_tensor emulated on gloo: checks the
    bucket padding, the per-rank shard views and the folded average of FlatGradAllReduce's own arithmetic."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def reduce_scatter_tensor(output, input, op=dist.ReduceOp.SUM, group=None, async_op=False):
        n = output.numel()
        assert input.numel() == n * world and output.data_ptr() == input.data_ptr() + rank * n * input.element_size()
        tmp = input.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
        if op == dist.ReduceOp.AVG:
            tmp /= world
        output.copy_(tmp[rank * n:(rank + 1) * n])
        return _Done()

    def all_gather_into_tensor(output, input, group=None, async_op=False):
        parts = [torch.empty_like(input) for _ in range(world)]
        dist.all_gather(parts, input.clone())
        output.copy_(torch.cat(parts))
        return _Done()

    dist.reduce_scatter_tensor, dist.all_gather_into_tensor = reduce_scatter_tensor, all_gather_into_tensor
    from stereoscene_amd.dp import FlatGradAllReduce
    m = _model()
    red = FlatGradAllReduce(m, bucket_mb=0.001, exchange="rs_ag")
    assert red.exchange == "rs_ag" and all((e - s) % world == 0 for s, e, _ in red.buckets)
    g = torch.Generator().manual_seed(100 + rank)
    for step in range(2):
        red.zero_grad()
        x = torch.randn(5, 16, generator=g)
        m(x).square().mean().backward()
        red.finish()
    out[rank] = [p.grad.clone() for p in m.parameters()]
    dist.destroy_process_group()


def test_flat_bucket_reduce_scatter_all_gather_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_rs_ag, args=(world, port, out), nprocs=world, join=True)
    ref = []
    for rank in range(world):
        m = _model()
        g = torch.Generator().manual_seed(100 + rank)
        for step in range(2):
            m.zero_grad()
            x = torch.randn(5, 16, generator=g)
            m(x).square().mean().backward()
        ref.append([p.grad.clone() for p in m.parameters()])
    want = [(a + b) / 2 for a, b in zip(*ref)]
    for r in range(world):
        for got, w in zip(out[r], want):
            assert torch.allclose(got, w, atol=1e-6)


def test_optimizer_state_is_world_size_independent():
    """ADVICE r2 (dp.py:62): bucket ends are padded to a multiple of the world size, so the flat moment buffers have a
    world-dependent length; the checkpointed state must not (train on 8 GPUs, resume on 1)."""
    from stereoscene_amd.dp import FlatGradAllReduce
    from stereoscene_amd.train import FlatAdamW
    ma, mb = _model(), _model()
    opt8 = FlatAdamW(ma, reducer=FlatGradAllReduce(ma, bucket_mb=0.001, align=8))      # the layout of an 8-rank run
    opt1 = FlatAdamW(mb, reducer=FlatGradAllReduce(mb, bucket_mb=64, align=1))         # ... resumed on one GPU
    assert opt8.m.numel() != opt1.m.numel()
    torch.manual_seed(3)
    opt8.m.normal_()
    opt8.v.uniform_()
    opt8.step_count = 7
    sd = opt8.state_dict()
    n = sum(p.numel() for p in ma.parameters())
    assert sd["m"].numel() == n and sd["v"].numel() == n
    opt1.load_state_dict(sd)
    assert opt1.step_count == 7
    for pa, pb in zip(opt8.reducer.params, opt1.reducer.params):
        oa, ob, k = opt8.reducer._offsets[pa], opt1.reducer._offsets[pb], pa.numel()
        assert torch.equal(opt8.m[oa:oa + k], opt1.m[ob:ob + k])
        assert torch.equal(opt8.v[oa:oa + k], opt1.v[ob:ob + k])
    # and back: a 1-GPU checkpoint loads into the padded layout, padding stays zero
    opt8b = FlatAdamW(_model(), reducer=None)
    opt8b.load_state_dict(opt1.state_dict())
    # a pre-r3 checkpoint (padded buffers, same layout) still loads
    legacy = {"m": opt8.m.clone(), "v": opt8.v.clone(), "step": 3, "lr": 1e-4}
    opt8.m.zero_()
    opt8.load_state_dict(legacy)
    assert torch.equal(opt8.m, legacy["m"])


def _bf16_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stereoscene_amd.dp import FlatGradAllReduce
    res = {}
    for cd in ("fp32", "bf16"):
        m = _model()
        red = FlatGradAllReduce(m, bucket_mb=0.001, comm_dtype=cd)
        g = torch.Generator().manual_seed(100 + rank)
        red.zero_grad()
        m(torch.randn(5, 16, generator=g)).square().mean().backward()
        nbytes = red.finish()
        res[cd] = (red.flat.clone(), nbytes)
    out[rank] = res
    dist.destroy_process_group()


def test_bf16_wire_format_halves_the_bytes_and_stays_within_bf16_rounding():
    """SSBEV_DP_COMM_DTYPE=bf16: the exchanged mean gradient equals the fp32 exchange to bf16 rounding of
    the operands and of the sum (<= 2^-7 relative to the largest gradient of a bucket), at half the bytes on the wire."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_bf16_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        (f32, b32), (bf, b16) = out[r]["fp32"], out[r]["bf16"]
        assert b16 * 2 == b32
        assert torch.equal(out[0]["bf16"][0], bf)                      # ranks agree bit for bit
        err = (bf - f32).abs().max().item()
        assert 0 < err <= 2.0 ** -7 * f32.abs().max().item(), err
