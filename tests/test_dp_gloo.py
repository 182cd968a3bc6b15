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
