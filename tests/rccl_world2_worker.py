"""Worker of tests/test_gpu_rccl.py::test_rccl_world2_*: one rank of a 2-rank RCCL job on two GPUs of one node (launched by
``python -m torch.distributed.run --nproc-per-node 2``).  Each rank runs a fwd+bwd step of the small configuration on its OWN
synthetic sample; the flat gradient buffer after dp.FlatGradAllReduce's exchange must equal the mean of the two ranks'
local gradients (gathered with a plain RCCL all_gather), for both exchange realisations and the bf16 wire format.
Reference: MMDistributedDataParallel's gradient all-reduce(mean), mmdet_train.py:70-79."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from stereoscene_amd import dp, model_zoo, synthetic as S
    cfg = S.CFG_S
    model = model_zoo.build_detector(cfg).train()           # fill-by-key weights: identical on every rank
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    smp = S.synthetic_sample(cfg, B=1, tag=f"w2rank{rank}")  # a different sample per rank
    inputs = model_zoo.img_inputs_from_sample(smp)
    gt = smp["gt_occ"].cuda()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def step(red):
        model.load_state_dict(sd0)
        red.zero_grad()
        losses = model.forward_train(img_inputs=inputs, gt_occ=gt)
        sum(v for k, v in losses.items() if k.startswith("loss")).backward()
        n = red.finish()
        torch.cuda.synchronize()
        return red.flat.detach().clone(), n

    # local gradients: a reducer without a process group view (exchange switched off by asking for a one-rank layout)
    red0 = dp.FlatGradAllReduce(model, bucket_mb=16, process_group=None, align=world)
    red0.active = False
    local_flat, _ = step(red0)
    red0.remove()
    both = [torch.empty_like(local_flat) for _ in range(world)]
    dist.all_gather(both, local_flat)
    want = (both[0].double() + both[1].double()) / 2.0
    assert not torch.equal(both[0], both[1]), "the two ranks must see different samples"
    scale = want.abs().max().item()
    for exchange, wire, tol in (("rs_ag", "fp32", 2e-6), ("all_reduce", "fp32", 2e-6), ("rs_ag", "bf16", 2.0 ** -7)):
        red = dp.FlatGradAllReduce(model, bucket_mb=16, exchange=exchange, comm_dtype=wire)
        assert red.active and red.world == world and red.native_avg and len(red.buckets) >= 4
        got, nbytes = step(red)
        assert nbytes == red.flat.numel() * (4 if wire == "fp32" else 2), (nbytes, red.flat.numel())
        err = (got.double() - want).abs().max().item()
        assert err <= tol * scale, (exchange, wire, err, scale)
        # every rank holds the same averaged gradient, bit for bit
        ref = got.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(ref, got), (exchange, wire)
        red.remove()
    dist.barrier()
    if rank == 0:
        print("RCCL_WORLD2_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
