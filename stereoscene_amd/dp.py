"""Data-parallel gradient exchange for the hot path: one process per GPU, RCCL over xGMI
(torch.distributed backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests).

The reference wraps the model in MMDistributedDataParallel (mmdet_train.py:70-79, launched by
tools/dist_train.sh:9-19) = bucketed all-reduce(mean) of gradients overlapped with backward,
broadcast_buffers=False.  This is the same exchange, laid out for MI355X: all gradients live in ONE flat
fp32 buffer (after a step ``param.grad`` are views into it), cut into a few large buckets in reverse-forward
order; a bucket's exchange is launched from the autograd hook of its last-arriving parameter, so it overlaps
the remaining backward.  xGMI is point-to-point (7 links/GPU): few, large messages.

Exchange per bucket (``exchange=``, env SSBEV_DP_EXCHANGE):
  * "rs_ag" (default on RCCL): in-place reduce-scatter(AVG) of the bucket into this rank's
    1/world slice, then in-place all-gather of the slices -- every GPU sends a distinct slice to each peer over
    its own xGMI link instead of pushing the whole bucket around one ring.  Bucket boundaries are multiples of
    ``world`` elements so the slices tile the bucket exactly.
  * "all_reduce": one all-reduce(AVG) per bucket (what gloo runs, with SUM + a scale pass: gloo has no AVG /
    reduce-scatter).
The 1/world of the mean is folded into the reduction (ReduceOp.AVG); there is no separate scaling pass on RCCL.

Gradients are moved into the flat buffer per BUCKET, not per parameter: during backward ``param.grad`` is
None, so autograd simply hands over the tensor our kernels produced (no ``grad += new`` launch per
parameter -- ~300 five-microsecond kernels per step on this model); when the last gradient of a bucket has
arrived, one multi-tensor copy packs the bucket and re-points ``param.grad`` at the flat views.
(A parameter used twice in one graph accumulates into its view after packing; with world_size > 1 such
a parameter must not be split from its second use by a bucket boundary.  The hot path has none.)

Parameters that received no gradient in a step (sub-modules switched off by an ablation mode) get a zeroed
slice for the exchange and are listed in ``no_grad_ranges``: the fused optimizer skips those ranges, like
torch.optim.AdamW skips ``grad is None``.
"""
import os

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, module, bucket_mb=64, process_group=None, average=True, exchange=None, align=None, comm_dtype=None,
                 timeline=False, exchange_at_world1=False):
        self.group = process_group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.active else 1
        # A process group of ONE rank has nothing to exchange: the collectives would be identity copies of the whole flat buffer
        # on RCCL's stream next to backward (measured on MI355X, 353 MB per step: +6 ms on a 72 ms step).  They are skipped
        # unless asked for (tests/test_gpu_rccl.py runs the real calls at world size 1; bench.py times them after the step).
        if self.active and self.world == 1 and not exchange_at_world1:
            self.active = False
        self.rank = dist.get_rank(process_group) if self.active else 0
        self.average = average
        backend = dist.get_backend(process_group) if self.active else "none"
        self.native_avg = backend == "nccl"                   # RCCL: AVG + reduce-scatter available
        explicit = exchange is not None                       # an explicit argument is honoured on any backend (tests emulate it on gloo)
        exchange = exchange or os.environ.get("SSBEV_DP_EXCHANGE", "rs_ag")
        if exchange not in ("rs_ag", "all_reduce"):
            raise ValueError(f"exchange must be 'rs_ag' or 'all_reduce', got {exchange!r}")
        self.exchange = exchange if (self.native_avg or explicit) else "all_reduce"
        # Optional bf16 wire format (env SSBEV_DP_COMM_DTYPE=bf16): a bucket is rounded to bf16, exchanged (half the bytes over
        # xGMI) and widened back into the fp32 flat buffer.  The SUM then happens in bf16 on the wire: relative error of the
        # averaged gradient <= ~2^-8 per element (tests/test_dp_gloo.py measures it) -- opt-in, like the reference's fp16
        # all-reduce under Fp16OptimizerHook; master gradients / moments stay fp32.
        # default: fp32 on the wire in the fp32 mode, bf16 in the bf16 storage mode ("bf16 exchange on by default in that
        # mode" -- the counterpart of the reference's fp16 gradient all-reduce under Fp16OptimizerHook, mmdet_train.py:131-134)
        from . import functional as _F
        cd = comm_dtype or os.environ.get("SSBEV_DP_COMM_DTYPE", "bf16" if _F.storage_bf16() else "fp32")
        if cd not in ("fp32", "bf16"):
            raise ValueError(f"comm_dtype must be 'fp32' or 'bf16', got {cd!r}")
        self.comm_dtype = cd
        self._wire = []                            # (bucket, bf16 buffer) pairs to widen back after the wait
        # Optional per-bucket "ready" timeline (tools/bucket_timeline.py): one event when a bucket's last gradient has arrived
        self.timeline = [] if timeline else None
        params = [p for p in module.parameters() if p.requires_grad]
        # gradients become ready roughly in reverse registration order (head -> ... -> stereo net)
        self.params = list(reversed(params))
        dev = self.params[0].device
        cap = max(1, int(bucket_mb * (1 << 20) // 4))
        align = max(self.world, 1) if align is None else int(align)   # (tests pass `align` to emulate another world size's layout)
        self.buckets, self._bucket_of = [], {}
        self._offsets = {}
        off, start, count = 0, 0, 0
        for p in self.params:
            self._offsets[p] = off
            self._bucket_of[p] = len(self.buckets)
            off += p.numel()
            count += 1
            if off - start >= cap:
                off = -(-off // align) * align               # pad: the bucket splits into `world` equal slices
                self.buckets.append((start, off, count))
                start, count = off, 0
        if count:
            off = -(-off // align) * align
            self.buckets.append((start, off, count))
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        for p in self.params:
            o = self._offsets[p]
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        self._views = {p: p.grad for p in self.params}
        self._members = [[] for _ in self.buckets]
        for p in self.params:
            self._members[self._bucket_of[p]].append(p)
        self._arrived = [0] * len(self.buckets)
        self._packed = [False] * len(self.buckets)
        self._seen = set()
        self._handles = []
        self.no_grad_ranges = []          # [(start, end)] element ranges of the flat buffer without a gradient this step
        self.bytes_exchanged = 0
        # the stream the flat buffer belongs to: the caller's stream NOW (a lazily resolved home could become the side
        # stream when the first bucket completes inside DepthNet's side-stream backward); zero_grad() re-captures it
        self._home = torch.cuda.current_stream(dev) if self.flat.is_cuda else None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _home_stream(self):
        """The stream the flat buffer belongs to: the caller's stream at ``zero_grad()`` (or at construction).  Packing and
        the exchange always run THERE, whatever stream the autograd hook happens to fire on (DepthNet's backward -- and with
        it the AccumulateGrad nodes of its parameters and this hook -- runs on the side stream, streams.py)."""
        return self._home

    def _pack(self, b):
        """Move the stolen gradient tensors of bucket b into the flat buffer (one multi-tensor copy)."""
        if not self.flat.is_cuda:
            return self._pack_on_current(b)
        from . import streams
        home = self._home_stream()
        here = torch.cuda.current_stream(self.flat.device)
        if here != home:
            home.wait_stream(here)              # gradients produced by the node whose hook this is
        # ... and everything the side stream has been given so far: weight gradients sent there by the convolution wrappers
        # AND gradients computed there in place (DepthNet's), which never mark it dirty
        streams.wait_side(home)
        with torch.cuda.stream(home):
            self._pack_on_current(b)

    def _pack_on_current(self, b):
        dst, src = [], []
        for p in self._members[b]:
            view = self._views[p]
            if p.grad is None:
                view.zero_()                       # no gradient reached this parameter in this step
                o = self._offsets[p]
                self.no_grad_ranges.append((o, o + p.numel()))
            elif p.grad.data_ptr() != view.data_ptr():
                dst.append(view)
                src.append(p.grad.detach().reshape(view.shape))
            p.grad = view
        if dst:
            if self.flat.is_cuda:
                home = torch.cuda.current_stream(self.flat.device)
                for t in src:                      # allocated on another stream, read here: tell the caching allocator
                    t.record_stream(home)
            torch._foreach_copy_(dst, src)
        self._packed[b] = True

    def _exchange(self, b, async_op):
        """Launch the exchange of bucket b; returns the work handles (in issue order)."""
        s, e, _ = self.buckets[b]
        buf = self.flat[s:e]
        if self.comm_dtype == "bf16":
            wire = buf.to(torch.bfloat16)
            self.bytes_exchanged += wire.numel() * 2
            self._wire.append((b, wire))
            buf = wire
        else:
            self.bytes_exchanged += buf.numel() * 4
        if self.exchange == "rs_ag":
            n = (e - s) // self.world
            mine = buf[self.rank * n:(self.rank + 1) * n]
            op = dist.ReduceOp.AVG if self.average else dist.ReduceOp.SUM
            h1 = dist.reduce_scatter_tensor(mine, buf, op=op, group=self.group, async_op=async_op)
            h2 = dist.all_gather_into_tensor(buf, mine, group=self.group, async_op=async_op)
            return [h1, h2] if async_op else []
        op = dist.ReduceOp.AVG if (self.average and self.native_avg) else dist.ReduceOp.SUM
        h = dist.all_reduce(buf, op=op, group=self.group, async_op=async_op)
        return [h] if async_op else []

    def _on_grad(self, p):
        if p in self._seen:                        # second use of a shared parameter: already counted
            return
        self._seen.add(p)
        b = self._bucket_of[p]
        self._arrived[b] += 1
        if self._arrived[b] == self.buckets[b][2]:
            self._pack(b)
            if not self.flat.is_cuda:
                if self.active:
                    self._handles += self._exchange(b, async_op=True)
                return
            with torch.cuda.stream(self._home_stream()):       # the collective is ordered behind the pack on the home stream
                if self.timeline is not None:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    self.timeline.append((b, ev))
                if self.active:
                    self._handles += self._exchange(b, async_op=True)

    def zero_grad(self):
        if self.flat.is_cuda:
            from . import streams
            self._home = torch.cuda.current_stream(self.flat.device)
            streams.new_step()
        for p in self.params:
            p.grad = None                          # autograd will hand over its tensors; nothing to clear
        self._arrived = [0] * len(self.buckets)
        self._packed = [False] * len(self.buckets)
        self._seen.clear()
        self.no_grad_ranges = []
        self.bytes_exchanged = 0
        self._wire = []
        if self.timeline is not None:
            self.timeline = []

    def finish(self):
        """Wait for the in-flight buckets (call after backward()); returns bytes exchanged per rank."""
        for h in self._handles:
            h.wait()
        # buckets with a parameter that received no gradient never completed: pack (and reduce) them now
        late = []
        for b in range(len(self.buckets)):
            if not self._packed[b]:
                self._pack(b)
                if self.active:
                    self._exchange(b, async_op=False)
                    late.append(b)
        for b, wire in self._wire:                 # bf16 wire format: widen the exchanged buckets back into the fp32 buffer
            s, e, _ = self.buckets[b]
            self.flat[s:e].copy_(wire)
        self._wire = []
        if self.active and self.average and self.world > 1 and not self.native_avg and self.exchange == "all_reduce":
            self.flat.div_(self.world)             # gloo (CPU tests): SUM + scale
        self._handles = []
        self._arrived = [0] * len(self.buckets)
        self._seen.clear()
        return self.bytes_exchanged if self.active else 0

    def live_ranges(self):
        """Complement of ``no_grad_ranges`` in [0, flat.numel()): the element ranges the optimizer should update."""
        out, pos = [], 0
        for s, e in sorted(self.no_grad_ranges):
            if s > pos:
                out.append((pos, s))
            pos = max(pos, e)
        if pos < self.flat.numel():
            out.append((pos, self.flat.numel()))
        return out

    def remove(self):
        for h in self._hooks:
            h.remove()
