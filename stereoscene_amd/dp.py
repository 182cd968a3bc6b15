"""Data-parallel gradient exchange for the hot path: one process per GPU, RCCL over xGMI
(torch.distributed backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests).

The reference wraps the model in MMDistributedDataParallel (mmdet_train.py:75-79) = bucketed
all-reduce(mean) of gradients overlapped with backward, broadcast_buffers=False.  This is the same
exchange, laid out for MI355X: all gradients live in ONE flat fp32 buffer (``param.grad`` are views
into it: no gather/scatter copies), cut into a few large buckets in reverse-forward order; a bucket's
all-reduce is launched from the autograd hook of its last-arriving parameter, so the exchange
overlaps the remaining backward.  xGMI is point-to-point (7 links/GPU): few, large messages.
"""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, module, bucket_mb=64, process_group=None, average=True):
        self.group = process_group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.active else 1
        self.average = average
        params = [p for p in module.parameters() if p.requires_grad]
        # gradients become ready roughly in reverse registration order (head -> ... -> stereo net)
        self.params = list(reversed(params))
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        cap = max(1, int(bucket_mb * (1 << 20) // 4))
        self.buckets, self._bucket_of, self._pending = [], {}, []
        off, start, count = 0, 0, 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self._bucket_of[p] = len(self.buckets)
            off += n
            count += 1
            if off - start >= cap:
                self.buckets.append((start, off, count))
                start, count = off, 0
        if count:
            self.buckets.append((start, off, count))
        self._arrived = [0] * len(self.buckets)
        self._handles = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _on_grad(self, p):
        b = self._bucket_of[p]
        self._arrived[b] += 1
        if self._arrived[b] == self.buckets[b][2] and self.active:
            s, e, _ = self.buckets[b]
            self._handles.append(dist.all_reduce(self.flat[s:e], group=self.group, async_op=True))

    def zero_grad(self):
        self.flat.zero_()
        self._arrived = [0] * len(self.buckets)

    def finish(self):
        """Wait for the in-flight buckets (call after backward()); returns bytes exchanged per rank."""
        if self.active:
            for h in self._handles:
                h.wait()
            # parameters that received no gradient this step never fire their hook: reduce their buckets now
            for b, (s, e, c) in enumerate(self.buckets):
                if self._arrived[b] != c:
                    dist.all_reduce(self.flat[s:e], group=self.group)
            if self.average and self.world > 1:
                self.flat.div_(self.world)
        self._handles = []
        self._arrived = [0] * len(self.buckets)
        return self.flat.numel() * 4

    def remove(self):
        for h in self._hooks:
            h.remove()
