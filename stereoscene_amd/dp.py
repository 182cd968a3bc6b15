"""Data-parallel gradient exchange for the hot path: one process per GPU, RCCL over xGMI
(torch.distributed backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests).

The reference wraps the model in MMDistributedDataParallel (mmdet_train.py:75-79) = bucketed
all-reduce(mean) of gradients overlapped with backward, broadcast_buffers=False.  This is the same
exchange, laid out for MI355X: all gradients live in ONE flat fp32 buffer (after a step ``param.grad``
are views into it), cut into a few large buckets in reverse-forward order; a bucket's all-reduce is
launched from the autograd hook of its last-arriving parameter, so the exchange overlaps the remaining
backward.  xGMI is point-to-point (7 links/GPU): few, large messages.

Gradients are moved into the flat buffer per BUCKET, not per parameter: during backward ``param.grad`` is
None, so autograd simply hands over the tensor our kernels produced (no ``grad += new`` launch per
parameter -- ~300 five-microsecond kernels per step on this model); when the last gradient of a bucket has
arrived, one multi-tensor copy packs the bucket and re-points ``param.grad`` at the flat views.
(A parameter used twice in one graph accumulates into its view after packing; with world_size > 1 such
a parameter must not be split from its second use by a bucket boundary.  The hot path has none.)
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
        self._views = {p: p.grad for p in self.params}
        self._members = [[] for _ in self.buckets]
        for p in self.params:
            self._members[self._bucket_of[p]].append(p)
        self._arrived = [0] * len(self.buckets)
        self._packed = [False] * len(self.buckets)
        self._seen = set()
        self._handles = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _pack(self, b):
        """Move the stolen gradient tensors of bucket b into the flat buffer (one multi-tensor copy)."""
        dst, src = [], []
        for p in self._members[b]:
            view = self._views[p]
            if p.grad is None:
                view.zero_()                       # no gradient reached this parameter in this step
            elif p.grad.data_ptr() != view.data_ptr():
                dst.append(view)
                src.append(p.grad.detach().reshape(view.shape))
            p.grad = view
        if dst:
            torch._foreach_copy_(dst, src)
        self._packed[b] = True

    def _on_grad(self, p):
        if p in self._seen:                        # second use of a shared parameter: already counted
            return
        self._seen.add(p)
        b = self._bucket_of[p]
        self._arrived[b] += 1
        if self._arrived[b] == self.buckets[b][2]:
            self._pack(b)
            if self.active:
                s, e, _ = self.buckets[b]
                self._handles.append(dist.all_reduce(self.flat[s:e], group=self.group, async_op=True))

    def zero_grad(self):
        for p in self.params:
            p.grad = None                          # autograd will hand over its tensors; nothing to clear
        self._arrived = [0] * len(self.buckets)
        self._packed = [False] * len(self.buckets)
        self._seen.clear()

    def finish(self):
        """Wait for the in-flight buckets (call after backward()); returns bytes exchanged per rank."""
        for h in self._handles:
            h.wait()
        # buckets with a parameter that received no gradient never completed: pack (and reduce) them now
        for b, (s, e, c) in enumerate(self.buckets):
            if not self._packed[b]:
                self._pack(b)
                if self.active:
                    dist.all_reduce(self.flat[s:e], group=self.group)
        if self.active and self.average and self.world > 1:
            self.flat.div_(self.world)
        self._handles = []
        self._arrived = [0] * len(self.buckets)
        self._seen.clear()
        return self.flat.numel() * 4

    def remove(self):
        for h in self._hooks:
            h.remove()
