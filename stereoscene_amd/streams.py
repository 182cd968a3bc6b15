"""Second HIP stream for work that hangs off the critical chain of the backward pass.

The data-gradient chain of backward is strictly sequential (layer L's needs layer L+1's), and between its matrix-core
kernels sit the HBM-bound normalisation passes.  The WEIGHT gradients depend only on (saved input, incoming gradient): they
are leaves of that chain.  Launched on a second stream they run next to the chain -- an MFMA-bound weight-gradient kernel
shares the CUs with a streaming GroupNorm pass of the chain, and the idle tail of one kernel is filled by the head of the
other -- instead of in front of it.  (HIP orders kernels of ONE stream with a barrier even when they are independent.)

ONE side stream per device serves both users (the weight gradients and view_transformer's DepthNet branch): with two side
streams next to the caller's -- three HIP streams waiting on one another's events -- the step dead-locked on the device in
about half of the runs on ROCm 7.2 (never with two streams; bisected in profiles/r3_stream_hang_bisect.txt), although every
wait refers to work enqueued earlier.  Work that is already ON the side stream (DepthNet's backward) runs in place.

Contract (what keeps this safe with autograd and the caching allocator):
  * ``with on_side(*reads)``: the side stream first waits for everything queued on the caller's stream (the tensors in
    ``reads`` were produced there; they are also registered with the allocator as used on the side stream);
  * tensors allocated inside the block and handed back to the caller's stream are passed through ``publish`` (registered as
    used on the caller's stream);
  * the caller's stream does NOT wait when the block ends.  It waits (``join``) when somebody is about to READ a result:
    dp.FlatGradAllReduce._pack (gradient bucket complete) and an end-of-backward callback queued on the autograd engine, so
    after ``backward()`` returns every gradient is ordered before later work on the caller's stream.
  * a weight gradient is only computed on the side stream when autograd will merely STORE it (a leaf parameter whose
    ``.grad is None``, used ONCE in the graph, without tensor hooks); an existing ``.grad`` would be read by AccumulateGrad
    on the caller's stream straight away, so would the gradient of a non-leaf weight by the node that produced it, and the
    two gradients of a weight used twice are summed by the engine on the caller's stream (``note_use`` / ``wgrad_on_side``).
"""
import os

import torch

WGRAD_STREAM = os.environ.get("SSBEV_WGRAD_STREAM", "1") != "0"

_SIDE = {}
_DIRTY = set()          # device indices whose side stream holds work the caller's stream has not waited for
_CB_QUEUED = set()


# Optional CU mask of the side stream ("stop the side stream from taxing the critical path"): the weight
# gradient / DepthNet kernels are then confined to a subset of the 256 CUs (hipExtStreamCreateWithCUMask) and leave the rest to the
# data-gradient chain.  SSBEV_SIDE_CU_MASK = "<n>" (the n lowest mask bits) or "<n>s" (n bits spread evenly over the 256);
# unset = an ordinary stream.  Measured in round 4: profiles/r4_side_stream_cu_mask.txt.
SIDE_CU_MASK = os.environ.get("SSBEV_SIDE_CU_MASK", "")


def _masked_stream(idx, spec):
    import ctypes
    n = int(spec.rstrip("s"))
    total = torch.cuda.get_device_properties(idx).multi_processor_count
    bits = [0] * ((total + 31) // 32)
    picks = [int(i * total / n) for i in range(n)] if spec.endswith("s") else list(range(n))
    for c in picks:
        bits[c // 32] |= 1 << (c % 32)
    arr = (ctypes.c_uint32 * len(bits))(*bits)
    hip = ctypes.CDLL("libamdhip64.so")
    h = ctypes.c_void_p()
    with torch.cuda.device(idx):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(len(bits)), arr)
    if rc != 0 or not h.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(h.value, device=idx)


def side_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _SIDE.get(idx)
    if st is None:
        st = _SIDE[idx] = _masked_stream(idx, SIDE_CU_MASK) if SIDE_CU_MASK else torch.cuda.Stream(device=idx)
    return idx, st


def join(device=None):
    """Make the current stream wait for the side stream(s) (no-op when they are idle)."""
    for idx in list(_DIRTY):
        if device is None or device.index in (None, idx):
            cur = torch.cuda.current_stream(idx)
            if cur == _SIDE[idx]:
                continue        # called from work that lives ON the side stream: nothing was ordered for the caller's stream
            cur.wait_stream(_SIDE[idx])
            _DIRTY.discard(idx)


def wait_side(stream):
    """Unconditionally order ``stream`` behind everything queued on its device's side stream so far (dp._pack: the gradients
    of a bucket may have been produced there inline -- DepthNet's backward -- without ever marking it dirty)."""
    idx = stream.device.index
    st = _SIDE.get(idx)
    if st is not None and st != stream:
        stream.wait_stream(st)
        if torch.cuda.current_stream(idx) == stream:
            _DIRTY.discard(idx)


def new_step():
    """Start of an optimisation step (dp.FlatGradAllReduce.zero_grad): forget per-graph state that a failed or abandoned
    backward may have left behind -- the engine drops its callbacks when backward() raises, and forward passes whose graph was
    never run backward leave their weight-use counts."""
    global _EPOCH
    _CB_QUEUED.clear()
    _EPOCH_CB[0] = False
    _EPOCH += 1


def _end_of_backward():
    _CB_QUEUED.clear()
    join()


_EPOCH_CB = [False]


def _epoch_end_of_backward():
    """End of ANY backward run that asked ``wgrad_on_side``: the weight-use counts of this graph are spent, and counts left by
    forward passes that were never run backward (validation without no_grad, an exception between forward and backward) must
    not make every later step look like a shared-weight graph."""
    global _EPOCH
    _EPOCH_CB[0] = False
    _EPOCH += 1


class on_side:
    def __init__(self, device, *reads):
        self.device, self.reads = device, reads

    def __enter__(self):
        self.idx, self.side = side_stream(self.device)
        self.main = torch.cuda.current_stream(self.idx)
        self.inline = self.main == self.side       # already on the side stream (DepthNet's backward lives there): run in place
        if self.inline:
            return self
        self.side.wait_stream(self.main)
        for t in self.reads:
            if t is not None:
                t.record_stream(self.side)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def publish(self, *tensors):
        if self.inline:
            return
        for t in tensors:
            if t is not None:
                t.record_stream(self.main)

    def __exit__(self, *exc):
        if self.inline:
            return False
        self.ctx.__exit__(*exc)
        _DIRTY.add(self.idx)
        if self.idx not in _CB_QUEUED:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
                _CB_QUEUED.add(self.idx)
            except RuntimeError:          # not inside an engine run (a backward() called by hand): order it right here
                join()
        return False


# ---- how often does one graph use a weight? -----------------------------------------------------------------------------
# A gradient computed on the side stream is only safe while autograd merely STORES it.  A weight that feeds two convolution
# calls of one graph gets its two gradients summed in the engine's input buffer on the caller's stream, unordered against the
# side stream.  The convolution wrappers therefore count the uses of a weight in forward (``note_use``) and backward asks
# ``wgrad_on_side``: more than one use since the weight's counter was last at rest -> every one of them stays on the caller's
# stream.  The counter rests again when all counted uses have run backward, at the end of every backward run that consulted it, or at the
# next ``new_step()``.
_EPOCH = 0


def note_use(weight):
    """Forward of a convolution wrapper that may send this weight's gradient to the side stream."""
    if not WGRAD_STREAM:                  # (callers ask only when this graph needs the weight's gradient)
        return
    st = getattr(weight, "_ssbev_uses", None)
    if st is None or st[0] != _EPOCH:
        st = weight._ssbev_uses = [_EPOCH, 0, False]          # epoch, uses not yet run backward, shared?
    st[1] += 1
    if st[1] > 1:
        st[2] = True


def wgrad_on_side(weight):
    """Should this weight gradient go to the side stream?  (Only when autograd will merely store it, see above.)"""
    # a LEAF parameter: its gradient goes to AccumulateGrad, which keeps the tensor when .grad is None; the gradient of a
    # computed weight (CA3D folds its channel gate into the weights) is read by the next backward node at once
    shared = False
    if WGRAD_STREAM and not _EPOCH_CB[0]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_epoch_end_of_backward)
            _EPOCH_CB[0] = True
        except RuntimeError:              # a backward() called by hand, outside an engine run: nothing to hang the reset on
            pass
    st = getattr(weight, "_ssbev_uses", None)
    if st is not None:
        shared = st[2]
        st[1] -= 1
        if st[1] <= 0:
            st[1], st[2] = 0, False
    # tensor hooks on the weight read the gradient on the caller's stream before AccumulateGrad sees it
    hooked = bool(getattr(weight, "_backward_hooks", None))
    return WGRAD_STREAM and weight.is_cuda and weight.is_leaf and weight.grad is None and not shared and not hooked
