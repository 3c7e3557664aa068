"""Side stream for weight gradients.

In a backward pass the data gradient of a layer is on the critical path (the next node needs it), its weight gradient
is not: nobody reads ``dW`` before the parameter-server pipeline packs the bucket.  The convolution and linear nodes
(``DTF_WGRAD_STREAM=0`` switches it off) launch their weight-gradient kernels on this second stream, so a compute-bound weight gradient
(K14 / K6T-conv: LDS + MFMA) can share the chip with the memory-bound BatchNorm backward passes of the next node instead
of running in series with them.  Consumers order themselves after it: ``ShardedPS`` makes its comm stream wait for this
stream too (its hooks fire in the middle of backward), and an autograd-engine callback makes the stream that ran backward
rejoin it when the backward pass ends -- whoever reads ``p.grad`` after ``loss.backward()`` sees finished gradients.
Inputs that the allocator might hand out again while the side stream still reads them are kept alive until that
rejoin (a reference in a list: no ``record_stream`` bookkeeping on the host -- 288 GB of HBM make holding one backward
pass worth of gradient tensors a non-issue, and ResNet-152's step is bound by host enqueue time).
"""
from __future__ import annotations

import os

import torch

ENABLED = os.environ.get("DTF_WGRAD_STREAM", "1") != "0"
_DEBUG = int(os.environ.get("DTF_WGRAD_DEBUG", "0"))      # bisecting aid: 1 fresh event per call, 2 rejoin after every call
_PRIORITY = int(os.environ.get("DTF_WGRAD_PRIORITY", "0"))
_streams = {}
_join_queued = set()
_held = {}          # (device index, stream) -> tensors the side stream may still be reading
_events = {}
_checks = []


def wgrad_stream(device, create=True):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    s = _streams.get(key)
    if s is None and create:
        # DTF_WGRAD_PRIORITY: stream priority of the side stream (torch: lower number = served first; 0 = like the compute
        # stream).  A positive value asks the hardware scheduler to give free CUs to the compute stream's kernels first.
        s = _streams[key] = torch.cuda.Stream(device=device, priority=_PRIORITY)
    return s


def join_into(stream, device):
    """``stream`` waits for everything launched on the weight-gradient stream so far (no-op if it was never used)."""
    s = wgrad_stream(device, create=False)
    if s is not None:
        stream.wait_stream(s)


layout_copies = 0        # how often a weight gradient came back in a foreign layout (diagnostics / tests)


def _in_param_layout(out, weight):
    """AccumulateGrad keeps a gradient as it is only when its strides are the parameter's; otherwise it stores a CLONE --
    made on the stream backward runs on, which has not waited for this one: the clone would read memory the weight-gradient
    kernel has not written yet (found with K21, whose result was a strided view: the first step's gradient of
    Inception-v3's stem layer came out as zeros, later steps "worked" because the allocator handed back the block that
    still held the previous step's values).  So the copy into the parameter's layout, where one is needed, is made
    here, on the side stream."""
    if (isinstance(out, torch.Tensor) and weight is not None and out.shape == weight.shape and out.stride() != weight.stride()):
        global layout_copies
        layout_copies += 1
        return torch.empty_like(weight, dtype=out.dtype).copy_(out)
    return out


def run_wgrad(fn, weight, *inputs):
    """``fn()`` -- the weight gradient of ``weight`` -- on the weight-gradient stream, ordered after the current stream;
    returns its result.

    Only when the result goes STRAIGHT to the parameter: ``weight`` must be a leaf whose ``.grad`` is still empty.  A
    derived weight (``F.pad(lin.weight, ...)``, a slice of a packed tensor) hands the gradient to further autograd nodes
    that run on the current stream right away, and a second contribution to a shared parameter is ADDED to the first
    on the current stream -- both would read a gradient that is still being computed over here.  Those cases run on
    the current stream, after it has caught up with this one."""
    if not ENABLED or not inputs or not inputs[0].is_cuda:
        return fn()
    dev = inputs[0].device
    cur = torch.cuda.current_stream(dev)
    if weight is None or not weight.is_leaf or weight.grad is not None:
        join_into(cur, dev)
        return fn()
    side = wgrad_stream(dev)
    key = (dev.index, cur.cuda_stream)
    # host cost matters (ResNet-152's step is ~95 % host enqueue time): one reused event instead of the fresh one
    # ``wait_stream`` creates, ``set_stream`` instead of the ``torch.cuda.stream`` context manager
    if _DEBUG == 1:
        side.wait_stream(cur)               # a fresh event per call
    else:
        ev = _events.get(key)
        if ev is None:
            ev = _events[key] = torch.cuda.Event()
        ev.record(cur)
        side.wait_event(ev)
    if key not in _join_queued:
        # once per backward pass: when the engine is done, the stream backward ran on waits for the side stream
        _join_queued.add(key)

        def _rejoin(cur=cur, side=side, key=key):
            _join_queued.discard(key)
            cur.wait_stream(side)
            if _DEBUG == 3:
                torch.cuda.synchronize()
                for copies, orig, shapes in _checks:
                    for c, o, sh in zip(copies, orig, shapes):
                        if not torch.equal(c, o):
                            print("[wgrad debug] operand %s changed during backward (%.3g of its elements)" % (
                                tuple(sh), float((c != o).float().mean())), flush=True)
                _checks.clear()
            _held.pop(key, None)        # freed in stream order AFTER the wait: safe for the allocator to reuse

        try:
            torch.autograd.Variable._execution_engine.queue_callback(_rejoin)
        except RuntimeError:        # not inside a backward pass (a direct call of a backward function in a test)
            _join_queued.discard(key)
            torch.cuda.set_stream(side)
            try:
                out = _in_param_layout(fn(), weight)
            finally:
                torch.cuda.set_stream(cur)
            cur.wait_stream(side)
            return out
    if _DEBUG == 3:
        # were the operands still the same when backward ended?  (copies taken on the compute stream first)
        _checks.append((tuple(i.clone() for i in inputs), inputs, tuple(i.shape for i in inputs)))
    torch.cuda.set_stream(side)
    try:
        out = _in_param_layout(fn(), weight)
    finally:
        torch.cuda.set_stream(cur)
    _held.setdefault(key, []).append(inputs)
    if _DEBUG == 2:
        cur.wait_stream(side)               # side stream, but no overlap with what follows
    return out
