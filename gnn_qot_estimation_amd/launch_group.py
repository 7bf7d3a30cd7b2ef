"""Host side of the multi-role launch (``qot_run_roles``, ``csrc/roles.hip``): independent small jobs of a train step
share ONE kernel launch instead of costing 4-9 us of launch latency each.

Two users:

* ``LaunchGroup`` -- an explicit group for the FORWARD prologue of ``TopologicalGNN`` (graph index of the batch +
  embedding-table projection + NNConv operand packing: none reads what another writes).  Callers register jobs whose
  outputs they have already allocated and call ``run()`` before the first consumer.
* the BACKWARD epilogue queue (``defer`` / ``flush``): the second-stage sums that autograd Functions used to launch one by
  one from inside their ``backward`` (read-out partials, NNConv slabs, ``lin_edge`` partials, table-gradient row sum) are
  registered instead, with their outputs allocated but not yet filled, and launched together -- by
  ``TableProjectFn.backward`` (table mode: it is the last node and the only in-graph consumer of a deferred result) or
  by an end-of-backward callback of the autograd engine.  Stage-2 jobs run in a second launch behind stage 1 (a job may
  read what stage 1 wrote).  Deferred outputs are leaf gradients (or feed ``TableProjectFn``), so nothing reads them
  before the flush; what is returned to autograd are VIEWS of buffers the queue keeps alive, so ``AccumulateGrad``
  still takes them without a copy.

``QOT_NO_LAUNCH_GROUPS=1`` restores one launch per job (A/B measurements, bisecting).

Threading: the backward queue is ONE module-level object armed through an autograd-engine callback -- right for the
single-threaded step the reference runs (``topological_training/train.py:107-116``), NOT for two backward passes in flight
on different host threads (their jobs would share a queue and a stream).  Such callers set ``QOT_NO_LAUNCH_GROUPS=1``
("Threads").  A deferred gradient is only handed out when nothing can read it before the flush:
``can_defer`` refuses receivers with an existing ``.grad``, tensor hooks, post-accumulate hooks, a non-contiguous layout, or
one Parameter appearing twice.
"""
from __future__ import annotations

import os

import torch

from . import _lib


def enabled() -> bool:
    return os.environ.get("QOT_NO_LAUNCH_GROUPS", "0") != "1"


def can_defer(*receivers) -> bool:
    """A backward-epilogue job may be deferred only if nothing reads its output before the flush.  Its outputs are
    gradients handed to autograd: for a LEAF whose ``.grad`` is ``None`` the engine just keeps the tensor (no kernel),
    but an existing ``.grad`` is accumulated into AT ONCE (an add kernel that would read the still unfilled buffer),
    and a non-leaf input (``F.pad`` of a parameter, a view, ...) hands the gradient to the next backward node at once.
    ``receivers``: the forward inputs that receive the deferred gradients (``None`` entries are ignored)."""
    if not enabled():
        return False
    for t in receivers:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_leaf or t.grad is not None:
            return False
        # a tensor hook / post-accumulate-grad hook runs on the gradient AT ONCE (on the still unfilled buffer), and a
        # receiver that is not contiguous makes AccumulateGrad clone the buffer at once (gradient layout contract)
        if getattr(t, "_backward_hooks", None) or getattr(t, "_post_accumulate_grad_hooks", None) or not t.is_contiguous():
            return False
    # one Parameter used by two nodes (tied weights): the engine's InputBuffer ADDS the two gradients as they arrive
    ids = [id(t) for t in receivers if t is not None]
    if len(ids) != len(set(ids)):
        return False
    return not torch.is_grad_enabled()   # create_graph=True: AccumulateGrad clones what it is given


class LaunchGroup:
    """Jobs launched together by ``run()``; ``post`` callables run after the launch (e.g. a status read-back)."""

    def __init__(self):
        self.roles, self.keep, self.post = [], [], []
        self.on_success = []         # run after the launch AND the post checks went through (e.g. cache insertions)

    def add(self, kind: int, ptrs, ints, keep=()):
        self.roles.append(_lib.make_role(kind, ptrs, ints))
        self.keep.extend(t for t in ptrs if isinstance(t, torch.Tensor))
        self.keep.extend(keep)

    def run(self):
        roles, post, done = self.roles, self.post, self.on_success
        self.roles, self.post, self.on_success = [], [], []
        try:
            _lib.run_roles(roles)
        finally:
            self.keep = []
        for fn in post:
            fn()                     # may raise (status read-back): nothing below runs then
        for fn in done:
            fn()


class _BackwardQueue:
    def __init__(self):
        self.clear()

    def pending(self) -> bool:
        return bool(self.stages[0] or self.stages[1] or self.forked)

    def clear(self):
        self.stages = ([], [])
        self.keep = []
        self.stream = None           # raw handle of the stream the backward nodes run on
        self.stream_obj = None
        self.forked = []             # side streams with work the epilogue has to wait for
        self.armed = False


_Q = _BackwardQueue()


def _arm():
    q = _Q
    if not q.armed:
        q.clear()                        # anything left over belongs to a backward pass that died: its buffers are gone
        q.stream = _lib.stream()         # node execution runs on the forward's stream; the callback may not
        q.stream_obj = torch.cuda.current_stream()
        torch.autograd.Variable._execution_engine.queue_callback(flush)
        q.armed = True
    return q


_SIDE = {}


def fork(fn, keep=()):
    """Run ``fn()`` (kernel launches through ``_lib.call``) on a side stream, ordered behind everything enqueued so far on
    the current stream, and let the current stream go on; the backward epilogue waits for it.  For a kernel whose result
    only the epilogue consumes and that uses other hardware than what follows it on the main stream (NNConv's grad-h
    kernel -- matrix cores, LDS -- next to the latency-bound TransformerConv backward kernels): the two then share the
    CUs instead of running back to back.  ``keep``: every tensor the forked work touches -- the caching allocator must
    not hand their blocks to main-stream allocations before the join.  Inside a backward pass only; capture-safe (the
    side stream joins the capture at the fork and leaves it at the flush)."""
    q = _arm()
    cur = torch.cuda.current_stream()
    key = (cur.device.index, cur.cuda_stream)
    side = _SIDE.get(key)
    if side is None:
        side = _SIDE[key] = torch.cuda.Stream(device=cur.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        fn()
    if side not in q.forked:
        q.forked.append(side)
    q.keep.extend(keep)


def defer(kind: int, ptrs, ints, stage: int = 1, keep=()):
    """Register a backward-epilogue job (``stage`` 1 or 2).  Only valid inside an autograd backward pass: the first job
    of a pass installs the engine's end-of-backward callback that launches whatever is still pending."""
    q = _arm()
    q.stages[stage - 1].append(_lib.make_role(kind, ptrs, ints))
    q.keep.extend(t for t in ptrs if isinstance(t, torch.Tensor))
    q.keep.extend(keep)


def flush():
    """Launch the pending jobs: one launch for stage 1, one for stage 2 (if any)."""
    q = _Q
    if not q.pending():
        q.armed = False
        return
    s1, s2 = q.stages
    stream, forked = q.stream, q.forked
    q.stages, q.forked = ([], []), []
    try:
        for side in forked:              # join: the jobs below read what the forked kernels wrote
            q.stream_obj.wait_stream(side)
        _lib.run_roles(s1, stream)
        _lib.run_roles(s2, stream)
    finally:
        q.keep = []
        q.armed = False
        q.stream = q.stream_obj = None


def drop_stale():
    """Called where no backward can be in flight (start of a forward): jobs still queued were registered by a backward
    pass that raised before its callback ran; their buffers may be gone -- forget them."""
    if _Q.pending() or _Q.armed:
        gid = getattr(torch._C, "_current_graph_task_id", None)
        if gid is not None and gid() != -1:
            return                       # a forward recomputed INSIDE a backward pass (checkpointing): the queue is live
        _Q.clear()
