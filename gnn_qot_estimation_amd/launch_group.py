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
"""
from __future__ import annotations

import os

import torch

from . import _lib


def enabled() -> bool:
    return os.environ.get("QOT_NO_LAUNCH_GROUPS", "0") != "1"


class LaunchGroup:
    """Jobs launched together by ``run()``; ``post`` callables run after the launch (e.g. a status read-back)."""

    def __init__(self):
        self.roles, self.keep, self.post = [], [], []

    def add(self, kind: int, ptrs, ints, keep=()):
        self.roles.append(_lib.make_role(kind, ptrs, ints))
        self.keep.extend(t for t in ptrs if isinstance(t, torch.Tensor))
        self.keep.extend(keep)

    def run(self):
        roles, post = self.roles, self.post
        self.roles, self.post = [], []
        try:
            _lib.run_roles(roles)
        finally:
            self.keep = []
        for fn in post:
            fn()


class _BackwardQueue:
    def __init__(self):
        self.stages = ([], [])
        self.keep = []
        self.stream = None
        self.armed = False

    def pending(self) -> bool:
        return bool(self.stages[0] or self.stages[1])

    def clear(self):
        self.stages = ([], [])
        self.keep = []
        self.stream = None
        self.armed = False


_Q = _BackwardQueue()


def defer(kind: int, ptrs, ints, stage: int = 1, keep=()):
    """Register a backward-epilogue job (``stage`` 1 or 2).  Only valid inside an autograd backward pass: the first job
    of a pass installs the engine's end-of-backward callback that launches whatever is still pending."""
    q = _Q
    if not q.armed:
        q.clear()                        # anything left over belongs to a backward pass that died: its buffers are gone
        q.stream = _lib.stream()         # node execution runs on the forward's stream; the callback may not
        torch.autograd.Variable._execution_engine.queue_callback(flush)
        q.armed = True
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
    stream = q.stream
    q.stages = ([], [])
    try:
        _lib.run_roles(s1, stream)
        _lib.run_roles(s2, stream)
    finally:
        q.keep = []
        q.armed = False
        q.stream = None


def drop_stale():
    """Called where no backward can be in flight (start of a forward): jobs still queued were registered by a backward
    pass that raised before its callback ran; their buffers may be gone -- forget them."""
    if _Q.pending() or _Q.armed:
        gid = getattr(torch._C, "_current_graph_task_id", None)
        if gid is not None and gid() != -1:
            return                       # a forward recomputed INSIDE a backward pass (checkpointing): the queue is live
        _Q.clear()
