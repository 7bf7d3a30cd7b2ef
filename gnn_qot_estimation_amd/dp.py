"""Data-parallel step support: one flat fp32 parameter/gradient buffer, one all-reduce.

The reference is single-process (no ``torch.distributed`` anywhere); graphs
in a batch are independent, so the batch shards by contiguous graph ranges and the only
exchange per step is the gradient sum.  Messages are 21 KB - 5 MB, i.e.
latency-bound on xGMI: a single bucket, one RCCL call, no per-tensor collectives.

``FlatModel`` re-homes every parameter as a view into one contiguous buffer (and every
``.grad`` as a view into a second one) so that
  * zero_grad is one memset,
  * the all-reduce is one collective over the whole gradient,
  * SGD is one fused update over one tensor.
``state_dict`` keys/shapes are untouched (views keep their names and shapes).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


_REDUCE_MODE = {}


def _avg_capable(like: torch.Tensor, group=None) -> int:
    """1 when this rank's backend runs ``ReduceOp.AVG`` on tensors like ``like``: the nccl (= RCCL) backend on device
    tensors (``ncclAvg``, NCCL >= 2.10 -- every RCCL that ships with ROCm 5+); gloo has no AVG.  A LOCAL capability
    test: no collective is issued here.  (Round 2 probed by issuing a real AVG all-reduce inside try/except: a rank
    whose backend rejected the op would have left the accepting ranks waiting in a collective it never joined.)"""
    if not like.is_cuda or not hasattr(dist.ReduceOp, "AVG"):
        return 0
    return 1 if dist.get_backend(group) == "nccl" else 0


def reduce_mode(like: torch.Tensor, group=None) -> str:
    """``"avg"`` when every rank of ``group`` can run ``ReduceOp.AVG`` on tensors like ``like`` (RCCL), else
    ``"sum"`` (sum, then divide).  Decided once per (group, device type): each rank evaluates a local capability flag
    and the flags are combined with ONE MIN all-reduce (an op every backend has), so all ranks take the same branch
    on every later step even if one build differs."""
    key = (id(group) if group is not None else 0, like.device.type)
    mode = _REDUCE_MODE.get(key)
    if mode is None:
        flag = torch.tensor([_avg_capable(like, group)], dtype=torch.int32, device=like.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        mode = _REDUCE_MODE[key] = "avg" if int(flag.item()) == 1 else "sum"
    return mode


class FlatModel:
    def __init__(self, model: torch.nn.Module):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        dev, dt = params[0].device, params[0].dtype
        total = sum(p.numel() for p in params)
        self.flat_param = torch.zeros(total, dtype=dt, device=dev)
        self.flat_grad = torch.zeros(total, dtype=dt, device=dev)
        off = 0
        for p in params:
            n = p.numel()
            self.flat_param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + n].view(p.shape)
            p.grad = self.flat_grad[off:off + n].view(p.shape)
            off += n
        self.params = params
        self.numel = total
        # a single leaf the optimizer sees; shares storage with every parameter view
        self.leaf = torch.nn.Parameter(self.flat_param, requires_grad=True)
        self.leaf.grad = self.flat_grad

    def zero_grad(self):
        self.flat_grad.zero_()

    def detach_grads(self):
        """Drop the ``.grad`` views so autograd ASSIGNS fresh gradient tensors instead of launching
        one accumulate-add kernel per parameter; ``gather_grads`` then packs them with one kernel."""
        for p in self.params:
            p.grad = None

    def gather_grads(self):
        """Pack the per-parameter gradients into the flat buffer (one ``cat`` kernel)."""
        parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params]
        torch.cat(parts, out=self.flat_grad)

    def rebind_grads(self):
        """Autograd may replace ``p.grad`` (e.g. after ``zero_grad(set_to_none=True)``);
        re-attach the views so accumulation lands in the flat buffer again."""
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + off * self.flat_grad.element_size():
                p.grad = self.flat_grad[off:off + n].view(p.shape)
            off += n

    def all_reduce_grads(self, weight: Optional[torch.Tensor] = None, group=None, force: bool = False):
        """Average gradients over ranks.  ``weight`` (0-dim tensor, e.g. the local number of
        loss rows) gives a weighted mean: needed where shards contribute different counts
        (LightpathGNN: n_lut differs per shard, cf. lightpath_training/train.py:133).  Weighting after
        backward is exact only when ranks are independent inside backward; with synchronised
        BatchNorm statistics scale the local loss with ``loss_scale`` instead and average plainly.
        ``force``: issue the collective even in a single-rank group (where it is the identity) -- how the one-GPU box
        exercises the RCCL call itself (``tests/test_gpu_dp.py``)."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size(group) == 1 and not force:
            return
        world = dist.get_world_size(group)
        if weight is None:
            # RCCL averages inside the collective (ncclAvg): no separate division launch.  Whether the
            # group's backend takes the op is decided ONCE per group, collectively (``reduce_mode``), so
            # every rank issues the same collective on every step.
            if reduce_mode(self.flat_grad, group) == "avg":
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.AVG, group=group)
            else:
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=group)
                self.flat_grad.div_(world)
        else:
            w = weight.to(self.flat_grad.dtype).reshape(1)
            buf = torch.cat([self.flat_grad * w, w])
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            self.flat_grad.copy_(buf[:-1] / buf[-1].clamp_min(1e-12))

    def broadcast_params(self, src: int = 0, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.flat_param, src=src, group=group)


class FusedSGD:
    """SGD(lr, momentum) over a ``FlatModel``: ONE kernel updates every parameter
    (``qot_sgd_momentum``; same arithmetic as ``torch.optim.SGD`` without dampening / nesterov /
    weight decay, the reference's optimizer at ``topological_training/train.py:66``)."""

    def __init__(self, flat: "FlatModel", lr: float, momentum: float = 0.0, device_lr: bool = False):
        """``device_lr``: keep the learning rate in device memory (``set_lr`` fills it) so a step captured in
        a HIP graph follows a scheduler without re-capture."""
        self.flat, self.momentum = flat, float(momentum)
        self.buf = torch.zeros_like(flat.flat_param)
        self.steps = 0          # host side; a captured graph bakes in "not the first step"
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=flat.flat_param.device) if device_lr else None
        self.lr = float(lr)

    @property
    def lr(self) -> float:
        return self._lr

    @lr.setter
    def lr(self, value: float):
        self._lr = float(value)
        if self.lr_dev is not None:
            self.lr_dev.fill_(self._lr)

    def step(self, grads=None):
        """One update from ``flat.flat_grad``.  ``grads`` (one tensor or ``None`` per parameter, in
        ``flat.params`` order; ``True`` = each parameter's ``.grad``) folds ``FlatModel.gather_grads`` into the
        update kernel: every gradient is read from its own tensor and the packed copy is still written to
        ``flat.flat_grad`` -- one launch instead of two when no all-reduce sits between backward and update."""
        from . import _lib
        if not self.flat.flat_param.is_cuda:
            raise _lib.QotError("FusedSGD runs on the GPU only (use torch.optim.SGD on CPU)")
        if grads is not None and len(self.flat.params) <= 48:
            import ctypes
            params = self.flat.params
            if grads is True:
                grads = [p.grad for p in params]
            keep, ptrs, offs, off = [], [], [0], 0
            for p, g in zip(params, grads):
                if g is not None:
                    if g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != p.numel() or g.device != p.device:
                        g = g.to(device=p.device, dtype=torch.float32).contiguous()
                    keep.append(g)
                ptrs.append(None if g is None else g.data_ptr())
                off += p.numel()
                offs.append(off)
            n = len(params)
            parr = (ctypes.c_void_p * n)(*ptrs)
            oarr = (ctypes.c_int64 * (n + 1))(*offs)
            _lib.call("qot_sgd_momentum_multi", _lib.ptr(self.flat.flat_param), ctypes.addressof(parr),
                      ctypes.addressof(oarr), n, _lib.ptr(self.flat.flat_grad), _lib.ptr(self.buf), self.flat.numel,
                      self.lr, None if self.lr_dev is None else _lib.ptr(self.lr_dev), self.momentum,
                      int(self.steps == 0))
            self.steps += 1
            return
        if grads is not None:
            if grads is True:
                grads = [p.grad for p in self.flat.params]
            torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1)
                       for g, p in zip(grads, self.flat.params)], out=self.flat.flat_grad)
        if self.lr_dev is not None:
            _lib.call("qot_sgd_momentum_dev", _lib.ptr(self.flat.flat_param), _lib.ptr(self.flat.flat_grad),
                      _lib.ptr(self.buf), self.flat.numel, _lib.ptr(self.lr_dev), self.momentum, int(self.steps == 0))
        else:
            _lib.call("qot_sgd_momentum", _lib.ptr(self.flat.flat_param), _lib.ptr(self.flat.flat_grad),
                      _lib.ptr(self.buf), self.flat.numel, self.lr, self.momentum, int(self.steps == 0))
        self.steps += 1


def loss_scale(n_local: int, device, group=None) -> torch.Tensor:
    """Factor that turns ``mean over the local loss rows`` into this rank's share of the mean over the
    GLOBAL rows when gradients are afterwards averaged over ranks: ``n_local * world / sum(n_local)``.

    Scaling the local loss BEFORE backward (instead of weighting gradients after it) is what stays
    exact when ranks are coupled inside the backward pass (synchronised BatchNorm statistics,
    ``functional.BnFn``): every rank then backpropagates its share of the same global loss.  One scalar
    all-reduce; returns 1 without ``torch.distributed``."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return torch.ones((), dtype=torch.float32, device=device)
    t = torch.tensor([float(n_local)], dtype=torch.float64, device=device)
    tot = t.clone()
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    return (t * dist.get_world_size(group) / tot.clamp_min(1.0)).float().reshape(())


def graph_range(num_graphs: int, rank: int, world: int, costs=None):
    """Contiguous graph range [lo, hi) owned by ``rank``: equal counts, or -- with per-graph ``costs`` (e.g. edges +
    nodes) -- equal work (``batch.balanced_ranges``; SURVEY 8(e) "balanced by sum e")."""
    if costs is not None:
        from .batch import balanced_ranges
        return balanced_ranges(costs, world)[rank]
    return (num_graphs * rank) // world, (num_graphs * (rank + 1)) // world
