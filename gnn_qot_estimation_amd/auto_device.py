"""``QOT_AUTO_DEVICE=1``: run a model that the caller left on the CPU through the HIP engine anyway.

``topological_training/train.py:62`` of the reference pins ``device = torch.device("cpu")`` and then does
``model.to(device)`` / ``data.to(device)`` (``:63,108``): with that script UNCHANGED the modules of this repo would see
CPU parameters and a CPU batch, and the HIP path has no CPU fallback (it raises ``QotError``).  BASELINE.json's
``configs[0]`` ("train.py on CPU ... plumbing") is exactly that situation.  Two documented behaviours:

* default: the loud failure (``QotError`` naming this switch and the one line of ``train.py`` that pins the CPU);
* opt-in, ``QOT_AUTO_DEVICE=1`` in the environment: ``forward`` keeps the caller's parameters where they are (so
  ``torch.optim.SGD(model.parameters())``, ``state_dict()``, ``torch.save`` behave as on any CPU model), uploads them
  and the batch to ``cuda:current`` (``Tensor.to`` is differentiable: gradients arrive back on the CPU leaves), runs a
  device-resident shadow copy of the module through ``torch.func.functional_call`` and returns the outputs on the
  caller's device.  BatchNorm running statistics are copied back after a training forward.  The compute still runs on
  the HIP kernels only -- nothing here computes on the CPU; the switch costs one parameter upload and one gradient
  download per step (21 KB at the reference's own width), which is plumbing, not a fast path.

The uploaded batch is cached on the batch object (the reference's loop hands the same object back from
``data.to("cpu")``), keyed by the identity / version of its tensors.
"""
from __future__ import annotations

import copy
import os

import torch

ENV = "QOT_AUTO_DEVICE"
_SKIP_BUFFERS = ("_qot_step",)          # device-side dropout counter: the shadow keeps its own


def enabled() -> bool:
    return os.environ.get(ENV, "0") == "1"


def cpu_model_error():
    from . import _lib
    return _lib.QotError(
        "model parameters are on the CPU: the HIP message-passing path has no CPU fallback.  Either move the model and "
        "the batch to the GPU (the reference's topological_training/train.py:62 pins torch.device('cpu'); its other "
        "scripts already select cuda), or set QOT_AUTO_DEVICE=1 to keep the script unchanged: forward() then uploads "
        "parameters and batch, runs on the GPU and returns CPU outputs (gnn_qot_estimation_amd/auto_device.py).")


def _gpu_batch(data, dev):
    tensors = [getattr(data, k, None) for k in ("x", "edge_index", "edge_attr", "batch", "node_ids", "ptr")]
    tag = tuple((t.data_ptr(), t._version, tuple(t.shape)) if isinstance(t, torch.Tensor) else None for t in tensors)
    cached = getattr(data, "_qot_gpu_batch", None)
    if cached is not None and cached[0] == tag and cached[1] == dev:
        return cached[2]
    moved = data.to(dev)
    if moved is data:                   # a container whose .to() works in place: keep the caller's object on its device
        raise RuntimeError("QOT_AUTO_DEVICE: data.to(device) moved the caller's batch in place")
    try:
        data._qot_gpu_batch = (tag, dev, moved)
    except Exception:
        pass
    return moved


def forward(model, data):
    """``model(data)`` for a CPU-resident ``model`` / ``data`` on the GPU engine; see the module docstring."""
    if not torch.cuda.is_available():
        from . import _lib
        raise _lib.QotError("QOT_AUTO_DEVICE=1 but no GPU is visible: the HIP path has no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    holder = model.__dict__.get("_qot_gpu_shadow")
    if holder is None:
        shadow = copy.deepcopy(model)
        shadow.__dict__.pop("_qot_gpu_shadow", None)
        shadow.to(dev)
        for p in shadow.parameters():
            p.requires_grad_(False)
        holder = (shadow,)              # in a tuple: not a registered submodule, state_dict unchanged
        model.__dict__["_qot_gpu_shadow"] = holder
    shadow = holder[0]
    shadow.train(model.training)
    for m_src, m_dst in zip(model.modules(), shadow.modules()):
        if isinstance(m_src, torch.nn.Dropout):
            m_dst.p = m_src.p
    for attr in ("allow_empty_lut", "is_lut_index", "_qot_fold_head"):
        if hasattr(model, attr):
            setattr(shadow, attr, getattr(model, attr))
    params = {k: p.to(dev) for k, p in model.named_parameters()}
    buffers = {k: b.to(dev) for k, b in model.named_buffers() if k.rsplit(".", 1)[-1] not in _SKIP_BUFFERS}
    out = torch.func.functional_call(shadow, {**params, **buffers}, (_gpu_batch(data, dev),))
    if model.training and buffers:
        with torch.no_grad():
            own = dict(model.named_buffers())
            for k, b in buffers.items():
                own[k].copy_(b)
    home = next(model.parameters()).device
    if isinstance(out, tuple):
        return tuple(o.to(home) for o in out)
    return out.to(home)
