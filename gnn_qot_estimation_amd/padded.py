"""Any ``hidden_channels``: widths the kernels are not instantiated for run on the next supported width with
zero-padded parameters.

The HIP kernels exist for hidden widths 16 / 32 / 64 / 128 / 256 (per-head GAT widths 4 ... 256); the reference's
constructors take any ``hidden_channels`` (``topological_training/models.py:7-9``, ``lightpath_training/models.py:8-10``).
A model built with another width keeps its parameters, buffers and ``state_dict`` exactly as the reference shapes them;
its forward runs a SHADOW model of the padded width through ``torch.func.functional_call`` with parameters padded on the
fly (``F.pad`` is differentiable: gradients arrive in the true shapes).  Padded channels carry exact zeros through every
layer -- zero rows / columns of the weights, zero biases, ``leaky_relu(0) = relu(0) = 0``, BatchNorm of a constant-zero
channel with zero bias is zero -- so the first ``H`` channels are the unpadded model's.  Two places need more than zeros:
TransformerConv scales its logits by ``1/sqrt(out_channels)``, so the padded query projection is pre-scaled by
``sqrt(H_pad / H)``; BatchNorm running statistics are copied back into the true buffers after a training forward.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

WIDTHS = (16, 32, 64, 128, 256)
GAT_WIDTHS = (4, 8, 16, 32, 64, 128, 256)


def padded_width(h: int, allowed=WIDTHS) -> int:
    for w in allowed:
        if h <= w:
            return w
    raise ValueError(f"hidden width {h} exceeds the largest supported width {allowed[-1]}")


def _pad_last(t, new):           # zero-pad the last dimension to `new`
    return F.pad(t, (0, new - t.shape[-1]))


def _pad2(t, rows, cols):        # [r, c] -> [rows, cols]
    return F.pad(t, (0, cols - t.shape[1], 0, rows - t.shape[0]))


def topological_params(model, hp: int) -> Dict[str, torch.Tensor]:
    """Parameter dictionary of a width-``hp`` TopologicalGNN from a width-``h`` one (keys: App. A)."""
    sd = dict(model.named_parameters())
    h = model.node_embeddings.embedding_dim
    out = {}
    qscale = math.sqrt(hp / h)
    for name, p in sd.items():
        if name == "node_embeddings.weight":
            out[name] = _pad_last(p, hp)
        elif name.startswith("conv1.lin_") and name.endswith(".weight"):
            if "lin_edge" in name:
                q = F.pad(p, (0, 0, 0, hp - h))                       # [H, D] -> [Hp, D]
            else:
                q = _pad2(p, hp, hp)
            out[name] = q * qscale if "lin_query" in name else q
        elif name.startswith("conv1.lin_") and name.endswith(".bias"):
            q = _pad_last(p, hp)
            out[name] = q * qscale if "lin_query" in name else q
        elif ".nn.0." in name:                                        # Linear(D, 2D): unchanged
            out[name] = p
        elif name.endswith(".nn.2.weight"):                           # [H*H, K], flat index a*H + o
            k = p.shape[1]
            out[name] = F.pad(p.view(h, h, k), (0, 0, 0, hp - h, 0, hp - h)).reshape(hp * hp, k)
        elif name.endswith(".nn.2.bias"):
            out[name] = _pad2(p.view(h, h), hp, hp).reshape(hp * hp)
        elif name.endswith(".lin.weight") or name == "mlp.0.weight":  # [H, H]
            out[name] = _pad2(p, hp, hp)
        elif name == "mlp.3.weight":                                  # [out, H]
            out[name] = _pad_last(p, hp)
        elif name == "mlp.3.bias":
            out[name] = p
        else:                                                         # conv biases, mlp.0.bias: [H]
            out[name] = _pad_last(p, hp)
    return out


def _pad_heads(t, c, cp, heads=4):      # [..., heads*c] -> [..., heads*cp], each head padded separately
    lead = t.shape[:-1]
    return F.pad(t.reshape(*lead, heads, c), (0, cp - c)).reshape(*lead, heads * cp)


def lightpath_params(model, cp: int):
    """(parameters, buffers) of a per-head-width-``cp`` LightpathGNN from a width-``c`` one."""
    c = model.conv1.out_channels
    params, buffers = {}, {}
    for name, p in model.named_parameters():
        if name.endswith(".att_src") or name.endswith(".att_dst"):     # [1, 4, C]
            params[name] = _pad_last(p, cp)
        elif name.endswith(".lin.weight"):                             # [4C, in]; in = F (layer 1) or 4C
            w = _pad_heads(p.t(), c, cp).t()                           # rows per head
            params[name] = w if name.startswith("conv1.") else _pad_heads(w, c, cp)
        elif name == "mlp.0.weight":                                   # [C, 4C]
            params[name] = F.pad(_pad_heads(p, c, cp), (0, 0, 0, cp - c))
        elif name == "mlp.0.bias":
            params[name] = _pad_last(p, cp)
        elif name == "mlp.3.weight":                                   # [out, C]
            params[name] = _pad_last(p, cp)
        elif name == "mlp.3.bias":
            params[name] = p
        else:                                                          # conv bias, BatchNorm weight / bias: [4C]
            params[name] = _pad_heads(p, c, cp)
    for name, b in model.named_buffers():
        if name.endswith("num_batches_tracked"):
            buffers[name] = b
        elif name.endswith("running_var"):
            buffers[name] = _pad_heads(b - 1.0, c, cp) + 1.0           # padded channels: variance 1
        elif name.endswith("running_mean"):
            buffers[name] = _pad_heads(b, c, cp)
    return params, buffers


def lightpath_copy_back(model, buffers, c: int, cp: int, heads: int = 4):
    """Running statistics of the true channels out of the padded buffers (after a training forward)."""
    with torch.no_grad():
        for name, b in model.named_buffers():
            if name.endswith("running_mean") or name.endswith("running_var"):
                b.copy_(buffers[name].view(heads, cp)[:, :c].reshape(heads * c))
