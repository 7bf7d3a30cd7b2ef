"""Independent fp64 restatement, per-destination-node loops (no scatter/index_select).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Shares no code with
``oracle.sparse``: every operator walks destination nodes in Python, gathers that node's
in-edge list, and evaluates the SURVEY.md Appendix B formula literally with dense fp64
tensor arithmetic.  Slow by design; used on small graphs only, and as the
``gradcheck`` subject.  Parameters are passed explicitly as a ``state_dict`` so this file
never touches ``oracle.sparse`` module classes.
"""
from __future__ import annotations

import math

import torch


def _in_edges(edge_index: torch.Tensor, n: int):
    lists = [[] for _ in range(n)]
    src = edge_index[0].tolist()
    dst = edge_index[1].tolist()
    for e, (j, i) in enumerate(zip(src, dst)):
        lists[i].append((e, j))
    return lists


def _lin(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def transformer_conv(sd, prefix, x, edge_index, edge_attr):
    """App. B.1 evaluated node by node."""
    g = lambda k: sd[prefix + k].double()
    x = x.double()
    ea = edge_attr.double()
    q = _lin(x, g("lin_query.weight"), g("lin_query.bias"))
    k = _lin(x, g("lin_key.weight"), g("lin_key.bias"))
    v = _lin(x, g("lin_value.weight"), g("lin_value.bias"))
    skip = _lin(x, g("lin_skip.weight"), g("lin_skip.bias"))
    we = g("lin_edge.weight")
    c = q.shape[1]
    rows = []
    for i, lst in enumerate(_in_edges(edge_index, x.shape[0])):
        if not lst:
            rows.append(skip[i])
            continue
        eps = torch.stack([we @ ea[e] for e, _ in lst])          # [deg, C]
        kj = torch.stack([k[j] for _, j in lst]) + eps
        vj = torch.stack([v[j] for _, j in lst]) + eps
        s = (kj @ q[i]) / math.sqrt(c)
        p = torch.exp(s - s.detach().max())
        alpha = p / (p.sum() + 1e-16)
        rows.append(alpha @ vj + skip[i])
    return torch.stack(rows)


def nn_conv(sd, prefix, x, edge_index, edge_attr):
    """App. B.2 evaluated node by node with the full ``[H_in, H_out]`` matrix per edge."""
    g = lambda k: sd[prefix + k].double()
    x = x.double()
    ea = edge_attr.double()
    w1, b1, w2, b2 = g("nn.0.weight"), g("nn.0.bias"), g("nn.2.weight"), g("nn.2.bias")
    wroot, bias = g("lin.weight"), g("bias")
    hin = x.shape[1]
    hout = wroot.shape[0]
    rows = []
    for i, lst in enumerate(_in_edges(edge_index, x.shape[0])):
        acc = torch.zeros(hout, dtype=torch.float64)
        for e, j in lst:
            h = torch.relu(w1 @ ea[e] + b1)
            theta = (w2 @ h + b2).view(hin, hout)
            acc = acc + x[j] @ theta
        rows.append(acc / max(len(lst), 1) + wroot @ x[i] + bias)
    return torch.stack(rows)


def gat_conv(sd, prefix, x, edge_index, heads=4):
    """App. B.3 evaluated node by node; self-loop handling restated from the definition
    (drop existing ``j == i`` edges, then every node gets exactly one self edge, last)."""
    g = lambda k: sd[prefix + k].double()
    x = x.double()
    n = x.shape[0]
    w, att_s, att_d, bias = g("lin.weight"), g("att_src")[0], g("att_dst")[0], g("bias")
    c = w.shape[0] // heads
    z = (x @ w.t()).view(n, heads, c)
    rows = []
    for i, lst in enumerate(_in_edges(edge_index, n)):
        nbrs = [j for _, j in lst if j != i] + [i]
        zi = z[i]
        a_d = (zi * att_d).sum(-1)                                 # [heads]
        zj = torch.stack([z[j] for j in nbrs])                     # [deg, heads, C]
        a_s = (zj * att_s).sum(-1)                                 # [deg, heads]
        s = a_s + a_d
        s = torch.where(s > 0, s, 0.2 * s)
        p = torch.exp(s - s.detach().max(dim=0).values)
        alpha = p / (p.sum(0) + 1e-16)
        rows.append((alpha.unsqueeze(-1) * zj).sum(0).reshape(-1) + bias)
    return torch.stack(rows)


def batch_norm_eval(sd, prefix, x):
    g = lambda k: sd[prefix + k].double()
    return (x.double() - g("running_mean")) / torch.sqrt(g("running_var") + 1e-5) * g("weight") + g("bias")


def batch_norm_train(sd, prefix, x):
    """Returns (y, new_running_mean, new_running_var) -- App. B.4."""
    g = lambda k: sd[prefix + k].double()
    x = x.double()
    n = x.shape[0]
    mean = x.mean(0)
    var_b = ((x - mean) ** 2).mean(0)
    y = (x - mean) / torch.sqrt(var_b + 1e-5) * g("weight") + g("bias")
    var_u = var_b * n / max(n - 1, 1)
    return y, 0.9 * g("running_mean") + 0.1 * mean, 0.9 * g("running_var") + 0.1 * var_u


def mean_pool(x, batch, num_graphs):
    rows = []
    b = batch.tolist()
    for gidx in range(num_graphs):
        members = [n for n, bb in enumerate(b) if bb == gidx]
        if members:
            rows.append(torch.stack([x[n] for n in members]).mean(0))
        else:
            rows.append(torch.zeros(x.shape[1], dtype=x.dtype))
    return torch.stack(rows)


def _leaky(x, slope=0.01):
    return torch.where(x > 0, x, slope * x)


def _head(sd, x):
    g = lambda k: sd[k].double()
    h = _leaky(_lin(x, g("mlp.0.weight"), g("mlp.0.bias")))
    return _lin(h, g("mlp.3.weight"), g("mlp.3.bias"))


def topological_forward(sd, data, num_graphs=None, extra_nnconv=()):
    """Eval-mode ``topological_training/models.py:43-64`` (dropout is identity)."""
    if data.x is None or data.x.numel() == 0:
        x = sd["node_embeddings.weight"].double()[data.node_ids]
    else:
        x = data.x.double()
    x = _leaky(transformer_conv(sd, "conv1.", x, data.edge_index, data.edge_attr))
    x = _leaky(nn_conv(sd, "conv2.", x, data.edge_index, data.edge_attr))
    for pfx in extra_nnconv:
        x = _leaky(nn_conv(sd, pfx, x, data.edge_index, data.edge_attr))
    if num_graphs is None:
        num_graphs = int(data.batch.max()) + 1
    return _head(sd, mean_pool(x, data.batch, num_graphs))


def lightpath_forward(sd, data, is_lut_index, train_stats=False, extra_layers=()):
    """``lightpath_training/models.py:26-45``; BN in eval mode unless ``train_stats``."""
    x = data.x
    layers = (("conv1.", "norm1.module."),) + tuple(extra_layers)
    for cpfx, npfx in layers:
        x = gat_conv(sd, cpfx, x, data.edge_index)
        x = batch_norm_train(sd, npfx, x)[0] if train_stats else batch_norm_eval(sd, npfx, x)
        x = torch.relu(x)
    mask = data.x[:, is_lut_index] == 1.0
    if not bool(mask.any()):
        raise ValueError("No LUT node found in the batch.")
    return _head(sd, x[mask]), data.batch[mask]
