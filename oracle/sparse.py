"""Sparse (edge-list) CPU restatement of the PyG operators the reference composes.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Every function cites the reference
call site it stands in for and the SURVEY.md Appendix B definition it follows.  The torch
primitives are the ones PyG's ``MessagePassing.propagate`` / ``utils.softmax`` /
``utils.scatter`` dispatch to when ``torch_scatter`` is absent, so timing this module is
the closest available stand-in for "the reference's CPU PyTorch path" (BASELINE.md section 2).

Module classes carry parameters under the exact ``state_dict`` keys of the shipped
checkpoints, so weights are shared with the HIP path by
``load_state_dict``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------- utils
def scatter_sum(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """PyG ``utils.scatter(reduce='sum')``: zeros + ``scatter_add_`` (App. B preamble)."""
    shape = (dim_size,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    return src.new_zeros(shape).scatter_add_(0, idx, src)


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """PyG ``utils.scatter(reduce='mean')``: sum / clamp(count, min=1)."""
    out = scatter_sum(src, index, dim_size)
    cnt = src.new_zeros(dim_size).scatter_add_(0, index, src.new_ones(index.numel()))
    cnt = cnt.clamp_(min=1).view(-1, *([1] * (src.dim() - 1)))
    return out / cnt


def scatter_amax(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """PyG ``utils.scatter(reduce='max')``: zeros + ``scatter_reduce_('amax', include_self=False)``."""
    shape = (dim_size,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    return src.new_zeros(shape).scatter_reduce_(0, idx, src, "amax", include_self=False)


def segment_softmax(src: torch.Tensor, index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """PyG ``utils.softmax`` (App. B.1/B.3): max on detached values, ``+1e-16`` in the sum."""
    src_max = scatter_amax(src.detach(), index, num_nodes)
    out = (src - src_max.index_select(0, index)).exp()
    out_sum = scatter_sum(out, index, num_nodes) + 1e-16
    return out / out_sum.index_select(0, index)


def global_mean_pool(x: torch.Tensor, batch: torch.Tensor, size: int | None = None) -> torch.Tensor:
    """``topological_training/models.py:61``; App. B.5 (``B = batch.max()+1``)."""
    if size is None:
        size = int(batch.max()) + 1 if batch.numel() else 0
    return scatter_mean(x, batch, size)


def _pyg_linear_reset(lin: nn.Linear) -> None:
    # App. B.7: PyG Linear default = kaiming_uniform(a=sqrt(5)); bias U(+-1/sqrt(fan_in)).
    nn.init.kaiming_uniform_(lin.weight, a=math.sqrt(5))
    if lin.bias is not None:
        bound = 1.0 / math.sqrt(lin.weight.shape[1])
        nn.init.uniform_(lin.bias, -bound, bound)


# --------------------------------------------------------------------------- B.1
class TransformerConv(nn.Module):
    """App. B.1; stands in for ``topological_training/models.py:15-17,53``.

    heads=1, concat=True, beta=False, root_weight=True, attention dropout 0.
    """

    def __init__(self, in_channels: int, out_channels: int, edge_dim: int):
        super().__init__()
        self.in_channels, self.out_channels, self.edge_dim = in_channels, out_channels, edge_dim
        self.lin_key = nn.Linear(in_channels, out_channels)
        self.lin_query = nn.Linear(in_channels, out_channels)
        self.lin_value = nn.Linear(in_channels, out_channels)
        self.lin_edge = nn.Linear(edge_dim, out_channels, bias=False)
        self.lin_skip = nn.Linear(in_channels, out_channels)
        for m in (self.lin_key, self.lin_query, self.lin_value, self.lin_edge, self.lin_skip):
            _pyg_linear_reset(m)

    def forward(self, x, edge_index, edge_attr):
        n = x.shape[0]
        src, dst = edge_index[0], edge_index[1]
        q = self.lin_query(x)
        k = self.lin_key(x)
        v = self.lin_value(x)
        q_i = q.index_select(0, dst)
        k_j = k.index_select(0, src)
        v_j = v.index_select(0, src)
        e = self.lin_edge(edge_attr)
        k_j = k_j + e
        alpha = (q_i * k_j).sum(-1) / math.sqrt(self.out_channels)
        alpha = segment_softmax(alpha, dst, n)
        msg = (v_j + e) * alpha.unsqueeze(-1)
        out = scatter_sum(msg, dst, n)
        return out + self.lin_skip(x)


# --------------------------------------------------------------------------- B.2
class NNConv(nn.Module):
    """App. B.2; stands in for ``topological_training/models.py:20-30,57``.

    aggr="mean", root_weight=True, bias=True.  Materialises ``[E, H_in*H_out]`` exactly as
    PyG's ``message`` does -- that is the reference's dominant CPU cost.
    """

    def __init__(self, in_channels: int, out_channels: int, nn_module: nn.Module):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.nn = nn_module
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        # App. B.7: root weight_initializer='uniform' -> U(+-1/sqrt(fan_in))
        bound = 1.0 / math.sqrt(in_channels)
        nn.init.uniform_(self.lin.weight, -bound, bound)

    def forward(self, x, edge_index, edge_attr):
        n = x.shape[0]
        src, dst = edge_index[0], edge_index[1]
        x_j = x.index_select(0, src)
        theta = self.nn(edge_attr).view(-1, self.in_channels, self.out_channels)
        msg = torch.matmul(x_j.unsqueeze(1), theta).squeeze(1)
        out = scatter_mean(msg, dst, n)
        return out + self.lin(x) + self.bias


# --------------------------------------------------------------------------- B.3
def gat_edge_set(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """``remove_self_loops`` then ``add_self_loops`` (App. B.3): one (n,n) per node, appended."""
    keep = edge_index[0] != edge_index[1]
    ei = edge_index[:, keep]
    loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([ei, torch.stack([loops, loops])], dim=1)


class GATConv(nn.Module):
    """App. B.3; stands in for ``lightpath_training/models.py:13,30``.

    concat=True, negative_slope=0.2, add_self_loops=True, bias=True, dropout 0.
    """

    def __init__(self, in_channels: int, out_channels: int, heads: int = 4):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels))
        nn.init.xavier_uniform_(self.lin.weight)  # glorot (App. B.7)
        for a in (self.att_src, self.att_dst):
            stdv = math.sqrt(6.0 / (a.size(-2) + a.size(-1)))
            nn.init.uniform_(a, -stdv, stdv)

    def forward(self, x, edge_index):
        n, h, c = x.shape[0], self.heads, self.out_channels
        z = self.lin(x).view(n, h, c)
        a_src = (z * self.att_src).sum(-1)
        a_dst = (z * self.att_dst).sum(-1)
        ei = gat_edge_set(edge_index, n)
        src, dst = ei[0], ei[1]
        alpha = a_src.index_select(0, src) + a_dst.index_select(0, dst)
        alpha = F.leaky_relu(alpha, 0.2)
        alpha = segment_softmax(alpha, dst, n)
        msg = z.index_select(0, src) * alpha.unsqueeze(-1)
        out = scatter_sum(msg, dst, n)
        return out.view(n, h * c) + self.bias


# --------------------------------------------------------------------------- B.4
class BatchNorm(nn.Module):
    """App. B.4; PyG ``BatchNorm`` wraps ``BatchNorm1d`` as ``.module`` (key prefix
    ``norm1.module.``); stands in for ``lightpath_training/models.py:14,31``."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.module = nn.BatchNorm1d(in_channels, eps=1e-5, momentum=0.1)

    def forward(self, x):
        return self.module(x)


# --------------------------------------------------------------------------- models
class TopologicalGNN(nn.Module):
    """Restates ``topological_training/models.py:6-64`` over the oracle operators.

    ``num_layers`` mirrors the build extension of ``gnn_qot_estimation_amd/topological.py`` (default 2 =
    the reference): every layer beyond the second is one more ``NNConv`` with its own edge network
    (``conv3.*`` ...), each followed by the same ``leaky_relu`` + ``Dropout`` as ``models.py:58-59``.
    """

    def __init__(self, num_nodes, hidden_channels, out_channels, edge_dim, dropout_p=0.5, num_layers=2):
        super().__init__()
        if num_layers < 2:
            raise ValueError("num_layers >= 2")
        self.node_embeddings = nn.Embedding(num_nodes, hidden_channels)
        self.conv1 = TransformerConv(hidden_channels, hidden_channels, edge_dim=edge_dim)
        for layer in range(2, num_layers + 1):
            edge_nn = nn.Sequential(
                nn.Linear(edge_dim, edge_dim * 2),
                nn.ReLU(),
                nn.Linear(edge_dim * 2, hidden_channels * hidden_channels),
            )
            setattr(self, f"conv{layer}", NNConv(hidden_channels, hidden_channels, edge_nn))
        self.mlp = nn.Sequential(
            nn.Linear(hidden_channels, hidden_channels),
            nn.LeakyReLU(),
            nn.Dropout(p=dropout_p),
            nn.Linear(hidden_channels, out_channels),
        )
        self.dropout = nn.Dropout(p=dropout_p)
        self.num_layers = num_layers

    def forward(self, data):
        x = data.x
        if x is None or x.numel() == 0:
            x = self.node_embeddings(data.node_ids)
        x = self.conv1(x, data.edge_index, data.edge_attr)
        x = self.dropout(F.leaky_relu(x))
        for layer in range(2, self.num_layers + 1):
            x = getattr(self, f"conv{layer}")(x, data.edge_index, data.edge_attr)
            x = self.dropout(F.leaky_relu(x))
        x = global_mean_pool(x, data.batch)
        return self.mlp(x)


class LightpathGNN(nn.Module):
    """Restates ``lightpath_training/models.py:7-45`` over the oracle operators.

    ``num_layers`` mirrors ``gnn_qot_estimation_amd/lightpath.py`` (default 1 = the reference): extra
    ``GATConv(4C, C, heads=4) -> BatchNorm -> relu`` blocks named ``conv2/norm2`` ... in front of the LUT select.
    """

    def __init__(self, in_channels, hidden_channels, output_dim, is_lut_index, dropout_p=0.5, num_layers=1):
        super().__init__()
        if num_layers < 1:
            raise ValueError("num_layers >= 1")
        width = hidden_channels * 4
        for layer in range(1, num_layers + 1):
            setattr(self, f"conv{layer}", GATConv(in_channels if layer == 1 else width, hidden_channels, heads=4))
            setattr(self, f"norm{layer}", BatchNorm(width))
        self.mlp = nn.Sequential(
            nn.Linear(width, hidden_channels),
            nn.LeakyReLU(),
            nn.Dropout(p=dropout_p),
            nn.Linear(hidden_channels, output_dim),
        )
        self.is_lut_index = is_lut_index
        self.num_layers = num_layers

    def forward(self, data):
        x = data.x
        for layer in range(1, self.num_layers + 1):
            x = F.relu(getattr(self, f"norm{layer}")(getattr(self, f"conv{layer}")(x, data.edge_index)))
        lut_mask = data.x[:, self.is_lut_index] == 1.0
        if not lut_mask.any():
            raise ValueError("No LUT node found in the batch.")
        return self.mlp(x[lut_mask]), data.batch[lut_mask]
