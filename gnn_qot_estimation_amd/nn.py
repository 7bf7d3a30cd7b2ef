"""Operator modules with the constructor surface, ``forward`` signatures and ``state_dict``
keys of the torch_geometric classes the reference imports
(``topological_training/models.py:3``: ``global_mean_pool, TransformerConv, NNConv``;
``lightpath_training/models.py:3``: ``GATConv, BatchNorm``) -- SURVEY.md Appendix A/B.

Only the configurations the reference instantiates are implemented (heads=1 concat
TransformerConv with ``edge_dim``; NNConv ``aggr="mean"``; GATConv ``heads=4, concat=True``);
anything else raises ``NotImplementedError`` rather than silently computing something else.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from . import functional as QF
from .graph import GraphIndex, batch_index_for, build_graph_index


def _pyg_linear_init(lin: nn.Linear):
    nn.init.kaiming_uniform_(lin.weight, a=math.sqrt(5))
    if lin.bias is not None:
        bound = 1.0 / math.sqrt(lin.weight.shape[1])
        nn.init.uniform_(lin.bias, -bound, bound)


class TransformerConv(nn.Module):
    """``TransformerConv(in, out, edge_dim=D)`` (heads=1, concat, root_weight, beta=False)."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True,
                 beta: bool = False, dropout: float = 0.0, edge_dim: Optional[int] = None,
                 bias: bool = True, root_weight: bool = True):
        super().__init__()
        if heads != 1 or not concat or beta or dropout != 0.0 or edge_dim is None or not bias or not root_weight:
            raise NotImplementedError("only the reference's TransformerConv(H, H, edge_dim=D) configuration")
        self.in_channels, self.out_channels, self.edge_dim = in_channels, out_channels, edge_dim
        self.lin_key = nn.Linear(in_channels, out_channels)
        self.lin_query = nn.Linear(in_channels, out_channels)
        self.lin_value = nn.Linear(in_channels, out_channels)
        self.lin_edge = nn.Linear(edge_dim, out_channels, bias=False)
        self.lin_skip = nn.Linear(in_channels, out_channels)
        self.reset_parameters()

    def reset_parameters(self):
        for m in (self.lin_key, self.lin_query, self.lin_value, self.lin_edge, self.lin_skip):
            _pyg_linear_init(m)

    def packed_weight(self):
        w = torch.cat([self.lin_query.weight, self.lin_key.weight, self.lin_value.weight, self.lin_skip.weight], 0)
        b = torch.cat([self.lin_query.bias, self.lin_key.bias, self.lin_value.bias, self.lin_skip.bias], 0)
        return w, b

    def forward(self, x, edge_index, edge_attr=None, graph: Optional[GraphIndex] = None, act=None):
        """``act = (slope, p, seed, step_counter)`` fuses ``dropout(leaky_relu(.))`` into the kernel
        epilogue (build extension; ``None`` = the plain PyG operator)."""
        if graph is None:
            graph = build_graph_index(edge_index, x.shape[0])
        w, b = self.packed_weight()
        qkvs = QF.LinearFn.apply(x, w, b)             # one MFMA GEMM for q|k|v|skip
        return QF.TConvFn.apply(qkvs, edge_attr, self.lin_edge.weight, graph, None, act)

    def forward_table(self, table, edge_attr, graph: GraphIndex, maps, act=None, step_pair=None, t4=None):
        """``conv(table[node_ids], ...)`` without materialising per-node inputs: project the
        ``[V, H]`` embedding table once (tiny GEMM, plain autograd) and gather projected rows.
        ``step_pair``: see ``QF.TableProjectFn``."""
        if t4 is None:
            t4 = self.project_table(table, step_pair)
        return QF.TConvFn.apply(t4, edge_attr, self.lin_edge.weight, graph, maps, act)

    def graph_form(self, table, graph: GraphIndex, maps):
        """``(n, B, max_e)`` when this batch runs in the graph form of table mode (``QF.tconv_graph_plan``), else ``None``."""
        return QF.tconv_graph_plan(table, self.out_channels, self.edge_dim, graph, maps)

    def prepare_graph_form(self, table, plan, step_pair=None, group=None):
        """The graph form's table-level inputs ``(t4, M, P)`` from the current parameters; with ``group`` the two jobs join the
        caller's multi-role launch."""
        return QF.tconv_graph_prepare(table, self.lin_query.weight, self.lin_query.bias, self.lin_key.weight,
                                      self.lin_key.bias, self.lin_value.weight, self.lin_value.bias, self.lin_skip.weight,
                                      self.lin_skip.bias, self.lin_edge.weight, plan[0], step_pair, group)

    def forward_table_graph(self, table, edge_attr, graph: GraphIndex, maps, plan, pre, act=None):
        """``conv(table[node_ids], ...)`` in the graph form: whole graphs per workgroup, the score matrix and the value
        table in LDS (``csrc/tconv_graph.hip``)."""
        return QF.TConvGraphFn.apply(table, self.lin_query.weight, self.lin_query.bias, self.lin_key.weight,
                                     self.lin_key.bias, self.lin_value.weight, self.lin_value.bias, self.lin_skip.weight,
                                     self.lin_skip.bias, self.lin_edge.weight, edge_attr, pre, graph, maps, plan, act)

    def project_table(self, table, step_pair=None, group=None):
        """``[q|k|v|skip]`` rows of the embedding table, ``[V, 4H]``; with ``group`` the launch is shared."""
        return QF.TableProjectFn.apply(table, self.lin_query.weight, self.lin_query.bias, self.lin_key.weight,
                                       self.lin_key.bias, self.lin_value.weight, self.lin_value.bias,
                                       self.lin_skip.weight, self.lin_skip.bias, step_pair, group)


class NNConv(nn.Module):
    """``NNConv(in, out, nn=Seq(Linear(D,2D), ReLU, Linear(2D,in*out)), aggr="mean")``."""

    def __init__(self, in_channels: int, out_channels: int, nn: nn.Module, aggr: str = "add",
                 root_weight: bool = True, bias: bool = True):
        super().__init__()
        if aggr != "mean" or not root_weight or not bias:
            raise NotImplementedError("only the reference's NNConv(aggr='mean', root_weight, bias)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.nn = nn
        self.lin = torch.nn.Linear(in_channels, out_channels, bias=False)
        self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.in_channels)
        torch.nn.init.uniform_(self.lin.weight, -bound, bound)
        torch.nn.init.zeros_(self.bias)

    def _edge_mlp(self):
        seq = self.nn
        ok = (isinstance(seq, torch.nn.Sequential) and len(seq) == 3 and isinstance(seq[0], torch.nn.Linear)
              and isinstance(seq[1], torch.nn.ReLU) and isinstance(seq[2], torch.nn.Linear)
              and seq[2].out_features == self.in_channels * self.out_channels
              and seq[0].bias is not None and seq[2].bias is not None)
        if not ok:
            raise NotImplementedError("edge network must be Seq(Linear, ReLU, Linear(., in*out)) as in "
                                      "topological_training/models.py:20-24")
        return seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias

    def prepack(self, group=None):
        """The (Wcat, WcatT, Wk^T) operands of the fused kernels from the current parameters (edge_dim <= 4), or ``None``
        where the operator does not use them; with ``group`` the gather joins the caller's multi-role launch."""
        w1, b1, w2, b2 = self._edge_mlp()
        K, D = w1.shape
        h = self.in_channels
        if D > 4 or K != 2 * D or h != self.out_channels or (h != 64 and h not in QF.GEN_WIDTHS) or not w2.is_cuda:
            return None
        return QF.nnconv_pack(QF._f32c(w2.detach()), QF._f32c(b2.detach()), QF._f32c(self.lin.weight.detach()), h, K, group)

    def forward(self, x, edge_index, edge_attr=None, graph: Optional[GraphIndex] = None, act=None, side=None, packed=None):
        if graph is None:
            graph = build_graph_index(edge_index, x.shape[0])
        w1, b1, w2, b2 = self._edge_mlp()
        return QF.NNConvFn.apply(x, edge_attr, w1, b1, w2, b2, self.lin.weight, self.bias, graph, act, side, packed)


class GATConv(nn.Module):
    """``GATConv(in, out, heads=4, concat=True)`` (negative_slope 0.2, self loops, bias)."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True,
                 negative_slope: float = 0.2, dropout: float = 0.0, add_self_loops: bool = True,
                 edge_dim: Optional[int] = None, bias: bool = True):
        super().__init__()
        if not concat or dropout != 0.0 or not add_self_loops or edge_dim is not None or not bias:
            raise NotImplementedError("only the reference's GATConv(F, C, heads=4, concat=True) configuration")
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.negative_slope = negative_slope
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.empty(heads * out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.lin.weight)
        for a in (self.att_src, self.att_dst):
            stdv = math.sqrt(6.0 / (a.size(-2) + a.size(-1)))
            nn.init.uniform_(a, -stdv, stdv)
        nn.init.zeros_(self.bias)

    def forward(self, x, edge_index, graph: Optional[GraphIndex] = None, bn_stats: bool = False):
        """``bn_stats=True`` (build extension): returns ``(out, partials)`` where ``partials`` are the column partials of
        ``out - bias`` a following ``BatchNorm(..., partials=(partials, conv.bias))`` uses for its batch statistics."""
        n = x.shape[0]
        if self.heads != 4:
            raise NotImplementedError("only the reference's GATConv(F, C, heads=4, concat=True) configuration")
        if graph is None:
            graph = build_graph_index(edge_index, n, gat_self_loops=True)
        if not graph.gat_self_loops:
            raise ValueError("GATConv needs a GraphIndex built with gat_self_loops=True")
        if self.thin_ok(x):
            return self.attend_thin(x, graph, bn_stats)
        return self.attend(self.project(x), graph, bn_stats)

    def project(self, x, with_logits: bool = False):
        """``z = x W^T`` (``lin``, no bias): own MFMA kernel where the inner width allows (``csrc/gemm.hip``), the library
        for the first layer's handful of input features.  ``with_logits``: returns ``(z, logits)`` where ``logits`` is
        ``(a_src, a_dst)`` when the kernel formed them in its epilogue (128 channels per head), else ``None``."""
        logits = None
        if x.is_cuda and x.dtype == torch.float32 and x.shape[0] > 0:
            if QF.gemm_ok(x.shape[1], self.lin.out_features):
                z = QF.GemmFn.apply(x, self.lin.weight)
            elif QF.skinny_ok(x.shape[1], self.lin.out_features) and not x.requires_grad:
                # first layer: F = 5 raw node features
                if with_logits and QF.logits_ok(self.lin.out_features, self.heads) and self.lin.out_features in (128, 256, 512, 1024):
                    z, a_s, a_d = QF.SkinnyLinearFn.apply(x, self.lin.weight, self.att_src, self.att_dst)
                    logits = (a_s, a_d)
                else:
                    z = QF.SkinnyLinearFn.apply(x, self.lin.weight)
            else:
                z = self.lin(x)
        else:
            z = self.lin(x)
        return (z, logits) if with_logits else z

    def thin_ok(self, x) -> bool:
        """The first layer's form: projection inside the attention kernels (``QF.GatThinFn``)."""
        return self.heads == 4 and QF.gat_thin_ok(x, self.heads, self.out_channels)

    def attend_thin(self, x, graph: GraphIndex, bn_stats: bool = False):
        return QF.GatThinFn.apply(x, self.lin.weight, self.att_src, self.att_dst, self.bias, graph, self.negative_slope,
                                  bn_stats)

    def attend(self, z, graph: GraphIndex, bn_stats: bool = False, logits=None):
        # attention logits a[n,h] = <z[n,h,:], att[h,:]> are formed from z inside the operator (one pass over z each
        # way) unless the projection left them behind (``logits``); their gradient returns into grad_z in the source pass
        # of the backward either way
        a_s, a_d = logits if logits is not None else (None, None)
        return QF.GatFn.apply(z, self.att_src, self.att_dst, self.bias, graph, self.negative_slope, bn_stats, a_s, a_d)


class BatchNorm(nn.Module):
    """PyG ``BatchNorm``: wraps ``BatchNorm1d`` as ``.module`` (keys ``<name>.module.*``)."""

    def __init__(self, in_channels: int, eps: float = 1e-5, momentum: float = 0.1, affine: bool = True,
                 track_running_stats: bool = True):
        super().__init__()
        if not affine or not track_running_stats or momentum is None:
            raise NotImplementedError("only affine BatchNorm with running statistics")
        self.module = nn.BatchNorm1d(in_channels, eps=eps, momentum=momentum)
        # one process per GPU: statistics over the global batch, as in the single-process reference
        self.sync_stats = True

    def forward(self, x, relu: bool = False, partials=None):
        """``partials``: ``(column partials, shift)`` handed over by the producer of ``x`` (``GATConv(bn_stats=True)``)."""
        m = self.module
        if self.training:
            m.num_batches_tracked.add_(1)
        if partials is not None and not (self.training and QF._dist_world()[1] == 1):
            partials = None
        return QF.BnFn.apply(x, m.weight, m.bias, m.running_mean, m.running_var, self.training,
                             m.momentum, m.eps, relu, self.sync_stats, partials)

    def rows(self, x, idx32, relu: bool = False, partials=None):
        """``self(x, relu)[idx32]`` for unique rows ``idx32`` with only those rows of the normalised matrix ever formed
        (``QF.BnRowsFn``); statistics and running statistics are those of all rows, as in ``forward``."""
        m = self.module
        if self.training:
            m.num_batches_tracked.add_(1)
        if partials is not None and not (self.training and QF._dist_world()[1] == 1):
            partials = None
        return QF.BnRowsFn.apply(x, m.weight, m.bias, m.running_mean, m.running_var, self.training, m.momentum, m.eps, relu,
                                 self.sync_stats, partials, idx32)

    def project_relu(self, x, lin_weight, partials=None, att=None):
        """``relu(self(x)) @ lin_weight^T`` without materialising the normalised activations (``QF.BnLinearFn``): this
        layer's BatchNorm + ReLU ride in the operand load of the next layer's projection.  ``att = (att_src, att_dst)`` of
        that next GATConv: returns ``(z, (a_src, a_dst))`` with its attention logits from the product's epilogue."""
        m = self.module
        if self.training:
            m.num_batches_tracked.add_(1)
        if partials is not None and not (self.training and QF._dist_world()[1] == 1):
            partials = None
        if att is None:
            return QF.BnLinearFn.apply(x, m.weight, m.bias, m.running_mean, m.running_var, self.training, m.momentum, m.eps,
                                       self.sync_stats, partials, lin_weight)
        z, a_s, a_d = QF.BnLinearFn.apply(x, m.weight, m.bias, m.running_mean, m.running_var, self.training, m.momentum,
                                          m.eps, self.sync_stats, partials, lin_weight, att[0], att[1])
        return z, (a_s, a_d)


def global_mean_pool(x, batch, size: Optional[int] = None, data=None):
    """``global_mean_pool(x, batch)`` -- ``topological_training/models.py:61``.

    With only ``(x, batch)`` the graph count is ``batch.max()+1`` exactly as PyG computes
    it (one device sync); pass ``size`` or the batch object to avoid the sync.
    """
    if data is not None and getattr(data, "batch", None) is batch:
        b32, ptr, B = batch_index_for(data, x.shape[0])
    else:
        class _Tmp:
            pass
        t = _Tmp()
        t.batch = batch
        if size is not None:
            t.num_graphs = int(size)
        b32, ptr, B = batch_index_for(t, x.shape[0])
    return QF.PoolFn.apply(x, b32, ptr, B)
