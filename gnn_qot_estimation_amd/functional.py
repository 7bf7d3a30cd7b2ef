"""``torch.autograd.Function`` wrappers around the C ABI (``include/qot_gnn.h``).

These own save-for-backward and output allocation (through torch's caching allocator);
the library itself never allocates.  Dense projections stay ``torch`` GEMMs (rocBLAS /
hipBLASLt on the matrix cores) -- the north star reserves MFMA for exactly those.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib, launch_group as LG
from .graph import GraphIndex, require_cuda

P = _lib.ptr


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError("the HIP path computes in fp32; got %s" % t.dtype)
    return t.contiguous()


def _off(t: torch.Tensor, floats: int) -> int:
    return t.data_ptr() + 4 * floats


# ------------------------------------------------------------------ embedding (a1)
class EmbedFn(torch.autograd.Function):
    """``table[node_ids]`` -- ``topological_training/models.py:51-52``."""

    @staticmethod
    def forward(ctx, table, ids32):
        require_cuda(table, ids32)
        table = _f32c(table)
        N, (V, H) = ids32.numel(), table.shape
        out = torch.empty(N, H, dtype=torch.float32, device=table.device)
        _lib.call("qot_embed_fwd", P(table), P(ids32), P(out), N, V, H)
        ctx.save_for_backward(ids32)
        ctx.vh = (V, H)
        return out

    @staticmethod
    def backward(ctx, g):
        (ids32,) = ctx.saved_tensors
        V, H = ctx.vh
        g = _f32c(g)
        gt = torch.zeros(V, H, dtype=torch.float32, device=g.device)
        _lib.call("qot_embed_bwd", P(g), P(ids32), P(gt), ids32.numel(), V, H)
        return gt, None


def _act_args(act):
    """(enabled, slope, p, seed, step_ptr) for the fused leaky_relu+dropout epilogue; ``act`` is
    ``None`` or ``(slope, p, seed, step_counter_or_None)``."""
    if act is None:
        return 0, 0.0, 0.0, 0, None
    slope, p, seed, step = act
    return 1, float(slope), float(p if step is not None else 0.0), int(seed), P(step)


def _act_backward(g, y, act):
    """grad wrt the pre-activation given grad wrt ``y = dropout(leaky_relu(pre))``."""
    slope, p, seed, step = act
    g = _f32c(g)
    gx = torch.empty_like(g)
    _lib.call("qot_act_bwd", P(g), P(y), P(gx), g.numel(), float(slope), float(p if step is not None else 0.0),
              int(seed), P(step))
    return gx


def colsum(x: torch.Tensor) -> torch.Tensor:
    """``x.sum(0)`` for a contiguous fp32 ``[N, C]`` CUDA matrix (bias gradients)."""
    n, c = x.shape
    if n == 0 or c % 4 or c > 1024:
        return x.sum(0)
    x = x.contiguous()
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = torch.empty(_lib.load().qot_colsum_workspace_floats(c), dtype=torch.float32, device=x.device)
    _lib.call("qot_colsum", P(x), c, n, c, P(out), P(ws))
    return out


def act_backward_colsum(g, y, act):
    """``(g', g'.sum(0))`` with ``g' = _act_backward(g, y, act)`` in one launch."""
    slope, p, seed, step = act
    g = _f32c(g)
    n, c = g.shape
    if n == 0 or c % 4 or c > 1024:
        gx = _act_backward(g, y, act)
        return gx, gx.sum(0)
    gx = torch.empty_like(g)
    out = torch.empty(c, dtype=torch.float32, device=g.device)
    ws = torch.empty(_lib.load().qot_colsum_workspace_floats(c), dtype=torch.float32, device=g.device)
    _lib.call("qot_act_bwd_colsum", P(g), P(y), P(gx), n, c, float(slope), float(p if step is not None else 0.0),
              int(seed), P(step), P(out), P(ws))
    return gx, out


# ------------------------------------------------------------------ node-level Linear
class LinearFn(torch.autograd.Function):
    """``x @ W^T + b`` over the node matrix.  Forward / grad_x are library GEMMs (MFMA); the
    weight gradient ``g^T x`` ([out, N] x [N, in], K = N ~ 1e5) goes through ``qot_gemm_tn`` when
    ``in == 64`` -- the library runs that shape at ~12 TFLOP/s."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = gemm_tn(g, x)                  # [out, in] = g^T x
        return gx, gw, colsum(g)


def _small_gemm(a, b, bias, m, n, k, sam, sak, sbk, sbn):
    # latency-bound problems: give every block at most two 32-deep K tiles (planes summed afterwards)
    split = max(1, min(16, (k + 63) // 64)) if m * n <= 256 * 256 else 1
    if split <= 1:
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
        _lib.call("qot_small_gemm", P(a), sam, sak, P(b), sbk, sbn, P(bias), P(out), n, m, n, k, 1)
        return out
    parts = torch.empty(split, m, n, dtype=torch.float32, device=a.device)
    _lib.call("qot_small_gemm", P(a), sam, sak, P(b), sbk, sbn, P(bias), P(parts), n, m, n, k, split)
    if (m * n) % 4:
        return parts.sum(0)
    out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    ws = torch.empty(_lib.load().qot_rowsum_wide_workspace_floats(m * n), dtype=torch.float32, device=a.device)
    _lib.call("qot_rowsum_wide", P(parts), split, m * n, P(out), P(ws))
    return out


class SmallLinearFn(torch.autograd.Function):
    """``x @ W^T + b`` for the small dense layers (head MLP ``topological_training/models.py:33-38``,
    embedding-table projection): one ``qot_small_gemm`` launch each for forward, grad_x, grad_W."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        require_cuda(x, weight, bias)
        x, weight, bias = _f32c(x), _f32c(weight), _f32c(bias)
        m, k = x.shape
        n = weight.shape[0]
        ctx.save_for_backward(x, weight)
        ctx.big = m * n * k >= (1 << 28)        # e.g. 65 536 LUT rows x 512 x 128: a real GEMM (csrc/gemm.hip, else the library)
        ctx.own = ctx.big and gemm_ok(k, n) and gemm_ok(n, k)
        if ctx.own:
            return gemm_nt(x, weight, bias=bias)
        if ctx.big:
            return torch.addmm(bias, x, weight.t())
        return _small_gemm(x, weight, bias, m, n, k, k, 1, 1, k)           # B(k,n) = W[n,k]

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _f32c(g)
        m, k = x.shape
        n = weight.shape[0]
        if ctx.own:
            gx = gemm_nt(g, weight.t().contiguous()) if ctx.needs_input_grad[0] else None
            return gx, gemm_tn_planes(g, x), colsum(g)
        if ctx.big:
            return (g @ weight if ctx.needs_input_grad[0] else None), g.t() @ x, colsum(g)
        gx = _small_gemm(g, weight, None, m, k, n, n, 1, k, 1) if ctx.needs_input_grad[0] else None   # g @ W
        if m >= 4096 and skinny_ok(n, k):
            # a handful of outputs over many rows (LightpathGNN's last Linear: 65 536 LUT rows x 128 -> 1): g^T x is the
            # skinny weight gradient with the operands' roles swapped, (x^T g)[k, n] = gw[n, k]^T -- row-parallel partial
            # sums in a fixed order instead of 16 workgroups walking 4096 rows each (394 -> ~10 us at cfg3)
            nblk = _lib.load().qot_skinny_linear_dw_blocks(m)
            part = torch.empty(nblk, k * n, dtype=torch.float32, device=g.device)
            _lib.call("qot_skinny_linear_dw", P(x), P(g), P(part), m, n, k)
            gwt = torch.empty(k * n, dtype=torch.float32, device=g.device)
            _lib.run_roles([_lib.make_role(_lib.ROLE_SUM_ROWS, (part, gwt), (nblk, k * n, 0))])
            gw = gwt.view(k, n).t().contiguous() if n > 1 else gwt.view(1, k)
        else:
            gw = _small_gemm(g, x, None, n, k, m, 1, n, k, 1)              # g^T @ x: A(i,r) = g[r,i]
        return gx, gw, colsum(g) if m >= 64 else g.sum(0)


class TableProjectFn(torch.autograd.Function):
    """``[q|k|v|skip] = table @ W^T + b`` for the ``[V, H]`` embedding table from the four Linear
    parameters in place (one launch; backward one launch for all nine gradients)."""

    @staticmethod
    def forward(ctx, table, wq, bq, wk, bk, wv, bv, ws, bs, step_pair=None, group=None):
        """``step_pair = (counter, snapshot)``: the launch also advances the dropout step counter (it is the
        forward's first kernel in table mode; see ``qot_table_project_fwd``).  ``group`` (a ``LaunchGroup``): the job
        joins the caller's multi-role launch instead of launching here; the output is valid after ``group.run()``."""
        require_cuda(table, wq, bq, wk, bk, wv, bv, ws, bs)
        table, wq, bq, wk, bk, wv, bv, ws, bs = (_f32c(t) for t in (table, wq, bq, wk, bk, wv, bv, ws, bs))
        V, H = table.shape
        out = torch.empty(V, 4 * H, dtype=torch.float32, device=table.device)
        cnt, snap = step_pair if step_pair is not None else (None, None)
        ctx.heavy = table_gemm_ok(V, H)
        if ctx.heavy:
            # a large table (cfg5: 1000 x 256): one product on the matrix cores instead of the blocked projection kernel
            # (68 -> ~25 us); the dropout counter advances first, as inside that kernel
            if cnt is not None:
                _lib.call("qot_step_advance", P(cnt), P(snap))
            w4 = torch.cat([wq, wk, wv, ws], 0)
            b4 = torch.cat([bq, bk, bv, bs], 0)
            _lib.call("qot_gemm_nt", P(table), H, P(w4), H, P(out), 4 * H, V, 4 * H, H, None, None, P(b4))
        elif group is not None and V > 0:
            group.add(_lib.ROLE_TABLE_PROJECT_FWD, (table, wq, bq, wk, bk, wv, bv, ws, bs, out, cnt, snap), (V, H))
        else:
            _lib.call("qot_table_project_fwd", P(table), P(wq), P(bq), P(wk), P(bk), P(wv), P(bv), P(ws), P(bs), P(out),
                      V, H, P(cnt), P(snap))
        ctx.save_for_backward(table, wq, wk, wv, ws)
        return out

    @staticmethod
    def backward(ctx, g):
        table, wq, wk, wv, ws = ctx.saved_tensors
        g = _f32c(g)
        V, H = table.shape
        dev = g.device
        gt = torch.empty(V * H, dtype=torch.float32, device=dev)
        gw = torch.empty(4 * H * H, dtype=torch.float32, device=dev)
        gb = torch.empty(4 * H, dtype=torch.float32, device=dev)
        if ctx.heavy:
            # large tables: grad_table = g W4 and grad_W = g^T table as split products on the matrix cores, planes summed in
            # order in one launch (cfg5: 161 -> ~45 us for the blocked kernel's job)
            if LG.enabled():
                LG.flush()                                  # g may be a row sum still waiting in the epilogue
            lib = _lib.load()
            w4t = torch.cat([wq.t(), wk.t(), wv.t(), ws.t()], 1).contiguous()          # [H, 4H]
            ks = 8
            planes_t = torch.empty(ks, V * H, dtype=torch.float32, device=dev)
            _lib.call("qot_gemm_nt_planes", P(g), 4 * H, P(w4t), 4 * H, P(planes_t), V, H, 4 * H, ks)
            splits = lib.qot_gemm_tn_splits(4 * H, H, V)
            planes_w = torch.empty(splits, 4 * H * H, dtype=torch.float32, device=dev)
            _lib.call("qot_gemm_tn_planes", P(g), 4 * H, P(table), H, P(planes_w), 4 * H, H, V, splits, None, None)
            _lib.run_roles([_lib.make_role(_lib.ROLE_SUM_ROWS, (planes_t, gt), (ks, V * H, 0)),
                            _lib.make_role(_lib.ROLE_SUM_ROWS, (planes_w, gw), (splits, 4 * H * H, 0))])
            gb = colsum(g)
        elif LG.enabled():
            # ``g`` (the table gradient) may itself be a deferred row sum of the TransformerConv backward: this job
            # rides in stage 2 of the backward epilogue, and the epilogue is launched now -- in table mode this node
            # is the last one of the pass
            LG.defer(_lib.ROLE_TABLE_PROJECT_BWD, (g, table, wq, wk, wv, ws, gt, gw, gb), (V, H), stage=2)
            LG.flush()
        else:
            _lib.call("qot_table_project_bwd", P(g), P(table), P(wq), P(wk), P(wv), P(ws), P(gt), P(gw), P(gb), V, H)
        gw2 = gw.view(4 * H, H)
        return (gt.view(V, H), gw2[:H], gb[:H], gw2[H:2 * H], gb[H:2 * H], gw2[2 * H:3 * H], gb[2 * H:3 * H], gw2[3 * H:],
                gb[3 * H:], None, None)


_LOSS_WS = {}


def smooth_l1_loss_and_grad(pred: torch.Tensor, target: torch.Tensor, beta: float = 1.0,
                            loss_out: Optional[torch.Tensor] = None):
    """``SmoothL1Loss()(pred, target)`` (mean) and ``d loss / d pred`` from ONE kernel
    (criterion of ``topological_training/train.py:69``).  Use as
    ``loss, g = smooth_l1_loss_and_grad(out, y); out.backward(g)`` -- the separate loss / mean /
    ones-fill / backward-elementwise launches of the autograd route are gone."""
    require_cuda(pred, target)
    p, t = _f32c(pred.detach()), _f32c(target.detach())
    if p.shape != t.shape:
        raise ValueError(f"shape mismatch {tuple(p.shape)} vs {tuple(t.shape)}")
    dev = p.device
    ws = _LOSS_WS.get(dev)
    if ws is None:
        ws = _LOSS_WS[dev] = torch.zeros(_lib.load().qot_smooth_l1_workspace_floats(), dtype=torch.float32, device=dev)
    loss = loss_out if loss_out is not None else torch.empty((), dtype=torch.float32, device=dev)
    grad = torch.empty_like(p)
    _lib.call("qot_smooth_l1", P(p), P(t), p.numel(), float(beta), P(loss), P(grad), P(ws))
    return loss, grad


def table_gemm_ok(V: int, H: int) -> bool:
    """The embedding-table projection and its backward run as products on ``qot_gemm_nt`` / ``_planes`` / ``qot_gemm_tn_planes``
    (V * H >= 1000 * 256: cfg5's table; at cfg4's 1000 x 128 the blocked one-kernel form of ``roles.hip`` measured the same
    or better): that form is latency-bound there."""
    return V * H >= 1000 * 256 and V % 4 == 0 and H % 32 == 0 and not os.environ.get("QOT_NO_TABLE_GEMM")


# ------------------------------------------------------------------ TransformerConv (a2)
def tconv_scores_ok(qkvs: torch.Tensor, maps, num_nodes: int) -> bool:
    """Table mode whose logits can come from the V x V score matrix ``T_q T_k^T`` (``qot_tconv_fwd_scores``): a table of
    4 <= V <= 4096 rows (16 MB of scores at most: L2 / Infinity Cache resident), V % 4 == 0, H % 32 == 0, and enough
    nodes that the V^2 H product is small next to the per-edge dots it replaces."""
    if maps is None or os.environ.get("QOT_NO_TCONV_SCORES"):
        return False
    V, H = qkvs.shape[0], qkvs.shape[1] // 4
    return 4 <= V <= 4096 and V % 4 == 0 and H % 32 == 0 and num_nodes >= 4 * V


class TConvFn(torch.autograd.Function):
    """Fused edge-softmax-aggregate of TransformerConv on packed projections.

    ``qkvs`` is ``[R, 4H]`` = ``[q | k | v | skip]`` (one GEMM); returns ``[N, H]``.
    Node mode (``maps is None``): R == N, one projected row per node.
    Table mode (``maps = (rowmap, colf, colf_t, tiles)``): ``qkvs`` is the projected EMBEDDING TABLE
    ``[V, 4H]`` and rows are gathered through ``node_ids`` -- ``(emb W^T + b)[ids]`` equals
    ``emb[ids] W^T + b`` (``topological_training/models.py:51-53``), so the four node-level GEMMs
    and the embedding gather/scatter disappear and the kernels' inputs are L2-resident.
    ``tiles = (B, n)``: node_ids is ``arange(n)`` repeated B times, so the table gradient is the sum
    over graphs of the per-node gradient ``[B, n, 4H]``.
    """

    @staticmethod
    def forward(ctx, qkvs, edge_attr, w_edge, graph: GraphIndex, maps, act=None):
        require_cuda(qkvs, edge_attr, w_edge)
        ctx.receivers = (w_edge,)
        # the table gradient may wait for the epilogue only when its consumer is TableProjectFn.backward (which flushes)
        ctx.table_from_project = type(qkvs.grad_fn).__name__ == "TableProjectFnBackward"
        qkvs, edge_attr, w_edge = _f32c(qkvs), _f32c(edge_attr), _f32c(w_edge)
        ctx.scores = None
        H4 = qkvs.shape[1]
        H = H4 // 4
        D = w_edge.shape[1]
        N = graph.num_nodes
        if maps is None and qkvs.shape[0] != N:
            raise ValueError("qkvs must have one row per node")
        if edge_attr.shape != (graph.num_edges_in, D):
            raise ValueError(f"edge_attr must be [{graph.num_edges_in}, {D}], got {tuple(edge_attr.shape)}")
        rowmap, colf = (maps[0], maps[1]) if maps is not None else (None, graph.col)
        out = torch.empty(N, H, dtype=torch.float32, device=qkvs.device)
        stats = torch.empty(N, 2, dtype=torch.float32, device=qkvs.device)
        # Tile form (node r of a group of graphs per workgroup, the logits' dense part from one row of T_q T_k^T in LDS): the
        # workgroup pays n dots for that row where the plain form pays one dot per in-edge of its rows_per_block destinations,
        # so it is taken only while n stays within ~2x of that (cfg2: 100 vs 64; at 1000-node graphs it cost 8x the kernel).
        rpb_ = _lib.load().qot_tconv_rows_per_block(H)
        tile_ok = (maps is not None and N > 0 and maps[3][1] <= 2 * rpb_ * max(2, graph.num_edges_in // max(N, 1))
                   and not os.environ.get("QOT_NO_TCONV_TILE"))
        if tile_ok:
            B_, n_ = maps[3]
            _lib.call("qot_tconv_fwd_tile", _off(qkvs, 0), _off(qkvs, H), _off(qkvs, 2 * H), _off(qkvs, 3 * H), H4,
                      P(edge_attr), P(w_edge), P(graph.rowptr), P(colf), P(graph.eid), P(rowmap), P(out), P(stats),
                      N, H, D, int(n_), int(B_), *_act_args(act))
        elif tconv_scores_ok(qkvs, maps, N):
            # table mode at large V: <q_i, k_j> looked up in T_q T_k^T (one small product on the matrix cores, L2-resident)
            # instead of a key-row gather and an H-term dot per edge
            V = qkvs.shape[0]
            scores = ctx.scores = torch.empty(V, V, dtype=torch.float32, device=qkvs.device)
            _lib.call("qot_gemm_nt", _off(qkvs, 0), H4, _off(qkvs, H), H4, P(scores), V, V, V, H, None, None, None)
            if tconv_rows_ok(qkvs, maps, H, D) and not os.environ.get("QOT_NO_TCONV_FWD_ROWS"):
                # node_ids == arange(n) in every graph: a workgroup per table row and slice of the graphs (csrc/tconv_rows.hip)
                B_, n_ = maps[3]
                _lib.call("qot_tconv_fwd_rows", _off(qkvs, 0), _off(qkvs, 2 * H), _off(qkvs, 3 * H), H4, P(scores), V,
                          P(edge_attr), P(w_edge), P(graph.rowptr), P(colf), P(graph.eid), P(out), P(stats), int(n_), int(B_),
                          int(max(1, min(B_, 8))), H, D, *_act_args(act))
            else:
                _lib.call("qot_tconv_fwd_scores", _off(qkvs, 0), _off(qkvs, 2 * H), _off(qkvs, 3 * H), H4, P(scores), V,
                          P(edge_attr), P(w_edge), P(graph.rowptr), P(colf), P(graph.eid), P(rowmap), P(out), P(stats),
                          N, H, D, *_act_args(act))
        else:
            _lib.call("qot_tconv_fwd", _off(qkvs, 0), _off(qkvs, H), _off(qkvs, 2 * H), _off(qkvs, 3 * H), H4,
                      P(edge_attr), P(w_edge), P(graph.rowptr), P(colf), P(graph.eid), P(rowmap), P(out), P(stats),
                      N, H, D, *_act_args(act))
        ctx.save_for_backward(qkvs, edge_attr, w_edge, stats, out if act is not None else None,
                              act[3] if act is not None else None)
        ctx.graph, ctx.maps = graph, maps
        ctx.act = None if act is None else (act[0], act[1], act[2])
        return out

    @staticmethod
    def backward(ctx, g):
        qkvs, edge_attr, w_edge, stats, y, act_step = ctx.saved_tensors
        graph, maps = ctx.graph, ctx.maps
        g = _f32c(g)
        H4 = qkvs.shape[1]
        H = H4 // 4
        D = w_edge.shape[1]
        N = graph.num_nodes
        dev = qkvs.device
        rowmap, colf, colf_t = (maps[0], maps[1], maps[2]) if maps is not None else (None, graph.col, None)
        if ctx.scores is not None and tconv_rows_ok(qkvs, maps, H, D):
            return _tconv_backward_rows(ctx, g, qkvs, edge_attr, w_edge, stats, y, act_step)
        escr = torch.empty(max(graph.cap, 1), 2, dtype=torch.float32, device=dev)
        delta = torch.empty(N, dtype=torch.float32, device=dev)
        pds = torch.empty(N, D, dtype=torch.float32, device=dev)
        pal = torch.empty(N, D, dtype=torch.float32, device=dev)
        gwe_flat = torch.empty(H * D, dtype=torch.float32, device=dev)
        rpb = _lib.load().qot_tconv_rows_per_block(H)     # destinations per workgroup
        tiled = maps is not None
        if tiled:
            # table gradient = sum over graphs: pre-reduced over `rpb` graphs inside the kernels
            B, n = maps[3]
            gb = (B + rpb - 1) // rpb
            gpart = torch.empty(gb, n, H4, dtype=torch.float32, device=dev)
            gskip = torch.empty(N, H, dtype=torch.float32, device=dev)
            tile_args = (n, B, P(gpart))
            gq_ptr, gs_ptr, gk_ptr, gv_ptr, ldg = None, P(gskip), None, None, H
            ws_rows = n * gb * rpb
        else:
            gnode = torch.empty(N, H4, dtype=torch.float32, device=dev)    # per-node [gq | gk | gv | gskip]
            tile_args = (0, 0, None)
            gq_ptr, gs_ptr, gk_ptr, gv_ptr, ldg = _off(gnode, 0), _off(gnode, 3 * H), _off(gnode, H), _off(gnode, 2 * H), H4
            ws_rows = N
        ws = torch.empty(_lib.load().qot_tconv_bwd_dst_workspace_floats(ws_rows, H, D), dtype=torch.float32, device=dev)
        # one launch: back through the fused leaky_relu+dropout (mask regenerated), destination pass,
        # and the lin_edge weight gradient; gskip then holds the gradient wrt the conv output
        if ctx.act is not None:
            slope, p, seed = ctx.act
            act_args = (P(y), float(slope), float(p if act_step is not None else 0.0), int(seed), P(act_step))
        else:
            act_args = (None, 0.0, 0.0, 0, None)
        grouped = LG.can_defer(*ctx.receivers)
        # grouped: the per-workgroup lin_edge partials stay in `ws`; their sums (two levels above 256 workgroups) and the
        # table-gradient row sum join the backward epilogue's multi-role launches instead of three launches of their own
        _lib.call("qot_tconv_bwd_dst", P(g), _off(qkvs, 0), _off(qkvs, H), _off(qkvs, 2 * H), H4,
                  P(edge_attr), P(w_edge), P(stats), P(graph.rowptr), P(colf), P(graph.eid), P(rowmap),
                  gq_ptr, gs_ptr, ldg, P(escr), P(delta), P(pds), P(pal), *act_args,
                  None if grouped else P(gwe_flat), P(ws), *tile_args, N, H, D)
        _lib.call("qot_tconv_bwd_src", gs_ptr, ldg, _off(qkvs, 0), H4, P(escr), P(delta),
                  P(graph.rowptr_t), P(graph.col_t), P(graph.pos_t), P(colf_t), gk_ptr, gv_ptr,
                  ldg, *tile_args, N, H)
        if grouped:
            blocks = int(_lib.load().qot_tconv_bwd_dst_blocks(N, H, tile_args[0], tile_args[1]))
            if blocks > 256:
                per = 128
                groups = (blocks + per - 1) // per
                level1 = torch.empty(groups, H * D, dtype=torch.float32, device=dev)
                LG.defer(_lib.ROLE_SUM_ROWS, (ws, level1), (blocks, H * D, per), stage=1)
                LG.defer(_lib.ROLE_SUM_ROWS, (level1, gwe_flat), (groups, H * D, 0), stage=2)
            else:
                LG.defer(_lib.ROLE_SUM_ROWS, (ws, gwe_flat), (blocks, H * D, 0), stage=1)
        if not tiled:
            gq = gnode
        else:
            if gb == 1:
                gq = gpart[0]
            else:
                gq_flat = torch.empty(n * H4, dtype=torch.float32, device=dev)   # table rows = sum over graph groups
                if LG.enabled() and ctx.table_from_project:
                    LG.defer(_lib.ROLE_SUM_ROWS, (gpart, gq_flat), (gb, n * H4, 0), stage=1)
                else:
                    wsr = torch.empty(_lib.load().qot_rowsum_wide_workspace_floats(n * H4), dtype=torch.float32, device=dev)
                    _lib.call("qot_rowsum_wide", P(gpart), gb, n * H4, P(gq_flat), P(wsr))
                gq = gq_flat.view(n, H4)
            if n < qkvs.shape[0]:                              # table rows no node refers to
                if LG.enabled():
                    LG.flush()                                 # (rare: the concatenation below reads the row sum)
                gq = torch.cat([gq, gq.new_zeros(qkvs.shape[0] - n, H4)], 0)
        return gq, None, gwe_flat.view(H, D), None, None, None


def tconv_rows_ok(qkvs: torch.Tensor, maps, H: int, D: int) -> bool:
    """The row form of the backward (``csrc/tconv_rows.hip``) takes this batch: table mode with ``node_ids == arange(n)`` in
    every graph, the score matrix kept from the forward, a table of exactly those n rows."""
    if maps is None or os.environ.get("QOT_NO_TCONV_ROWS"):
        return False
    B, n = maps[3]
    return (B >= 1 and n == qkvs.shape[0] and bool(_lib.load().qot_tconv_rows_supported(int(n), int(H), int(D))))


def _tconv_backward_rows(ctx, g, qkvs, edge_attr, w_edge, stats, y, act_step):
    """``TConvFn.backward`` in the row form: destination pass (one workgroup per table row and slice of the graphs: grad M
    row, grad P, grad T_skip), source pass (grad T_v), the slices summed in order, then grad T_q / grad T_k from two small
    products on the matrix cores.  Returns what ``backward`` returns."""
    graph, maps = ctx.graph, ctx.maps
    lib = _lib.load()
    dev = qkvs.device
    H4 = qkvs.shape[1]
    H = H4 // 4
    D = w_edge.shape[1]
    N = graph.num_nodes
    B, n = maps[3]
    colf = maps[1]
    scores = ctx.scores
    parts = int(max(1, min(B, 8)))
    npad, ldrow = lib.qot_tconv_rows_npad(n), lib.qot_tconv_rows_ld(n, H, D)
    f32 = dict(dtype=torch.float32, device=dev)
    escr = torch.empty(max(graph.cap, 1), 2, **f32)
    delta = torch.empty(N, **f32)
    gskip = torch.empty(N, H, **f32)
    drows = torch.empty(parts, n * ldrow, **f32)
    wparts = torch.empty(parts * n, H * D, **f32)
    vrows = torch.empty(parts, n * H, **f32)
    if ctx.act is not None:
        slope, p, seed = ctx.act
        act_args = (P(y), float(slope), float(p if act_step is not None else 0.0), int(seed), P(act_step))
    else:
        act_args = (None, 0.0, 0.0, 0, None)
    _lib.call("qot_tconv_bwd_dst_rows", P(g), _off(qkvs, 0), _off(qkvs, 2 * H), H4, P(edge_attr), P(w_edge), P(stats),
              P(graph.rowptr), P(colf), P(graph.eid), P(scores), scores.shape[1], P(gskip), P(escr), P(delta), *act_args,
              int(n), int(B), parts, P(drows), P(wparts), H, D)
    _lib.call("qot_tconv_bwd_src_rows", P(gskip), P(escr), P(graph.rowptr_t), P(graph.col_t), P(graph.pos_t), int(n), int(B),
              parts, P(vrows), H)
    # the slices and the lin_edge partials, each summed in a fixed order (one multi-role launch); K1 floats of slack behind S:
    # the product below reads K1 >= npad + 32 columns from the grad M column on (its operand B is zero past grad P: the
    # columns it meets there -- the row's zero padding, the next row's head -- are finite)
    K1 = 32 * (((npad // 32) + 1 + 7) // 8 * 8)            # grad M, grad P and padding up to eight equal K slices
    S_buf = torch.zeros(n * ldrow + K1, **f32)
    S = S_buf[:n * ldrow]
    gv = torch.empty(n * H, **f32)
    gwe_g = torch.empty(H * D, **f32)
    roles = [_lib.make_role(_lib.ROLE_SUM_ROWS, (drows, S), (parts, n * ldrow, 0)),
             _lib.make_role(_lib.ROLE_SUM_ROWS, (vrows, gv), (parts, n * H, 0))]
    blocks = parts * n
    two_level = blocks > 256
    if two_level:
        per = 128
        groups = (blocks + per - 1) // per
        level1 = torch.empty(groups, H * D, **f32)
        roles.append(_lib.make_role(_lib.ROLE_SUM_ROWS, (wparts, level1), (blocks, H * D, per)))
    else:
        roles.append(_lib.make_role(_lib.ROLE_SUM_ROWS, (wparts, gwe_g), (blocks, H * D, 0)))
    _lib.run_roles(roles)
    S = S.view(n, ldrow)
    rs = 1.0 / float(H) ** 0.5
    gM = S[:, H:H + npad]                                   # [n, npad], columns >= n are zero
    gP = S[:, H + npad:H + npad + D]                        # [n, D]
    tq, tk = qkvs[:, 0:H], qkvs[:, H:2 * H]
    # grad T_q = rs (grad M T_k + grad P W_e^T) as ONE product over K = npad + 32: A = [grad M | grad P | ...] (adjacent
    # in S), B = rs [T_k^T | W_e | 0]; few output tiles, long K: eight K slices, planes summed in order
    Bq = torch.zeros(H, K1, **f32)
    Bq[:, :n] = tk.t()
    Bq[:, npad:npad + D] = w_edge
    Bq.mul_(rs)
    ks = 8
    planes_q = torch.empty(ks, n * H, **f32)
    _lib.call("qot_gemm_nt_planes", P(gM), ldrow, P(Bq), K1, P(planes_q), n, H, K1, ks)
    # grad T_k = rs grad M^T T_q: TN product (inner dimension = the n table rows), split over it
    splits = lib.qot_gemm_tn_splits(npad, H, n)
    planes_k = torch.empty(splits, npad * H, **f32)
    _lib.call("qot_gemm_tn_planes", P(gM), ldrow, _off(qkvs, 0), H4, P(planes_k), npad, H, n, splits, None, None)
    gq_, gk_ = torch.empty(n * H, **f32), torch.empty(npad * H, **f32)
    roles = [_lib.make_role(_lib.ROLE_SUM_ROWS, (planes_q, gq_), (ks, n * H, 0)),
             _lib.make_role(_lib.ROLE_SUM_ROWS, (planes_k, gk_), (splits, npad * H, 0))]
    if two_level:
        roles.append(_lib.make_role(_lib.ROLE_SUM_ROWS, (level1, gwe_g), (groups, H * D, 0)))
    _lib.run_roles(roles)
    g4 = torch.cat([gq_.view(n, H), gk_.view(npad, H)[:n] * rs, gv.view(n, H), S[:, 0:H]], dim=1)
    # grad lin_edge = sum_i q_i/sqrt(H) (x) pd_i + g_i (x) p2_i; the first part is a function of the table row only
    tqp = _small_gemm(tq, gP, None, H, D, n, 1, H4, ldrow, 1)          # T_q^T grad P: A(c, r) = tq[r, c], B(r, d) = gP[r, d]
    gwe = torch.add(gwe_g.view(H, D), tqp, alpha=rs)
    return g4, None, gwe, None, None, None


def tconv_graph_plan(table, H: int, D: int, graph: GraphIndex, maps):
    """``(n, B, max_e)`` when TransformerConv's GRAPH form takes this batch, else ``None``: table mode with ``node_ids ==
    arange(n)`` in every graph (``maps``), the index built graph by graph (block-diagonal, slices verified), ``n <= 128``
    and the LDS images of ``csrc/tconv_graph.hip`` fit, a table the one-row projection bodies take (V < 512)."""
    if maps is None or graph.sizes is None or os.environ.get("QOT_NO_TCONV_GRAPH"):
        return None
    B, n = maps[3]
    if B <= 0 or n <= 0 or graph.num_nodes != B * n or table.shape[0] >= 512 or n > table.shape[0]:
        return None
    max_e = int(graph.sizes[1])
    if not _lib.load().qot_tconv_graph_supported(int(n), max_e, int(H), int(D)):
        return None
    return int(n), int(B), max_e


def tconv_graph_prepare(table, wq, bq, wk, bk, wv, bv, ws, bs, w_edge, n: int, step_pair=None, group=None):
    """``(t4, M, P)`` of the graph form from the current parameters: the projected table ``[V, 4H]`` (as
    ``TableProjectFn``; the launch also advances the dropout counter of ``step_pair``), the score matrix ``M = T_q T_k^T /
    sqrt(H)`` ``[n, ldm]`` and ``P = T_q W_e / sqrt(H)`` ``[n, D]``.  Two independent jobs: with ``group`` they join the
    forward prologue's multi-role launch (valid after ``group.run()``)."""
    ts = [_f32c(t.detach()) for t in (table, wq, bq, wk, bk, wv, bv, ws, bs, w_edge)]
    require_cuda(*ts)
    table, wq, bq, wk, bk, wv, bv, ws, bs, w_edge = ts
    V, H = table.shape
    D = w_edge.shape[1]
    dev = table.device
    t4 = torch.empty(V, 4 * H, dtype=torch.float32, device=dev)
    M = torch.empty(n, _lib.load().qot_tconv_graph_ldm(n), dtype=torch.float32, device=dev)
    Pm = torch.empty(n, D, dtype=torch.float32, device=dev)
    cnt, snap = step_pair if step_pair is not None else (None, None)
    if group is not None:
        group.add(_lib.ROLE_TABLE_PROJECT_FWD, (table, wq, bq, wk, bk, wv, bv, ws, bs, t4, cnt, snap), (V, H))
        group.add(_lib.ROLE_TABLE_SCORES, (table, wq, bq, wk, bk, w_edge, M, Pm), (n, H, D))
    else:
        _lib.call("qot_table_project_fwd", table, wq, bq, wk, bk, wv, bv, ws, bs, t4, V, H, P(cnt), P(snap))
        _lib.call("qot_table_scores", table, wq, bq, wk, bk, w_edge, M, Pm, n, H, D)
    return t4, M, Pm


class TConvGraphFn(torch.autograd.Function):
    """TransformerConv of ``x = table[arange(n)]`` per graph in the GRAPH form (``csrc/tconv_graph.hip``): embedding
    lookup, the four projections, the edge softmax / aggregation, the fused ``dropout(leaky_relu(.))`` and the whole
    backward down to the gradients of the table and of the nine projection parameters
    (``topological_training/models.py:51-55`` and its autograd).  ``pre = (t4, M, P)`` from ``tconv_graph_prepare`` (the
    forward prologue's launch).  One kernel forward; backward one kernel + the two epilogue launches of the step."""

    @staticmethod
    def forward(ctx, table, wq, bq, wk, bk, wv, bv, ws, bs, w_edge, edge_attr, pre, graph: GraphIndex, maps, plan, act=None):
        t4, M, Pm = pre
        n, B, max_e = plan
        edge_attr, w_edge_c = _f32c(edge_attr), _f32c(w_edge.detach())
        require_cuda(edge_attr, t4, M, Pm)
        H, D = table.shape[1], w_edge.shape[1]
        N = graph.num_nodes
        if edge_attr.shape != (graph.num_edges_in, D):
            raise ValueError(f"edge_attr must be [{graph.num_edges_in}, {D}], got {tuple(edge_attr.shape)}")
        dev = t4.device
        out = torch.empty(N, H, dtype=torch.float32, device=dev)
        E = max(graph.num_edges_in, 1)
        # left behind for the backward: attention weights and edge features in CSR slot order, sum alpha ea per node
        alpha = torch.empty(E, dtype=torch.float32, device=dev)
        ea_csr = torch.empty(E, D, dtype=torch.float32, device=dev)
        aa = torch.empty(max(N, 1), D, dtype=torch.float32, device=dev)
        _lib.call("qot_tconv_fwd_graph", t4, 4 * H, M, Pm, w_edge_c, edge_attr, graph.rowptr, maps[1], graph.eid, graph.row,
                  out, alpha, ea_csr, aa, n, B, max_e, H, D, *_act_args(act))
        ctx.save_for_backward(table, wq, wk, wv, ws, w_edge_c, t4, alpha, ea_csr, aa, out if act is not None else None,
                              act[3] if act is not None else None)
        ctx.graph, ctx.maps, ctx.plan = graph, maps, plan
        ctx.act = None if act is None else (act[0], act[1], act[2])
        return out

    @staticmethod
    def backward(ctx, g):
        table, wq, wk, wv, ws, w_edge, t4, alpha, ea_csr, aa, y, act_step = ctx.saved_tensors
        graph, maps = ctx.graph, ctx.maps
        n, B, max_e = ctx.plan
        g = _f32c(g)
        V, H = table.shape
        D = w_edge.shape[1]
        dev = g.device
        lib = _lib.load()
        blocks, rowlen = int(lib.qot_tconv_bwd_graph_blocks(B)), int(lib.qot_tconv_graph_row_floats(n, H, D))
        partials = torch.empty(blocks, rowlen, dtype=torch.float32, device=dev)
        if ctx.act is not None:
            slope, p, seed = ctx.act
            act_args = (y, float(slope), float(p if act_step is not None else 0.0), int(seed), act_step)
        else:
            act_args = (None, 0.0, 0.0, 0, None)
        _lib.call("qot_tconv_bwd_graph", g, *act_args, t4, 4 * H, w_edge, ea_csr, alpha, aa, graph.rowptr, maps[1],
                  graph.row, graph.rowptr_t, graph.col_t, graph.pos_t, partials, n, B, max_e, H, D)
        S = partials[0] if blocks == 1 else torch.empty(rowlen, dtype=torch.float32, device=dev)
        gt = torch.empty(V * H, dtype=torch.float32, device=dev)
        gw = torch.empty(4 * H * H, dtype=torch.float32, device=dev)
        gb = torch.empty(4 * H, dtype=torch.float32, device=dev)
        gwe = torch.empty(H, D, dtype=torch.float32, device=dev)
        wq_, wk_, wv_, ws_ = (_f32c(t.detach()) for t in (wq, wk, wv, ws))
        tab = _f32c(table.detach())
        if LG.enabled():
            # this node is the last of the pass in table mode: its two jobs ride in the backward epilogue's launches,
            # which are issued now (everything returned below is filled by them)
            if blocks > 1:
                LG.defer(_lib.ROLE_SUM_ROWS, (partials, S), (blocks, rowlen, 0), stage=1)
            LG.defer(_lib.ROLE_TABLE_PROJECT_BWD_SCORES, (S, t4, w_edge, tab, wq_, wk_, wv_, ws_, gt, gw, gb, gwe),
                     (V, n, H, D), stage=2)
            LG.flush()
        else:
            if blocks > 1:
                wsr = torch.empty(lib.qot_rowsum_wide_workspace_floats(rowlen), dtype=torch.float32, device=dev)
                _lib.call("qot_rowsum_wide", partials, blocks, rowlen, S, wsr)
            _lib.call("qot_table_project_bwd_scores", S, t4, w_edge, tab, wq_, wk_, wv_, ws_, gt, gw, gb, gwe, V, n, H, D)
        gw2 = gw.view(4 * H, H)
        return (gt.view(V, H), gw2[:H], gb[:H], gw2[H:2 * H], gb[H:2 * H], gw2[2 * H:3 * H], gb[2 * H:3 * H], gw2[3 * H:],
                gb[3 * H:], gwe, None, None, None, None, None, None)


# ------------------------------------------------------------------ NNConv (a4)
def nnconv_wcat(w2, b2, wroot, hin, hout, k):
    """[(K+2)*Hin, Hout]: K blocks of the edge-MLP's second layer, its bias block, root^T."""
    return torch.cat([
        w2.view(hin, hout, k).permute(2, 0, 1).reshape(k * hin, hout),
        b2.view(hin, hout),
        wroot.t(),
    ], dim=0)


def nnconv_wcat_t(w2, b2, wroot, hin, hout, k):
    """[(K+2)*Hout, Hin]: the per-block transposes (adjoint GEMM operand)."""
    return torch.cat([
        w2.view(hin, hout, k).permute(2, 1, 0).reshape(k * hout, hin),
        b2.view(hin, hout).t(),
        wroot,
    ], dim=0)


def gemm_tn(a: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """``a.t() @ g`` for ``a [N, KT]``, ``g [N, 64]`` through ``qot_gemm_tn`` (streaming split-K
    MFMA kernel); other shapes use the library GEMM."""
    n, kt = a.shape
    if g.shape[1] != 64 or kt % 128 != 0 or kt > 1280 or n == 0 or os.environ.get("QOT_DISABLE_GEMM_TN"):
        return a.t() @ g
    a, g = a.contiguous(), g.contiguous()
    out = torch.empty(kt, 64, dtype=torch.float32, device=a.device)
    ws = torch.empty(_lib.load().qot_gemm_tn_workspace_floats(kt), dtype=torch.float32, device=a.device)
    _lib.call("qot_gemm_tn", P(a), kt, P(g), 64, n, kt, P(out), P(ws))
    return out


_PERM_CACHE = {}


def nnconv_perm_index(kt: int, device) -> torch.Tensor:
    """Gather index that lays a [KT, 64] GEMM operand out in MFMA fragment order for
    ``qot_nnconv_fused`` (layout: csrc/nnconv_mfma.hip)."""
    key = (kt, str(device))
    if key not in _PERM_CACHE:
        nh, g, l, r = torch.meshgrid(torch.arange(2), torch.arange(kt // 8), torch.arange(64), torch.arange(4),
                                     indexing="ij")
        k = 8 * g + 2 * r + (l >> 5)
        n = nh * 32 + (l & 31)
        _PERM_CACHE[key] = (k * 64 + n).reshape(-1).to(device)
    return _PERM_CACHE[key]


def nnconv_gradh_perm_index(k: int, device) -> torch.Tensor:
    """Index into wk = Wcat[:K*64] ([K*64, 64], wk[n, o]) that lays Wk^T out in fragment order
    for ``qot_nnconv_gradh_fused``."""
    key = ("gradh", k, str(device))
    if key not in _PERM_CACHE:
        nb, gq, l, r = torch.meshgrid(torch.arange(2 * k), torch.arange(8), torch.arange(64), torch.arange(4),
                                      indexing="ij")
        o = 8 * gq + 2 * r + (l >> 5)
        n = nb * 32 + (l & 31)
        _PERM_CACHE[key] = (n * 64 + o).reshape(-1).to(device)
    return _PERM_CACHE[key]


def nnconv_fused_indices(k: int, device):
    """Gather indices into ``pflat = cat([nn.2.weight.flatten(), nn.2.bias, lin.weight.flatten()])``
    (H = 64) that produce, with ONE gather each, the three fragment-ordered operands of the fused
    NNConv kernels: Wcat (forward), WcatT (adjoint) and Wk^T (grad-h).  Replaces per-step
    permute/reshape/cat chains (six small kernels) by host-side index composition, done once."""
    key = ("fusedidx", k, str(device))
    if key not in _PERM_CACHE:
        h = 64
        kk, a, o = torch.meshgrid(torch.arange(k), torch.arange(h), torch.arange(h), indexing="ij")
        w2_off = (a * h + o) * k + kk                                   # Wcat[k*H + a, o] = W2[a*H+o, k]
        a2, o2 = torch.meshgrid(torch.arange(h), torch.arange(h), indexing="ij")
        b2_off = h * h * k + a2 * h + o2                                # block K: b2[a*H + o]
        root_off = h * h * k + h * h + o2 * h + a2                      # block K+1: wroot[o, a]
        wcat_idx = torch.cat([w2_off.reshape(k * h, h), b2_off, root_off], 0)          # [(K+2)H, H]
        wcat_t_idx = wcat_idx.view(k + 2, h, h).transpose(1, 2).reshape((k + 2) * h, h)
        fwd = wcat_idx.reshape(-1)[nnconv_perm_index((k + 2) * h, "cpu")]
        adj = wcat_t_idx.reshape(-1)[nnconv_perm_index((k + 2) * h, "cpu")]
        gh = wcat_idx[:k * h].reshape(-1)[nnconv_gradh_perm_index(k, "cpu")]
        allidx = torch.cat([fwd, adj, gh]).to(torch.int32).contiguous().to(device)
        _PERM_CACHE[key] = (allidx, fwd.numel(), adj.numel(), gh.numel())
    return _PERM_CACHE[key]


def nnconv_pack_operands(w2, b2, wroot, k: int):
    """(Wcat, WcatT, Wk^T) in MFMA fragment order straight from the three parameters: one launch."""
    allidx, n_f, n_a, n_g = nnconv_fused_indices(k, w2.device)
    packed = torch.empty(allidx.numel(), dtype=torch.float32, device=w2.device)
    _lib.call("qot_gather3", P(w2), w2.numel(), P(b2), b2.numel(), P(wroot), P(allidx), P(packed), allidx.numel())
    return packed[:n_f], packed[n_f:n_f + n_a], packed[n_f + n_a:]


def nnconv_pack(w2, b2, wroot, h: int, k: int, group=None):
    """(Wcat, WcatT, Wk^T) in the fragment orders of the width's kernels.  ``group`` (a ``LaunchGroup``): the gather
    joins the caller's multi-role launch; the operands are valid after ``group.run()``."""
    allidx, n_f, n_a, n_g = nnconv_fused_indices(k, w2.device) if h == 64 else nnconv_gen_indices(h, k, w2.device)
    packed = torch.empty(allidx.numel(), dtype=torch.float32, device=w2.device)
    if group is not None:
        group.add(_lib.ROLE_GATHER3, (w2, b2, wroot, allidx, packed), (w2.numel(), b2.numel(), allidx.numel()))
    else:
        _lib.call("qot_gather3", P(w2), w2.numel(), P(b2), b2.numel(), P(wroot), P(allidx), P(packed), allidx.numel())
    return packed[:n_f], packed[n_f:n_f + n_a], packed[n_f + n_a:]


GEN_WIDTHS = (16, 32, 128, 256)       # csrc/nnconv_gen.hip; 64 has its own tuned kernels (csrc/nnconv_mfma.hip)


def nnconv_gen_indices(h: int, k: int, device, cw: int = 0):
    """As ``nnconv_fused_indices`` for the width-generic kernels: gather indices into
    ``cat([nn.2.weight.flatten(), nn.2.bias, lin.weight.flatten()])`` (-1 = zero padding) giving Wcat and WcatT in
    the per-pass fragment order of ``nnconv_gen_kernel`` and Wk^T in that of ``nnconv_gradh_gen_kernel``
    (layouts: csrc/nnconv_gen.hip).  Host-side index composition, done once per (width, edge_dim)."""
    key = ("genidx", h, k, str(device), cw)
    if key not in _PERM_CACHE:
        cw = cw or min(h, 64)            # input channels per pass (32: the three-workgroups-per-CU variant)
        n_pass, ncb, gall = h // cw, (h + 31) // 32, (k + 2) * cw // 8
        P_, CB, G_, L_, R_ = torch.meshgrid(torch.arange(n_pass), torch.arange(ncb), torch.arange(gall), torch.arange(64),
                                            torch.arange(4), indexing="ij")
        kl = 8 * G_ + 2 * R_ + (L_ >> 5)                 # local inner index of the pass
        kk, a = kl // cw, P_ * cw + kl % cw              # block, channel
        col = CB * 32 + (L_ & 31)

        def wcat_off(kk, a, o):                          # offset of Wcat[kk*h + a][o] in the flat parameter triple
            return torch.where(kk < k, (a * h + o) * k + kk,
                               torch.where(kk == k, h * h * k + a * h + o, h * h * (k + 1) + o * h + a))
        pad = col >= h
        colc = col.clamp(max=h - 1)
        fwd = torch.where(pad, torch.full_like(col, -1), wcat_off(kk, a, colc))
        adj = torch.where(pad, torch.full_like(col, -1), wcat_off(kk, colc, a))      # WcatT[kk*h + a][o] = Wcat[kk*h + o][a]
        cwg = min(h, 32)
        npg, nbg, gh_ = h // cwg, k * cwg // 32, h // 8
        P2, NB, GQ, L2, R2 = torch.meshgrid(torch.arange(npg), torch.arange(nbg), torch.arange(gh_), torch.arange(64),
                                            torch.arange(4), indexing="ij")
        o = 8 * GQ + 2 * R2 + (L2 >> 5)
        n = NB * 32 + (L2 & 31)
        kq, al = n // cwg, n % cwg
        gh = ((P2 * cwg + al) * h + o) * k + kq
        allidx = torch.cat([fwd.reshape(-1), adj.reshape(-1), gh.reshape(-1)]).to(torch.int32).contiguous().to(device)
        _PERM_CACHE[key] = (allidx, fwd.numel(), adj.numel(), gh.numel())
    return _PERM_CACHE[key]


def nnconv_pack_operands_gen(w2, b2, wroot, h: int, k: int):
    """(Wcat, WcatT, Wk^T) in the generic kernels' fragment orders straight from the three parameters: one launch."""
    allidx, n_f, n_a, n_g = nnconv_gen_indices(h, k, w2.device)
    packed = torch.empty(allidx.numel(), dtype=torch.float32, device=w2.device)
    _lib.call("qot_gather3", P(w2), w2.numel(), P(b2), b2.numel(), P(wroot), P(allidx), P(packed), allidx.numel())
    return packed[:n_f], packed[n_f:n_f + n_a], packed[n_f + n_a:]


class NNConvFn(torch.autograd.Function):
    """NNConv(aggr='mean') = aggregate-then-GEMM (see ``csrc/nnconv.hip``)."""

    @staticmethod
    def forward(ctx, x, edge_attr, w1, b1, w2, b2, wroot, bias, graph: GraphIndex, act=None, side=None, packed=None):
        """``side`` (dict shared with the consumer, see ``HeadFn``): the consumer's backward delivers the
        gradient wrt the PRE-activation output and this layer's bias gradient in ``side["gbias"]``.
        ``packed``: ``nnconv_pack(...)``'s result when the caller has already packed the operands (in the forward
        prologue's multi-role launch)."""
        require_cuda(x, edge_attr, w1, b1, w2, b2, wroot, bias)
        ctx.receivers = (w1, b1, w2, b2, wroot)   # who gets the (possibly deferred) parameter gradients
        if side is not None and act is not None:
            side["bias_param"] = bias             # the read-out's backward produces this parameter's gradient
        x, edge_attr = _f32c(x), _f32c(edge_attr)
        w1, b1, w2, b2, wroot, bias = (_f32c(t) for t in (w1, b1, w2, b2, wroot, bias))
        N, hin = x.shape
        hout = wroot.shape[0]
        K, D = w1.shape
        if K != 2 * D:
            raise _lib.QotError("NNConv edge MLP must be Linear(D, 2D) -> ReLU -> Linear(2D, Hin*Hout)")
        if edge_attr.shape != (graph.num_edges_in, D):
            raise ValueError(f"edge_attr must be [{graph.num_edges_in}, {D}], got {tuple(edge_attr.shape)}")
        if hin != hout:
            raise _lib.QotError("NNConv HIP path needs in_channels == out_channels")
        if hin != 64 and hin not in GEN_WIDTHS:
            raise _lib.QotError(f"NNConv HIP path: hidden width {hin} not in (16, 32, 64, 128, 256)")
        if D > 8:
            raise _lib.QotError("NNConv HIP path: edge_dim <= 8 (include/qot_gnn.h)")
        if D > 4:
            # edge_dim 5..8 (the reference takes edge_dim = len(dataset.FEATURES), topological_training/dataset.py:40):
            # the fused tile kernels are built for K = 2D <= 8 operand blocks; wider edge MLPs materialise the operand
            # A [N, (K+2)H] (qot_nnconv_agg) and multiply it with library GEMMs -- correct at every width, slower
            return NNConvFn._forward_wide_edge(ctx, x, edge_attr, w1, b1, w2, b2, wroot, bias, graph, act, side)
        # (Wcat, WcatT, Wk^T) in MFMA fragment order; every width runs gather -> LDS tile -> fp32 MFMA, the operand
        # [N, (K+2)H] never exists in HBM
        wp, wp_adj, bp = packed if packed is not None else nnconv_pack(w2, b2, wroot, hin, K)
        out = torch.empty(N, hout, dtype=torch.float32, device=x.device)
        _lib.call("qot_nnconv_fused", P(x), hin, P(edge_attr), P(w1), P(b1), P(graph.rowptr), P(graph.col),
                  P(graph.eid), P(graph.invdeg), 0, P(wp), P(bias), P(out), N, hin, D, *_act_args(act))
        ctx.save_for_backward(x, edge_attr, w1, b1, w2, b2, wroot, None, wp_adj, bp, out if act is not None else None,
                              act[3] if act is not None else None)
        ctx.graph = graph
        ctx.act = None if act is None else (act[0], act[1], act[2])
        ctx.side = side if act is not None else None
        return out

    @staticmethod
    def _forward_wide_edge(ctx, x, edge_attr, w1, b1, w2, b2, wroot, bias, graph, act, side):
        N, hin = x.shape
        hout = wroot.shape[0]
        K, D = w1.shape
        A = torch.empty(N, (K + 2) * hin, dtype=torch.float32, device=x.device)
        _lib.call("qot_nnconv_agg", P(x), hin, P(edge_attr), P(w1), P(b1), P(graph.rowptr), P(graph.col),
                  P(graph.eid), None, P(graph.invdeg), 0, P(A), N, hin, D)
        out = torch.addmm(bias, A, nnconv_wcat(w2, b2, wroot, hin, hout, K))
        if act is not None:                  # separate activation kernel, same mask as the fused epilogue
            pre = out
            out = torch.empty_like(pre)
            _lib.call("qot_act_fwd", P(pre), P(out), pre.numel(), *_act_args(act)[1:])
        ctx.save_for_backward(x, edge_attr, w1, b1, w2, b2, wroot, A, None, None, out if act is not None else None,
                              act[3] if act is not None else None)
        ctx.graph = graph
        ctx.act = None if act is None else (act[0], act[1], act[2])
        ctx.side = side if act is not None else None
        return out

    @staticmethod
    def _backward_wide_edge(ctx, g, gbias):
        x, edge_attr, w1, b1, w2, b2, wroot, A, _, _, _, _ = ctx.saved_tensors
        graph = ctx.graph
        N, hin = x.shape
        hout = wroot.shape[0]
        K, D = w1.shape
        dev = x.device
        gwcat = gemm_tn(A, g)                                # [(K+2)Hin, Hout] = A^T g
        gx = None
        if ctx.needs_input_grad[0]:
            U = torch.empty(N, (K + 2) * hout, dtype=torch.float32, device=dev)
            _lib.call("qot_nnconv_agg", P(g), hout, P(edge_attr), P(w1), P(b1), P(graph.rowptr_t), P(graph.col_t),
                      P(graph.pos_t), P(graph.eid), P(graph.invdeg), 1, P(U), N, hout, D)
            gx = U @ nnconv_wcat_t(w2, b2, wroot, hin, hout, K)
        gw2 = gwcat[:K * hin].view(K, hin, hout).permute(1, 2, 0).reshape(hin * hout, K)
        gb2 = gwcat[K * hin:(K + 1) * hin].reshape(hin * hout)
        gwroot = gwcat[(K + 1) * hin:].t()
        wk = w2.view(hin, hout, K).permute(2, 0, 1).reshape(K * hin, hout)
        GA = (g @ wk.t()).contiguous()                       # [N, K*Hin]
        gw1 = torch.zeros(K, D, dtype=torch.float32, device=dev)
        gb1 = torch.zeros(K, dtype=torch.float32, device=dev)
        _lib.call("qot_nnconv_bwd_edge", P(GA), K * hin, P(x), hin, P(edge_attr), P(w1), P(b1),
                  P(graph.rowptr), P(graph.col), P(graph.eid), P(graph.invdeg), P(gw1), P(gb1), N, hin, D)
        return gx, None, gw1, gb1, gw2, gb2, gwroot, gbias, None, None, None, None

    @staticmethod
    def backward(ctx, g):
        x, edge_attr, w1, b1, w2, b2, wroot, A, wp_adj, bp, y, act_step = ctx.saved_tensors
        graph = ctx.graph
        g = _f32c(g)
        if ctx.side is not None:
            # folded mode (decided at forward time): the consumer's backward (HeadFn) has gone back through this
            # layer's activation and left the bias gradient here; ``g`` is wrt the PRE-activation output.  Every
            # HeadFn.backward leaves a fresh entry, so a second backward over a retained graph stays correct; it is
            # popped (not read) so that autograd holds the only reference and assigns it as .grad without a copy kernel.
            if "gbias" not in ctx.side:
                raise RuntimeError("NNConv output was folded into the fused read-out, but the read-out's backward "
                                   "has not run: the conv output must feed only the read-out head "
                                   "(set model._qot_fold_head = False for other graphs)")
            gbias = ctx.side.pop("gbias")
        elif ctx.act is not None:
            g, gbias = act_backward_colsum(g, y, ctx.act + (act_step,))
        else:
            gbias = colsum(g)
        if A is not None:
            return NNConvFn._backward_wide_edge(ctx, g, gbias)
        N, hin = x.shape
        hout = wroot.shape[0]
        K, D = w1.shape
        dev = x.device
        hh = hin * hout
        # (flat buffers the epilogue queue may keep alive; autograd gets views of them)
        gw1f = torch.empty(K * D, dtype=torch.float32, device=dev)
        gb1f = torch.empty(K, dtype=torch.float32, device=dev)
        gw1, gb1 = gw1f.view(K, D), gb1f.view(K)
        wsh = torch.empty(_lib.load().qot_nnconv_gradh_workspace_floats(D), dtype=torch.float32, device=dev)
        gpar = torch.empty((K + 2) * hh, dtype=torch.float32, device=dev)
        gradh_args = (P(g), hout, P(x), hin, P(edge_attr), P(w1), P(b1), P(graph.rowptr), P(graph.col), P(graph.eid),
                      P(graph.invdeg), P(bp))
        if hin == 64 and not os.environ.get("QOT_SPLIT_NNCONV_BWD"):
            # one gather feeds both products: grad_x = U @ WcatT and gWcat = X^T U; the slab sum of the weight gradient
            # and the block sum of the grad-h kernel share one launch behind both kernels
            gx = torch.empty(N, hin, dtype=torch.float32, device=dev)
            ws = torch.empty(_lib.load().qot_nnconv_adjoint_dw_workspace_floats(D), dtype=torch.float32, device=dev)
            deferred = LG.can_defer(*ctx.receivers)
            # QOT_FORK (experiment, default off): the grad-h kernel on a side stream, "before" = next to the adjoint
            # kernel, "after" = next to the TransformerConv backward that follows on the main stream
            fork_mode = os.environ.get("QOT_FORK", "off") if deferred else "off"
            launch_gradh = lambda: _lib.call("qot_nnconv_gradh_fused", *gradh_args, None, None, P(wsh), N, hin, D)
            keep = (g, x, edge_attr, w1, b1, graph.rowptr, graph.col, graph.eid, graph.invdeg, bp, wsh)
            if fork_mode == "before":
                LG.fork(launch_gradh, keep=keep)
            _lib.call("qot_nnconv_adjoint_dw", P(g), hout, P(x), hin, P(edge_attr), P(w1), P(b1), P(graph.rowptr_t),
                      P(graph.col_t), P(graph.eid_t), P(graph.invdeg), P(wp_adj), P(gx), P(gpar), 2, P(ws), N, hin, D)
            if fork_mode == "after":
                LG.fork(launch_gradh, keep=keep)
            elif fork_mode != "before":
                launch_gradh()
            if deferred:             # both second-stage sums join the backward epilogue's multi-role launch
                LG.defer(_lib.ROLE_NNCONV_FINALIZE64, (ws, wsh, gpar, gw1f, gb1f), (N, D), stage=1)
            else:
                _lib.call("qot_nnconv_bwd_finalize", P(ws), P(wsh), P(gpar), P(gw1f), P(gb1f), N, hin, D)
        else:
            # grad_x: the forward kernel over the transposed graph with the per-block transposed weights;
            # weight gradient: A^T g by slices of the result, the operand gathered 32 channels at a time
            gx = None
            if ctx.needs_input_grad[0]:
                gx = torch.empty(N, hin, dtype=torch.float32, device=dev)
                _lib.call("qot_nnconv_fused", P(g), hout, P(edge_attr), P(w1), P(b1), P(graph.rowptr_t),
                          P(graph.col_t), P(graph.eid_t), P(graph.invdeg), 1, P(wp_adj), None, P(gx), N, hout, D,
                          0, 0.0, 0.0, 0, None)
            ws = torch.empty(_lib.load().qot_nnconv_dw_workspace_floats(N, hin, D), dtype=torch.float32, device=dev)
            _lib.call("qot_nnconv_dw", P(x), hin, P(g), hout, P(edge_attr), P(w1), P(b1), P(graph.rowptr), P(graph.col),
                      P(graph.eid), P(graph.invdeg), P(gpar), P(ws), N, hin, D)
            # grad of the edge MLP's first layer
            _lib.call("qot_nnconv_gradh_fused", *gradh_args, P(gw1), P(gb1), P(wsh), N, hin, D)
        # already in the parameters' own layouts (no permute / copy kernels)
        gw2, gb2, gwroot = gpar[:hh * K].view(hh, K), gpar[hh * K:hh * (K + 1)], gpar[hh * (K + 1):].view(hout, hin)
        return gx, None, gw1, gb1, gw2, gb2, gwroot, gbias, None, None, None, None


# ------------------------------------------------------------------ leaky_relu + dropout (a3)
class ActFn(torch.autograd.Function):
    """``dropout(leaky_relu(x, slope), p)`` in one pass; mask regenerated in backward."""

    @staticmethod
    def forward(ctx, x, slope, p, seed, step_counter):
        require_cuda(x)
        x = _f32c(x)
        y = torch.empty_like(x)
        _lib.call("qot_act_fwd", P(x), P(y), x.numel(), float(slope), float(p), int(seed), P(step_counter))
        ctx.save_for_backward(y, step_counter if step_counter is not None else torch.empty(0))
        ctx.cfg = (float(slope), float(p), int(seed), step_counter is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        y, counter = ctx.saved_tensors
        slope, p, seed, has_counter = ctx.cfg
        g = _f32c(g)
        gx = torch.empty_like(g)
        _lib.call("qot_act_bwd", P(g), P(y), P(gx), g.numel(), slope, p, seed, P(counter) if has_counter else None)
        return gx, None, None, None, None


# ------------------------------------------------------------------ global mean pool (a5)
class PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, batch32, ptr32, B):
        require_cuda(x)
        x = _f32c(x)
        N, H = x.shape
        out = torch.empty(B, H, dtype=torch.float32, device=x.device)
        _lib.call("qot_pool_fwd", P(x), P(ptr32), P(out), B, H)
        ctx.save_for_backward(batch32, ptr32)
        ctx.dims = (N, H, B)
        return out

    @staticmethod
    def backward(ctx, g):
        batch32, ptr32 = ctx.saved_tensors
        N, H, B = ctx.dims
        g = _f32c(g)
        gx = torch.empty(N, H, dtype=torch.float32, device=g.device)
        _lib.call("qot_pool_bwd", P(g), P(ptr32), P(batch32), P(gx), N, B, H)
        return gx, None, None, None


# ------------------------------------------------------------------ fused read-out head (a5 + a6)
class HeadFn(torch.autograd.Function):
    """``mlp(global_mean_pool(x, batch))`` of ``topological_training/models.py:61-63`` in one kernel
    each way: pool -> Linear -> LeakyReLU -> Dropout -> Linear, and the whole backward (incl. pool
    backward and all four parameter gradients)."""

    @staticmethod
    def forward(ctx, x, ptr32, w0, b0, w3, b3, B, act, in_act=None, side=None, loss=None):
        """``loss = (target [B, O], beta, loss_out 0-dim)``: the train step's criterion (``SmoothL1Loss(mean, beta)``,
        ``topological_training/train.py:69,113-115``) rides in the same kernel: returns ``(out, grad_out)`` with
        ``grad_out = d loss / d out`` (feed it to ``out.backward``); ``loss_out`` receives the loss value with the
        backward epilogue (it is the sum of per-graph shares, summed where the other partials are summed).
        ``in_act`` / ``side``: the producer of ``x`` applied ``dropout(leaky_relu(.))`` in its epilogue
        (``in_act = (slope, p, seed, step)``) and has agreed (same ``side`` dict handed to its Function) to
        receive the gradient wrt its PRE-activation output: backward then folds that activation backward
        and the producer's bias gradient (``side["gbias"]``) into the pool-backward pass."""
        require_cuda(x, w0, b0, w3, b3)
        ctx.receivers = (w0, b0, w3, b3)          # who gets the (possibly deferred) parameter gradients
        x, w0, b0, w3, b3 = (_f32c(t) for t in (x, w0, b0, w3, b3))
        N, H = x.shape
        O = w3.shape[0]
        slope, p, seed, step = act
        dev = x.device
        pooled = torch.empty(B, H, dtype=torch.float32, device=dev)
        hidden = torch.empty(B, H, dtype=torch.float32, device=dev)
        out = torch.empty(B, O, dtype=torch.float32, device=dev)
        gout = loss_rows = None
        fold = in_act is not None and side is not None
        ctx.train_fused = None
        if loss is not None:
            target, beta, loss_out = loss[:3]
            target = _f32c(target.detach())
            if target.shape != (B, O):
                raise ValueError(f"target must be [{B}, {O}], got {tuple(target.shape)}")
            gout = torch.empty(B, O, dtype=torch.float32, device=dev)
            loss_rows = torch.empty(B, dtype=torch.float32, device=dev)
            ctx.loss = (loss_rows, loss_out)
            if len(loss) > 3 and loss[3] and not os.environ.get("QOT_NO_HEAD_TRAIN"):
                # train step: forward, criterion AND backward of the read-out in one kernel (the graph's rows are read
                # once and stay in LDS for the pool backward); backward() only hands the results on
                nb = H * H + H + O * H + O
                gx = torch.empty(N, H, dtype=torch.float32, device=dev)
                ws = torch.empty(_lib.load().qot_head_bwd_workspace_floats(H, O), dtype=torch.float32, device=dev)
                fa = (1, float(in_act[0]), float(in_act[1] if in_act[3] is not None else 0.0), int(in_act[2]), P(in_act[3])) \
                    if fold else (0, 0.0, 0.0, 0, None)
                _lib.call("qot_head_train", P(x), P(ptr32), P(w0), P(b0), P(w3), P(b3), P(target), float(beta), P(out), P(gout),
                          P(loss_rows), P(gx), P(ws), B, H, O, float(slope), float(p if step is not None else 0.0), int(seed),
                          P(step), *fa)
                ctx.train_fused = (gx, ws, gout, nb + (H if fold else 0))
                ctx.cfg = (N, H, O, B, float(slope), float(p if step is not None else 0.0), int(seed))
                ctx.fold = fold
                ctx.side = side if fold else None
                ctx.mark_non_differentiable(gout)
                ctx.set_materialize_grads(False)
                return out, gout
            _lib.call("qot_head_fwd_loss", P(x), P(ptr32), P(w0), P(b0), P(w3), P(b3), P(pooled), P(hidden), P(out), B, H,
                      O, float(slope), float(p if step is not None else 0.0), int(seed), P(step), P(target), float(beta),
                      P(gout), P(loss_rows))
        else:
            _lib.call("qot_head_fwd", P(x), P(ptr32), P(w0), P(b0), P(w3), P(b3), P(pooled), P(hidden), P(out), B, H, O,
                      float(slope), float(p if step is not None else 0.0), int(seed), P(step))
            ctx.loss = None
        ctx.save_for_backward(ptr32, w0, w3, pooled, hidden, step, x if fold else None,
                              in_act[3] if fold else None)
        ctx.cfg = (N, H, O, B, float(slope), float(p if step is not None else 0.0), int(seed))
        ctx.fold = (float(in_act[0]), float(in_act[1] if in_act[3] is not None else 0.0), int(in_act[2])) if fold else None
        ctx.side = side if fold else None
        if loss is not None:
            ctx.mark_non_differentiable(gout)
            ctx.set_materialize_grads(False)     # no zeros_like(gout) fill launch for the output that carries no gradient
            return out, gout
        return out

    @staticmethod
    def _backward_train_fused(ctx, g):
        """The kernel of the forward has already gone backward with ``grad_out = d loss / d out``: valid only for exactly
        that gradient -- ``out.backward(g)`` with the tensor ``forward_loss`` returned."""
        gx, ws, gout, ntot = ctx.train_fused
        N, H, O, B, slope, p, seed = ctx.cfg
        if g is None or g.data_ptr() != gout.data_ptr() or g.shape != gout.shape:
            raise RuntimeError("the read-out ran its backward inside forward_loss with the criterion's own gradient: call "
                               "out.backward(g) with the g forward_loss returned (or use forward() + a criterion)")
        nb = H * H + H + O * H + O
        grads = torch.empty(ntot, dtype=torch.float32, device=gx.device)
        loss_rows, loss_out = ctx.loss
        roles = [(_lib.ROLE_SUM_ROWS, (ws, grads), (_lib.load().qot_head_train_blocks(B, H), ntot, 0)),
                 (_lib.ROLE_SUM_ROWS, (loss_rows, loss_out), (B, 1, 0))]
        if LG.can_defer(*ctx.receivers, ctx.side.get("bias_param") if ctx.fold else None):
            for r in roles:
                LG.defer(*r, stage=1)
        else:
            _lib.run_roles([_lib.make_role(*r) for r in roles])
        if ctx.fold:
            ctx.side["gbias"] = grads[nb:]
        gw0 = grads[:H * H].view(H, H)
        gb0 = grads[H * H:H * H + H]
        gw3 = grads[H * H + H:H * H + H + O * H].view(O, H)
        gb3 = grads[H * H + H + O * H:H * H + H + O * H + O]
        return gx, None, gw0, gb0, gw3, gb3, None, None, None, None, None

    @staticmethod
    def backward(ctx, g, _g_gout=None):
        if ctx.train_fused is not None:
            return HeadFn._backward_train_fused(ctx, g)
        ptr32, w0, w3, pooled, hidden, step, x_in, in_step = ctx.saved_tensors
        N, H, O, B, slope, p, seed = ctx.cfg
        if g is None:                            # (only with set_materialize_grads(False): no gradient reached `out`)
            g = torch.zeros(B, O, dtype=torch.float32, device=pooled.device)
        g = _f32c(g)
        dev = g.device
        gx = torch.empty(N, H, dtype=torch.float32, device=dev)
        nb = H * H + H + O * H + O
        ntot = nb + (H if ctx.fold else 0)
        grads = torch.empty(ntot, dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.load().qot_head_bwd_workspace_floats(H, O), dtype=torch.float32, device=dev)
        fold_args = (P(x_in), ctx.fold[0], ctx.fold[1], ctx.fold[2], P(in_step)) if ctx.fold else (None, 0.0, 0.0, 0, None)
        # the sum of the workgroup partials joins the backward epilogue's launch -- if nothing reads it before then
        grouped = LG.can_defer(*ctx.receivers, ctx.side.get("bias_param") if ctx.fold else None)
        _lib.call("qot_head_bwd", P(g), P(pooled), P(hidden), P(ptr32), P(w0), P(w3), P(gx), None if grouped else P(grads),
                  P(ws), B, H, O, slope, p, seed, P(step), *fold_args)
        if grouped:
            LG.defer(_lib.ROLE_SUM_ROWS, (ws, grads), (_lib.load().qot_head_bwd_blocks(B), ntot, 0), stage=1)
        if ctx.loss is not None:         # the loss value: sum of the per-graph shares the forward left behind
            loss_rows, loss_out = ctx.loss
            role = (_lib.ROLE_SUM_ROWS, (loss_rows, loss_out), (B, 1, 0))
            if LG.enabled():
                LG.defer(*role, stage=1)
            else:
                _lib.run_roles([_lib.make_role(*role)])
        if ctx.fold:
            ctx.side["gbias"] = grads[nb:]          # the producer's bias gradient; gx is wrt its pre-activation
        gw0 = grads[:H * H].view(H, H)
        gb0 = grads[H * H:H * H + H]
        gw3 = grads[H * H + H:H * H + H + O * H].view(O, H)
        gb3 = grads[H * H + H + O * H:H * H + H + O * H + O]
        return gx, None, gw0, gb0, gw3, gb3, None, None, None, None, None


# ------------------------------------------------------------------ GATConv (a7)
class GatFn(torch.autograd.Function):
    """GATConv after the dense projection: attention logits from ``z`` (``a[n,h] = <z[n,h,:], att[h,:]>``), fused edge
    softmax + aggregation (+bias).  ``graph`` carries the self loops.  ``bn_stats=True`` also returns the
    per-workgroup column partials of the output that the BatchNorm behind it turns into its batch statistics
    (``BnFn`` ``partials=``): one pass over ``[N, 4C]`` fewer per layer."""

    @staticmethod
    def forward(ctx, z, att_src, att_dst, bias, graph: GraphIndex, neg_slope: float, bn_stats: bool = False,
                a_src=None, a_dst=None):
        """``a_src`` / ``a_dst``: the logits, when the projection that made ``z`` left them behind (``gemm_nt(att=...)``);
        their gradient still returns through ``grad_z`` in ``backward``."""
        require_cuda(z, att_src, att_dst, bias)
        z, bias = _f32c(z), _f32c(bias)
        heads, C = att_src.shape[-2], att_src.shape[-1]
        att_s, att_d = _f32c(att_src).reshape(-1), _f32c(att_dst).reshape(-1)
        N, HC = z.shape
        dev = z.device
        if a_src is None or a_dst is None:
            a_src = torch.empty(N, heads, dtype=torch.float32, device=dev)
            a_dst = torch.empty(N, heads, dtype=torch.float32, device=dev)
            _lib.call("qot_gat_logits", P(z), P(att_s), P(att_d), P(a_src), P(a_dst), N, heads, C)
        else:
            a_src, a_dst = _f32c(a_src), _f32c(a_dst)
        out = torch.empty(N, HC, dtype=torch.float32, device=dev)
        stats = torch.empty(N, heads, 2, dtype=torch.float32, device=dev)
        partials = None
        if bn_stats and N > 0:
            partials = torch.empty(_lib.load().qot_gat_bn_partials_floats(N, heads, C), dtype=torch.float32, device=dev)
        _lib.call("qot_gat_fwd", P(z), P(a_src), P(a_dst), P(bias), P(graph.rowptr), P(graph.col), P(out),
                  P(stats), N, heads, C, float(neg_slope), P(partials))
        ctx.save_for_backward(z, a_src, a_dst, stats, att_s, att_d)
        ctx.graph, ctx.ns, ctx.att_shape = graph, float(neg_slope), tuple(att_src.shape)
        if bn_stats:
            if partials is None:
                partials = torch.empty(0, dtype=torch.float32, device=dev)
            ctx.mark_non_differentiable(partials)
            return out, partials
        return out

    @staticmethod
    def backward(ctx, g, _gp=None):
        z, a_src, a_dst, stats, att_s, att_d = ctx.saved_tensors
        graph, ns = ctx.graph, ctx.ns
        g = _f32c(g)
        N, HC = z.shape
        heads = a_src.shape[1]
        C = HC // heads
        dev = z.device
        gad = torch.empty(N, heads, dtype=torch.float32, device=dev)
        gas = torch.empty(N, heads, dtype=torch.float32, device=dev)
        gz = torch.empty(N, HC, dtype=torch.float32, device=dev)
        escr = torch.empty(max(graph.cap, 1), heads, 2, dtype=torch.float32, device=dev)
        delta = torch.empty(N, heads, dtype=torch.float32, device=dev)
        # (the bias gradient -- column sums of g -- rides along: every g row passes through the destination pass once)
        g_bias = torch.empty(HC, dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.load().qot_gat_bn_partials_floats(N, heads, C), dtype=torch.float32, device=dev)
        _lib.call("qot_gat_bwd_dst", P(g), P(z), P(a_src), P(a_dst), P(stats), P(graph.rowptr), P(graph.col),
                  P(gad), P(escr), P(delta), N, heads, C, ns, P(g_bias), P(ws))
        # the source pass also sends the logit gradients back into grad_z through the attention vectors
        _lib.call("qot_gat_bwd_src", P(g), P(a_src), P(a_dst), P(escr), P(delta), P(graph.rowptr_t),
                  P(graph.col_t), P(graph.pos_t), P(gz), P(gas), N, heads, C, ns, P(att_s), P(att_d), P(gad))
        g_att = torch.empty(2, HC, dtype=torch.float32, device=dev)
        _lib.call("qot_gat_att_grad", P(z), P(gas), P(gad), _off(g_att, 0), _off(g_att, HC), P(ws), N, heads, C)
        return gz, g_att[0].view(ctx.att_shape), g_att[1].view(ctx.att_shape), g_bias, None, None, None, None, None


def gat_thin_ok(x: torch.Tensor, heads: int, C: int) -> bool:
    """GATConv with a handful of input features (LightpathGNN's first layer: 5) on the thin kernels -- the projection formed
    inside the attention kernels, ``z`` never materialised (``GatThinFn``).  Large batches only: the handful of tiny launches
    the logits and the attention-vector gradients take costs more than it saves when the step is launch-bound
    (``QOT_GAT_THIN_MIN_ROWS``, default 32768); ``QOT_NO_GAT_THIN=1`` switches it off."""
    if os.environ.get("QOT_NO_GAT_THIN", "0") == "1" or not x.is_cuda or x.dtype != torch.float32 or x.requires_grad:
        return False
    if x.dim() != 2 or x.shape[0] < int(os.environ.get("QOT_GAT_THIN_MIN_ROWS", "32768")) or not skinny_ok(x.shape[1], 4):
        return False
    return bool(_lib.load().qot_gat_thin_supported(heads, C, x.shape[1]))


def _skinny_dw(g, x):
    """``g^T x`` for ``x [N, F <= 8]`` (fixed-order per-workgroup partials + row sum), ``[C, F]``."""
    N, C = g.shape
    F = x.shape[1]
    nblk = _lib.load().qot_skinny_linear_dw_blocks(N)
    part = torch.empty(nblk, C * F, dtype=torch.float32, device=g.device)
    _lib.call("qot_skinny_linear_dw", P(g), P(x), P(part), N, F, C)
    gw = torch.empty(C * F, dtype=torch.float32, device=g.device)
    _lib.run_roles([_lib.make_role(_lib.ROLE_SUM_ROWS, (part, gw), (nblk, C * F, 0))])
    return gw.view(C, F)


class GatThinFn(torch.autograd.Function):
    """``GATConv`` on raw node features ``x [N, F <= 8]`` with the projection inside the attention kernels
    (``qot_gat_fwd_thin`` / ``qot_gat_bwd_dst_thin``): ``z = x W^T`` is never written or read -- the kernels form a row where
    the dense ones read it (F scalars of ``x`` and 4 F FMAs per float4, ``W^T`` in LDS).  The logits are
    ``a[n, h] = x_n . (W_h^T att_h)`` (two products of width 4), the attention vectors' gradient
    ``sum_n ga[n, h] z[n, h, :] = W_h (ga[:, h]^T x)``; the weight gradient stays ``grad_z^T x`` with ``grad_z`` from the source
    pass (which carries the logits' share).  Same mathematics as ``SkinnyLinearFn`` + ``GatFn``, other rounding of ``z``
    (its sum runs over the F features in another order).  ``x`` gets no gradient."""

    @staticmethod
    def forward(ctx, x, lin_weight, att_src, att_dst, bias, graph: GraphIndex, neg_slope: float, bn_stats: bool = False):
        require_cuda(x, lin_weight, att_src, att_dst, bias)
        x, w, bias = _f32c(x), _f32c(lin_weight), _f32c(bias)
        heads, C = att_src.shape[-2], att_src.shape[-1]
        N, F = x.shape
        HC = heads * C
        dev = x.device
        w3 = w.view(heads, C, F)
        v_src = (w3 * _f32c(att_src).view(heads, C, 1)).sum(1).contiguous()          # [heads, F]
        v_dst = (w3 * _f32c(att_dst).view(heads, C, 1)).sum(1).contiguous()
        a_src = torch.empty(N, heads, dtype=torch.float32, device=dev)
        a_dst = torch.empty(N, heads, dtype=torch.float32, device=dev)
        _lib.call("qot_skinny_linear_fwd", P(x), P(v_src), P(a_src), N, F, heads)
        _lib.call("qot_skinny_linear_fwd", P(x), P(v_dst), P(a_dst), N, F, heads)
        out = torch.empty(N, HC, dtype=torch.float32, device=dev)
        stats = torch.empty(N, heads, 2, dtype=torch.float32, device=dev)
        partials = None
        if bn_stats and N > 0:
            partials = torch.empty(_lib.load().qot_gat_bn_partials_floats(N, heads, C), dtype=torch.float32, device=dev)
        _lib.call("qot_gat_fwd_thin", P(x), F, P(w), P(a_src), P(a_dst), P(bias), P(graph.rowptr), P(graph.col), P(out),
                  P(stats), N, heads, C, float(neg_slope), P(partials))
        ctx.save_for_backward(x, w, a_src, a_dst, stats, _f32c(att_src).reshape(-1), _f32c(att_dst).reshape(-1))
        ctx.graph, ctx.ns, ctx.att_shape = graph, float(neg_slope), tuple(att_src.shape)
        if bn_stats:
            if partials is None:
                partials = torch.empty(0, dtype=torch.float32, device=dev)
            ctx.mark_non_differentiable(partials)
            return out, partials
        return out

    @staticmethod
    def backward(ctx, g, _gp=None):
        x, w, a_src, a_dst, stats, att_s, att_d = ctx.saved_tensors
        graph, ns = ctx.graph, ctx.ns
        g = _f32c(g)
        N, F = x.shape
        heads = a_src.shape[1]
        HC = w.shape[0]
        C = HC // heads
        dev = x.device
        gad = torch.empty(N, heads, dtype=torch.float32, device=dev)
        gas = torch.empty(N, heads, dtype=torch.float32, device=dev)
        gz = torch.empty(N, HC, dtype=torch.float32, device=dev)
        escr = torch.empty(max(graph.cap, 1), heads, 2, dtype=torch.float32, device=dev)
        delta = torch.empty(N, heads, dtype=torch.float32, device=dev)
        g_bias = torch.empty(HC, dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.load().qot_gat_bn_partials_floats(N, heads, C), dtype=torch.float32, device=dev)
        _lib.call("qot_gat_bwd_dst_thin", P(g), P(x), F, P(w), P(a_src), P(a_dst), P(stats), P(graph.rowptr), P(graph.col),
                  P(gad), P(escr), P(delta), N, heads, C, ns, P(g_bias), P(ws))
        _lib.call("qot_gat_bwd_src", P(g), P(a_src), P(a_dst), P(escr), P(delta), P(graph.rowptr_t),
                  P(graph.col_t), P(graph.pos_t), P(gz), P(gas), N, heads, C, ns, P(att_s), P(att_d), P(gad))
        g_lin = _skinny_dw(gz, x)                                                    # [HC, F]
        w3 = w.view(heads, C, F)
        g_att_src = (w3 * _skinny_dw(gas, x).view(heads, 1, F)).sum(-1)              # [heads, C]
        g_att_dst = (w3 * _skinny_dw(gad, x).view(heads, 1, F)).sum(-1)
        return (None, g_lin, g_att_src.view(ctx.att_shape), g_att_dst.view(ctx.att_shape), g_bias, None, None, None)


# ------------------------------------------------------------------ BatchNorm (+ReLU) (a8)
def _dist_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_world_size()
    return None, 1


def _bn_statistics(x, running_mean, running_var, training, momentum, eps, sync, partials):
    """``(mean, rstd, n_tot or None, world)`` of BatchNorm over the node matrix ``x`` -- batch statistics (+ in-place
    running-statistics update) in training mode, running statistics in eval mode; see ``BnFn``."""
    N, C = x.shape
    dev = x.device
    dist, world = _dist_world() if (sync and training) else (None, 1)
    n_tot = None
    if training and world > 1:
        mean_l = torch.zeros(C, dtype=torch.float32, device=dev)
        rstd_l = torch.ones(C, dtype=torch.float32, device=dev)
        if N > 0:
            part = torch.empty(_lib.load().qot_bn_partials_floats(N, C), dtype=torch.float32, device=dev)
            _lib.call("qot_bn_stats", P(x), N, C, float(eps), 0.0, P(mean_l), P(rstd_l), None, None, P(part))
        m64 = mean_l.double()
        ex2 = (1.0 / rstd_l.double().square() - eps).clamp_min(0.0) + m64 * m64 if N > 0 else m64 * 0.0
        buf = torch.cat([m64 * N, ex2 * N, torch.tensor([float(N)], dtype=torch.float64, device=dev)])
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        n_tot = buf[-1].clamp_min(1.0)       # (no host read: an all-empty global batch normalises nothing)
        mean64 = buf[:C] / n_tot
        var64 = (buf[C:2 * C] / n_tot - mean64 * mean64).clamp_min(0.0)
        mean = mean64.float()
        rstd = torch.rsqrt(var64 + eps).float()
        with torch.no_grad():
            unbiased = var64 * (n_tot / (n_tot - 1).clamp_min(1.0))
            running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
            running_var.mul_(1.0 - momentum).add_(unbiased.float(), alpha=momentum)
    elif training:
        if N == 0:
            raise ValueError("BatchNorm in training mode needs at least one row")
        if N == 1:      # as torch.nn.BatchNorm1d in training mode
            raise ValueError(f"Expected more than 1 value per channel when training, got input size [1, {C}]")
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        rstd = torch.empty(C, dtype=torch.float32, device=dev)
        if partials is not None and partials[0].numel() > 0 and C % 4 == 0:
            part, shift = partials
            nblk = _lib.load().qot_gat_blocks(N, 4, C // 4)
            chunk = _lib.load().qot_gat_chunk_rows(N, 4, C // 4)
            _lib.call("qot_bn_stats_from_partials", P(_f32c(shift)), P(part), nblk, chunk, N, C, float(eps),
                      float(momentum), P(mean), P(rstd), P(running_mean), P(running_var))
        else:
            part = torch.empty(_lib.load().qot_bn_partials_floats(N, C), dtype=torch.float32, device=dev)
            _lib.call("qot_bn_stats", P(x), N, C, float(eps), float(momentum), P(mean), P(rstd),
                      P(running_mean), P(running_var), P(part))
    else:
        mean = running_mean.detach().to(torch.float32).contiguous()
        rstd = torch.rsqrt(running_var.detach().to(torch.float32) + eps).contiguous()
    return mean, rstd, n_tot, world


class BnFn(torch.autograd.Function):
    """BatchNorm1d over the node matrix with an optional fused ReLU.

    Training: batch statistics + in-place running-stat update (momentum, unbiased var).
    Eval: running statistics.  App. B.4.

    ``sync``: under ``torch.distributed`` (one process per GPU) the batch statistics are
    those of the GLOBAL batch, as in the single-process reference: forward all-reduces
    ``(n*mean, n*E[x^2], n)`` (one ``[2C+1]`` fp64 message), backward all-reduces ``(sum dy*xhat, sum dy)``
    (``[2C]``).  The parameter gradients returned stay local sums; the flat-gradient all-reduce
    averages them like every other parameter.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, relu, sync=False,
                partials=None):
        """``partials = (column partials, shift)`` from the producer of ``x`` (``GatFn(bn_stats=True)``; shift = its
        bias): the batch statistics come from them instead of a pass over ``x`` (single process, training mode)."""
        require_cuda(x, weight, bias)
        x, weight, bias = _f32c(x), _f32c(weight), _f32c(bias)
        N, C = x.shape
        mean, rstd, n_tot, world = _bn_statistics(x, running_mean, running_var, training, momentum, eps, sync, partials)
        y = torch.empty_like(x)
        if N > 0:
            _lib.call("qot_bn_apply", P(x), P(mean), P(rstd), P(weight), P(bias), P(y), N, C, int(relu))
        # the output is NOT kept for backward: the ReLU mask is recomputed from x (same expression, same bits)
        ctx.save_for_backward(x, bias, mean, rstd, weight, n_tot if n_tot is not None else torch.empty(0))
        ctx.cfg = (bool(training), bool(relu), world > 1)
        return y

    @staticmethod
    def backward(ctx, g):
        x, bias, mean, rstd, weight, n_tot = ctx.saved_tensors
        training, relu, synced = ctx.cfg
        gx, gw, gb = _bn_backward(_f32c(g), x, bias, mean, rstd, weight, n_tot, training, relu, synced)
        return gx, gw, gb, None, None, None, None, None, None, None, None


class BnRowsFn(torch.autograd.Function):
    """``act(BatchNorm(x))[idx]`` -- LightpathGNN normalises the whole node matrix and keeps the LUT nodes' rows
    (``lightpath_training/models.py:31-32`` then ``:35-40``); one row in ten at cfg3.  Same statistics, running-statistics
    update and distributed behaviour as ``BnFn`` (they are those of ALL rows); the normalised matrix is never formed: the
    forward writes the ``n`` consumed rows only, the backward takes its column sums over those rows (the others' gradient
    is zero) and writes the dense ``grad_x`` -- every row feels the batch statistics -- without a zero-filled ``[N, C]``
    gradient ever being scattered into or read.  ``grad_x`` is bit for bit ``BnFn`` + ``RowsGatherFn``'s for the same column
    sums; those differ from the dense path's in the last bit (same terms, other order).  ``idx32``: unique rows."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, relu, sync, partials, idx32):
        require_cuda(x, weight, bias, idx32)
        x, weight, bias = _f32c(x), _f32c(weight), _f32c(bias)
        N, C = x.shape
        n = idx32.numel()
        mean, rstd, n_tot, world = _bn_statistics(x, running_mean, running_var, training, momentum, eps, sync, partials)
        y = torch.empty(n, C, dtype=torch.float32, device=x.device)
        if n > 0:
            _lib.call("qot_bn_apply_rows", P(x), P(idx32), n, P(mean), P(rstd), P(weight), P(bias), P(y), C, int(relu))
        ctx.save_for_backward(x, bias, mean, rstd, weight, n_tot if n_tot is not None else torch.empty(0), idx32)
        ctx.cfg = (bool(training), bool(relu), world > 1)
        return y

    @staticmethod
    def backward(ctx, g):
        x, bias, mean, rstd, weight, n_tot, idx32 = ctx.saved_tensors
        training, relu, synced = ctx.cfg
        g = _f32c(g)
        N, C = x.shape
        n = idx32.numel()
        dev = x.device
        gw = torch.zeros(C, dtype=torch.float32, device=dev)
        gb = torch.zeros(C, dtype=torch.float32, device=dev)
        if n > 0:
            part = torch.empty(_lib.load().qot_bn_partials_floats(n, C), dtype=torch.float32, device=dev)
            _lib.call("qot_bn_bwd_reduce_rows", P(g), P(idx32), n, P(x), P(mean), P(rstd), P(gw), P(gb), C, int(relu),
                      P(part), P(weight), P(bias))
        if not training:              # running statistics: rows without a gradient get none
            gfull = torch.zeros(N, C, dtype=torch.float32, device=dev)
            if n > 0:
                _lib.call("qot_rows_scatter", P(g), P(idx32), P(gfull), n, C)
            gx = torch.empty_like(x)
            _lib.call("qot_bn_bwd_apply", P(gfull), None, P(x), P(mean), P(rstd), P(weight), P(gw), P(gb), P(gx), N, C,
                      int(relu), 0, P(bias))
            return gx, gw, gb, None, None, None, None, None, None, None, None, None
        gw_use, gb_use = gw, gb
        if synced:
            dist, _ = _dist_world()
            tot = torch.cat([gw, gb]).double()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            scale = float(N) / n_tot  # the kernel divides by the LOCAL row count: pre-scale the global sums
            gw_use = (tot[:C] * scale).float().contiguous()
            gb_use = (tot[C:] * scale).float().contiguous()
        gx = torch.empty_like(x)
        _lib.call("qot_bn_bwd_apply_rows", P(g), P(idx32), n, P(x), P(mean), P(rstd), P(weight), P(gw_use), P(gb_use), P(gx),
                  N, C, int(relu), P(bias))
        return gx, gw, gb, None, None, None, None, None, None, None, None, None


def _bn_backward(g, x, bias, mean, rstd, weight, n_tot, training, relu, synced):
    """``(grad_x, grad_weight, grad_bias)`` of ``y = [relu](BatchNorm(x))`` given ``g = d/dy``; the ReLU mask is recomputed
    from ``x``.  Shared by ``BnFn`` and ``BnLinearFn``."""
    N, C = x.shape
    dev = x.device
    gw = torch.zeros(C, dtype=torch.float32, device=dev)
    gb = torch.zeros(C, dtype=torch.float32, device=dev)
    gx = torch.empty_like(x)
    if N > 0:
        part = torch.empty(_lib.load().qot_bn_partials_floats(N, C), dtype=torch.float32, device=dev)
        _lib.call("qot_bn_bwd_reduce", P(g), None, P(x), P(mean), P(rstd), P(gw), P(gb), N, C, int(relu),
                  P(part), P(weight), P(bias))
    gw_use, gb_use = gw, gb
    if synced:
        dist, _ = _dist_world()
        tot = torch.cat([gw, gb]).double()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        # the kernel divides by the LOCAL row count: pre-scale the global sums by N_local / n_total
        scale = float(N) / n_tot
        gw_use = (tot[:C] * scale).float().contiguous()
        gb_use = (tot[C:] * scale).float().contiguous()
    if N > 0:
        _lib.call("qot_bn_bwd_apply", P(g), None, P(x), P(mean), P(rstd), P(weight), P(gw_use), P(gb_use), P(gx),
                  N, C, int(relu), int(training), P(bias))
    return gx, gw, gb


def gemm_nt(a, b, scale=None, shift=None, bias=None, att=None):
    """``a' @ b.T (+ bias)`` on ``qot_gemm_nt`` (``a'`` = ``relu(a * scale + shift)`` when given); ``a [M, K]``, ``b [N, K]``
    contiguous fp32, K a multiple of 32.  ``att = (att_src, att_dst)`` (flat ``[N]``, N = heads * 128): also returns
    GATConv's attention logits ``(a_src, a_dst)`` ``[M, heads]`` from the epilogue (``qot_gemm_nt_logits``)."""
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    if att is None:
        _lib.call("qot_gemm_nt", P(a), K, P(b), K, P(out), N, M, N, K, P(scale), P(shift), P(bias))
        return out
    heads = N // 128
    a_src = torch.empty(M, heads, dtype=torch.float32, device=a.device)
    a_dst = torch.empty(M, heads, dtype=torch.float32, device=a.device)
    _lib.call("qot_gemm_nt_logits", P(a), K, P(b), K, P(out), N, M, N, K, P(scale), P(shift), P(bias), P(att[0]), P(att[1]),
              P(a_src), P(a_dst))
    return out, a_src, a_dst


def logits_ok(out_features: int, heads: int) -> bool:
    """The projection's epilogue can form GATConv's attention logits: one 128-column output tile per head."""
    return heads >= 1 and out_features == heads * 128 and os.environ.get("QOT_NO_FUSED_LOGITS", "0") != "1"


def gemm_tn_planes(a_t, b_t, scale=None, shift=None):
    """``a_t.T @ b_t'`` for ``a_t [K, M]``, ``b_t [K, N]`` (``b_t'`` = ``relu(b_t * scale + shift)`` per column when given):
    split-K planes on ``qot_gemm_tn_planes`` summed in a fixed order."""
    K, M = a_t.shape
    N = b_t.shape[1]
    dev = a_t.device
    splits = _lib.load().qot_gemm_tn_splits(M, N, K)
    part = torch.empty(splits, M * N, dtype=torch.float32, device=dev)
    _lib.call("qot_gemm_tn_planes", P(a_t), M, P(b_t), N, P(part), M, N, K, splits, P(scale), P(shift))
    if splits == 1:
        return part.view(M, N)
    out = torch.empty(M * N, dtype=torch.float32, device=dev)
    _lib.run_roles([_lib.make_role(_lib.ROLE_SUM_ROWS, (part, out), (splits, M * N, 0))])
    return out.view(M, N)


def gemm_ok(k: int, n: int) -> bool:
    """Shapes ``csrc/gemm.hip`` takes: inner dimension a multiple of 32, widths multiples of 4."""
    return k % 32 == 0 and n % 4 == 0 and k >= 32


class SkinnyLinearFn(torch.autograd.Function):
    """``x @ W^T`` for a handful of input features (GATConv's first-layer ``lin``, ``in_channels`` = 5): a bandwidth
    kernel each way (``csrc/skinny.hip``) instead of padded library GEMMs.  ``x`` (the raw node features) gets no
    gradient."""

    @staticmethod
    def forward(ctx, x, weight, att_src=None, att_dst=None):
        """``att_src`` / ``att_dst`` (GATConv's, ``[1, heads, 128]``): also returns the attention logits ``(a_src, a_dst)``
        (non-differentiable here: ``GatFn.backward`` sends their gradient into ``grad_z``)."""
        require_cuda(x, weight)
        x, weight = _f32c(x), _f32c(weight)
        N, F = x.shape
        C = weight.shape[0]
        out = torch.empty(N, C, dtype=torch.float32, device=x.device)
        ctx.save_for_backward(x)
        ctx.dims = (N, F, C)
        if att_src is None:
            _lib.call("qot_skinny_linear_fwd", P(x), P(weight), P(out), N, F, C)
            return out
        a_src = torch.empty(N, C // 128, dtype=torch.float32, device=x.device)
        a_dst = torch.empty(N, C // 128, dtype=torch.float32, device=x.device)
        _lib.call("qot_skinny_linear_fwd_logits", P(x), P(weight), P(out), N, F, C, P(_f32c(att_src.detach()).reshape(-1)),
                  P(_f32c(att_dst.detach()).reshape(-1)), P(a_src), P(a_dst))
        ctx.mark_non_differentiable(a_src, a_dst)
        return out, a_src, a_dst

    @staticmethod
    def backward(ctx, g, *_unused):
        (x,) = ctx.saved_tensors
        N, F, C = ctx.dims
        g = _f32c(g)
        nblk = _lib.load().qot_skinny_linear_dw_blocks(N)
        part = torch.empty(nblk, C * F, dtype=torch.float32, device=g.device)
        _lib.call("qot_skinny_linear_dw", P(g), P(x), P(part), N, F, C)
        gw = torch.empty(C * F, dtype=torch.float32, device=g.device)
        _lib.run_roles([_lib.make_role(_lib.ROLE_SUM_ROWS, (part, gw), (nblk, C * F, 0))])
        return None, gw.view(C, F), None, None


def skinny_ok(f: int, c: int) -> bool:
    return 1 <= f <= 8 and c % 4 == 0 and 4 <= c <= 1024 and (256 // (c // 4)) * c * f * 4 <= 64 * 1024


def gemm_nt_ok(a: torch.Tensor, b: torch.Tensor) -> bool:
    """``a @ b.T`` fits ``qot_gemm_nt`` (``a [M, K]``, ``b [N, K]``)."""
    return a.is_cuda and a.dtype == torch.float32 and gemm_ok(a.shape[1], b.shape[0])


def _library_gemm() -> bool:
    """``QOT_LIBRARY_GEMM=1``: the large projections' forward / grad_x products through the BLAS library instead of
    ``qot_gemm_nt`` (comparison runs; ``tools/bench_gemm.py`` has both at cfg3's shape)."""
    return bool(os.environ.get("QOT_LIBRARY_GEMM"))


def grad_x_product(g: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """``g @ W`` ([N, out] x [out, in]) as ``qot_gemm_nt(g, W^T)``: the weight is small (<= 1 MB), its transpose one
    tiny copy, and both operands of the product are then k-contiguous."""
    if _library_gemm() or not gemm_nt_ok(g, weight.t()):
        return g @ weight
    return gemm_nt(g, weight.t().contiguous())


class GemmFn(torch.autograd.Function):
    """``x @ W^T`` (no bias) for GATConv's projection: forward and grad_x on ``qot_gemm_nt`` (256 x 256 tiles at cfg3's
    shape, ``csrc/gemm256.hip``), the weight gradient ``g^T x`` -- inner dimension = number of nodes -- on
    ``qot_gemm_tn_planes``."""

    @staticmethod
    def forward(ctx, x, weight):
        require_cuda(x, weight)
        x, weight = _f32c(x), _f32c(weight)
        ctx.save_for_backward(x, weight)
        if _library_gemm() or not gemm_nt_ok(x, weight):
            return x @ weight.t()
        return gemm_nt(x, weight)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _f32c(g)
        gx = grad_x_product(g, weight) if ctx.needs_input_grad[0] else None
        return gx, gemm_tn_planes(g, x)


class BnLinearFn(torch.autograd.Function):
    """``relu(BatchNorm(x)) @ W^T`` -- ``F.relu(norm(x))`` of one layer feeding ``GATConv.lin`` of the next
    (``lightpath_training/models.py:31-32`` then ``:13,30`` of the following block) with the normalised activations NEVER
    materialised: the projection applies ``relu(x * scale + shift)`` while it loads its operand (``qot_gemm_nt``), the
    weight gradient recomputes it the same way (``qot_gemm_tn_planes``), the BatchNorm backward recomputes the ReLU mask
    from ``x`` as ``BnFn`` does.  Statistics, running-statistics update and distributed behaviour are ``BnFn``'s."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, sync, partials, lin_weight,
                att_src=None, att_dst=None):
        """``att_src`` / ``att_dst`` (the next GATConv's, 128 channels per head): also returns that layer's attention logits
        ``(a_src, a_dst)`` from the product's epilogue (non-differentiable here, see ``SkinnyLinearFn``)."""
        require_cuda(x, weight, bias, lin_weight)
        x, weight, bias, lin_weight = _f32c(x), _f32c(weight), _f32c(bias), _f32c(lin_weight)
        N, C = x.shape
        mean, rstd, n_tot, world = _bn_statistics(x, running_mean, running_var, training, momentum, eps, sync, partials)
        scale = (rstd * weight).contiguous()                 # y = relu(x * scale + shift)
        shift = torch.addcmul(bias, mean, scale, value=-1.0).contiguous()
        ctx.save_for_backward(x, bias, mean, rstd, weight, n_tot if n_tot is not None else torch.empty(0), lin_weight,
                              scale, shift)
        ctx.cfg = (bool(training), world > 1)
        if att_src is not None and N > 0:
            z, a_src, a_dst = gemm_nt(x, lin_weight, scale, shift,
                                      att=(_f32c(att_src.detach()).reshape(-1), _f32c(att_dst.detach()).reshape(-1)))
            ctx.mark_non_differentiable(a_src, a_dst)
            return z, a_src, a_dst
        z = gemm_nt(x, lin_weight, scale, shift) if N > 0 else x.new_empty(0, lin_weight.shape[0])
        if att_src is not None:
            e = x.new_empty(0, lin_weight.shape[0] // 128)
            return z, e, e.clone()
        return z

    @staticmethod
    def backward(ctx, gz, *_unused):
        x, bias, mean, rstd, weight, n_tot, lin_weight, scale, shift = ctx.saved_tensors
        training, synced = ctx.cfg
        gz = _f32c(gz)
        N, C = x.shape
        if N == 0:
            if synced:      # an empty shard still takes part in the statistics' all-reduce the other ranks are waiting in
                dist, _ = _dist_world()
                dist.all_reduce(torch.zeros(2 * C, dtype=torch.float64, device=x.device), op=dist.ReduceOp.SUM)
            return (torch.zeros_like(x), torch.zeros_like(weight), torch.zeros_like(bias), None, None, None, None, None,
                    None, None, torch.zeros_like(lin_weight), None, None)
        gy = grad_x_product(gz, lin_weight)                  # [N, C]
        g_lin = gemm_tn_planes(gz, x, scale, shift)          # [out, C] = gz^T relu(bn(x))
        gx, gw, gb = _bn_backward(gy, x, bias, mean, rstd, weight, n_tot, training, True, synced)
        return gx, gw, gb, None, None, None, None, None, None, None, g_lin, None, None


# ------------------------------------------------------------------ LUT rows (a9)
class RowsGatherFn(torch.autograd.Function):
    """``x[idx]`` for a unique index list (boolean-mask selection) and its adjoint."""

    @staticmethod
    def forward(ctx, x, idx32):
        require_cuda(x, idx32)
        x = _f32c(x)
        n, C = idx32.numel(), x.shape[1]
        out = torch.empty(n, C, dtype=torch.float32, device=x.device)
        _lib.call("qot_rows_gather", P(x), P(idx32), P(out), n, C)
        ctx.save_for_backward(idx32)
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx32,) = ctx.saved_tensors
        g = _f32c(g)
        gx = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        _lib.call("qot_rows_scatter", P(g), P(idx32), P(gx), idx32.numel(), ctx.shape[1])
        return gx, None
