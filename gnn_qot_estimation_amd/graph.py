"""Device-side graph index (CSR by destination + CSC by source), built once per batch.

The reference re-derives this bookkeeping inside every PyG conv call
(``topological_training/models.py:53,57``; ``lightpath_training/models.py:30`` also rebuilds
the self-loop edge list each call).  Topology is static across layers and across
forward/backward, so the index is cached on the batch object.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import os

import torch

from . import _lib


@dataclass
class GraphIndex:
    num_nodes: int
    num_edges_in: int      # E of the caller's edge_index
    cap: int               # slots allocated (E, or E + N with GAT self loops)
    rowptr: torch.Tensor   # [N+1] int32, in-edges of node i are slots rowptr[i]:rowptr[i+1]
    col: torch.Tensor      # [cap] int32 source node of each slot
    eid: torch.Tensor      # [cap] int32 original edge id (-1 = inserted self loop)
    row: torch.Tensor      # [cap] int32 destination of each slot
    rowptr_t: torch.Tensor  # [N+1] int32, out-edges of node j
    col_t: torch.Tensor    # [cap] int32 destination of each out-edge
    pos_t: torch.Tensor    # [cap] int32 CSR slot of each out-edge
    eid_t: torch.Tensor    # [cap] int32 original edge id of each out-edge
    invdeg: torch.Tensor   # [N] fp32 1/max(in_degree,1)
    gat_self_loops: bool
    # by-products of the one-launch build of block-diagonal batches (None otherwise)
    ids32: Optional[torch.Tensor] = None    # [N] int32 node_ids
    colf: Optional[torch.Tensor] = None     # [E] int32 node_ids[col]
    colf_t: Optional[torch.Tensor] = None   # [E] int32 node_ids[col_t]
    ptr32: Optional[torch.Tensor] = None    # [B+1] int32 graph boundaries
    sizes: Optional[tuple] = None           # (max nodes, max edges) of one graph: the per-graph build has checked every slice
    status_pending: bool = False            # built with check=False: the caller still owes a check_index_status()


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.QotError(
                "the HIP message-passing path needs CUDA (ROCm) tensors; got a CPU tensor. "
                "There is no CPU fallback -- move the model and batch to the GPU."
            )


_STATUS = {}
# Every eager per-graph index build reads its status word back (one 4-byte host read, ~30 us).  A pipelined loader
# loop that must not synchronise per batch may set this False and call ``check_index_status(device)`` once per epoch.
CHECK_INDEX_STATUS = True


def _index_status(dev) -> torch.Tensor:
    """Per-device int32 flag word of ``qot_csr_build_by_graph`` (bit 0: an edge leaves its graph's node range, bit 1: a
    slice lies outside the arrays or exceeds the stated maximum).  Zeroed once and sticky: the kernel only ORs into
    it, so no fill launch rides in a captured step."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _STATUS.get(key)
    if st is None:
        st = _STATUS[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return st


def check_index_status(dev) -> None:
    """Raise if a per-graph index build on ``dev`` flagged its slices (the kernel then wrote nothing for that graph).
    One 4-byte host read; skipped while a stream capture is running (a captured step has been through an eager step
    with the same batch object, and the flag is sticky: the next eager build or an explicit call reports it)."""
    try:
        if torch.cuda.is_current_stream_capturing():
            return
    except Exception:
        pass
    st = _index_status(dev)
    code = int(st.item())
    if code:
        st.zero_()
        what = []
        if code & 1:
            what.append("an edge leaves its graph's node range")
        if code & 2:
            what.append("a graph slice lies outside the node / edge arrays or exceeds (max_nodes, max_edges)")
        if code & 4:
            what.append("edge_index holds a self loop although the batch says has_self_loops = False")
        raise _lib.QotError("qot_csr_build_by_graph: inconsistent batch slices (" + "; ".join(what) + "): ptr / edge_ptr "
                            "do not describe edge_index -- pass slices=None for the general build")


def build_graph_index(edge_index: torch.Tensor, num_nodes: int, gat_self_loops: bool = False,
                      slices=None, node_ids: Optional[torch.Tensor] = None, group=None, check: bool = True) -> GraphIndex:
    """``slices = (node_ptr, edge_ptr, max_nodes, max_edges)`` (int64 device tensors ``[B+1]`` and host
    ints) marks a block-diagonal batch whose graphs keep nodes and edges contiguous: the index is then
    built by one workgroup per graph in a single launch (``qot_csr_build_by_graph``).  ``group`` (a
    ``launch_group.LaunchGroup``): that launch is shared with the caller's other jobs -- the index is valid after
    ``group.run()`` (the general multi-launch build ignores ``group`` and runs at once)."""
    require_cuda(edge_index)
    if edge_index.dtype != torch.long or edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("edge_index must be int64 [2, E]")
    ei = edge_index.contiguous()
    dev = ei.device
    E = ei.shape[1]
    N = int(num_nodes)
    cap = E + (N if gat_self_loops else 0)
    i32 = dict(dtype=torch.int32, device=dev)
    g = GraphIndex(
        num_nodes=N, num_edges_in=E, cap=cap,
        rowptr=torch.empty(N + 1, **i32), col=torch.empty(max(cap, 1), **i32),
        eid=torch.empty(max(cap, 1), **i32), row=torch.empty(max(cap, 1), **i32),
        rowptr_t=torch.empty(N + 1, **i32), col_t=torch.empty(max(cap, 1), **i32),
        pos_t=torch.empty(max(cap, 1), **i32), eid_t=torch.empty(max(cap, 1), **i32),
        invdeg=torch.empty(max(N, 1), dtype=torch.float32, device=dev),
        gat_self_loops=gat_self_loops,
    )
    lib = _lib.load()
    if slices is not None and not gat_self_loops:
        node_ptr, edge_ptr, max_n, max_m = slices
        B = node_ptr.numel() - 1
        if (B >= 1 and edge_ptr.numel() == B + 1 and node_ptr.is_cuda and edge_ptr.is_cuda
                and node_ptr.dtype == torch.long and edge_ptr.dtype == torch.long
                and int(max_n) <= 65535 and (5 * int(max_n) + 2 + 6 * int(max_m)) * 4 <= 144 * 1024):
            g.ptr32 = torch.empty(B + 1, **i32)
            ids = None
            if node_ids is not None and node_ids.is_cuda and node_ids.dtype == torch.long and node_ids.numel() == N:
                ids = node_ids.contiguous()
                g.ids32 = torch.empty(N, **i32)
                g.colf, g.colf_t = torch.empty(max(E, 1), **i32), torch.empty(max(E, 1), **i32)
            # Named, so that both outlive the launch: `ptr(node_ptr.contiguous()), ptr(edge_ptr.contiguous())` hands the
            # kernel two pointers into ONE block when the inputs are strided views (the first temporary is freed and its
            # block reused by the second) -- the round-2 GPU fault, DESIGN.md section 8.  Tensors go to `_lib.call`
            # as tensors; it takes the pointers itself while its argument tuple keeps them alive.
            np_c, ep_c = node_ptr.contiguous(), edge_ptr.contiguous()
            status = _index_status(dev)
            g.sizes = (int(max_n), int(max_m))
            if group is not None and B >= 1:
                group.add(_lib.ROLE_CSR_BY_GRAPH, (ei, np_c, ep_c, g.rowptr, g.col, g.eid, g.row, g.rowptr_t, g.col_t,
                                                   g.pos_t, g.eid_t, g.invdeg, status, ids, g.ids32, g.colf, g.colf_t,
                                                   g.ptr32), (E, N, B, int(max_n), int(max_m)))
                if CHECK_INDEX_STATUS:
                    group.post.append(lambda: check_index_status(dev))
                return g
            _lib.call("qot_csr_build_by_graph", ei, E, N, np_c, ep_c, B, int(max_n), int(max_m), g.rowptr, g.col,
                      g.eid, g.row, g.rowptr_t, g.col_t, g.pos_t, g.eid_t, g.invdeg, status, ids, g.ids32, g.colf,
                      g.colf_t, g.ptr32)
            if CHECK_INDEX_STATUS:
                check_index_status(dev)
            return g
    if slices is not None and gat_self_loops:
        # GATConv's self-looped index of a batch of small graphs without self loops (the caller vouches for that:
        # graph_index_for passes slices in GAT mode only for has_self_loops == False): one launch, one wave per graph
        node_ptr, edge_ptr, max_n, max_m = slices
        B = node_ptr.numel() - 1
        if (B >= 1 and edge_ptr.numel() == B + 1 and node_ptr.is_cuda and edge_ptr.is_cuda and node_ptr.dtype == torch.long
                and edge_ptr.dtype == torch.long and lib.qot_csr_gat_by_graph_supported(int(max_n), int(max_m))):
            g.ptr32 = torch.empty(B + 1, **i32)
            np_c, ep_c = node_ptr.contiguous(), edge_ptr.contiguous()     # named: see above
            status = _index_status(dev)
            _lib.call("qot_csr_build_gat_by_graph", ei, E, N, np_c, ep_c, B, int(max_n), int(max_m), g.rowptr, g.col, g.eid,
                      g.row, g.rowptr_t, g.col_t, g.pos_t, g.eid_t, g.invdeg, status, g.ptr32)
            if CHECK_INDEX_STATUS and check:     # check=False: the caller reads the flag itself, behind its next host sync
                check_index_status(dev)
            else:
                g.status_pending = CHECK_INDEX_STATUS
            return g
    ws_bytes = lib.qot_csr_workspace_bytes(E, N, int(gat_self_loops))
    if ws_bytes == 0:
        raise _lib.QotError("qot_csr_workspace_bytes failed")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.call("qot_csr_build", _lib.ptr(ei), E, N, int(gat_self_loops), _lib.ptr(g.rowptr), _lib.ptr(g.col),
              _lib.ptr(g.eid), _lib.ptr(g.row), _lib.ptr(g.rowptr_t), _lib.ptr(g.col_t), _lib.ptr(g.pos_t),
              _lib.ptr(g.eid_t), _lib.ptr(g.invdeg), _lib.ptr(ws), ws_bytes)
    return g


def _cache(data) -> Optional[dict]:
    c = getattr(data, "_qot_cache", None)
    if c is None:
        c = {}
        try:
            setattr(data, "_qot_cache", c)
        except Exception:
            return None
    return c


def graph_index_for(data, num_nodes: int, gat_self_loops: bool = False, group=None, check: bool = True) -> GraphIndex:
    """Cached ``GraphIndex`` of ``data.edge_index`` (rebuilt if the tensor changed).  ``group``: see ``build_graph_index``.
    ``check=False`` (GAT mode): the one-launch build's status word is not read back here -- the caller calls
    ``check_index_status`` itself, behind a host synchronisation it has anyway."""
    ei = data.edge_index
    key = ("graph", bool(gat_self_loops))
    tag = (ei.data_ptr(), ei._version, tuple(ei.shape), int(num_nodes))
    c = _cache(data)
    if c is not None and key in c and c[key][0] == tag:
        return c[key][1]
    slices = None
    ptr, eptr, sizes = getattr(data, "ptr", None), getattr(data, "edge_ptr", None), getattr(data, "graph_sizes", None)
    if eptr is None and not gat_self_loops and ptr is not None:
        # a PyG Batch keeps the same per-graph edge slices in `_slice_dict["edge_index"]` (host tensor)
        sd = getattr(data, "_slice_dict", None)
        es = sd.get("edge_index") if isinstance(sd, dict) else None
        if isinstance(es, torch.Tensor) and es.dim() == 1 and es.numel() == ptr.numel() and es.numel() >= 2:
            es = es.to(torch.long).cpu()
            pc = ptr.detach().to("cpu", torch.long) if c is None or "ptr_host" not in c else c["ptr_host"]
            eptr = es.to(ei.device)
            sizes = (int((pc[1:] - pc[:-1]).max()), int((es[1:] - es[:-1]).max()))
            if c is not None:
                c["ptr_host"] = pc
    if (ptr is not None and eptr is not None and sizes is not None and ptr.numel() == eptr.numel() and ptr.is_cuda
            and eptr.is_cuda and (not gat_self_loops or (getattr(data, "has_self_loops", None) is False
                                                         and os.environ.get("QOT_NO_GAT_BY_GRAPH", "0") != "1"))):
        slices = (ptr, eptr, sizes[0], sizes[1])
    ids = getattr(data, "node_ids", None) if (slices is not None and getattr(data, "uniform_node_ids", None)) else None
    g = build_graph_index(ei, num_nodes, gat_self_loops, slices, ids, group=group, check=check)
    if c is not None:
        def remember():
            c[key] = (tag, g)
            if g.ptr32 is not None:       # read-out boundaries came with the index
                c["ptr32"] = ((ptr.data_ptr(), ptr._version, tuple(ptr.shape)), (g.ptr32, ptr.numel() - 1))
        if group is not None and group.roles:
            # the build is still WAITING in the caller's multi-role launch: the index enters the cache only after that
            # launch has run and its status read-back has passed -- a batch with inconsistent slices (QotError) or a forward
            # that raises before the launch must not leave torch.empty arrays behind for the next forward to gather through
            group.on_success.append(remember)
        else:
            remember()
    return g


def to_i32(t: torch.Tensor) -> torch.Tensor:
    require_cuda(t)
    t = t.contiguous()
    if t.dtype == torch.int32:
        return t
    if t.dtype != torch.long:
        raise ValueError("expected an int64 index tensor")
    out = torch.empty(t.shape, dtype=torch.int32, device=t.device)
    _lib.call("qot_i64_to_i32", _lib.ptr(t), _lib.ptr(out), t.numel())
    return out


def batch_index_for(data, num_nodes: int):
    """(batch32 [N], ptr32 [B+1], B).  ``B`` comes from ``data.num_graphs`` when the batch
    object carries it; otherwise ``batch.max()+1`` as PyG's global_mean_pool does
    (``topological_training/models.py:61``) -- that one costs a device sync."""
    batch = data.batch
    tag = (batch.data_ptr(), batch._version, tuple(batch.shape))
    c = _cache(data)
    if c is not None and "batch" in c and c["batch"][0] == tag:
        return c["batch"][1]
    B = getattr(data, "num_graphs", None)
    if B is None:
        B = int(batch.max().item()) + 1 if batch.numel() else 0
    B = int(B)
    b32 = to_i32(batch)
    ptr = torch.empty(B + 1, dtype=torch.int32, device=batch.device)
    _lib.call("qot_batch_ptr", _lib.ptr(b32), int(num_nodes), B, _lib.ptr(ptr))
    res = (b32, ptr, B)
    if c is not None:
        c["batch"] = (tag, res)
    return res


def batch_ptr_for(data, num_nodes: int):
    """(ptr32 [B+1], B): graph boundaries for the pooled read-out.  A batch object that carries
    ``ptr`` (ours and PyG's do) only needs it narrowed to int32; otherwise it is derived from
    ``batch`` as in ``batch_index_for``."""
    ptr = getattr(data, "ptr", None)
    B = getattr(data, "num_graphs", None)
    if ptr is None or B is None or ptr.numel() != int(B) + 1 or not ptr.is_cuda:
        _, p32, B = batch_index_for(data, num_nodes)
        return p32, B
    tag = (ptr.data_ptr(), ptr._version, tuple(ptr.shape))
    c = _cache(data)
    if c is not None and "ptr32" in c and c["ptr32"][0] == tag:
        return c["ptr32"][1]
    res = (to_i32(ptr), int(B))
    if c is not None:
        c["ptr32"] = (tag, res)
    return res


def cached_i32(data, name: str) -> torch.Tensor:
    t = getattr(data, name)
    tag = (t.data_ptr(), t._version, tuple(t.shape))
    c = _cache(data)
    key = ("i32", name)
    if c is not None and key in c and c[key][0] == tag:
        return c[key][1]
    out = to_i32(t)
    if c is not None:
        c[key] = (tag, out)
    return out


def _detect_uniform_node_ids(data, N: int) -> None:
    """A batch object without the ``uniform_node_ids`` hint (e.g. a PyG ``Batch``): look at the device
    tensors once -- equal graph sizes and ``node_ids == arange(n)`` in every graph, what the reference's
    dataset emits (``topological_training/dataset.py:78``) -- and remember the answer on the object.
    One host read per batch object (~30 us) against the ~0.2 ms table mode saves at batch 1024; skipped
    while a stream capture is running (a capture must not synchronise)."""
    ids, ptr, B = getattr(data, "node_ids", None), getattr(data, "ptr", None), getattr(data, "num_graphs", None)
    found = None
    try:
        capturing = torch.cuda.is_current_stream_capturing()
    except Exception:
        capturing = False
    if (not capturing and ids is not None and ptr is not None and B and ids.is_cuda and ptr.is_cuda
            and ids.numel() == N and N % int(B) == 0 and ptr.numel() == int(B) + 1):
        n = N // int(B)
        ar = torch.arange(n, device=ids.device, dtype=ids.dtype)
        ok = (ids.view(int(B), n) == ar).all() & (ptr == torch.arange(int(B) + 1, device=ptr.device, dtype=ptr.dtype) * n).all()
        if bool(ok.item()):
            found = n
    if capturing:
        return
    try:
        data.uniform_node_ids = found
    except Exception:
        pass


def table_maps_for(data, graph: GraphIndex, group=None):
    """(rowmap, colf, colf_t, (B, n)) for TransformerConv's table mode, or None.

    Table mode needs ``node_ids == arange(n)`` repeated for every graph (what the reference's
    dataset emits: ``topological_training/dataset.py:78``) -- then the gradient of the projected
    table is a plain sum over graphs.  The batch object advertises it as ``uniform_node_ids = n``
    (set by ``Batch.from_data_list`` on the host); otherwise the general per-node path runs.
    One launch builds the int32 node ids and both gathered column maps.
    """
    if not hasattr(data, "uniform_node_ids"):
        _detect_uniform_node_ids(data, graph.num_nodes)
    n = getattr(data, "uniform_node_ids", None)
    if not n:
        return None
    N = graph.num_nodes
    if N % n != 0:
        return None
    ids = data.node_ids
    tag = (ids.data_ptr(), ids._version, graph.col.data_ptr())
    c = _cache(data)
    if c is not None and "tmaps" in c and c["tmaps"][0] == tag:
        return c["tmaps"][1]
    if graph.colf is not None and graph.ids32 is not None:      # built together with the index
        res = (graph.ids32, graph.colf, graph.colf_t, (N // n, int(n)))
        if c is not None:
            if group is not None and group.roles:              # as the index itself: cached once the launch went through
                group.on_success.append(lambda: c.__setitem__("tmaps", (tag, res)))
            else:
                c["tmaps"] = (tag, res)
        return res
    require_cuda(ids)
    if group is not None and group.roles:
        group.run()                   # the map kernel below reads graph.col: the index build must not wait in the group
    if ids.dtype != torch.int64:
        ids = ids.long()
    ids = ids.contiguous()
    E = graph.num_edges_in            # table mode is used without GAT self loops: every slot is live
    dev = ids.device
    ids32 = torch.empty(N, dtype=torch.int32, device=dev)
    colf = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    colf_t = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    _lib.call("qot_table_maps", _lib.ptr(ids), _lib.ptr(graph.col), _lib.ptr(graph.col_t), _lib.ptr(ids32),
              _lib.ptr(colf), _lib.ptr(colf_t), N, E)
    res = (ids32, colf, colf_t, (N // n, int(n)))
    if c is not None:
        c["tmaps"] = (tag, res)
    return res
