"""Batch packer / loader: the build-owned counterpart of the PyG ``DataLoader`` the reference
harness iterates (``topological_training/train.py:93-95,107``; ``lightpath_training/train.py``),
SURVEY.md section 8(f) rank 1.

What the reference does per step: 4 worker processes unpickle graphs and run PyG's ``Collater``
(``Batch.from_data_list``), the main process then calls ``data.to(device)`` -- a pageable,
synchronous H2D copy in front of every step.  Here:

* collate follows the same layout rules (``batch.Batch.from_data_list``) but
  writes straight into PINNED, preallocated staging buffers (two sets, ping-pong), so there is no
  per-step host allocation and the copy engine can DMA from them;
* the H2D copies of batch t+1 are issued on a side stream while step t computes; ``__next__`` only
  makes the consumer stream wait on the copy event (no host sync);
* device tensors come from a small ring of preallocated buffers, reused every ``depth`` batches.

The loader yields ``Batch`` objects with the attributes the models read
(``x/edge_index/edge_attr/batch/node_ids/y/ptr/num_graphs`` and the host-side
``uniform_node_ids`` hint).  With ``device`` on the CPU it degrades to plain collate (used by the
CPU tests and by the oracle).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .batch import Batch, Data

_FIELDS = ("x", "edge_index", "edge_attr", "y", "node_ids", "batch", "ptr", "edge_ptr")


class _Staging:
    """Growable pinned host buffers + matching device buffers for one in-flight batch."""

    def __init__(self, device: torch.device, pin: bool):
        self.device, self.pin = device, pin
        self.host = {}
        self.dev = {}
        self.event = torch.cuda.Event() if device.type == "cuda" else None

    def host_view(self, name: str, shape, dtype) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= int(s)
        buf = self.host.get(name)
        if buf is None or buf.dtype != dtype or buf.numel() < n:
            cap = max(n, 1)
            buf = torch.empty(int(cap * 1.25) + 16, dtype=dtype, pin_memory=self.pin)
            self.host[name] = buf
        return buf[:n].view(*shape)

    def dev_view(self, name: str, shape, dtype) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= int(s)
        buf = self.dev.get(name)
        if buf is None or buf.dtype != dtype or buf.numel() < n:
            buf = torch.empty(int(max(n, 1) * 1.25) + 16, dtype=dtype, device=self.device)
            self.dev[name] = buf
        return buf[:n].view(*shape)


def collate_into(graphs: Sequence[Data], st: Optional[_Staging]) -> Batch:
    """``Batch.from_data_list`` semantics, writing into staging buffers when given."""
    if st is None:
        return Batch.from_data_list(graphs)
    sizes = [g.num_nodes for g in graphs]
    edges = [g.num_edges for g in graphs]
    N, E, B = sum(sizes), sum(edges), len(graphs)
    out = Batch()
    out.num_graphs, out._num_nodes = B, N
    ptr = st.host_view("ptr", (B + 1,), torch.long)
    eptr = st.host_view("edge_ptr", (B + 1,), torch.long)
    eptr[0] = 0
    batch = st.host_view("batch", (N,), torch.long)
    ei = st.host_view("edge_index", (2, E), torch.long)
    off = 0
    eo = 0
    ptr[0] = 0
    has = lambda name: bool(graphs) and all(getattr(g, name, None) is not None for g in graphs)
    cat_fields = {}
    for name in ("x", "edge_attr", "node_ids", "y"):
        if has(name):
            first = getattr(graphs[0], name)
            rows = sum(getattr(g, name).shape[0] for g in graphs)
            cat_fields[name] = (st.host_view(name, (rows,) + tuple(first.shape[1:]), first.dtype), 0)
    for gi, g in enumerate(graphs):
        n, e = sizes[gi], edges[gi]
        batch[off:off + n] = gi
        if e:
            torch.add(g.edge_index, off, out=ei[:, eo:eo + e])
        for name, (buf, pos) in list(cat_fields.items()):
            t = getattr(g, name)
            buf[pos:pos + t.shape[0]] = t
            cat_fields[name] = (buf, pos + t.shape[0])
        off += n
        eo += e
        ptr[gi + 1] = off
        eptr[gi + 1] = eo
    out.ptr, out.batch, out.edge_index = ptr, batch, ei
    out.edge_ptr = eptr
    out.graph_sizes = (max(sizes, default=0), max(edges, default=0))
    out.has_self_loops = bool((ei[0] == ei[1]).any())
    for name in ("x", "edge_attr", "node_ids", "y"):
        setattr(out, name, cat_fields[name][0] if name in cat_fields else None)
    out.uniform_node_ids = None
    if graphs and out.node_ids is not None and len(set(sizes)) == 1 and sizes[0] > 0:
        n0 = sizes[0]
        if bool((out.node_ids.view(B, n0) == torch.arange(n0)).all()):
            out.uniform_node_ids = n0
    return out


class PackedGraphs:
    """Pre-tensorised shard: the whole dataset as flat tensors + per-graph offsets
    (SURVEY.md 8(f) rank 2).  The reference keeps ~1.5 M one-graph pickles and rebuilds every
    sample with per-edge Python loops (``topological_training/dataset.py:85-104``); converting once
    to this layout makes a batch of consecutive graphs a CONTIGUOUS SLICE of every attribute:

        node_ptr[G+1], edge_ptr[G+1]           graph g owns nodes node_ptr[g]:node_ptr[g+1], ...
        edge_index[2, E_total] (int64)         numbered within the shard (graph g's nodes start at node_ptr[g])
        edge_attr[E_total, D], node_ids[N_total], x[N_total, F], y[G * rows_per_graph, ...]

    ``pin()`` page-locks the storage so ``GraphLoader`` can DMA slices straight to the GPU: no
    per-step host collate, no staging copy; re-basing to batch-local numbering is one subtraction on
    the device.  ``PackedGraphs.from_data_list`` converts any list of ``Data``; ``__getitem__`` gives
    back single graphs (so it is also a drop-in ``dataset``).
    """

    def __init__(self, node_ptr, edge_ptr, edge_index, edge_attr=None, node_ids=None, x=None, y=None,
                 uniform_node_ids=None):
        self.node_ptr, self.edge_ptr, self.edge_index = node_ptr, edge_ptr, edge_index
        self.edge_attr, self.node_ids, self.x, self.y = edge_attr, node_ids, x, y
        self.uniform_node_ids = uniform_node_ids
        self.y_rows = 0 if y is None else y.shape[0] // max(len(self), 1)
        ncnt, ecnt = node_ptr[1:] - node_ptr[:-1], edge_ptr[1:] - edge_ptr[:-1]
        self.graph_sizes = (int(ncnt.max()) if ncnt.numel() else 0, int(ecnt.max()) if ecnt.numel() else 0)
        self.has_self_loops = bool((edge_index[0] == edge_index[1]).any()) if not edge_index.is_cuda else None
        self.pinned = False
        self.device = None          # set by to_device(): the shard lives in HBM

    def __len__(self):
        return self.node_ptr.numel() - 1

    @classmethod
    def from_data_list(cls, graphs: Sequence[Data]) -> "PackedGraphs":
        b = Batch.from_data_list(graphs)
        edges = torch.tensor([0] + [g.num_edges for g in graphs], dtype=torch.long).cumsum(0)
        return cls(b.ptr.clone(), edges, b.edge_index, b.edge_attr, b.node_ids, b.x, b.y, b.uniform_node_ids)

    @classmethod
    def from_batch(cls, b: Batch) -> "PackedGraphs":
        """A collated ``Batch`` (host tensors, with its ``edge_ptr``) as a shard: the same flat tensors, no copy."""
        if getattr(b, "edge_ptr", None) is None:
            raise ValueError("the batch carries no per-graph edge slices (edge_ptr)")
        return cls(b.ptr.clone(), b.edge_ptr.clone(), b.edge_index, b.edge_attr, b.node_ids, b.x, b.y,
                   getattr(b, "uniform_node_ids", None))

    def pin(self) -> "PackedGraphs":
        for name in ("edge_index", "edge_attr", "node_ids", "x", "y"):
            t = getattr(self, name)
            if t is not None and not t.is_pinned():
                setattr(self, name, t.contiguous().pin_memory())
        self.pinned = True
        return self

    # ---- HBM-resident shard -------------------------------------------------------------------
    def to_device(self, device="cuda") -> "PackedGraphs":
        """Keep the WHOLE shard in device memory (the reference's 1.5 M-graph dataset is < 10 GB as
        flat tensors; an MI355X has 288 GB).  ``GraphLoader`` then hands out batches that are views /
        one-kernel re-basings of it -- no host collate, no H2D copy, no staging -- and, with
        ``cache_batches=True``, keeps every batch object (and the graph index the model attaches to
        it) for the following epochs: the reference trains 35 epochs over fixed, unshuffled chunks
        (``train.py:79-95``), so from the second visit a step does no graph preparation at all."""
        dev = torch.device(device)
        out = PackedGraphs(self.node_ptr, self.edge_ptr,
                           self.edge_index.to(dev), None if self.edge_attr is None else self.edge_attr.to(dev),
                           None if self.node_ids is None else self.node_ids.to(dev),
                           None if self.x is None else self.x.to(dev), None if self.y is None else self.y.to(dev),
                           self.uniform_node_ids)
        sizes = (self.node_ptr[1:] - self.node_ptr[:-1])
        out.graph_of_node = torch.repeat_interleave(torch.arange(len(self)), sizes).to(dev)
        out.node_ptr_dev = self.node_ptr.to(dev)
        out.edge_ptr_dev = self.edge_ptr.to(dev)
        out.device = dev
        out._batch_cache = {}
        return out

    def device_batch(self, lo: int, hi: int, cache: bool = False) -> Batch:
        """Graphs [lo, hi) of a device-resident shard as a ``Batch`` (three small kernels, or none
        when the batch object is cached)."""
        if cache and (lo, hi) in self._batch_cache:
            return self._batch_cache[(lo, hi)]
        n0, n1 = int(self.node_ptr[lo]), int(self.node_ptr[hi])
        e0, e1 = int(self.edge_ptr[lo]), int(self.edge_ptr[hi])
        out = Batch()
        out.num_graphs, out._num_nodes = hi - lo, n1 - n0
        out.uniform_node_ids = self.uniform_node_ids
        out.edge_index = self.edge_index[:, e0:e1] - n0 if n0 else self.edge_index[:, e0:e1]
        out.edge_attr = None if self.edge_attr is None else self.edge_attr[e0:e1]
        out.node_ids = None if self.node_ids is None else self.node_ids[n0:n1]
        out.x = None if self.x is None else self.x[n0:n1]
        out.y = None if self.y is None else self.y[lo * self.y_rows:hi * self.y_rows]
        out.ptr = self.node_ptr_dev[lo:hi + 1] - n0
        out.edge_ptr = self.edge_ptr_dev[lo:hi + 1] - e0
        out.graph_sizes = self.graph_sizes
        out.has_self_loops = False if self.has_self_loops is False else None
        out.batch = self.graph_of_node[n0:n1] - lo
        if cache:
            self._batch_cache[(lo, hi)] = out
        return out

    def __getitem__(self, g: int) -> Data:
        n0, n1 = int(self.node_ptr[g]), int(self.node_ptr[g + 1])
        e0, e1 = int(self.edge_ptr[g]), int(self.edge_ptr[g + 1])
        d = Data(edge_index=self.edge_index[:, e0:e1] - n0, num_nodes=n1 - n0)
        d.edge_attr = None if self.edge_attr is None else self.edge_attr[e0:e1]
        d.node_ids = None if self.node_ids is None else self.node_ids[n0:n1]
        d.x = None if self.x is None else self.x[n0:n1]
        d.y = None if self.y is None else self.y[g * self.y_rows:(g + 1) * self.y_rows]
        return d

    def slice_batch(self, lo: int, hi: int, st: "_Staging", copy_stream) -> Batch:
        """Graphs [lo, hi) as a device ``Batch``: slices DMA'd on ``copy_stream`` into the slot's
        device buffers, indices re-based there (one kernel each for edge_index and batch)."""
        dev = st.device
        n0, n1 = int(self.node_ptr[lo]), int(self.node_ptr[hi])
        e0, e1 = int(self.edge_ptr[lo]), int(self.edge_ptr[hi])
        out = Batch()
        out.num_graphs, out._num_nodes = hi - lo, n1 - n0
        out.uniform_node_ids = self.uniform_node_ids
        with torch.cuda.stream(copy_stream):
            ei = st.dev_view("edge_index", (2, e1 - e0), torch.long)
            ei[0].copy_(self.edge_index[0, e0:e1], non_blocking=True)
            ei[1].copy_(self.edge_index[1, e0:e1], non_blocking=True)
            if n0:
                ei.sub_(n0)
            out.edge_index = ei
            for name, a, b_ in (("edge_attr", e0, e1), ("node_ids", n0, n1), ("x", n0, n1),
                                ("y", lo * self.y_rows, hi * self.y_rows)):
                src = getattr(self, name)
                if src is None:
                    setattr(out, name, None)
                    continue
                d = st.dev_view(name, (b_ - a,) + tuple(src.shape[1:]), src.dtype)
                d.copy_(src[a:b_], non_blocking=True)
                setattr(out, name, d)
            ptr = st.dev_view("ptr", (hi - lo + 1,), torch.long)
            ptr.copy_(self.node_ptr[lo:hi + 1], non_blocking=True)
            if n0:
                ptr.sub_(n0)
            out.ptr = ptr
            eptr = st.dev_view("edge_ptr", (hi - lo + 1,), torch.long)
            eptr.copy_(self.edge_ptr[lo:hi + 1], non_blocking=True)
            if e0:
                eptr.sub_(e0)
            out.edge_ptr = eptr
            out.graph_sizes = self.graph_sizes
            out.has_self_loops = False if self.has_self_loops is False else None
            counts = ptr[1:] - ptr[:-1]
            bt = st.dev_view("batch", (n1 - n0,), torch.long)
            bt.copy_(torch.repeat_interleave(torch.arange(hi - lo, device=dev), counts, output_size=n1 - n0))
            out.batch = bt
            st.event.record(copy_stream)
        out._qot_ready = st.event
        return out


class GraphLoader:
    """Iterates ``dataset`` (a sequence of ``Data``) in batches, device-resident and prefetched.

    ``GraphLoader(dataset, batch_size, shuffle=False, drop_last=False, device="cuda", depth=2)``
    mirrors ``DataLoader(dataset, batch_size=..., shuffle=...)`` of the reference harness; the
    number of worker processes has no counterpart (collation is a handful of tensor copies).
    """

    def __init__(self, dataset: Sequence[Data], batch_size: int, shuffle: bool = False, drop_last: bool = False,
                 device="cuda", depth: int = 2, generator: Optional[torch.Generator] = None,
                 indices: Optional[Sequence[int]] = None, cache_batches: bool = False,
                 batches: Optional[Sequence[Sequence[int]]] = None):
        """``indices`` restricts the loader to a subset of ``dataset`` (the role of
        ``torch.utils.data.Subset`` at ``topological_training/train.py:34-36,89``) without wrapping
        it, so a pinned ``PackedGraphs`` keeps its zero-collate path for consecutive ranges.
        ``batches`` gives the batches explicitly (one index list per yielded batch, in order; an EMPTY
        list yields ``None``): a data-parallel rank uses it to take its share of every global batch, so
        that all ranks iterate the same number of steps (``harness.run_epoch``)."""
        if batch_size < 1:
            raise ValueError("batch_size must be >= 1")
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        self.device = torch.device(device)
        self.generator = generator
        self.indices = None if indices is None else list(indices)
        self.cache_batches = cache_batches
        self.batches = None if batches is None else [list(b) for b in batches]
        self.cuda = self.device.type == "cuda"
        self.depth = max(2, depth) if self.cuda else 1
        self._stages = [_Staging(self.device, pin=True) for _ in range(self.depth)] if self.cuda else []
        self._copy_stream = torch.cuda.Stream(self.device) if self.cuda else None

    @property
    def num_samples(self) -> int:
        return len(self.dataset) if self.indices is None else len(self.indices)

    def __len__(self) -> int:
        if self.batches is not None:
            return len(self.batches)
        n = self.num_samples
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _order(self) -> List[int]:
        base = list(range(len(self.dataset))) if self.indices is None else self.indices
        if not self.shuffle:
            return list(base)
        return [base[i] for i in torch.randperm(len(base), generator=self.generator).tolist()]

    def _stage(self, idx: Sequence[int], slot: int) -> Batch:
        packed = isinstance(self.dataset, PackedGraphs)
        if self.cuda and packed and self.dataset.pinned and list(idx) == list(range(idx[0], idx[0] + len(idx))):
            # consecutive graphs of a pinned shard: DMA the slices, nothing to collate on the host
            st = self._stages[slot]
            self._copy_stream.wait_stream(torch.cuda.current_stream(self.device))
            return self.dataset.slice_batch(idx[0], idx[0] + len(idx), st, self._copy_stream)
        graphs = [self.dataset[i] for i in idx]
        if not self.cuda:
            return Batch.from_data_list(graphs)
        st = self._stages[slot]
        if st.event is not None and st.host:
            st.event.synchronize()     # the DMA that last read this slot's pinned buffers (depth batches ago)
        host = collate_into(graphs, st)
        dev = Batch()
        dev.num_graphs, dev._num_nodes, dev.uniform_node_ids = host.num_graphs, host._num_nodes, host.uniform_node_ids
        dev.graph_sizes = host.graph_sizes
        dev.has_self_loops = host.has_self_loops
        # the device buffers of this slot were last read `depth` batches ago: the copy stream must
        # not overwrite them before that step's kernels are done
        self._copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._copy_stream):
            for name in _FIELDS:
                h = getattr(host, name, None)
                if h is None:
                    setattr(dev, name, None)
                    continue
                d = st.dev_view(name, tuple(h.shape), h.dtype)
                d.copy_(h, non_blocking=True)
                setattr(dev, name, d)
            st.event.record(self._copy_stream)
        dev._qot_ready = st.event
        return dev

    def __iter__(self):
        if self.batches is not None:
            chunks = self.batches
        else:
            order = self._order()
            chunks = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
            if self.drop_last and chunks and len(chunks[-1]) < self.batch_size:
                chunks.pop()
        consecutive = lambda c: len(c) == 0 or list(c) == list(range(c[0], c[0] + len(c)))
        resident = isinstance(self.dataset, PackedGraphs) and self.dataset.device is not None
        if resident and all(consecutive(c) for c in chunks):
            # HBM-resident shard, consecutive graphs: batches are views of it (nothing to stage)
            for c in chunks:
                yield self.dataset.device_batch(c[0], c[0] + len(c), cache=self.cache_batches) if c else None
            return
        if not self.cuda:
            for c in chunks:
                yield self._stage(c, 0) if c else None
            return
        pending = []
        nxt = 0
        stage = lambda k: self._stage(chunks[k], k % self.depth) if chunks[k] else None
        for _ in range(min(self.depth - 1, len(chunks))):      # prime the pipeline
            pending.append(stage(nxt))
            nxt += 1
        while pending:
            cur = pending.pop(0)
            if cur is not None:
                torch.cuda.current_stream(self.device).wait_event(cur._qot_ready)   # device-side wait only
            if nxt < len(chunks):
                pending.append(stage(nxt))
                nxt += 1
            yield cur
