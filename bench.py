#!/usr/bin/env python3
"""Headline benchmark: graphs/sec of a full TopologicalGNN train step on MI355X.

Workload (BASELINE.json configs[1]): 2-layer GNN (TransformerConv + NNConv),
hidden=64, synthetic 100-node / 400-directed-edge topologies, batch=1024 graphs PER GPU
(weak scaling), fp32, dropout p=0.5 on, SGD(lr .1, momentum .9), SmoothL1 loss.
One "step" = {CSR build, zero_grad, forward, loss, backward, (gradient all-reduce), SGD}
-- the body of topological_training/train.py:109-116 -- with the batch resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  ``roofline`` is for the dominant hand-written kernel, timed
live with events on the launch stream; ``cpu_baseline`` is the oracle's PyG-style CPU step
(kind "port": torch_geometric itself is not installable here) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_*_f32)

CFG = dict(cfg=2, B=1024, n=100, e=400, H=64, D=4, out=3, layers=2, dropout=0.5)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)       # ~30 ms: the part needs ~12 ms of load to reach its clock
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--steps-per-graph", type=int, default=4,
                    help="train steps captured into one HIP graph at N=1 (a replay costs ~8 us of launch gap on this "
                         "part whatever the graph holds; with an exchange step between backward and update -- N>1 -- it is 1)")
    ap.add_argument("--cpu-sample-graphs", type=int, default=64)
    ap.add_argument("--no-lightpath", action="store_true", help="skip the separate LightpathGNN (configs[2]) measurement")
    ap.add_argument("--lightpath-steps", type=int, default=10)
    ap.add_argument("--no-reference-scale", action="store_true",
                    help="skip the reference-scale (V=75 H=16 B=512 / F=5 C=32 B=512) eager and replayed step times")
    return ap.parse_args()


def make_batch(rank, device, B=None):
    from gnn_qot_estimation_amd import synthetic as S
    B = B or CFG["B"]
    # every graph distinct (rng seed 1234 + 1000 cfg + graph index; ranks take disjoint index ranges).
    # Host-side generation of 1024 graphs takes ~1 s, once, outside the timed region.
    return S.topological_batch(CFG["cfg"], B, n=CFG["n"], e=CFG["e"], edge_dim=CFG["D"],
                               first_graph=rank * B).to(device)


def build_model(device):
    import gnn_qot_estimation_amd as q
    torch.manual_seed(0)
    m = q.TopologicalGNN(CFG["n"], CFG["H"], CFG["out"], CFG["D"], dropout_p=CFG["dropout"],
                         num_layers=CFG["layers"]).to(device)
    m.train()
    return m


class TrainStep:
    """zero_grad -> forward -> SmoothL1 -> backward -> all-reduce -> SGD, eager or HIP-graph."""

    def __init__(self, model, batch, world, use_graph, steps_per_graph=1):
        from gnn_qot_estimation_amd.dp import FlatModel, FusedSGD
        global QF
        from gnn_qot_estimation_amd import functional as QF
        self.model, self.batch, self.world = model, batch, world
        self.flat = FlatModel(model)
        self.flat.broadcast_params()
        self.opt = FusedSGD(self.flat, lr=0.1, momentum=0.9)
        self.y = batch.y.view(-1, CFG["out"])
        self.loss = torch.zeros((), device=batch.y.device)
        self.graph_fb = None
        self.graph_opt = None
        self.graph_multi = None              # steps_per_graph whole steps in one graph
        # N > 1 (or BENCH_FORCE_DP=1: the same code path on a single-rank group, how the one-GPU box exercises it): the
        # step has an exchange -- pack, ONE flat-gradient all-reduce -- between backward and update
        self.dp = world > 1 or bool(os.environ.get("BENCH_FORCE_DP"))
        self.spg = max(1, int(steps_per_graph))
        self.use_graph = use_graph
        self.collective_in_graph = None      # True: the RCCL call is captured with the rest of the step
        self.capture_error = None

    def _fwd_bwd(self):
        if not os.environ.get("BENCH_PREP_OUTSIDE"):
            self.batch._qot_cache = {}       # graph prep (CSR/CSC build) is part of every step
        self.flat.detach_grads()             # zero_grad(set_to_none=True): autograd assigns, no add kernels
        # forward with the criterion (SmoothL1Loss, mean) folded into the read-out head's kernel: out, loss value
        # (written with the backward epilogue) and d loss / d out; then backward from the model output
        out, _, g = self.model.forward_loss(self.batch, self.y, loss_out=self.loss)
        out.backward(g)
        if self.dp:
            self.flat.gather_grads()         # one kernel packs all gradients into the flat buffer RCCL reduces

    def _update(self):
        # one rank: the pack rides in the update kernel (gradients read from their own tensors)
        self.opt.step(grads=None if self.dp else True)

    def _exchange(self):
        if self.dp:
            self.flat.all_reduce_grads(force=True)

    def _eager(self):
        self._fwd_bwd()
        self._exchange()
        self._update()

    def capture(self):
        """Capture WHOLE steps -- forward, backward, pack, the RCCL all-reduce of the flat gradient, update -- as one HIP
        graph holding ``steps_per_graph`` of them, at every N (r04: RCCL 2.26 takes part in stream capture;
        tests/test_gpu_dp.py).  A backend that cannot be captured (gloo in the CPU rehearsal) falls back to two graphs
        with the eager collective between them, one step per replay; the line says which (``collective_in_graph``)."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        capturable = not self.dp or (dist.is_initialized() and dist.get_backend() == "nccl")
        # with a collective inside, the capture is thread-local: the process group's watchdog thread polls its events while
        # this thread captures, and in the default (global) mode a call from ANOTHER thread can invalidate a capture; one
        # retry before falling back (the forced-DP bench test failed once in this round's ~25 suite runs, cause not seen)
        mode = dict(capture_error_mode="thread_local") if self.dp else {}
        for attempt in range(2 if (capturable and self.dp) else 1):
            if not capturable:
                break
            try:
                self.graph_fb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_fb, **mode):
                    self._eager()
                if self.spg > 1:             # several whole steps per replay: every step still runs every kernel
                    self.graph_multi = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph_multi, pool=self.graph_fb.pool(), **mode):
                        for _ in range(self.spg):
                            self._eager()
                self.collective_in_graph = self.dp
                return
            except Exception as e:           # noqa: BLE001 -- the runtime's refusal is recorded, the fallback measured
                if not self.dp:
                    raise
                self.capture_error = repr(e)[:400]
                self.graph_fb = self.graph_multi = None
                torch.cuda.synchronize()
        if not capturable:
            self.capture_error = f"backend {dist.get_backend() if dist.is_initialized() else None} is not stream-capturable"
        self.collective_in_graph = False
        self.spg = 1
        self.graph_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_fb):
            self._fwd_bwd()
        self.graph_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_opt):
            self._update()

    def __call__(self):
        if self.graph_fb is None:
            self._eager()
        else:
            self.graph_fb.replay()
            if self.graph_opt is not None:
                self._exchange()
                self.graph_opt.replay()

    def run(self, n):
        """``n`` train steps: replays of the multi-step graph, the remainder step by step."""
        if self.graph_multi is not None:
            while n >= self.spg:
                self.graph_multi.replay()
                n -= self.spg
        for _ in range(n):
            self()


def event_time_ms(fn, iters=20, warm=3, settle=False):
    """Average duration of ``fn``'s launches between two events on the launch stream.  The warm-up runs straight into the
    timed launches (no synchronisation in between).  ``settle``: after an idle gap the part takes 60-70 launches (~12 ms)
    to come back to its clock (rocprof trace of this file: the same kernel 192 -> 175 us over 70 standalone launches,
    175 us inside the replayed step), so chunks of ``iters`` launches are timed back to back until two consecutive chunks
    agree within 1 % (at most 10 chunks) and the last one is reported."""
    torch.cuda.synchronize()
    for _ in range(warm):
        fn()
    prev = None
    for _ in range(10 if settle else 1):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(iters):
            fn()
        en.record()
        en.synchronize()
        cur = st.elapsed_time(en) / iters
        if prev is not None and abs(cur - prev) <= 0.01 * cur:
            break
        prev = cur
    return cur


def kernel_table(model, batch):
    """Time each hand-written kernel of the step standalone (events on the launch stream) and
    price it with the ALGORITHMIC bytes / flops of DESIGN.md's kernel table (each distinct input
    element read once, each output written once, int32 graph structure; 2*M*N*K per GEMM)."""
    from gnn_qot_estimation_amd import _lib
    from gnn_qot_estimation_amd.functional import nnconv_perm_index, nnconv_gradh_perm_index
    from gnn_qot_estimation_amd.graph import build_graph_index
    P = _lib.ptr
    lib = _lib.load()
    dev = batch.edge_attr.device
    N, E, H, D = batch.num_nodes, batch.num_edges, CFG["H"], CFG["D"]
    K = 2 * D
    KT = (K + 2) * H
    g = build_graph_index(batch.edge_index, N)
    f = lambda *s: torch.randn(*s, device=dev)
    qkvs, gout, x = f(N, 4 * H), f(N, H), f(N, H)
    ea = batch.edge_attr
    we, w1, b1, bias = f(H, D), f(K, D), f(K), f(H)
    out, stats = torch.empty(N, H, device=dev), torch.empty(N, 2, device=dev)
    gq = torch.empty(N, 4 * H, device=dev)
    escr, delta = torch.empty(E, 2, device=dev), torch.empty(N, device=dev)
    pds, pal = torch.empty(N, D, device=dev), torch.empty(N, D, device=dev)
    wp = f(KT * H)[nnconv_perm_index(KT, dev)].contiguous()
    bp = f(K * H * H)[nnconv_gradh_perm_index(K, dev)].contiguous()
    gw1, gb1 = torch.zeros(K, D, device=dev), torch.zeros(K, device=dev)
    ws_h = torch.empty(lib.qot_nnconv_gradh_workspace_floats(D), device=dev)
    off = lambda t, k: t.data_ptr() + 4 * k
    csr = 4 * E + 4 * (N + 1)
    rows = []

    def add(name, fn, bytes_, flops=0, bound="hbm"):
        ms = event_time_ms(fn, iters=25, warm=5, settle=True)
        rows.append(dict(kernel=name, ms=ms, bound=bound, alg_bytes=bytes_, gbs=bytes_ / ms / 1e6, flops=flops,
                         tflops=flops / ms / 1e9))

    conv_in = N * H * 4 + E * D * 4 + csr + 4 * E + 4 * N        # x rows, edge features, CSR, eid, invdeg
    add("nnconv_fused_fwd", lambda: _lib.call("qot_nnconv_fused", P(x), H, P(ea), P(w1), P(b1), P(g.rowptr), P(g.col),
                                              P(g.eid), P(g.invdeg), 0, P(wp), P(bias), P(out), N, H, D, 0, 0.0, 0.0, 0, None),
        conv_in + N * H * 4 + KT * H * 4, 2.0 * N * KT * H, "mfma")
    gwt = torch.empty(KT, H, device=dev)
    ws_a = torch.empty(lib.qot_nnconv_adjoint_dw_workspace_floats(D), device=dev)
    add("nnconv_adjoint_dw", lambda: _lib.call("qot_nnconv_adjoint_dw", P(gout), H, P(x), H, P(ea), P(w1), P(b1),
                                               P(g.rowptr_t), P(g.col_t), P(g.eid_t), P(g.invdeg), P(wp), P(out),
                                               P(gwt), 2, P(ws_a), N, H, D),   # 2: main kernel only (what rocprof lists)
        conv_in + 2 * N * H * 4 + 2 * KT * H * 4, 4.0 * N * KT * H, "mfma")
    add("nnconv_gradh_fused", lambda: _lib.call("qot_nnconv_gradh_fused", P(gout), H, P(x), H, P(ea), P(w1), P(b1),
                                                P(g.rowptr), P(g.col), P(g.eid), P(g.invdeg), P(bp), P(gw1), P(gb1),
                                                P(ws_h), N, H, D),
        conv_in + N * H * 4 + K * H * H * 4, 2.0 * N * H * K * H + 2.0 * E * K * H, "mfma")
    add("tconv_fwd", lambda: _lib.call("qot_tconv_fwd", off(qkvs, 0), off(qkvs, H), off(qkvs, 2 * H), off(qkvs, 3 * H),
                                       4 * H, P(ea), P(we), P(g.rowptr), P(g.col), P(g.eid), None, P(out), P(stats), N, H, D,
                                       0, 0.0, 0.0, 0, None),
        4 * N * H * 4 + N * H * 4 + 8 * N + E * D * 4 + csr + 4 * E)
    # what the replayed step runs in table mode: the tile form (node r of 16 graphs per workgroup) on a projected table
    n_, B_ = CFG["n"], N // CFG["n"]
    t4 = f(n_, 4 * H)
    ids32 = (torch.arange(N, device=dev, dtype=torch.int32) % n_).contiguous()
    colf = ids32[g.col.long()].contiguous()
    add("tconv_fwd_tile", lambda: _lib.call("qot_tconv_fwd_tile", off(t4, 0), off(t4, H), off(t4, 2 * H), off(t4, 3 * H), 4 * H,
                                            P(ea), P(we), P(g.rowptr), P(colf), P(g.eid), P(ids32), P(out), P(stats), N, H, D,
                                            n_, B_, 0, 0.0, 0.0, 0, None),
        N * H * 4 + 8 * N + E * D * 4 + csr + 4 * E + 4 * N + n_ * 4 * H * 4)
    # ... and, since r04, what the replayed step ACTUALLY runs for graphs of <= 128 nodes: the graph form (whole graphs per
    # workgroup, score matrix and value table in LDS; csrc/tconv_graph.hip).  Algorithmic bytes as SURVEY 8(d) prices a conv
    # (the projected table is 100 KB) + what the forward leaves for the backward (alpha, CSR-ordered edge features, sum alpha ea)
    gsz = build_graph_index(batch.edge_index, N, slices=(batch.ptr, batch.edge_ptr) + tuple(batch.graph_sizes),
                            node_ids=batch.node_ids)
    max_e = int(batch.graph_sizes[1])
    ldm = lib.qot_tconv_graph_ldm(n_)
    Mm, Pm = f(n_, ldm), f(n_, D)
    alpha, ea_csr, aa = torch.empty(E, device=dev), torch.empty(E, D, device=dev), torch.empty(N, D, device=dev)
    step1 = torch.ones((), dtype=torch.long, device=dev)
    add("tconv_fwd_graph", lambda: _lib.call("qot_tconv_fwd_graph", t4, 4 * H, Mm, Pm, we, ea, gsz.rowptr, gsz.colf, gsz.eid,
                                             gsz.row, out, alpha, ea_csr, aa, n_, B_, max_e, H, D, 1, 0.01, 0.5, 1234, step1),
        N * H * 4 + E * D * 4 + csr + 8 * E + E * 4 + E * D * 4 + N * D * 4 + n_ * 4 * H * 4)
    blocks, rowlen = lib.qot_tconv_bwd_graph_blocks(B_), lib.qot_tconv_graph_row_floats(n_, H, D)
    partials = torch.empty(blocks, rowlen, device=dev)
    add("tconv_bwd_graph", lambda: _lib.call("qot_tconv_bwd_graph", gout, out, 0.01, 0.5, 1234, step1, t4, 4 * H, we, ea_csr,
                                             alpha, aa, gsz.rowptr, gsz.colf, gsz.row, gsz.rowptr_t, gsz.col_t, gsz.pos_t,
                                             partials, n_, B_, max_e, H, D),
        2 * N * H * 4 + E * D * 4 + 2 * csr + 12 * E + E * 4 + N * D * 4 + blocks * rowlen * 4)
    add("tconv_bwd_dst", lambda: _lib.call("qot_tconv_bwd_dst", P(gout), off(qkvs, 0), off(qkvs, H), off(qkvs, 2 * H),
                                           4 * H, P(ea), P(we), P(stats), P(g.rowptr), P(g.col), P(g.eid), None,
                                           off(gq, 0), off(gq, 3 * H), 4 * H, P(escr), P(delta), P(pds), P(pal),
                                           None, 0.0, 0.0, 0, None, None, None, 0, 0, None, N, H, D),
        4 * N * H * 4 + N * H * 4 + 8 * N + E * D * 4 + csr + 4 * E + 8 * E + 4 * N + 8 * N * D)
    add("tconv_bwd_src", lambda: _lib.call("qot_tconv_bwd_src", P(gout), H, off(qkvs, 0), 4 * H, P(escr), P(delta),
                                           P(g.rowptr_t), P(g.col_t), P(g.pos_t), None, off(gq, H), off(gq, 2 * H),
                                           4 * H, 0, 0, None, N, H),
        2 * N * H * 4 + 2 * N * H * 4 + 8 * E + 4 * N + csr + 4 * E)
    slices = (batch.ptr, batch.edge_ptr) + tuple(batch.graph_sizes) if getattr(batch, "edge_ptr", None) is not None else None
    add("csr_build", lambda: build_graph_index(batch.edge_index, N, slices=slices), 16 * E + 8 * 4 * E + 3 * 4 * N)
    return rows


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(sample_graphs):
    """Oracle (PyG-style sparse restatement) train step on the host cores, bounded sample.

    BASELINE.md's protocol says all cores; on this 256-core host that oversubscribes torch's
    intra-op pool catastrophically (0.7-2.7 graphs/s at 256 threads vs >100 at 16), so the best of
    a few thread counts is reported and ``cores`` is the count actually used."""
    from gnn_qot_estimation_amd import synthetic as S
    from oracle import sparse as O
    ncpu = os.cpu_count() or 1
    b = S.topological_batch(CFG["cfg"], sample_graphs, n=CFG["n"], e=CFG["e"], edge_dim=CFG["D"])
    y = b.y.view(-1, CFG["out"])
    best = None
    tried = []
    t_all = time.perf_counter()
    for thr in sorted({t for t in (8, 16, 32, ncpu) if t <= ncpu}):
        if thr > 64 and best is not None and time.perf_counter() - t_all > 20.0:
            break
        torch.set_num_threads(thr)
        torch.manual_seed(0)
        m = O.TopologicalGNN(CFG["n"], CFG["H"], CFG["out"], CFG["D"], dropout_p=CFG["dropout"])
        m.train()
        opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9)

        def step():
            opt.zero_grad()
            loss = F.smooth_l1_loss(m(b), y)
            loss.backward()
            opt.step()

        tw = time.perf_counter()
        step()                                   # warm-up (allocator, thread pool)
        tw = time.perf_counter() - tw
        if best is not None and tw > 6 * best[0]:        # hopeless configuration (oversubscribed pool)
            tried.append(f"{thr}t:{sample_graphs / tw:.1f}")
            continue
        t0 = time.perf_counter()
        iters = 0
        while iters < 2 or (time.perf_counter() - t0 < 3.0 and iters < 10):
            step()
            iters += 1
        dt = (time.perf_counter() - t0) / iters
        tried.append(f"{thr}t:{sample_graphs / dt:.1f}")
        if best is None or dt < best[0]:
            best = (dt, thr, iters)
        if thr >= 64 and dt > 4 * best[0]:
            break
    dt, thr, iters = best
    return dict(value=sample_graphs / dt, unit="graphs/s", cores=thr, kind="port", cpu_model=cpu_model(), host_cores=ncpu,
                sample=f"{iters} train steps of {sample_graphs} graphs (n=100,e=400,H=64; PyG-style [E,H*H] NNConv) "
                       f"after 1 warm-up, {dt:.2f} s/step at {thr} threads (best of {', '.join(tried)} graphs/s; "
                       f"host has {ncpu} cores), torch {torch.__version__} CPU")


LP = dict(cfg=3, B=65536, F=5, C=128, heads=4, layers=3, out=3)


def lightpath_measurement(device, steps):
    """BASELINE.json configs[2] -- the OTHER model north_star names (lightpath_training/models.py:26-45): 65 536 per-lightpath
    chain graphs (2..20 nodes), 3 x (GATConv heads=4 C=128 + BatchNorm + ReLU), LUT read-out, SmoothL1, SGD(momentum) --
    as a SEPARATE object of the JSON line (never ``value``).  Eager launches (the step is ~35 ms of GPU work: launch gaps
    do not matter), every graph distinct, graph index rebuilt in every step, median of per-step event times.
    Algorithmic figures as SURVEY.md 8(d) prices the path, at the GENERATED sizes (N nodes, E' = E + N edges with the
    self loops GATConv adds, B graphs, W = heads * C): per conv layer forward 8NW + 4E' + 4(N + B) bytes, backward
    12NW + 8E' + 4(N + B); BatchNorm as its own read + write pass each way: 16NW per layer; LUT rows and head: 24BW;
    flops: the dense projections, 2 N W (F + (L - 1) W) forward, twice that backward minus the first layer's grad_x."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import _lib, synthetic as S
    from gnn_qot_estimation_amd.dp import FlatModel, FusedSGD
    from gnn_qot_estimation_amd import functional as QFn
    batch = S.lightpath_batch(LP["B"], cfg=LP["cfg"]).to(device)
    torch.manual_seed(0)
    model = q.LightpathGNN(LP["F"], LP["C"], LP["out"], 1, dropout_p=0.5, num_layers=LP["layers"]).to(device).train()
    flat = FlatModel(model)
    opt = FusedSGD(flat, lr=0.01, momentum=0.9)
    loss_t = torch.zeros((), device=device)

    def step():
        batch._qot_cache = {}
        flat.detach_grads()
        out, lb = model(batch)
        _, g = QFn.smooth_l1_loss_and_grad(out, batch.y[lb], loss_out=loss_t)
        out.backward(g)
        opt.step(grads=True)

    for _ in range(3):
        step()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    evs[0].record()
    for i in range(steps):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    med = per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2])
    mean = evs[0].elapsed_time(evs[steps]) / steps
    loss = float(loss_t.item())
    N, B, W, L, F_ = batch.num_nodes, batch.num_graphs, LP["heads"] * LP["C"], LP["layers"], LP["F"]
    E = batch.num_edges + N
    conv = (8 * N * W + 4 * E + 4 * (N + B)) + (12 * N * W + 8 * E + 4 * (N + B))
    alg_bytes = L * (conv + 16 * N * W) + 24 * B * W
    fwd_flops = 2.0 * N * W * (F_ + (L - 1) * W)
    alg_flops = 3.0 * fwd_flops - 2.0 * N * W * F_
    # the dominant kernels of this step, alone: the forward projection with the previous layer's BatchNorm + ReLU in its
    # operand load and the attention logits in its epilogue (gemm256_nt_kernel<true, true>: the top entry of
    # profiles/r04_cfg3_kernel_stats.csv) and the split-K weight gradient g^T y (gemm_tn_full_kernel<true>); the one that
    # takes longer per launch is reported.  (No library GEMM is left in the step: DESIGN 4.7.)
    g_, x_ = torch.randn(N, W, device=device), torch.randn(N, W, device=device)
    sp = _lib.load().qot_gemm_tn_splits(W, W, N)
    part = torch.empty(sp, W * W, device=device)
    sc_, sh_ = torch.rand(W, device=device) + 0.5, torch.randn(W, device=device)
    ms_tn = event_time_ms(lambda: _lib.call("qot_gemm_tn_planes", g_, W, x_, W, part, W, W, N, sp, sc_, sh_), iters=5, warm=2,
                          settle=True)
    wgt, zout = torch.randn(W, W, device=device) / W ** 0.5, torch.empty(N, W, device=device)
    att = torch.randn(2, W, device=device)
    a_sd = torch.empty(2, N, LP["heads"], device=device)
    ms_nt = event_time_ms(lambda: _lib.call("qot_gemm_nt_logits", x_, W, wgt, W, zout, W, N, W, W, sc_, sh_, None, att[0], att[1],
                                            a_sd[0], a_sd[1]), iters=5, warm=2, settle=True)
    if ms_nt >= ms_tn:
        ms_k, kname, tkey = ms_nt, "gemm256_nt_kernel<true, true> (projection x' W^T, BatchNorm+ReLU in the operand load, logits in the epilogue)", "gemm256_nt_kernel<true, true>"
    else:
        ms_k, kname, tkey = ms_tn, "gemm_tn_full_kernel<true> (weight gradient g^T y of GATConv.lin)", "gemm_tn_full_kernel<true>"
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r04_cfg3_traffic.json")
    if os.path.exists(tpath):
        for r in json.load(open(tpath)).get("kernels", []):
            if r["kernel"] == tkey:
                traffic = r["hbm_bytes_per_launch"]
    kflops = 2.0 * N * W * W
    sec = med * 1e-3
    return {
        "workload": f"configs[2]: LightpathGNN {L} x (GATConv heads={LP['heads']} C={LP['C']} + BatchNorm + ReLU), LUT read-out, "
                    f"{B} distinct chain graphs (2..20 nodes: N = {N}, E = {batch.num_edges} + {N} self loops), dropout 0.5, "
                    "SGD momentum 0.9, SmoothL1; eager launches, graph index rebuilt in every step",
        "graphs": B, "steps": steps, "ms_per_step": mean, "ms_per_step_median": med,
        "graphs_per_s": B / (mean * 1e-3), "graphs_per_s_median": B / sec, "final_loss": loss, "dtype": "f32",
        "alg_bytes_per_step": alg_bytes, "alg_flops_per_step": alg_flops,
        "step_roofline": {"hbm_GBps": alg_bytes / sec / 1e9, "hbm_frac": alg_bytes / sec / 1e9 / HBM_PEAK_GBS,
                          "mfma_TFLOPs": alg_flops / sec / 1e12, "mfma_frac": alg_flops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS},
        "roofline": {"bound": "mfma", "kernel": kname, "achieved": kflops / ms_k / 1e9,
                     "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": kflops / ms_k / 1e9 / MFMA_F32_PEAK_TFLOPS,
                     "traffic": traffic, "alg_flops_per_launch": kflops, "alg_bytes_per_launch": 8 * N * W + 4 * W * W,
                     "ms_per_launch": ms_k, "other_kernel_ms": {"gemm256_nt_kernel<true, true>": ms_nt,
                                                                 "gemm_tn_full_kernel<true>": ms_tn},
                     "traffic_source": "profiles/r04_cfg3_traffic.json (per-kernel HBM bytes and GB/s of every cfg3 kernel)"},
    }


def reference_scale_measurement(device, steps=100):
    """SURVEY 8(d)'s "reference-scale sanity config" -- the sizes the reference's user actually runs: TopologicalGNN V = 75,
    H = 16, B = 512 (topological_training/train.py:38,50-52) and LightpathGNN F = 5, C = 32, B = 512
    (lightpath_training/train.py:39,52).  A SEPARATE object of the JSON line (never ``value``).  One train step = the step
    body of the reference's loop on a cached batch of an HBM-resident shard (``harness.StepReplayer._step``): eager launches,
    and the same step replayed as a HIP graph (what ``harness.fit(replay=True)`` does from a batch's third visit on)."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import harness as Hn, synthetic as S
    from gnn_qot_estimation_amd.dp import FlatModel, FusedSGD
    from gnn_qot_estimation_amd.loader import GraphLoader

    def timed(fn, n):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    out = {}
    torch.manual_seed(0)
    cases = (
        ("topological", "TopologicalGNN V=75 H=16 edge_dim=4, B=512 (75 nodes, 150 directed edges per graph)",
         lambda: q.TopologicalGNN(75, 16, 3, 4, dropout_p=0.5), lambda: S.topological_batch(2, 512, n=75, e=150)),
        ("lightpath", "LightpathGNN F=5 C=32 heads=4, B=512 (chain graphs of 2..20 nodes)",
         lambda: q.LightpathGNN(5, 32, 3, 1, dropout_p=0.5), lambda: S.lightpath_batch(512)),
    )
    for kind, what, make_model, make_batch_ in cases:
        b = make_batch_()
        graphs = b.num_graphs
        shard = q.PackedGraphs.from_batch(b).to_device(device)
        loader = GraphLoader(shard, graphs, shuffle=False, device=device, cache_batches=True)
        data = next(iter(loader))
        model = make_model().to(device).train()
        flat = FlatModel(model)
        opt = FusedSGD(flat, lr=0.01, momentum=0.9, device_lr=True)
        rp = Hn.StepReplayer(model, kind, 3, device, flat, opt)
        assert rp.run(data, True)                       # first visit: eager (builds and caches the graph index)
        eager_ms = timed(lambda: rp._step(data, True), steps)
        assert rp.run(data, True)                       # second visit: captured
        assert (id(data), True) in rp.graphs
        replay_ms = timed(lambda: rp.run(data, True), steps)
        loss = float(rp._loss.item())
        if not (loss == loss):
            raise SystemExit(f"reference scale {kind}: non-finite loss")
        out[kind] = {"workload": what, "graphs": graphs, "nodes": int(b.num_nodes), "edges": int(b.num_edges),
                     "eager_ms_per_step": eager_ms, "replayed_ms_per_step": replay_ms,
                     "eager_graphs_per_s": graphs / (eager_ms * 1e-3), "replayed_graphs_per_s": graphs / (replay_ms * 1e-3),
                     "final_loss": loss}
        del rp, flat, opt, model, loader, shard
        torch.cuda.empty_cache()
    out["note"] = ("cached batch of an HBM-resident shard, dropout 0.5, SGD momentum 0.9, SmoothL1; eager = host launches "
                   "(host-bound at this size), replayed = the whole step as one HIP graph")
    return out


def spawn_ranks(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes (one per device) from a
    parent that never touches the GPU, relay rank 0's JSON line, fail if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no torch.cuda call has happened in this process: the children are fresh processes, nothing is re-exec'ed
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("BENCH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    force_dp = bool(os.environ.get("BENCH_FORCE_DP"))
    if world > 1 or force_dp:
        # BENCH_BACKEND=gloo + BENCH_SHARE_GPU=1: rehearsal of the multi-rank path on a one-GPU box.
        # BENCH_FORCE_DP=1 (at N = 1): the N > 1 step -- pack, RCCL all-reduce inside the captured step, plain update -- on a
        # single-rank nccl group: the one-GPU box measures what the collective's launch costs the step (not a scaling number)
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = dict(rank=rank, world_size=world) if world == 1 else {}
        # RCCL prints a version banner on STDOUT when its communicator comes up (first collective): keep stdout for the one
        # JSON line -- the banner goes to stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device, **kw)
            else:
                dist.init_process_group(backend, **kw)
            warm = torch.zeros(1, device=device)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    from gnn_qot_estimation_amd import _lib
    _lib.load()

    batch = make_batch(rank, device)
    model = build_model(device)
    step = TrainStep(model, batch, world, use_graph=not args.no_graph, steps_per_graph=args.steps_per_graph)
    if step.use_graph:
        step.capture()

    def barrier():
        if world > 1:
            dist.barrier()

    # The part needs ~12 ms of load to reach its clock: whatever W the caller asks for, at least 50
    # untimed steps (~30 ms) run before the timed region; the extra ones are reported as config.prime_steps.
    prime_steps = max(0, 50 - args.warmup)
    step.run(prime_steps + args.warmup)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step.run(args.steps)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # SURVEY 8(d) asks for the MEDIAN step time: a second pass of the same K steps with an event between steps (kept out
    # of the timed region above so that `value` stays K back-to-back steps between two synchronisations)
    # (an event between replays: with steps_per_graph > 1 a sample is the mean of that many consecutive steps)
    spg = step.spg if step.graph_multi is not None else 1
    nsamp = max(1, args.steps // spg)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nsamp + 1)]
    evs[0].record()
    for i in range(nsamp):
        step.run(spg)
        evs[i + 1].record()
    torch.cuda.synchronize()
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) / spg for i in range(nsamp))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if world > 1:
        t = torch.tensor([median_ms], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        median_ms = float(t.item())
    loss = float(step.loss.item())
    if not (loss == loss) or loss in (float("inf"), float("-inf")):
        raise SystemExit(f"non-finite loss {loss}")
    # the exchange step alone (outside the timed region): events around the flat-gradient all-reduce
    all_reduce_us = None
    if step.dp and dist.is_initialized():
        all_reduce_us = event_time_ms(lambda: step.flat.all_reduce_grads(force=True), iters=20, warm=3) * 1e3
        t = torch.tensor([all_reduce_us], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        all_reduce_us = float(t.item())

    if rank == 0:
        graphs = CFG["B"] * world * args.steps
        res = {
            "metric": "graphs/sec (train step) on 100-node/400-edge synthetic topologies, batch=1024",
            "value": graphs / dt, "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "ms_per_step_median": median_ms,
            "graphs_per_s_median": CFG["B"] * world / (median_ms * 1e-3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: 2-layer TopologicalGNN (TransformerConv+NNConv) hidden=64, "
                                   "100-node/400-directed-edge topologies, batch=1024 graphs per GPU, dropout 0.5, "
                                   "SGD momentum 0.9, SmoothL1; " + ("graph index cached across steps (BENCH_PREP_OUTSIDE: the HBM-resident, cached-batch loader mode; not the headline)" if os.environ.get("BENCH_PREP_OUTSIDE") else "CSR build included in every step"),
                       "graphs_per_gpu": CFG["B"], "global_batch": CFG["B"] * world, "distinct_graphs_per_gpu": CFG["B"],
                       "launch": "eager" if step.graph_fb is None else (
                           f"hip-graph replay, {step.spg} whole steps (fwd+bwd" + ("+pack+all-reduce" if step.dp else "") + "+optimizer each) per graph"
                           if step.graph_multi is not None else
                           ("hip-graph replay, one whole step per graph" if step.graph_opt is None
                            else "hip-graph replay (fwd+bwd+pack | eager all-reduce | optimizer)")),
                       "steps_per_graph": step.spg if step.graph_multi is not None else 1,
                       "collective_in_graph": step.collective_in_graph, "collective_capture_error": step.capture_error,
                       "parallelism": f"dp{world}", "final_loss": loss, "prime_steps": prime_steps,
                       "collective_world_size": dist.get_world_size() if dist.is_initialized() else 1,
                       "collective_backend": dist.get_backend() if dist.is_initialized() else None,
                       "all_reduce_us": all_reduce_us,
                       "all_reduce_floats": int(step.flat.flat_grad.numel())},
        }
        rows = kernel_table(model, batch)
        dom = max(rows, key=lambda r: r["ms"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(dom["kernel"])
        if dom["bound"] == "mfma":
            res["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"],
                               "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": dom["tflops"] / MFMA_F32_PEAK_TFLOPS, "traffic": traffic,
                               "alg_flops_per_launch": dom["flops"], "alg_bytes_per_launch": dom["alg_bytes"],
                               "ms_per_launch": dom["ms"]}
        else:
            res["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["gbs"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": dom["gbs"] / HBM_PEAK_GBS, "traffic": traffic,
                               "alg_bytes_per_launch": dom["alg_bytes"], "ms_per_launch": dom["ms"]}
        # whole-step view against both rooflines, from SURVEY.md 8(d)'s per-graph algorithmic figures:
        # bytes: embed 2(4n+4nH) + convs (8nH+4e+4(n+1)+4eD fwd, 12nH+8e+4(n+1)+4eD bwd) + pool 2(4nH+4n+4H);
        # flops: 3 x fwd, fwd = TransformerConv 8nH^2+2eHD+4eH + NNConv 2nH^2(K+1)+2eH(K+1)+2nH^2
        n_, e_, H_, D_, K_, L_ = CFG["n"], CFG["e"], CFG["H"], CFG["D"], 2 * CFG["D"], CFG["layers"]
        conv_f = 8 * n_ * H_ + 4 * e_ + 4 * (n_ + 1) + 4 * e_ * D_
        conv_b = 12 * n_ * H_ + 8 * e_ + 4 * (n_ + 1) + 4 * e_ * D_
        step_bytes = (2 * (4 * n_ + 4 * n_ * H_) + L_ * (conv_f + conv_b) + 2 * (4 * n_ * H_ + 4 * n_ + 4 * H_)) * CFG["B"]
        fwd_flops = (8 * n_ * H_ * H_ + 2 * e_ * H_ * D_ + 4 * e_ * H_
                     + (L_ - 1) * (2 * n_ * H_ * H_ * (K_ + 1) + 2 * e_ * H_ * (K_ + 1) + 2 * n_ * H_ * H_))
        step_flops = 3.0 * fwd_flops * CFG["B"]
        sec = dt / args.steps
        res["step_roofline"] = {"alg_bytes_per_step": step_bytes, "hbm_GBps": step_bytes / sec / 1e9,
                                "hbm_frac": step_bytes / sec / 1e9 / HBM_PEAK_GBS,
                                "alg_flops_per_step": step_flops, "mfma_TFLOPs": step_flops / sec / 1e12,
                                "mfma_frac": step_flops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                "note": "per rank; algorithmic figures of SURVEY.md 8(d)"}
        res["kernels"] = [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]
        if world == 1 and not args.no_lightpath:
            del step, batch, model
            torch.cuda.empty_cache()
            res["lightpath"] = lightpath_measurement(device, args.lightpath_steps)
        if world == 1 and not args.no_reference_scale:
            res["reference_scale"] = reference_scale_measurement(device)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_sample_graphs)
        print(json.dumps(res))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
