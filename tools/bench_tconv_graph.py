"""TransformerConv graph form (csrc/tconv_graph.hip) piece by piece through the C ABI at cfg2's shape: score matrix job,
forward, backward, the row sum of the partials and the projection backward, next to the kernels they replace.  With the
diagnostic build (QOT_LIB_PATH=tools/diag/libqot_gnn_diag.so, `make -C gnn_qot_estimation_amd/csrc DIAG=1`) also the
in-kernel phase stamps (cycles of thread 0, summed over workgroups)."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_qot_estimation_amd import _lib, synthetic as S
from gnn_qot_estimation_amd.graph import build_graph_index

dev = torch.device("cuda:0")
lib = _lib.load()
B, n, e, H, D = int(os.environ.get("TG_B", 1024)), int(os.environ.get("TG_N", 100)), int(os.environ.get("TG_E", 400)), 64, 4
b = S.topological_batch(2, B, n=n, e=e).to(dev)
N, E = b.num_nodes, b.num_edges
g = build_graph_index(b.edge_index, N, slices=(b.ptr, b.edge_ptr) + tuple(b.graph_sizes), node_ids=b.node_ids)
max_e = int(b.graph_sizes[1])
f = lambda *s: torch.randn(*s, device=dev)
table, wq, bq, wk, bk, wv, bv, ws, bs, we = f(n, H), f(H, H) / 8, f(H), f(H, H) / 8, f(H), f(H, H) / 8, f(H), f(H, H) / 8, f(H), f(H, D)
t4 = torch.empty(n, 4 * H, device=dev)
ldm = lib.qot_tconv_graph_ldm(n)
M, Pm = torch.empty(n, ldm, device=dev), torch.empty(n, D, device=dev)
out, alpha = torch.empty(N, H, device=dev), torch.empty(E, device=dev)
ea_csr, aa = torch.empty(E, D, device=dev), torch.empty(N, D, device=dev)
gwe = torch.empty(H, D, device=dev)
gout = f(N, H)
step = torch.ones((), dtype=torch.long, device=dev)
blocks, rowlen = lib.qot_tconv_bwd_graph_blocks(B), lib.qot_tconv_graph_row_floats(n, H, D)
partials, Ssum = torch.empty(blocks, rowlen, device=dev), torch.empty(rowlen, device=dev)
gt, gw, gb = torch.empty(n * H, device=dev), torch.empty(4 * H * H, device=dev), torch.empty(4 * H, device=dev)
act = (1, 0.01, 0.5, 1234, step)
_lib.call("qot_table_project_fwd", table, wq, bq, wk, bk, wv, bv, ws, bs, t4, n, H, None, None)


def timeit(fn, it=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(it):
            fn()
        en.record()
        torch.cuda.synchronize()
        best = min(best, st.elapsed_time(en) / it * 1e3)
    return round(best, 2)


jobs = {
    "table_scores": lambda: _lib.call("qot_table_scores", table, wq, bq, wk, bk, we, M, Pm, n, H, D),
    "fwd_graph": lambda: _lib.call("qot_tconv_fwd_graph", t4, 4 * H, M, Pm, we, b.edge_attr, g.rowptr, g.colf, g.eid, g.row,
                                   out, alpha, ea_csr, aa, n, B, max_e, H, D, *act),
    "bwd_graph": lambda: _lib.call("qot_tconv_bwd_graph", gout, out, 0.01, 0.5, 1234, step, t4, 4 * H, we, ea_csr, alpha, aa,
                                   g.rowptr, g.colf, g.row, g.rowptr_t, g.col_t, g.pos_t, partials, n, B, max_e, H, D),
    "sum_rows": lambda: _lib.run_roles([_lib.make_role(_lib.ROLE_SUM_ROWS, (partials, Ssum), (blocks, rowlen, 0))]),
    "project_bwd_scores": lambda: _lib.call("qot_table_project_bwd_scores", Ssum, t4, we, table, wq, wk, wv, ws, gt, gw, gb, gwe,
                                            n, n, H, D),
}
res = {"B": B, "n": n, "E": E, "partials_MB": round(partials.numel() * 4 / 1e6, 1)}
for k, fn in jobs.items():
    res[k + "_us"] = timeit(fn)
# the kernels it replaces
off = lambda t, k: t.data_ptr() + 4 * k
stats = torch.empty(N, 2, device=dev)
res["fwd_tile_us"] = timeit(lambda: _lib.call("qot_tconv_fwd_tile", off(t4, 0), off(t4, H), off(t4, 2 * H), off(t4, 3 * H), 4 * H,
                                               b.edge_attr, we, g.rowptr, g.colf, g.eid, g.ids32, out, stats, N, H, D, n, B, *act))
print(json.dumps(res))
if hasattr(lib, "qot_debug_tg_variant") and not os.environ.get("TG_STAMPS"):
    lib.qot_debug_tg_variant.argtypes = [ctypes.c_int]
    abl = {}
    for name, v in (("full", 0), ("no edge dots / ge", 1), ("no source pass / gWe", 2), ("no 1c", 4), ("no act backward", 8),
                    ("none of the passes", 7), ("none + no act", 15)):
        lib.qot_debug_tg_variant(v)
        abl[name] = timeit(jobs["bwd_graph"])
    lib.qot_debug_tg_variant(0)
    print(json.dumps({"bwd_graph ablation (us)": abl}))
    abl = {}
    for name, v in (("full", 0), ("no stage C", 16), ("no stage B", 32), ("no edge staging", 64), ("no B, C", 48), ("nothing but the tables", 112),
                    ("C without its output stores", 128), ("C without the activation (hash)", 256),
                    ("nothing at all (launch, barriers, per-thread constants)", 624)):
        lib.qot_debug_tg_variant(v)
        abl[name] = timeit(jobs["fwd_graph"])
    lib.qot_debug_tg_variant(0)
    print(json.dumps({"fwd_graph ablation (us)": abl}))
if hasattr(lib, "qot_debug_tg_stamps") and os.environ.get("TG_STAMPS"):
    lib.qot_debug_tg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    names = ["fwd trip 1 (tables, row pointers) + barrier", "fwd A stage graph, logits", "fwd B softmax",
             "fwd C aggregate + store", "bwd wait at the top barrier", "bwd commit prefetched graph + barrier",
             "bwd 1a edge dots, ge (+ prefetch issue)", "bwd 2 source pass, gWe + barrier", "bwd 1c per-destination scalars",
             "bwd partial row"]
    for which in ("fwd_graph", "bwd_graph"):
        lib.qot_debug_tg_stamps(None, 1)
        jobs[which]()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        lib.qot_debug_tg_stamps(ctypes.cast(buf, ctypes.c_void_p), 0)
        wgs = blocks if which == "bwd_graph" else min(256, (B + 3) // 4)
        for i, nm in enumerate(names):
            if buf[i]:
                print(f"  {nm:32s} {buf[i] / wgs:10.0f} cycles per workgroup (s_memtime ticks of thread 0)")
