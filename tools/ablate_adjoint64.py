"""Phase ablation of nnconv_adjoint_dw64 (diagnostic build, QOT_LIB_PATH=tools/diag/libqot_gnn_diag.so): template variants
of the production kernel, interleaved rounds in one process.  cfg2 shape (1024 graphs, 100 nodes, 400 edges, H = 64)."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_qot_estimation_amd import _lib, functional as QF, synthetic as S
from gnn_qot_estimation_amd.graph import build_graph_index
dev = torch.device("cuda:0"); P = _lib.ptr; lib = _lib.load()
lib.qot_debug_set_variant.argtypes = [ctypes.c_int]
H, D, K = 64, 4, 8
b = S.tile_batch(S.topological_batch(2, 128, n=100, e=400), 8).to(dev)
N = b.num_nodes; g = build_graph_index(b.edge_index, N)
x, gout = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
w1, b1 = torch.randn(K, D, device=dev), torch.randn(K, device=dev)
w2, b2, wr = torch.randn(H * H, K, device=dev), torch.randn(H * H, device=dev), torch.randn(H, H, device=dev)
_, bp_t, _ = QF.nnconv_pack_operands(w2, b2, wr, K)
ws = torch.empty(lib.qot_nnconv_adjoint_dw_workspace_floats(D), device=dev)
gx = torch.empty(N, H, device=dev); gw = torch.empty((K + 2) * 64 * 64, device=dev)
f = lambda: _lib.call("qot_nnconv_adjoint_dw", P(gout), H, P(x), H, P(b.edge_attr), P(w1), P(b1), P(g.rowptr_t), P(g.col_t),
                      P(g.eid_t), P(g.invdeg), P(bp_t), P(gx), P(gw), 2, P(ws), N, H, D)
def t(it=20):
    f(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(it): f()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / it * 1e3
names = {0: "full", 11: "rows_hot", 12: "no_gradx_loop", 13: "no_dw_loop", 14: "no_gather", 15: "gather_only", 16: "one_of_K_sums", 17: "no_index_chain"}
res = {v: 1e9 for v in names.values()}
for rnd in range(4):
    for v, nm in names.items():
        lib.qot_debug_set_variant(v); res[nm] = min(res[nm], t())
lib.qot_debug_set_variant(0)
print(json.dumps({"N": N, "us": {k: round(v, 1) for k, v in res.items()}}))
