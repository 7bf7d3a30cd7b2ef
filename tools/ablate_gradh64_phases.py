"""Phase ablation + in-kernel phase stamps of nnconv_gradh64 (diagnostic build, QOT_LIB_PATH=tools/diag/libqot_gnn_diag.so)."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_qot_estimation_amd import _lib, functional as QF, synthetic as S
from gnn_qot_estimation_amd.graph import build_graph_index
dev = torch.device("cuda:0"); P = _lib.ptr; lib = _lib.load()
lib.qot_debug_set_variant.argtypes = [ctypes.c_int]
lib.qot_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
H, D, K = 64, 4, 8
b = S.tile_batch(S.topological_batch(2, 128, n=100, e=400), 8).to(dev)
N = b.num_nodes; g = build_graph_index(b.edge_index, N)
x, gout = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
w1, b1 = torch.randn(K, D, device=dev), torch.randn(K, device=dev)
w2, b2, wr = torch.randn(H * H, K, device=dev), torch.randn(H * H, device=dev), torch.randn(H, H, device=dev)
_, _, bp = QF.nnconv_pack_operands(w2, b2, wr, K)
ws = torch.empty(lib.qot_nnconv_gradh_workspace_floats(D), device=dev)
gw1, gb1 = torch.empty(K, D, device=dev), torch.empty(K, device=dev)
f = lambda: _lib.call("qot_nnconv_gradh_fused", P(gout), H, P(x), H, P(b.edge_attr), P(w1), P(b1), P(g.rowptr), P(g.col),
                      P(g.eid), P(g.invdeg), P(bp), P(gw1), P(gb1), P(ws), N, H, D)
def t(it=20):
    f(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(it): f()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / it * 1e3
names = {0: "full", 32: "no_dot_phase", 33: "no_ga_mfma_phase", 34: "ga_phase_without_lds_stores", 35: "ga_phase_weights_not_streamed"}
res = {v: 1e9 for v in names.values()}
for rnd in range(4):
    for v, nm in names.items():
        lib.qot_debug_set_variant(v); res[nm] = min(res[nm], t())
print(json.dumps({"N": N, "us": {k: round(v, 1) for k, v in res.items()}}))
lib.qot_debug_set_variant(31); f(); torch.cuda.synchronize()
lib.qot_debug_stamps(None, 1); f(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)(); lib.qot_debug_stamps(ctypes.cast(buf, ctypes.c_void_p), 0)
lib.qot_debug_set_variant(0)
tiles = (N + 31) // 32
nm = ["g tile load+store", "barrier", "GA MFMA phase", "barrier", "dot phase", "end barrier"]
tot = sum(buf[:6])
for n_, v in zip(nm, buf[:6]):
    print(f"  {n_:22s} {v / (tiles * 4):9.0f} cycles/wave/tile  {100.0 * v / tot:5.1f}%")
