"""A/B of the tuned grad-h kernel (H = 64) against the width-generic one at H = 64 (diagnostic build, QOT_LIB_PATH)."""
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
_, _, bp_t = QF.nnconv_pack_operands(w2, b2, wr, K)
allidx, n_f, n_a, n_g = QF.nnconv_gen_indices(64, K, dev)
flat = torch.cat([w2.reshape(-1), b2, wr.reshape(-1)])
bp_g = flat[allidx[n_f + n_a:].long()].contiguous()
ws = torch.empty(lib.qot_nnconv_gradh_workspace_floats(D), device=dev)
outs = {}
def mk(bp, name):
    gw1, gb1 = torch.empty(K, D, device=dev), torch.empty(K, device=dev)
    outs[name] = (gw1, gb1)
    return lambda: _lib.call("qot_nnconv_gradh_fused", P(gout), H, P(x), H, P(b.edge_attr), P(w1), P(b1), P(g.rowptr), P(g.col),
                             P(g.eid), P(g.invdeg), P(bp), P(gw1), P(gb1), P(ws), N, H, D)
def t(f, it=20):
    f(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(it): f()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / it * 1e3
ft, fg = mk(bp_t, "tuned"), mk(bp_g, "generic")
res = {"tuned": 1e9, "generic": 1e9}
for rnd in range(4):
    lib.qot_debug_set_variant(0); res["tuned"] = min(res["tuned"], t(ft))
    lib.qot_debug_set_variant(9); res["generic"] = min(res["generic"], t(fg))
lib.qot_debug_set_variant(0)
err = float((outs["generic"][0] - outs["tuned"][0]).abs().max() / outs["tuned"][0].abs().max())
print(json.dumps({"us": {k: round(v, 1) for k, v in res.items()}, "rel_diff": err}))
