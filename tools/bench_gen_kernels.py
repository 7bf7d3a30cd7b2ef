"""Times the four width-generic fused NNConv kernels (csrc/nnconv_gen.hip) one by one through the C ABI of the RELEASE
library, at the shapes of BASELINE configs[3] / configs[4]:  python tools/bench_gen_kernels.py [H] [graphs] [cfg]
(cfg 4: 1000-node / 4000-edge graphs, cfg 5: power-law in-degrees up to 64).  One JSON line per kernel."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_qot_estimation_amd import _lib, functional as QF, synthetic as S
from gnn_qot_estimation_amd.graph import build_graph_index

H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else (5 if H == 256 else 4)
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
dev = torch.device("cuda:0")
P = _lib.ptr
lib = _lib.load()
base = S.topological_batch(cfg, min(B, 16), n=1000, e=4000)
b = S.tile_batch(base, B // min(B, 16)).to(dev)
N, D, K = b.num_nodes, 4, 8
g = build_graph_index(b.edge_index, N)
x, gout = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
w1, b1 = torch.randn(K, D, device=dev), torch.randn(K, device=dev)
w2, b2, wr = torch.randn(H * H, K, device=dev) / 8, torch.randn(H * H, device=dev) / 8, torch.randn(H, H, device=dev) / 8
wp, wp_adj, bp = QF.nnconv_pack_operands_gen(w2, b2, wr, H, K)
bias, out = torch.randn(H, device=dev), torch.empty(N, H, device=dev)
gpar = torch.empty((K + 2) * H * H, device=dev)
ws = torch.empty(lib.qot_nnconv_dw_workspace_floats(N, H, D), device=dev)
gw1, gb1 = torch.empty(K, D, device=dev), torch.empty(K, device=dev)
wsh = torch.empty(lib.qot_nnconv_gradh_workspace_floats(D), device=dev)
ea = b.edge_attr
flops = 2.0 * N * (K + 2) * H * H
kernels = {
    "fwd": (flops, lambda: _lib.call("qot_nnconv_fused", P(x), H, P(ea), P(w1), P(b1), P(g.rowptr), P(g.col), P(g.eid), P(g.invdeg),
                                      0, P(wp), P(bias), P(out), N, H, D, 0, 0.0, 0.0, 0, None)),
    "adjoint": (flops, lambda: _lib.call("qot_nnconv_fused", P(gout), H, P(ea), P(w1), P(b1), P(g.rowptr_t), P(g.col_t), P(g.eid_t),
                                          P(g.invdeg), 1, P(wp_adj), None, P(out), N, H, D, 0, 0.0, 0.0, 0, None)),
    "dw": (flops, lambda: _lib.call("qot_nnconv_dw", P(x), H, P(gout), H, P(ea), P(w1), P(b1), P(g.rowptr), P(g.col), P(g.eid),
                                     P(g.invdeg), P(gpar), P(ws), N, H, D)),
    "gradh": (2.0 * N * K * H * H, lambda: _lib.call("qot_nnconv_gradh_fused", P(gout), H, P(x), H, P(ea), P(w1), P(b1), P(g.rowptr),
                                                      P(g.col), P(g.eid), P(g.invdeg), P(bp), P(gw1), P(gb1), P(wsh), N, H, D)),
}
def t(run, iters=10):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run(); torch.cuda.synchronize(); st.record()
    for _ in range(iters): run()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3
for name, (fl, run) in kernels.items():
    if only and name not in only: continue
    us = min(t(run) for _ in range(3))
    print(json.dumps({"kernel": name, "H": H, "N": N, "E": int(b.edge_index.shape[1]), "cfg": cfg, "us": round(us, 1),
                      "TFLOPs": round(fl / us / 1e6, 1)}), flush=True)
