"""Phase ablation of the width-generic fused NNConv forward kernel (diagnostic build: `make -C gnn_qot_estimation_amd/csrc DIAG=1`,
run with QOT_LIB_PATH=tools/diag/libqot_gnn_diag.so).  Variants: 0 production, 1 no gather, 2 weight fragments not streamed,
3 gather only, 4 LDS-fed MFMA only.  Interleaved rounds in one process; events on the launch stream."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_qot_estimation_amd import _lib, functional as QF, synthetic as S
from gnn_qot_estimation_amd.graph import build_graph_index

H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B, n, e = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (256, 1000, 4000)))
dev = torch.device("cuda:0")
P = _lib.ptr
lib = _lib.load()
lib.qot_debug_gen_variant.argtypes = [ctypes.c_int]
base = S.topological_batch(int(os.environ.get("ABLATE_CFG", "4")), min(B, 16), n=n, e=e)
b = S.tile_batch(base, B // min(B, 16)).to(dev)
N, D, K = b.num_nodes, 4, 8
g = build_graph_index(b.edge_index, N)
x = torch.randn(N, H, device=dev)
w1, b1 = torch.randn(K, D, device=dev), torch.randn(K, device=dev)
w2, b2, wr = torch.randn(H * H, K, device=dev), torch.randn(H * H, device=dev), torch.randn(H, H, device=dev)
wp, _, _ = QF.nnconv_pack_operands_gen(w2, b2, wr, H, K) if H != 64 else (None, None, None)
tflag = 0
if H == 64:
    allidx, n_f, _, _ = QF.nnconv_gen_indices(64, K, dev)
    flat = torch.cat([w2.reshape(-1), b2, wr.reshape(-1)])
    wp = flat[allidx[:n_f].long().clamp(min=0)].contiguous()
    tflag = 2
bias, out = torch.randn(H, device=dev), torch.empty(N, H, device=dev)
run = lambda: _lib.call("qot_nnconv_fused", P(x), H, P(b.edge_attr), P(w1), P(b1), P(g.rowptr), P(g.col), P(g.eid), P(g.invdeg),
                        tflag, P(wp), P(bias), P(out), N, H, D, 0, 0.0, 0.0, 0, None)
def t(iters=10):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); st.record()
    for _ in range(iters): run()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3
res = {v: [] for v in range(5)}
for rnd in range(4):
    for v in range(5):
        lib.qot_debug_gen_variant(v); run(); res[v].append(t())
lib.qot_debug_gen_variant(0)
flops = 2.0 * N * (K + 2) * H * H
names = {0: "full", 1: "no gather", 2: "no weight stream", 3: "gather only", 4: "LDS-fed MFMA only", 5: "consumers prio 2", 6: "producers prio 2"}
out_ = {"H": H, "N": N, "flops": flops}
for v in range(5):
    us = min(res[v])
    out_[names[v]] = {"us": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1) if v != 3 else None}
# weight-gradient kernel: full / no gather / gather only
gout = torch.randn(N, H, device=dev)
gpar = torch.empty((K + 2) * H * H, device=dev)
wsd = torch.empty(lib.qot_nnconv_dw_workspace_floats(N, H, D), device=dev)
keep = run
run = lambda: _lib.call("qot_nnconv_dw", P(x), H, P(gout), H, P(b.edge_attr), P(w1), P(b1), P(g.rowptr), P(g.col), P(g.eid),
                        P(g.invdeg), P(gpar), P(wsd), N, H, D)
dres = {0: [], 1: [], 3: [], 5: [], 6: []}
for rnd in range(3):
    for v in dres:
        lib.qot_debug_gen_variant(v); run(); dres[v].append(t())
lib.qot_debug_gen_variant(0)
out_["dw"] = {names[v]: {"us": round(min(dres[v]), 1), "TFLOPs": round(flops / min(dres[v]) / 1e6, 1) if v != 3 else None} for v in dres}
run = keep
if H == 64:      # the tuned tile kernel on the same inputs, and both with the dropout epilogue
    wpt, _, _ = QF.nnconv_pack_operands(w2, b2, wr, K)
    step = torch.tensor([3], dtype=torch.int64, device=dev)
    def mk(flag, w, act):
        a = (1, 0.01, 0.5, 1234, P(step)) if act else (0, 0.0, 0.0, 0, None)
        return lambda: _lib.call("qot_nnconv_fused", P(x), H, P(b.edge_attr), P(w1), P(b1), P(g.rowptr), P(g.col), P(g.eid),
                                 P(g.invdeg), flag, P(w), P(bias), P(out), N, H, D, *a)
    cases = {"tuned": mk(0, wpt, False), "generic": mk(2, wp, False), "tuned+dropout": mk(0, wpt, True), "generic+dropout": mk(2, wp, True)}
    best = {k: 1e9 for k in cases}
    for rnd in range(4):
        for k, f in cases.items():
            run = f; f(); best[k] = min(best[k], t())
    out_["h64_compare_us"] = {k: round(v, 1) for k, v in best.items()}
print(json.dumps(out_))
