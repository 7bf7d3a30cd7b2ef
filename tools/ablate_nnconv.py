"""Ablation timing of the fused NNConv kernel (interleaved rounds in one process)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_qot_estimation_amd import _lib, synthetic as S
from gnn_qot_estimation_amd.graph import build_graph_index
from gnn_qot_estimation_amd.functional import nnconv_perm_index
P = _lib.ptr
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base = S.topological_batch(2, 128, n=100, e=400)
b = S.tile_batch(base, B // 128).to(dev)
N, H, D, K = b.num_nodes, 64, 4, 8
g = build_graph_index(b.edge_index, N)
x = torch.randn(N, H, device=dev); w1 = torch.randn(K, D, device=dev); b1 = torch.randn(K, device=dev)
wcat = torch.randn((K + 2) * H, H, device=dev); wp = wcat.reshape(-1)[nnconv_perm_index((K + 2) * H, dev)].contiguous()
bias = torch.randn(H, device=dev); out = torch.empty(N, H, device=dev)
lib = _lib.load(); lib.qot_debug_set_variant.argtypes = [ctypes.c_int]
def run():
    _lib.call("qot_nnconv_fused", P(x), H, P(b.edge_attr), P(w1), P(b1), P(g.rowptr), P(g.col), P(g.eid),
              P(g.invdeg), 0, P(wp), P(bias), P(out), N, H, D, 0, 0.0, 0.0, 0, None)
def t(iters=20):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); st.record()
    for _ in range(iters): run()
    en.record(); torch.cuda.synchronize(); return st.elapsed_time(en) / iters * 1e3
names = {0: "full", 1: "mfma only (no gather)", 2: "gather only (no mfma)", 100: "full, 1 WG/CU", 101: "mfma only, 1 WG/CU",
         102: "gather only, 1 WG/CU", 4: "mfma only, B not streamed", 5: "mfma only, no A reads, no B", 104: "mfma only, B not streamed, 1 WG/CU",
         105: "mfma only, no A no B, 1 WG/CU", 6: "full, gathered rows hot", 7: "full, no index chain", 8: "full, one of K sums", 10: "cost shape of a rank-1 MFMA gather (4 edges / destination, no control flow)"}
res = {v: [] for v in names}
for rnd in range(5):
    for v in names:
        lib.qot_debug_set_variant(v); run(); res[v].append(t())
lib.qot_debug_set_variant(0)
flops = 2.0 * N * (K + 2) * H * H
for v, n in names.items():
    m = min(res[v]); print(f"{n:28s} min {m:8.1f} us  med {sorted(res[v])[2]:8.1f} us   ({flops / m / 1e6:6.1f} TFLOP/s if full GEMM)")

# per-phase stamps (diagnostic build: shares, not absolute time)
import numpy as np
lib.qot_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
tiles = (N + 31) // 32
names = ["gather (+ first weight requests)", "barrier after gather", "mfma main loop", "barrier before root", "root write+barrier+mfma",
         "reduce+epilogue", "end barrier"]
for v, title, wgs in ((3, "full kernel, 2 WG/CU", 512), (9, "no gather / A reads / B stream, 2 WG/CU", 512), (109, "no gather / A reads / B stream, 1 WG/CU", 256)):
    lib.qot_debug_set_variant(v); run(); torch.cuda.synchronize()
    lib.qot_debug_stamps(None, 1); run(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)(); lib.qot_debug_stamps(ctypes.cast(buf, ctypes.c_void_p), 0)
    tot = sum(buf[:7])
    print(title)
    for n, val in zip(names, buf[:7]):
        print(f"  {n:34s} {val / (tiles * 4):9.0f} ticks/wave/tile  {100.0 * val / tot:5.1f}%")
    print(f"  total {tot / (tiles * 4):9.0f} ticks/wave/tile; tiles per WG {tiles / wgs:.2f}")
lib.qot_debug_set_variant(0)
