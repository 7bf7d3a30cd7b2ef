"""Timing + per-phase stamps of the weight-stationary NNConv kernel next to the tile kernel."""
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
bias = torch.randn(H, device=dev); out = torch.empty(N, H, device=dev); out2 = torch.empty(N, H, device=dev)
lib = _lib.load()
def run(name, o):
    _lib.call(name, P(x), H, P(b.edge_attr), P(w1), P(b1), P(g.rowptr), P(g.col), P(g.eid),
              P(g.invdeg), 0, P(wp), P(bias), P(o), N, H, D, 0, 0.0, 0.0, 0, None)
def t(name, o, iters=20):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run(name, o); torch.cuda.synchronize(); st.record()
    for _ in range(iters): run(name, o)
    en.record(); torch.cuda.synchronize(); return st.elapsed_time(en) / iters * 1e3
flops = 2.0 * N * (K + 2) * H * H
for rnd in range(3):
    for name, o in (("qot_nnconv_fused", out), ("qot_nnconv_fused_ws", out2)):
        m = t(name, o); print(f"{name:24s} {m:8.1f} us  {flops / m / 1e6:6.1f} TFLOP/s")
print("max abs diff", float((out - out2).abs().max()), "ref scale", float(out.abs().max()))
lib.qot_debug_ws_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.qot_debug_ws_stamps(None, 1); run("qot_nnconv_fused_ws", out2); torch.cuda.synchronize()
lib.qot_debug_ws_stamps(None, 1); run("qot_nnconv_fused_ws", out2); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)(); lib.qot_debug_ws_stamps(ctypes.cast(buf, ctypes.c_void_p), 0)
lib.qot_debug_ws_stamps(None, 2)
tiles = (N + 31) // 32
names = ["index loads + operands", "16 MFMAs + previous epilogue", "feature loads + barrier", "dma issue + 64 MFMAs", "meta/partials to LDS", "dma wait", "barrier"]
tot = sum(buf[:7])
for n, v in zip(names, buf[:7]):
    print(f"  {n:28s} {v / (tiles * 8):9.0f} ticks/wave/tile  {100.0 * v / tot:5.1f}%")
