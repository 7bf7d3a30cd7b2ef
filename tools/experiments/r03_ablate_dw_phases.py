"""Phase stamps of the width-generic weight-gradient kernel (diagnostic build: `make -C gnn_qot_estimation_amd/csrc DIAG=1`,
QOT_LIB_PATH=tools/diag/libqot_gnn_diag.so).  Slots: producers 0 issue side, 1 plan, 2 fast FMAs (waits for the rows), 3 tail
loop, 4 tile stores, 5 barrier; consumers 6 multiply, 7 barrier.  Cycles per tile iteration of one wave, averaged."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_qot_estimation_amd import _lib, synthetic as S
from gnn_qot_estimation_amd.graph import build_graph_index
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
P = _lib.ptr
lib = _lib.load()
lib.qot_debug_gen_variant.argtypes = [ctypes.c_int]
lib.qot_debug_gen_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
b = S.tile_batch(S.topological_batch(cfg, 16, n=1000, e=4000), B // 16).to(dev)
N, D, K = b.num_nodes, 4, 8
g = build_graph_index(b.edge_index, N)
x, gout = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
w1, b1 = torch.randn(K, D, device=dev), torch.randn(K, device=dev)
gpar = torch.empty((K + 2) * H * H, device=dev)
ws = torch.empty(lib.qot_nnconv_dw_workspace_floats(N, H, D), device=dev)
run = lambda: _lib.call("qot_nnconv_dw", P(x), H, P(gout), H, P(b.edge_attr), P(w1), P(b1), P(g.rowptr), P(g.col), P(g.eid),
                        P(g.invdeg), P(gpar), P(ws), N, H, D)
run(); torch.cuda.synchronize()
lib.qot_debug_gen_variant(int(os.environ.get("DW_VARIANT", "7")))
lib.qot_debug_gen_stamps(None, 1)
run(); torch.cuda.synchronize()
host = (ctypes.c_ulonglong * 8)()
lib.qot_debug_gen_stamps(host, 0)
lib.qot_debug_gen_variant(0)
ntiles = (N + 31) // 32
nslice = (H // 32) * (2 if H == 256 else 1)
iters = ntiles * nslice          # tile iterations summed over all workgroups (one stamping wave per role)
names = ["issue_idx", "issue_x", "plan", "fast", "tail_stores", "p_barrier", "multiply", "c_barrier"]
print(json.dumps({"H": H, "N": N, "cycles_per_tile": {n: round(host[i] / iters, 1) for i, n in enumerate(names)}}))
