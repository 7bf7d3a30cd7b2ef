"""A/B of nnconv_adjoint_dw64 builds (interleaved rounds in ONE process, cfg2 shape): QOT_LIB_A / QOT_LIB_B paths."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_qot_estimation_amd import _lib, functional as QF, synthetic as S
from gnn_qot_estimation_amd.graph import build_graph_index
dev = torch.device("cuda:0")
b = S.topological_batch(2, 1024, n=100, e=400).to(dev)
N, H, D, K = b.num_nodes, 64, 4, 8
g = build_graph_index(b.edge_index, N)
f = lambda *s: torch.randn(*s, device=dev)
x, gout, w1, b1 = f(N, H), f(N, H), f(K, D), f(K)
wp = f((K + 2) * H * H)[QF.nnconv_perm_index((K + 2) * H, dev)].contiguous()
out, gwt = torch.empty(N, H, device=dev), torch.empty((K + 2) * H, H, device=dev)
libs = {}
for tag in ("A", "B"):
    path = os.environ.get("QOT_LIB_" + tag)
    lib = C.CDLL(path)
    fn = lib.qot_nnconv_adjoint_dw
    fn.restype = C.c_int
    fn.argtypes = _lib.SIGNATURES["qot_nnconv_adjoint_dw"][1]
    wsf = lib.qot_nnconv_adjoint_dw_workspace_floats
    wsf.restype = C.c_size_t; wsf.argtypes = [C.c_int]
    libs[tag] = (fn, torch.empty(wsf(D), device=dev), path)
def run(tag):
    fn, ws, _ = libs[tag]
    rc = fn(gout.data_ptr(), H, x.data_ptr(), H, b.edge_attr.data_ptr(), w1.data_ptr(), b1.data_ptr(), g.rowptr_t.data_ptr(),
            g.col_t.data_ptr(), g.eid_t.data_ptr(), g.invdeg.data_ptr(), wp.data_ptr(), out.data_ptr(), gwt.data_ptr(), 2,
            ws.data_ptr(), N, H, D, _lib.stream())
    assert rc == 0, rc
def t(tag, it=40):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(it): run(tag)
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / it * 1e3
for tag in libs:
    for _ in range(80): run(tag)
res = {"A": [], "B": []}
for rnd in range(6):
    for tag in ("A", "B"):
        res[tag].append(round(t(tag), 1))
print(json.dumps({k: {"lib": libs[k][2], "us": v, "min": min(v), "median": sorted(v)[len(v) // 2]} for k, v in res.items()}))
