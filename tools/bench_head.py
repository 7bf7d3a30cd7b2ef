"""qot_head_train (read-out forward + criterion + backward in one kernel) alone at cfg2's shape; with the diagnostic build
(QOT_LIB_PATH=tools/diag/libqot_gnn_diag.so) also its phase ablation."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_qot_estimation_amd import _lib
dev = torch.device("cuda:0")
lib = _lib.load()
B, n, H, O = int(os.environ.get("HD_B", 1024)), 100, 64, 3
N = B * n
f = lambda *s: torch.randn(*s, device=dev)
x, w0, b0, w3, b3, tgt = f(N, H), f(H, H) / 8, f(H), f(O, H) / 8, f(O), f(B, O)
ptr = (torch.arange(B + 1, device=dev, dtype=torch.int32) * n).contiguous()
out, gout, lrows, gx = torch.empty(B, O, device=dev), torch.empty(B, O, device=dev), torch.empty(B, device=dev), torch.empty(N, H, device=dev)
ws = torch.empty(lib.qot_head_bwd_workspace_floats(H, O), device=dev)
step = torch.ones((), dtype=torch.long, device=dev)
run = lambda: _lib.call("qot_head_train", x, ptr, w0, b0, w3, b3, tgt, 1.0, out, gout, lrows, gx, ws, B, H, O, 0.01, 0.5, 99, step,
                        1, 0.01, 0.5, 77, step)
def timeit(it=50):
    for _ in range(5): run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(it): run()
        en.record(); torch.cuda.synchronize()
        best = min(best, st.elapsed_time(en) / it * 1e3)
    return round(best, 2)
res = {"B": B, "head_train_us": timeit()}
if hasattr(lib, "qot_debug_head_variant"):
    lib.qot_debug_head_variant.argtypes = [ctypes.c_int]
    for name, v in (("no pool backward", 1), ("no dense phases", 2), ("no row loads", 4), ("no loads, no pool backward", 5), ("nothing", 7)):
        lib.qot_debug_head_variant(v); res[name] = timeit()
    lib.qot_debug_head_variant(0)
print(json.dumps(res))
