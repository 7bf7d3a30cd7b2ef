"""Capture one piece of the step in a HIP graph and replay it (isolates capture-unsafe code)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_qot_estimation_amd import synthetic as S
from gnn_qot_estimation_amd.graph import build_graph_index
from gnn_qot_estimation_amd.functional import gemm_tn
what = sys.argv[1]
dev = torch.device("cuda:0")
b = S.tile_batch(S.topological_batch(2, 128, n=100, e=400), 8).to(dev)
N = b.num_nodes
a = torch.randn(N, 640, device=dev); g = torch.randn(N, 64, device=dev)
def body():
    if what == "csr":
        gi = build_graph_index(b.edge_index, N)
        return gi.rowptr.sum() + gi.col.sum() + gi.pos_t.sum()
    if what == "gemm_tn":
        return gemm_tn(a, g).sum()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): ref = body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = body()
torch.cuda.synchronize()
for _ in range(3): gr.replay()
torch.cuda.synchronize()
print(what, "capture+replay ok", float(out), float(ref))
