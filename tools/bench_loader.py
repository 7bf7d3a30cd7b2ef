"""PCIe-inclusive train throughput: every step's batch comes from host memory through GraphLoader
(pinned pre-tensorised shard, slices DMA'd on a side stream while the previous step computes).
Eager launches (the batch buffers rotate, so the step is not graph-captured here)."""
import json, os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_qot_estimation_amd as q
from gnn_qot_estimation_amd import synthetic as S
from gnn_qot_estimation_amd.dp import FlatModel, FusedSGD

dev = torch.device("cuda:0")
B, NB = 1024, 8
base = S.topological_batch(2, 128, n=100, e=400)
big = S.tile_batch(base, B * NB // 128)
shard = q.PackedGraphs(big.ptr.clone(), torch.arange(0, big.num_edges + 1, 400), big.edge_index, big.edge_attr,
                       big.node_ids, None, big.y, 100).pin()
torch.manual_seed(0)
model = q.TopologicalGNN(100, 64, 3, 4, dropout_p=0.5).to(dev).train()
flat = FlatModel(model); opt = FusedSGD(flat, lr=0.1, momentum=0.9)

def step(batch):
    flat.detach_grads()
    loss = F.smooth_l1_loss(model(batch), batch.y.view(-1, 3))
    loss.backward(); flat.gather_grads(); opt.step()
    return loss

def epoch(loader):
    n = 0
    for batch in loader:
        step(batch); n += batch.num_graphs
    return n

loader = q.GraphLoader(shard, batch_size=B, device=dev)
epoch(loader); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
for _ in range(5): n += epoch(loader)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
# same eager step on a resident batch, for the ratio
res = next(iter(q.GraphLoader(shard, batch_size=B, device=dev)))
for _ in range(5): step(res)
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(40): step(res)
torch.cuda.synchronize(); dr = (time.perf_counter() - t1) / 40
mb = (big.edge_index.numel() * 8 + big.edge_attr.numel() * 4 + big.node_ids.numel() * 8 + big.batch.numel() * 8) / NB / 1e6
print(json.dumps({"pcie_inclusive_graphs_per_s": round(n / dt), "ms_per_step": round(dt / (n / B) * 1e3, 3),
                  "resident_eager_graphs_per_s": round(B / dr), "resident_eager_ms_per_step": round(dr * 1e3, 3),
                  "host_to_device_MB_per_step": round(mb, 1)}))

# loader alone (no training): how fast can batches be delivered?
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
for _ in range(5):
    for batch in q.GraphLoader(shard, batch_size=B, device=dev):
        n += batch.num_graphs
torch.cuda.synchronize(); dl = time.perf_counter() - t0
print(json.dumps({"loader_only_graphs_per_s": round(n / dl), "ms_per_batch": round(dl / (n / B) * 1e3, 3),
                  "GBps": round(mb * (n / B) / dl / 1e3, 1)}))
# raw pinned H2D bandwidth for reference
src = torch.empty(64 * 2**20, dtype=torch.uint8).pin_memory(); dst = torch.empty_like(src, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): dst.copy_(src, non_blocking=True)
torch.cuda.synchronize(); print(json.dumps({"pinned_h2d_GBps": round(10 * 64 * 2**20 / (time.perf_counter() - t0) / 1e9, 1)}))

# graph-replay variant: static input buffers (fixed-size batches), D2D from the loader slot, replay
static = q.Batch()
first = next(iter(q.GraphLoader(shard, batch_size=B, device=dev)))
for name in ("edge_index", "edge_attr", "node_ids", "batch", "y", "ptr"):
    setattr(static, name, getattr(first, name).clone())
static.x = None; static.num_graphs = B; static._num_nodes = first.num_nodes; static.uniform_node_ids = 100
ys = static.y.view(-1, 3)
def fwd_bwd():
    static._qot_cache = {}
    flat.detach_grads()
    loss = F.smooth_l1_loss(model(static), ys)
    loss.backward(); flat.gather_grads()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): fwd_bwd(); opt.step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1): fwd_bwd()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2): opt.step()
def epoch_graph(loader):
    n = 0
    for batch in loader:
        for name in ("edge_index", "edge_attr", "node_ids", "batch", "y"):
            getattr(static, name).copy_(getattr(batch, name), non_blocking=True)
        g1.replay(); g2.replay(); n += batch.num_graphs
    return n
epoch_graph(loader); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
for _ in range(10): n += epoch_graph(loader)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"pcie_inclusive_graph_replay_graphs_per_s": round(n / dt), "ms_per_step": round(dt / (n / B) * 1e3, 3)}))
