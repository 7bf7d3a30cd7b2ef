"""Host-side profile of eager train steps at the reference's own scale (V=75, H=16, B=512): where the
Python time of the drop-in path goes (the GPU work of such a step is ~0.2 ms)."""
import cProfile, os, pstats, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_qot_estimation_amd as q
from gnn_qot_estimation_amd import synthetic as S
from gnn_qot_estimation_amd.dp import FlatModel, FusedSGD

dev = torch.device("cuda:0")
torch.manual_seed(0)
b = S.tile_batch(S.topological_batch(2, 64, n=75, e=60), 8).to(dev)
model = q.TopologicalGNN(75, 16, 3, 4).to(dev).train()
flat = FlatModel(model)
opt = FusedSGD(flat, lr=0.01, momentum=0.9)


def step():
    b._qot_cache = {}
    flat.detach_grads()
    loss = F.smooth_l1_loss(model(b), b.y.view(-1, 3))
    loss.backward()
    opt.step(grads=True)


for _ in range(20):
    step()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step")
torch.autograd.set_multithreading_enabled(False)     # backward Functions on this thread: visible to cProfile
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats("gnn_qot_estimation_amd|built-in|method", 45)
