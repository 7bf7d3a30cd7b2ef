"""CPU oracle train-step time vs thread count (choose a fair cpu_baseline configuration)."""
import os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_qot_estimation_amd import synthetic as S
from oracle import sparse as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
b = S.topological_batch(2, B, n=100, e=400); y = b.y.view(-1, 3)
for thr in [int(t) for t in (sys.argv[2:] or ["8", "16", "32", "64", "128", "256"])]:
    torch.set_num_threads(thr)
    torch.manual_seed(0)
    m = O.TopologicalGNN(100, 64, 3, 4, dropout_p=0.5).train()
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9)
    def step():
        opt.zero_grad(); F.smooth_l1_loss(m(b), y).backward(); opt.step()
    step(); t0 = time.perf_counter(); step(); dt = time.perf_counter() - t0
    print(f"B={B} threads={thr:4d}  {dt:7.2f} s/step  {B / dt:7.2f} graphs/s", flush=True)
