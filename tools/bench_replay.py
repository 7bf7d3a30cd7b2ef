"""Reference-scale training loop (V=75, H=16, batch 512; 1-layer GAT C=32): eager steps vs steps replayed
as HIP graphs per cached batch of an HBM-resident shard (harness.fit(replay=...))."""
import copy, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_qot_estimation_amd as q
from gnn_qot_estimation_amd import harness as Hn, synthetic as S


def timed(model, shard, kind, replay, epochs=14, bs=512):
    marks = []
    def log(msg):
        torch.cuda.synchronize(); marks.append(time.perf_counter())
    Hn.fit(model, shard, kind=kind, batch_size=bs, num_epochs=epochs, chunk_fraction=0.5, patience=epochs + 1,
           log=log, replay=replay)
    # marks: [start msg, epoch1, epoch2, ...]; steady state = last 8 epochs
    dt = (marks[-1] - marks[-9]) / 8
    return dt


torch.manual_seed(0)
# topological: 12 batches of 512 graphs (75 nodes, 150 directed edges each)
tb = S.tile_batch(S.topological_batch(2, 512, n=75, e=150), 12)
nodes, edges = 75, 150
B = tb.num_graphs
node_ptr = torch.arange(B + 1) * nodes
edge_ptr = torch.arange(B + 1) * edges
ei = tb.edge_index
shard = q.PackedGraphs(node_ptr, edge_ptr, ei, tb.edge_attr, tb.node_ids, None, tb.y, uniform_node_ids=75).to_device("cuda")
train_batches = int(B * 0.7 * 0.5) // 512 + 1
val_batches = int(B * 0.15) // 512 + 1
res = {}
for replay in (False, True):
    m = q.TopologicalGNN(75, 16, 3, 4, dropout_p=0.5)
    dt = timed(m, shard, "topological", replay)
    res["replay" if replay else "eager"] = dt
steps = train_batches + val_batches
print(json.dumps({"config": "reference scale topological V=75 H=16 B=512, HBM-resident shard", "batches_per_epoch": steps,
                  "eager_ms_per_epoch": round(res["eager"] * 1e3, 2), "replay_ms_per_epoch": round(res["replay"] * 1e3, 2),
                  "speedup": round(res["eager"] / res["replay"], 2)}))
