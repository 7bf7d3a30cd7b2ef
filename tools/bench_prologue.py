"""The forward prologue's multi-role launch at cfg2, role by role and together (qot_run_roles through the Python layer)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_qot_estimation_amd as q
from gnn_qot_estimation_amd import _lib, functional as QF, launch_group as LG, synthetic as S
from gnn_qot_estimation_amd.graph import graph_index_for, table_maps_for
dev = torch.device("cuda:0")
b = S.topological_batch(2, 1024, n=100, e=400).to(dev)
m = q.TopologicalGNN(100, 64, 3, 4).to(dev)
def build(which):
    grp = LG.LaunchGroup()
    b._qot_cache = {}
    if "csr" in which:
        g = graph_index_for(b, b.num_nodes, group=grp)
    if "proj" in which or "scores" in which:
        c = m.conv1
        ts = [t.detach() for t in (m.node_embeddings.weight, c.lin_query.weight, c.lin_query.bias, c.lin_key.weight, c.lin_key.bias,
                                   c.lin_value.weight, c.lin_value.bias, c.lin_skip.weight, c.lin_skip.bias, c.lin_edge.weight)]
        t4 = torch.empty(100, 256, device=dev); M = torch.empty(100, 100, device=dev); Pm = torch.empty(100, 4, device=dev)
        if "proj" in which:
            grp.add(_lib.ROLE_TABLE_PROJECT_FWD, (*ts[:9], t4, None, None), (100, 64))
        if "scores" in which:
            grp.add(_lib.ROLE_TABLE_SCORES, (ts[0], ts[1], ts[2], ts[3], ts[4], ts[9], M, Pm), (100, 64, 4))
    if "pack" in which:
        m.conv2.prepack(grp)
    return grp
def timeit(which, it=40):
    for _ in range(5): build(which).run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        groups = [build(which) for _ in range(it)]
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for g in groups:
            roles, g.roles, g.post, g.on_success = g.roles, [], [], []
            _lib.run_roles(roles)
        en.record(); torch.cuda.synchronize()
        best = min(best, st.elapsed_time(en) / it * 1e3)
    return round(best, 2)
from gnn_qot_estimation_amd import graph as G
G.CHECK_INDEX_STATUS = False
res = {w: timeit(w.split("+")) for w in ("csr", "proj", "scores", "pack", "csr+proj", "csr+proj+pack", "csr+proj+scores+pack")}
print(json.dumps(res))
