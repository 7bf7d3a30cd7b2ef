"""Generates the golden fixtures under tests/golden/ (run HERE, where /root/reference is
mounted; the fixtures -- pure tensor data -- then travel to the GPU box).

  <name>.pt = {"model_params": {...}, "state_dict": {reference checkpoint tensors},
               "inputs": {...seeded synthetic batch...}, "expected": oracle eval-mode output,
               "expected_dense64": independent fp64 restatement on the same inputs}

The reference's three shipped checkpoints (topological_training/models/model_0.pth,
lightpath_training/models/model_{0,1}.pth) are the only artefacts that pin the
state_dict contract.  They are loaded with weights_only=True.
Expected outputs come from the CPU oracle (oracle.sparse) and are cross-checked against
oracle.dense64 before being written.  PyG itself is not importable here, so these are
"parity unpinned" with respect to the reference's own arithmetic.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import dense64 as D64, sparse as O            # noqa: E402
from gnn_qot_estimation_amd import synthetic as S         # noqa: E402
import gnn_qot_estimation_amd as q                       # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def topo():
    ck = torch.load(f"{REF}/topological_training/models/model_0.pth", map_location="cpu", weights_only=True)
    p = ck["model_params"]
    m = O.TopologicalGNN(p["num_nodes"], p["hidden_channels"], p["output_dim"], p["edge_dim"], dropout_p=0.0).eval()
    m.load_state_dict(ck["model_state_dict"], strict=True)
    # 75-node graphs, few active nodes (to_graph.py:133-135): 6 graphs, 40 directed edges among the first 30 nodes
    graphs = []
    for g in range(6):
        gen = torch.Generator().manual_seed(100 + g)
        links = torch.randint(0, 30, (20, 2), generator=gen)
        links = links[links[:, 0] != links[:, 1]]
        attr = torch.rand(links.shape[0], 4, generator=gen)
        ei = torch.cat([links.t(), links.t().flip(0)], dim=1)
        graphs.append(q.Data(edge_index=ei, edge_attr=torch.cat([attr, attr]), node_ids=torch.arange(75), num_nodes=75))
    b = q.Batch.from_data_list(graphs)
    out = m(b)
    d64 = D64.topological_forward(ck["model_state_dict"], b)
    assert rel(out, d64) < 1e-5, rel(out, d64)
    torch.save({"model_params": p, "state_dict": ck["model_state_dict"],
                "inputs": {"edge_index": b.edge_index, "edge_attr": b.edge_attr, "node_ids": b.node_ids,
                           "batch": b.batch, "num_graphs": b.num_graphs},
                "expected": out.detach(), "expected_dense64": d64.detach()},
               os.path.join(ROOT, "tests/golden/topological_model_0.pt"))
    print("topological_model_0", tuple(out.shape), rel(out, d64))


def lightpath(k):
    ck = torch.load(f"{REF}/lightpath_training/models/model_{k}.pth", map_location="cpu", weights_only=True)
    p = ck["model_params"]
    lut = p["feature_indices"]["is_lut"]
    m = O.LightpathGNN(p["in_channels"], p["hidden_channels"], p["output_dim"], lut, dropout_p=0.0).eval()
    m.load_state_dict(ck["model_state_dict"], strict=True)
    b = S.lightpath_batch(24, cfg=30 + k)
    out, lb = m(b)
    d64, lb64 = D64.lightpath_forward(ck["model_state_dict"], b, lut)
    assert torch.equal(lb, lb64) and rel(out, d64) < 1e-5, rel(out, d64)
    # train-mode BN: running stats after 3 steps on the same batch
    m.train()
    for _ in range(3):
        m(b)
    torch.save({"model_params": p, "state_dict": ck["model_state_dict"],
                "inputs": {"x": b.x, "edge_index": b.edge_index, "batch": b.batch, "num_graphs": b.num_graphs},
                "expected": out.detach(), "expected_lut_batch": lb, "expected_dense64": d64.detach(),
                "bn_after_3_train_steps": {kk: v.clone() for kk, v in m.norm1.module.state_dict().items()}},
               os.path.join(ROOT, f"tests/golden/lightpath_model_{k}.pt"))
    print(f"lightpath_model_{k}", tuple(out.shape), rel(out, d64))


if __name__ == "__main__":
    topo()
    lightpath(0)
    lightpath(1)
