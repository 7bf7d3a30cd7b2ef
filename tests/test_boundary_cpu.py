"""CPU-side checks of the drop-in boundary: state_dict contract, the
C-ABI library exporting every declared symbol, collate layout (App. C), loud failure."""
import ctypes
import os
import re

import pytest
import torch

import gnn_qot_estimation_amd as q
from gnn_qot_estimation_amd import _lib, synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

TOPO_KEYS = ["node_embeddings.weight", "conv1.lin_key.weight", "conv1.lin_key.bias", "conv1.lin_query.weight",
             "conv1.lin_query.bias", "conv1.lin_value.weight", "conv1.lin_value.bias", "conv1.lin_edge.weight",
             "conv1.lin_skip.weight", "conv1.lin_skip.bias", "conv2.bias", "conv2.nn.0.weight", "conv2.nn.0.bias",
             "conv2.nn.2.weight", "conv2.nn.2.bias", "conv2.lin.weight", "mlp.0.weight", "mlp.0.bias",
             "mlp.3.weight", "mlp.3.bias"]
LP_KEYS = ["conv1.att_src", "conv1.att_dst", "conv1.bias", "conv1.lin.weight", "norm1.module.weight",
           "norm1.module.bias", "norm1.module.running_mean", "norm1.module.running_var",
           "norm1.module.num_batches_tracked", "mlp.0.weight", "mlp.0.bias", "mlp.3.weight", "mlp.3.bias"]


def test_state_dict_keys_shapes_match_appendix_a():
    m = q.TopologicalGNN(75, 16, 3, 4)
    sd = m.state_dict()
    assert list(sd.keys()) == TOPO_KEYS
    assert sd["conv1.lin_edge.weight"].shape == (16, 4) and sd["conv2.nn.2.weight"].shape == (256, 8)
    assert sum(p.numel() for p in m.parameters()) == 5291
    lp = q.LightpathGNN(5, 32, 3, 1)
    sd = lp.state_dict()
    assert list(sd.keys()) == LP_KEYS
    assert sd["conv1.att_src"].shape == (1, 4, 32) and sd["norm1.module.num_batches_tracked"].dtype == torch.int64
    assert sum(p.numel() for p in lp.parameters()) == 5507
    assert not any(k.startswith("_qot") for k in sd)


def test_oracle_and_hip_modules_share_state_dict():
    from oracle import sparse as O
    a, b = O.TopologicalGNN(20, 8, 3, 4), q.TopologicalGNN(20, 8, 3, 4)
    b.load_state_dict(a.state_dict(), strict=True)
    a2, b2 = O.LightpathGNN(5, 8, 3, 1), q.LightpathGNN(5, 8, 3, 1)
    b2.load_state_dict(a2.state_dict(), strict=True)


@pytest.mark.parametrize("name", ["topological_model_0", "lightpath_model_0", "lightpath_model_1"])
def test_shipped_checkpoint_fixture_loads_strict(name):
    """Golden fixtures carry the reference checkpoints' tensors (data, not code); they must
    load strict=True into the build's modules (reference: */test.py:58-69)."""
    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=True)
    p = fx["model_params"]
    if name.startswith("topo"):
        m = q.TopologicalGNN(p["num_nodes"], p["hidden_channels"], p["output_dim"], p["edge_dim"], dropout_p=0.0)
    else:
        m = q.LightpathGNN(p["in_channels"], p["hidden_channels"], p["output_dim"],
                           p["feature_indices"]["is_lut"], dropout_p=0.0)
    m.load_state_dict(fx["state_dict"], strict=True)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "qot_gnn.h")).read()
    declared = set(re.findall(r"\b(qot_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().qot_abi_version() == _lib.ABI_VERSION
    assert b"unsupported" in _lib.load().qot_error_string(-1)


def test_collate_layout_appendix_c():
    g1 = q.Data(edge_index=torch.tensor([[0, 1], [1, 0]]), edge_attr=torch.ones(2, 4), node_ids=torch.arange(2),
                y=torch.zeros(3), num_nodes=2)
    g2 = q.Data(edge_index=torch.tensor([[0, 2], [2, 0]]), edge_attr=torch.zeros(2, 4), node_ids=torch.arange(3),
                y=torch.ones(3), num_nodes=3)
    b = q.Batch.from_data_list([g1, g2])
    assert b.edge_index.tolist() == [[0, 1, 2, 4], [1, 0, 4, 2]]       # offset
    assert b.node_ids.tolist() == [0, 1, 0, 1, 2]                      # NOT offset
    assert b.batch.tolist() == [0, 0, 1, 1, 1] and b.ptr.tolist() == [0, 2, 5]
    assert b.y.shape == (6,) and b.num_graphs == 2 and b.x is None
    s = q.shard_graphs(b, 1, 2)
    assert s.edge_index.tolist() == [[0, 2], [2, 0]] and s.node_ids.tolist() == [0, 1, 2] and s.num_graphs == 1


def test_hip_path_refuses_cpu_tensors():
    m = q.TopologicalGNN(14, 32, 3, 4)
    with pytest.raises(_lib.QotError, match="no CPU fallback"):
        m(S.topological_batch(1, 2))
    lp = q.LightpathGNN(5, 8, 3, 1)
    with pytest.raises(_lib.QotError):
        lp(S.lightpath_batch(2))


def test_cpu_model_error_names_the_opt_in_and_the_opt_in_never_computes_on_cpu(monkeypatch):
    """configs[0] / the reference's ``topological_training/train.py:62`` (device pinned to CPU): the default is a loud
    error that names ``QOT_AUTO_DEVICE``; with the switch set and no GPU visible it is still an error -- the switch
    uploads to the GPU, it is not a CPU path."""
    m = q.TopologicalGNN(14, 32, 3, 4)
    monkeypatch.delenv("QOT_AUTO_DEVICE", raising=False)
    with pytest.raises(_lib.QotError, match="QOT_AUTO_DEVICE=1"):
        m(S.topological_batch(1, 2))
    if not torch.cuda.is_available():
        monkeypatch.setenv("QOT_AUTO_DEVICE", "1")
        with pytest.raises(_lib.QotError, match="no GPU is visible"):
            m(S.topological_batch(1, 2))
        with pytest.raises(_lib.QotError, match="no GPU is visible"):
            q.LightpathGNN(5, 8, 3, 1)(S.lightpath_batch(2))


def test_unsupported_configurations_raise():
    with pytest.raises(NotImplementedError):
        q.TransformerConv(8, 8, heads=2, edge_dim=4)
    with pytest.raises(NotImplementedError):
        q.NNConv(8, 8, nn=torch.nn.Linear(4, 64), aggr="add")
    with pytest.raises(NotImplementedError):
        q.GATConv(5, 8, heads=4, concat=False)


def test_synthetic_generators_are_seeded_and_well_formed():
    a, b = S.topological_batch(2, 4, n=100, e=400), S.topological_batch(2, 4, n=100, e=400)
    assert torch.equal(a.edge_index, b.edge_index) and torch.equal(a.edge_attr, b.edge_attr)
    assert a.edge_index.shape == (2, 1600) and a.num_nodes == 400
    src, dst = a.edge_index
    assert (src != dst).all() and (src // 100 == dst // 100).all()                # no self loops, block diagonal
    key = src * 1000 + dst
    assert key.unique().numel() == key.numel()                                    # no duplicate directed edges
    p5 = S.topological_batch(5, 2, n=500)
    assert int(torch.bincount(p5.edge_index[1]).max()) <= 64
    lp = S.lightpath_batch(50)
    assert (lp.x[:, 1] == 1.0).sum() == 50 and lp.y.shape == (50, 3)
