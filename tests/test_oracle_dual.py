"""Dual-restatement agreement and gradcheck (iii).

``oracle.sparse`` (fp32 scatter form, PyG-style) vs ``oracle.dense64`` (fp64 per-node loops,
no shared code): <= 1e-5 rel on fp32-limited comparisons, <= 1e-10 when the sparse path is
itself run in fp64.
"""
import pytest
import torch

from helpers import rel_err
from oracle import dense64 as D64
from oracle import sparse as O
from gnn_qot_estimation_amd import synthetic as S
import gnn_qot_estimation_amd as q


def _topo_batch():
    b = S.topological_batch(2, 3, n=12, e=30)
    # add an isolated node set, duplicate edge and self loop to the mix
    extra = q.Data(edge_index=torch.tensor([[0, 0, 2, 3], [1, 1, 2, 0]]), edge_attr=torch.rand(4, 4),
                   node_ids=torch.arange(6), num_nodes=6, y=torch.rand(3))
    return q.Batch.from_data_list([extra] + [_g for _g in _split(b)])


def _split(batch):
    out = []
    for g in range(batch.num_graphs):
        out.append(q.shard_graphs(batch, g, batch.num_graphs))
    return [q.Data(edge_index=s.edge_index, edge_attr=s.edge_attr, node_ids=s.node_ids, num_nodes=s.num_nodes,
                   y=s.y) for s in out]


def test_topological_sparse_vs_dense64():
    torch.manual_seed(0)
    m = O.TopologicalGNN(12, 8, 3, 4, dropout_p=0.0).eval()
    b = _topo_batch()
    out = m(b)
    ref = D64.topological_forward(m.state_dict(), b)
    assert rel_err(out, ref) <= 1e-5
    m64 = O.TopologicalGNN(12, 8, 3, 4, dropout_p=0.0).double().eval()
    m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    b64 = b.to("cpu"); b64.edge_attr = b.edge_attr.double()
    assert rel_err(m64(b64), ref) <= 1e-10


@pytest.mark.parametrize("train", [False, True])
def test_lightpath_sparse_vs_dense64(train):
    torch.manual_seed(0)
    m = O.LightpathGNN(5, 8, 3, 1, dropout_p=0.0)
    with torch.no_grad():
        m.conv1.bias.uniform_(-0.5, 0.5)
        m.norm1.module.running_mean.uniform_(-0.2, 0.2); m.norm1.module.running_var.uniform_(0.5, 1.5)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    b = S.lightpath_batch(7)
    m.train(train)
    out, lb = m(b)
    ref, lb64 = D64.lightpath_forward(sd0, b, 1, train_stats=train)
    assert torch.equal(lb, lb64)
    assert rel_err(out, ref) <= 1e-5


def test_gat_existing_self_loops_dual():
    torch.manual_seed(0)
    m = O.GATConv(5, 4, heads=4)
    x = torch.randn(5, 5)
    ei = torch.tensor([[0, 1, 1, 2, 2, 3, 0], [1, 0, 1, 2, 3, 2, 1]])
    out = m(x, ei)
    ref = D64.gat_conv({"c." + k: v for k, v in m.state_dict().items()}, "c.", x, ei)
    assert rel_err(out, ref) <= 1e-5


def test_gradcheck_fp64_operators():
    """torch.autograd.gradcheck on the fp64 sparse operators (they are the gradient oracle)."""
    torch.manual_seed(0)
    ei = torch.tensor([[0, 1, 2, 2, 3, 0], [1, 0, 1, 3, 2, 0]])
    ea = torch.rand(6, 2, dtype=torch.float64)
    x = torch.randn(4, 4, dtype=torch.float64, requires_grad=True)
    tc = O.TransformerConv(4, 4, edge_dim=2).double()
    assert torch.autograd.gradcheck(lambda t: tc(t, ei, ea), (x,), atol=1e-6)
    enn = torch.nn.Sequential(torch.nn.Linear(2, 4), torch.nn.ReLU(), torch.nn.Linear(4, 16)).double()
    nc = O.NNConv(4, 4, enn).double()
    assert torch.autograd.gradcheck(lambda t: nc(t, ei, ea), (x,), atol=1e-6)
    gc = O.GATConv(4, 2, heads=4).double()
    assert torch.autograd.gradcheck(lambda t: gc(t, ei), (x,), atol=1e-6)


def test_factorised_nnconv_identity():
    """App. B.2 factorisation used by the HIP path: A @ Wcat == PyG-style message/mean."""
    from gnn_qot_estimation_amd.functional import nnconv_wcat
    torch.manual_seed(0)
    h, d = 8, 4
    enn = torch.nn.Sequential(torch.nn.Linear(d, 2 * d), torch.nn.ReLU(), torch.nn.Linear(2 * d, h * h))
    m = O.NNConv(h, h, enn)
    b = S.topological_batch(2, 2, n=10, e=24)
    x = torch.randn(b.num_nodes, h)
    ref = m(x, b.edge_index, b.edge_attr)
    src, dst = b.edge_index
    hh = torch.relu(enn[0](b.edge_attr))                                  # [E, K]
    n = x.shape[0]
    deg = torch.zeros(n).scatter_add_(0, dst, torch.ones(dst.numel())).clamp(min=1)
    blocks = [torch.zeros(n, h).index_add_(0, dst, hh[:, k:k + 1] * x[src]) / deg[:, None] for k in range(2 * d)]
    blocks.append(torch.zeros(n, h).index_add_(0, dst, x[src]) / deg[:, None])
    blocks.append(x)
    A = torch.cat(blocks, dim=1)
    out = A @ nnconv_wcat(enn[2].weight, enn[2].bias, m.lin.weight, h, h, 2 * d) + m.bias
    assert rel_err(out, ref) <= 1e-5


def test_topological_three_layers_sparse_vs_dense64():
    """``num_layers=3`` (cfg4/cfg5 architecture): TransformerConv + NNConv + NNConv."""
    torch.manual_seed(0)
    m = O.TopologicalGNN(12, 8, 3, 4, dropout_p=0.0, num_layers=3).eval()
    with torch.no_grad():
        m.conv3.bias.uniform_(-0.3, 0.3)
    b = _topo_batch()
    ref = D64.topological_forward(m.state_dict(), b, extra_nnconv=("conv3.",))
    assert rel_err(m(b), ref) <= 1e-5
    # the default still is the 2-layer reference model, same keys as before
    assert [k for k in O.TopologicalGNN(12, 8, 3, 4).state_dict() if k.startswith("conv3")] == []


@pytest.mark.parametrize("train", [False, True])
def test_lightpath_three_layers_sparse_vs_dense64(train):
    """``num_layers=3`` (cfg3 architecture): 3 x (GATConv(heads=4) + BatchNorm + relu)."""
    torch.manual_seed(0)
    m = O.LightpathGNN(5, 4, 3, 1, dropout_p=0.0, num_layers=3)
    with torch.no_grad():
        for l in (1, 2, 3):
            getattr(m, f"conv{l}").bias.uniform_(-0.5, 0.5)
            bn = getattr(m, f"norm{l}").module
            bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    b = S.lightpath_batch(6)
    m.train(train)
    out, lb = m(b)
    ref, lb64 = D64.lightpath_forward(sd0, b, 1, train_stats=train,
                                      extra_layers=(("conv2.", "norm2.module."), ("conv3.", "norm3.module.")))
    assert torch.equal(lb, lb64)
    assert rel_err(out, ref) <= 2e-5
