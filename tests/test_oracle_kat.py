"""Hand-derived known-answer tests pinning the oracle.

Expected values are computed here from the Appendix B formulas with scalar ``math`` /
explicit small-tensor arithmetic that shares no code with ``oracle.sparse`` or
``oracle.dense64``.  The reference has no tests of its own for this path ("parity
unpinned" by the reference); these KATs + the dual restatement are what pin the oracle.
"""
import math

import pytest
import torch

from oracle import dense64 as D64
from oracle import sparse as O


def _tconv(h=2, d=1):
    torch.manual_seed(1)
    m = O.TransformerConv(h, h, edge_dim=d)
    return m


def test_tconv_two_nodes_one_edge():
    """One edge 0->1: softmax over a single logit is 1/(1+1e-16) = 1 in fp32, so
    out_1 = v_0 + We ea + skip_1 and out_0 = skip_0 (zero in-degree)."""
    m = _tconv(h=2, d=1)
    x = torch.tensor([[1.0, 2.0], [-0.5, 0.25]])
    ei = torch.tensor([[0], [1]])
    ea = torch.tensor([[0.7]])
    out = m(x, ei, ea)
    sd = m.state_dict()
    v0 = sd["lin_value.weight"] @ x[0] + sd["lin_value.bias"]
    eps = sd["lin_edge.weight"] @ ea[0]
    skip = x @ sd["lin_skip.weight"].t() + sd["lin_skip.bias"]
    assert torch.allclose(out[1], v0 + eps + skip[1], atol=1e-6)
    assert torch.allclose(out[0], skip[0], atol=1e-7)


def test_tconv_star_softmax_weights():
    """3-node star into node 0 with identity-like weights chosen so logits are known:
    q_0 = [1,0], k_1 = [a,0], k_2 = [b,0], no edge term -> alpha = softmax([a,b]/sqrt(2))."""
    m = O.TransformerConv(2, 2, edge_dim=1)
    with torch.no_grad():
        for lin in (m.lin_query, m.lin_key, m.lin_value, m.lin_skip):
            lin.weight.copy_(torch.eye(2)); lin.bias.zero_()
        m.lin_edge.weight.zero_()
        m.lin_skip.weight.zero_()
    a, b = 0.3, -1.2
    x = torch.tensor([[1.0, 0.0], [a, 5.0], [b, -3.0]])
    ei = torch.tensor([[1, 2], [0, 0]])
    out = m(x, ei, torch.zeros(2, 1))
    sa, sb = a / math.sqrt(2), b / math.sqrt(2)
    za, zb = math.exp(sa - max(sa, sb)), math.exp(sb - max(sa, sb))
    wa, wb = za / (za + zb + 1e-16), zb / (za + zb + 1e-16)
    exp0 = torch.tensor([wa * a + wb * b, wa * 5.0 + wb * -3.0])
    assert torch.allclose(out[0], exp0, atol=1e-6)
    assert torch.allclose(out[1], torch.zeros(2)) and torch.allclose(out[2], torch.zeros(2))


def test_tconv_edge_term_enters_key_and_value():
    """With x = 0 and zero biases: k_j + e = e, v_j + e = e.  Two edges into node 0 with
    edge features f1, f2 and lin_edge = [[1],[0]], q bias = [c,0]:
    logits = c*f/sqrt(2); out_0 = sum alpha_e * [f_e, 0]."""
    m = O.TransformerConv(2, 2, edge_dim=1)
    c, f1, f2 = 2.0, 0.5, 1.5
    with torch.no_grad():
        for lin in (m.lin_query, m.lin_key, m.lin_value, m.lin_skip):
            lin.weight.zero_(); lin.bias.zero_()
        m.lin_query.bias.copy_(torch.tensor([c, 0.0]))
        m.lin_edge.weight.copy_(torch.tensor([[1.0], [0.0]]))
    x = torch.zeros(3, 2)
    out = m(x, torch.tensor([[1, 2], [0, 0]]), torch.tensor([[f1], [f2]]))
    s1, s2 = c * f1 / math.sqrt(2), c * f2 / math.sqrt(2)
    z1, z2 = math.exp(s1 - s2), 1.0
    w1, w2 = z1 / (z1 + z2), z2 / (z1 + z2)
    assert torch.allclose(out[0], torch.tensor([w1 * f1 + w2 * f2, 0.0]), atol=1e-6)


def test_nnconv_mean_and_theta_layout():
    """theta_e viewed [H_in, H_out] row-major (flat index a*H_out + o): choose the edge MLP
    so theta is a known matrix; two in-edges -> mean; zero in-degree -> root + bias only."""
    hin = hout = 2
    nn_ = torch.nn.Sequential(torch.nn.Linear(1, 2), torch.nn.ReLU(), torch.nn.Linear(2, hin * hout))
    m = O.NNConv(hin, hout, nn_)
    with torch.no_grad():
        nn_[0].weight.copy_(torch.tensor([[1.0], [-1.0]])); nn_[0].bias.zero_()     # h = [relu(f), relu(-f)]
        nn_[2].weight.copy_(torch.tensor([[1.0, 0.0], [2.0, 0.0], [3.0, 0.0], [4.0, 0.0]]))  # theta = f*[[1,2],[3,4]] for f>0
        nn_[2].bias.copy_(torch.tensor([0.5, 0.0, 0.0, -0.5]))
        m.lin.weight.copy_(torch.tensor([[1.0, 0.0], [0.0, 1.0]])); m.bias.copy_(torch.tensor([10.0, 20.0]))
    x = torch.tensor([[1.0, 1.0], [2.0, 0.0], [0.0, 3.0]])
    ei = torch.tensor([[1, 2], [0, 0]])
    ea = torch.tensor([[2.0], [-1.0]])      # second edge: f<0 -> h=[0,1] -> W2 column 1 is zero -> theta = bias only
    out = m(x, ei, ea)
    th1 = torch.tensor([[2.0 * 1 + 0.5, 2.0 * 2], [2.0 * 3, 2.0 * 4 - 0.5]])
    th2 = torch.tensor([[0.5, 0.0], [0.0, -0.5]])
    msg = (x[1] @ th1 + x[2] @ th2) / 2.0
    assert torch.allclose(out[0], msg + x[0] + torch.tensor([10.0, 20.0]), atol=1e-6)
    assert torch.allclose(out[1], x[1] + torch.tensor([10.0, 20.0]))


def test_gat_self_loop_rules():
    """Existing self loops are dropped and exactly one (n,n) is appended per node; a node
    with no other in-edge attends only to itself -> out = z_n + bias."""
    ei = torch.tensor([[0, 1, 1, 0], [1, 1, 1, 0]])      # 0->1, two self loops on 1, self loop on 0
    e2 = O.gat_edge_set(ei, 3)
    assert e2.tolist() == [[0, 0, 1, 2], [1, 0, 1, 2]]
    torch.manual_seed(3)
    m = O.GATConv(3, 2, heads=4)
    with torch.no_grad():
        m.bias.uniform_(-1, 1)
    x = torch.randn(3, 3)
    out = m(x, ei)
    z = x @ m.lin.weight.t()
    assert torch.allclose(out[2], z[2] + m.bias, atol=1e-6)     # isolated node: alpha = 1
    assert torch.allclose(out[0], z[0] + m.bias, atol=1e-6)     # only its own self loop
    # node 1: neighbours {0, 1}; per head softmax of leaky_relu(a_s[j] + a_d[1])
    zv = z.view(3, 4, 2)
    a_s = (zv * m.att_src[0]).sum(-1); a_d = (zv * m.att_dst[0]).sum(-1)
    exp = torch.zeros(4, 2)
    for h in range(4):
        s = [float(a_s[j, h] + a_d[1, h]) for j in (0, 1)]
        s = [v if v > 0 else 0.2 * v for v in s]
        mx = max(s); p = [math.exp(v - mx) for v in s]; den = sum(p) + 1e-16
        exp[h] = (p[0] / den) * zv[0, h] + (p[1] / den) * zv[1, h]
    assert torch.allclose(out[1], exp.reshape(-1) + m.bias, atol=1e-6)


def test_pool_counts_and_empty_graph():
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
    batch = torch.tensor([0, 0, 2])
    out = O.global_mean_pool(x, batch)
    assert out.tolist() == [[2.0, 3.0], [0.0, 0.0], [5.0, 6.0]]     # B = batch.max()+1, empty graph -> 0 (count clamp)


def test_batchnorm_running_stats_formula():
    """App. B.4: normalise with biased var, update running stats with UNBIASED var."""
    bn = O.BatchNorm(2)
    x = torch.tensor([[1.0, 10.0], [3.0, 30.0], [5.0, 50.0]])
    bn.train()
    y = bn(x)
    mean = torch.tensor([3.0, 30.0]); var_b = torch.tensor([8.0 / 3, 800.0 / 3]); var_u = torch.tensor([4.0, 400.0])
    assert torch.allclose(y, (x - mean) / torch.sqrt(var_b + 1e-5), atol=1e-5)
    assert torch.allclose(bn.module.running_mean, 0.1 * mean)
    assert torch.allclose(bn.module.running_var, 0.9 * torch.ones(2) + 0.1 * var_u)
    assert int(bn.module.num_batches_tracked) == 1


def test_lightpath_value_error_and_tuple():
    from gnn_qot_estimation_amd import synthetic as S
    m = O.LightpathGNN(5, 8, 3, 1)
    m.eval()
    out, lb = m(S.lightpath_batch(6))
    assert out.shape == (6, 3) and lb.tolist() == list(range(6))
    with pytest.raises(ValueError, match="No LUT node found in the batch."):
        m(S.lightpath_batch(3, lut=False))
