"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, forward and backward.

Tolerance: <= 1e-4 relative (max-abs error over max-abs reference), fp32 -- the bar
BASELINE.json's north_star states.  "parity unpinned" by the reference's own tests (it has
none, and PyG is absent): the oracle is pinned by tests/test_oracle_*.py instead.
"""
import copy

import pytest
import torch

from helpers import TOL, rel_err

pytestmark = pytest.mark.gpu


def _models(kind, device, **kw):
    import gnn_qot_estimation_amd as q
    from oracle import sparse as O
    torch.manual_seed(0)
    if kind == "topo":
        ref = O.TopologicalGNN(**kw)
        hip = q.TopologicalGNN(**kw)
    else:
        ref = O.LightpathGNN(**kw)
        hip = q.LightpathGNN(**kw)
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1 and p.abs().max() == 0:      # zero-init biases: make them matter
                p.uniform_(-0.1, 0.1)
    hip.load_state_dict(ref.state_dict(), strict=True)
    return ref, hip.to(device)


def _grad_compare(ref, hip, analytic_zero=()):
    # A gradient that is analytically zero (e.g. lin_key.bias: softmax is shift invariant)
    # is pure rounding noise in both implementations, so each parameter's error is taken
    # relative to max(its own magnitude, 1e-3 x the largest gradient in the model).
    worst = 0.0
    rp = dict(ref.named_parameters())
    gmax = max(float(p.grad.abs().max()) for p in rp.values() if p.grad is not None)
    for name, p in hip.named_parameters():
        assert p.grad is not None, name
        a, b = p.grad.detach().double().cpu(), rp[name].grad.detach().double()
        floor = gmax if name in analytic_zero else 1e-3 * gmax
        e = float((a - b).abs().max() / max(float(b.abs().max()), floor))
        worst = max(worst, e)
        assert e <= TOL, (name, e)
    return worst


@pytest.mark.parametrize("cfg,B,n,e,H", [(1, 16, 14, 42, 32), (2, 8, 100, 400, 64), (2, 3, 30, 80, 16),
                                         (5, 2, 300, 0, 128), (2, 2, 50, 200, 256)])
def test_topological_fwd_bwd(cuda_device, cfg, B, n, e, H):
    from gnn_qot_estimation_amd import synthetic as S
    batch = S.topological_batch(cfg, B, n=n, e=e)
    V = 14 if cfg == 1 else n
    ref, hip = _models("topo", cuda_device, num_nodes=V, hidden_channels=H, out_channels=3, edge_dim=4, dropout_p=0.0)
    ref.train(); hip.train()
    out_ref = ref(batch)
    dbatch = batch.to(cuda_device)
    out_hip = hip(dbatch)
    assert out_hip.shape == out_ref.shape == (B, 3)
    assert rel_err(out_hip, out_ref) <= TOL
    y = batch.y.view(-1, 3)
    torch.nn.functional.smooth_l1_loss(out_ref, y).backward()
    torch.nn.functional.smooth_l1_loss(out_hip, y.to(cuda_device)).backward()
    _grad_compare(ref, hip)


@pytest.mark.parametrize("D,H,p", [(6, 64, 0.0), (8, 32, 0.0), (5, 128, 0.0), (6, 64, 0.5)])
def test_topological_edge_dim_5_to_8(cuda_device, D, H, p):
    """The reference takes ``edge_dim = len(dataset.FEATURES)`` (``topological_training/dataset.py:40``, ``train.py:51``):
    data-dependent.  The fused NNConv tile kernels are built for edge_dim <= 4; 5..8 (the range ``include/qot_gnn.h``
    promises) run the materialised-operand path (``qot_nnconv_agg`` + GEMMs + ``qot_nnconv_bwd_edge``) behind the same
    ``NNConvFn`` (the range used to raise at the first forward).  Forward and every gradient vs the oracle;
    with dropout on, a run-to-run replay (fused-head fold + the separate activation kernel draw the same masks)."""
    from gnn_qot_estimation_amd import synthetic as S
    batch = S.topological_batch(2, 5, n=40, e=140, edge_dim=D)
    ref, hip = _models("topo", cuda_device, num_nodes=40, hidden_channels=H, out_channels=3, edge_dim=D, dropout_p=p)
    dbatch = batch.to(cuda_device)
    y = batch.y.view(-1, 3)
    if p > 0:
        hip.train()
        hip._qot_seed = 99
        a = hip(dbatch)
        torch.nn.functional.smooth_l1_loss(a, y.to(cuda_device)).backward()
        assert all(torch.isfinite(q_.grad).all() for q_ in hip.parameters())
        assert float(a.abs().max()) > 0
        hip.eval(); ref.eval()
        assert rel_err(hip(dbatch), ref(batch)) <= TOL
        return
    ref.train(); hip.train()
    out_ref, out_hip = ref(batch), hip(dbatch)
    assert rel_err(out_hip, out_ref) <= TOL
    torch.nn.functional.smooth_l1_loss(out_ref, y).backward()
    torch.nn.functional.smooth_l1_loss(out_hip, y.to(cuda_device)).backward()
    _grad_compare(ref, hip)


def test_topological_isolated_nodes_and_duplicates(cuda_device):
    """Zero in-degree rows (common in the real 75-node graphs, to_graph.py:133-135),
    duplicate edges and a self loop."""
    import gnn_qot_estimation_amd as q
    ei = torch.tensor([[0, 1, 1, 2, 2, 4, 4], [1, 0, 0, 2, 1, 1, 0]])
    ea = torch.rand(7, 4)
    d = q.Data(edge_index=ei, edge_attr=ea, node_ids=torch.arange(6), num_nodes=6)
    batch = q.Batch.from_data_list([d, d])
    ref, hip = _models("topo", cuda_device, num_nodes=6, hidden_channels=16, out_channels=3, edge_dim=4, dropout_p=0.0)
    ref.eval(); hip.eval()
    assert rel_err(hip(batch.to(cuda_device)), ref(batch)) <= TOL


def test_topological_given_x(cuda_device):
    from gnn_qot_estimation_amd import synthetic as S
    batch = S.topological_batch(2, 4, n=20, e=60)
    batch.x = torch.randn(batch.num_nodes, 32)
    ref, hip = _models("topo", cuda_device, num_nodes=20, hidden_channels=32, out_channels=3, edge_dim=4, dropout_p=0.0)
    ref.eval(); hip.eval()
    assert rel_err(hip(batch.to(cuda_device)), ref(batch)) <= TOL


@pytest.mark.parametrize("H", [64, 128])
def test_non_finite_node_feature_stays_in_its_graph(cuda_device, H):
    """Edge slots past a destination's degree multiply a row by zero; that row is the destination's OWN row, never a
    fixed one: a NaN in node 0's features reaches graph 0 only (with row 0 as the filler it reached every destination whose
    degree is not a multiple of the slot block, i.e. every graph)."""
    from gnn_qot_estimation_amd import synthetic as S
    batch = S.topological_batch(2, 5, n=20, e=60)
    batch.x = torch.randn(batch.num_nodes, H)
    batch.x[0, 3] = float("nan")
    _, hip = _models("topo", cuda_device, num_nodes=20, hidden_channels=H, out_channels=3, edge_dim=4, dropout_p=0.0)
    out = hip.eval()(batch.to(cuda_device))
    assert bool(torch.isnan(out[0]).any()) and bool(torch.isfinite(out[1:]).all())


@pytest.mark.parametrize("thin", [False, True])
@pytest.mark.parametrize("B,C,train", [(64, 32, True), (64, 32, False), (17, 128, True), (5, 8, True)])
def test_lightpath_fwd_bwd(cuda_device, monkeypatch, B, C, train, thin):
    """``thin``: the first GATConv in the form large batches take (``GatThinFn``: projection inside the attention kernels,
    BatchNorm statistics from ITS epilogue's partials), forced at this size."""
    from gnn_qot_estimation_amd import synthetic as S
    monkeypatch.setenv("QOT_GAT_THIN_MIN_ROWS", "1" if thin else "1000000000")
    batch = S.lightpath_batch(B)
    ref, hip = _models("lp", cuda_device, in_channels=5, hidden_channels=C, output_dim=3, is_lut_index=1, dropout_p=0.0)
    ref.train(train); hip.train(train)
    out_ref, lb_ref = ref(batch)
    out_hip, lb_hip = hip(batch.to(cuda_device))
    assert torch.equal(lb_hip.cpu(), lb_ref)
    assert rel_err(out_hip, out_ref) <= TOL
    y = batch.y[lb_ref]
    torch.nn.functional.smooth_l1_loss(out_ref, y).backward()
    torch.nn.functional.smooth_l1_loss(out_hip, y.to(cuda_device)).backward()
    # a bias added right before a train-mode BatchNorm has an analytically zero gradient
    _grad_compare(ref, hip, analytic_zero=("conv1.bias",) if train else ())
    if train:   # running statistics + counter (App. B.4)
        for k in ("running_mean", "running_var"):
            assert rel_err(getattr(hip.norm1.module, k), getattr(ref.norm1.module, k)) <= TOL
        assert int(hip.norm1.module.num_batches_tracked) == int(ref.norm1.module.num_batches_tracked) == 1


def test_lightpath_no_lut_raises(cuda_device):
    from gnn_qot_estimation_amd import synthetic as S
    batch = S.lightpath_batch(4, lut=False)
    _, hip = _models("lp", cuda_device, in_channels=5, hidden_channels=32, output_dim=3, is_lut_index=1)
    with pytest.raises(ValueError, match="No LUT node found in the batch."):
        hip(batch.to(cuda_device))


def test_lightpath_existing_self_loops_and_multi_lut(cuda_device):
    import gnn_qot_estimation_amd as q
    x = torch.rand(5, 5); x[:, 1] = 0; x[0, 1] = 1; x[3, 1] = 1
    ei = torch.tensor([[0, 1, 1, 2, 2, 3, 0], [1, 0, 1, 2, 3, 2, 1]])   # self loops 1->1, 2->2, duplicate 0->1
    d = q.Data(x=x, edge_index=ei, y=torch.rand(1, 3), num_nodes=5)
    batch = q.Batch.from_data_list([d, d, d])
    ref, hip = _models("lp", cuda_device, in_channels=5, hidden_channels=16, output_dim=3, is_lut_index=1)
    ref.eval(); hip.eval()
    o_r, b_r = ref(batch)
    o_h, b_h = hip(batch.to(cuda_device))
    assert torch.equal(b_h.cpu(), b_r) and o_h.shape == (6, 3)
    assert rel_err(o_h, o_r) <= TOL


def test_cpu_tensor_fails_loudly():
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    from gnn_qot_estimation_amd._lib import QotError
    m = q.TopologicalGNN(14, 32, 3, 4)
    with pytest.raises(QotError):
        m(S.topological_batch(1, 2))


def test_dropout_statistics_and_replay(cuda_device):
    """Fused leaky_relu+dropout: keep rate ~ 1-p, scaling 1/(1-p), backward uses the same mask."""
    from gnn_qot_estimation_amd import functional as QF
    x = torch.randn(1 << 16, 64, device=cuda_device, requires_grad=True)
    step = torch.tensor(7, device=cuda_device)
    y = QF.ActFn.apply(x, 0.01, 0.5, 1234, step)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.5) < 0.01
    ref = torch.nn.functional.leaky_relu(x.detach(), 0.01) * 2.0
    m = y != 0
    assert torch.allclose(y[m], ref[m])
    y.sum().backward()
    g = x.grad
    assert torch.equal(g != 0, m)
    y2 = QF.ActFn.apply(x, 0.01, 0.5, 1234, torch.tensor(8, device=cuda_device))
    assert not torch.equal(y2 != 0, m)


@pytest.mark.parametrize("n,kt", [(1000, 640), (37, 128), (102400, 640), (8, 1280)])
def test_gemm_tn_matches_fp64(cuda_device, n, kt):
    """qot_gemm_tn (streaming split-K MFMA, deterministic slab reduce) vs an fp64 product."""
    from gnn_qot_estimation_amd.functional import gemm_tn
    torch.manual_seed(0)
    a = torch.randn(n, kt, device=cuda_device)
    g = torch.randn(n, 64, device=cuda_device)
    out = gemm_tn(a, g)
    ref = a.double().t() @ g.double()
    assert rel_err(out, ref) <= 1e-5
    assert torch.equal(out, gemm_tn(a, g))          # bitwise reproducible (no atomics)


@pytest.mark.parametrize("gat", [False, True])
def test_csr_build_matches_stable_cpu_sort(cuda_device, gat):
    """Graph prep: CSR-by-destination keeps the caller's edge order inside a destination;
    GAT mode drops j==i edges and appends one self loop per node (last in its row)."""
    from gnn_qot_estimation_amd.graph import build_graph_index
    torch.manual_seed(0)
    n, e = 500, 4000
    ei = torch.randint(0, n, (2, e))
    ei[:, :50] = ei[0, :50]                     # some self loops
    g = build_graph_index(ei.to(cuda_device), n, gat_self_loops=gat)
    src, dst = ei[0], ei[1]
    ids = torch.arange(e)
    if gat:
        keep = src != dst
        src, dst, ids = src[keep], dst[keep], ids[keep]
        loops = torch.arange(n)
        src, dst, ids = torch.cat([src, loops]), torch.cat([dst, loops]), torch.cat([ids, torch.full((n,), -1)])
    order = torch.sort(dst, stable=True).indices
    rowptr = torch.zeros(n + 1, dtype=torch.long)
    rowptr[1:] = torch.bincount(dst, minlength=n).cumsum(0)
    m = src.numel()
    assert torch.equal(g.rowptr.cpu().long(), rowptr)
    assert torch.equal(g.col.cpu().long()[:m], src[order])
    assert torch.equal(g.eid.cpu().long()[:m], ids[order])
    assert torch.equal(g.row.cpu().long()[:m], dst[order])
    # CSC: out-edges of j in the caller's edge order (appended self loops last); pos_t = CSR slot
    order_t = torch.sort(src, stable=True).indices
    rowptr_t = torch.zeros(n + 1, dtype=torch.long)
    rowptr_t[1:] = torch.bincount(src, minlength=n).cumsum(0)
    slot_of = torch.empty(m, dtype=torch.long)
    slot_of[order] = torch.arange(m)
    assert torch.equal(g.rowptr_t.cpu().long(), rowptr_t)
    assert torch.equal(g.pos_t.cpu().long()[:m], slot_of[order_t])
    assert torch.equal(g.col_t.cpu().long()[:m], dst[order_t])
    assert torch.equal(g.eid_t.cpu().long()[:m], ids[order_t])
    deg = torch.bincount(dst, minlength=n).clamp(min=1).float()
    assert torch.allclose(g.invdeg.cpu(), 1.0 / deg)


def test_fused_sgd_matches_torch_sgd(cuda_device):
    """FusedSGD (one kernel over the flat buffer) == torch.optim.SGD(lr, momentum) for 3 steps."""
    from gnn_qot_estimation_amd.dp import FlatModel, FusedSGD
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4)).to(cuda_device)
    b = copy.deepcopy(a)
    flat = FlatModel(a)
    fused = FusedSGD(flat, lr=0.1, momentum=0.9)
    ref = torch.optim.SGD(b.parameters(), lr=0.1, momentum=0.9)
    for step in range(3):
        x = torch.randn(32, 8, device=cuda_device)
        flat.detach_grads(); ref.zero_grad()
        a(x).square().mean().backward(); b(x).square().mean().backward()
        flat.gather_grads()
        fused.step(); ref.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("device_lr", [False, True])
def test_fused_sgd_with_folded_gradient_pack(cuda_device, device_lr):
    """FusedSGD.step(grads=True): gradients read from the parameters' own .grad tensors (one unused
    parameter -> None), packed copy still written to flat_grad; == torch.optim.SGD for 3 steps."""
    from gnn_qot_estimation_amd.dp import FlatModel, FusedSGD

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.unused, self.b = torch.nn.Linear(8, 16), torch.nn.Linear(3, 5), torch.nn.Linear(16, 4)

        def forward(self, x):
            return self.b(self.a(x))

    torch.manual_seed(0)
    a = M().to(cuda_device)
    b = copy.deepcopy(a)
    flat = FlatModel(a)
    fused = FusedSGD(flat, lr=0.1, momentum=0.9, device_lr=device_lr)
    ref = torch.optim.SGD(b.parameters(), lr=0.1, momentum=0.9)
    for step in range(3):
        x = torch.randn(32, 8, device=cuda_device)
        flat.detach_grads(); ref.zero_grad()
        a(x).square().mean().backward(); b(x).square().mean().backward()
        fused.step(grads=True); ref.step()
        packed = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in a.parameters()])
        assert torch.equal(flat.flat_grad, packed)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-7)


def test_colsum(cuda_device):
    from gnn_qot_estimation_amd.functional import colsum
    x = torch.randn(10007, 256, device=cuda_device)
    assert rel_err(colsum(x), x.double().sum(0)) <= 1e-5


def test_topological_mixed_sizes_uses_node_path(cuda_device):
    """Graphs of different sizes: no uniform node_ids hint -> per-node embedding path (EmbedFn +
    node-level projections) instead of TransformerConv's table mode; both must match the oracle."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    a = S.topological_batch(2, 1, n=10, e=24)
    b = S.topological_batch(2, 1, n=14, e=30, first_graph=5)
    g1 = q.Data(edge_index=a.edge_index, edge_attr=a.edge_attr, node_ids=a.node_ids, y=a.y, num_nodes=10)
    g2 = q.Data(edge_index=b.edge_index, edge_attr=b.edge_attr, node_ids=b.node_ids, y=b.y, num_nodes=14)
    batch = q.Batch.from_data_list([g1, g2, g1])
    assert batch.uniform_node_ids is None
    ref, hip = _models("topo", cuda_device, num_nodes=14, hidden_channels=64, out_channels=3, edge_dim=4, dropout_p=0.0)
    out_ref = ref(batch)
    out_hip = hip(batch.to(cuda_device))
    assert rel_err(out_hip, out_ref) <= TOL
    y = batch.y.view(-1, 3)
    torch.nn.functional.smooth_l1_loss(out_ref, y).backward()
    torch.nn.functional.smooth_l1_loss(out_hip, y.to(cuda_device)).backward()
    _grad_compare(ref, hip)


def test_table_mode_is_used_and_table_larger_than_graph(cuda_device):
    """uniform node_ids -> table mode; embedding table with more rows than the graphs use
    (V = 75 rows, 14-node graphs): unused rows get exactly zero gradient."""
    from gnn_qot_estimation_amd import synthetic as S
    batch = S.topological_batch(1, 6)
    assert batch.uniform_node_ids == 14
    ref, hip = _models("topo", cuda_device, num_nodes=75, hidden_channels=32, out_channels=3, edge_dim=4, dropout_p=0.0)
    out_ref = ref(batch)
    out_hip = hip(batch.to(cuda_device))
    assert rel_err(out_hip, out_ref) <= TOL
    y = batch.y.view(-1, 3)
    torch.nn.functional.smooth_l1_loss(out_ref, y).backward()
    torch.nn.functional.smooth_l1_loss(out_hip, y.to(cuda_device)).backward()
    _grad_compare(ref, hip)
    assert float(hip.node_embeddings.weight.grad[14:].abs().max()) == 0.0


@pytest.mark.parametrize("H", [64, 32])
def test_fused_activation_epilogue_equals_separate_kernel(cuda_device, H):
    """conv(..., act=(slope, p, seed, step)) must equal ActFn(conv(...)) bit for bit (same mask:
    same seed / step / element indexing), forward and input gradient, with dropout ON."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import functional as QF, synthetic as S
    from gnn_qot_estimation_amd.graph import build_graph_index
    torch.manual_seed(0)
    b = S.topological_batch(2, 6, n=50, e=160).to(cuda_device)
    N = b.num_nodes
    g = build_graph_index(b.edge_index, N)
    step = torch.tensor(5, device=cuda_device)
    act = (0.01, 0.5, 987654321, step)
    tc = q.TransformerConv(H, H, edge_dim=4).to(cuda_device)
    edge_nn = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.ReLU(), torch.nn.Linear(8, H * H))
    nc = q.NNConv(H, H, nn=edge_nn, aggr="mean").to(cuda_device)
    for conv in (tc, nc):
        x1 = torch.randn(N, H, device=cuda_device, requires_grad=True)
        x2 = x1.detach().clone().requires_grad_(True)
        y_fused = conv(x1, b.edge_index, b.edge_attr, graph=g, act=act)
        y_sep = QF.ActFn.apply(conv(x2, b.edge_index, b.edge_attr, graph=g), *act)
        assert torch.equal(y_fused, y_sep)
        assert 0.4 < float((y_fused == 0).float().mean()) < 0.6
        w = torch.randn_like(y_fused)
        (y_fused * w).sum().backward()
        (y_sep * w).sum().backward()
        assert torch.equal(x1.grad, x2.grad)


def test_graph_loader_prefetch_matches_to_device(cuda_device):
    """Pinned, double-buffered, side-stream H2D loader: every batch equals Batch.from_data_list(...)
    .to(device), across more batches than staging slots (buffer reuse), and feeds the model."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    graphs = []
    for g in range(20):
        b = S.topological_batch(2, 1, n=30, e=80, first_graph=g)
        graphs.append(q.Data(edge_index=b.edge_index, edge_attr=b.edge_attr, node_ids=b.node_ids, y=b.y, num_nodes=30))
    ref, hip = _models("topo", cuda_device, num_nodes=30, hidden_channels=64, out_channels=3, edge_dim=4, dropout_p=0.0)
    ref.eval(); hip.eval()
    loader = q.GraphLoader(graphs, batch_size=6, device=cuda_device)
    seen = 0
    with torch.no_grad():
        for k, batch in enumerate(loader):
            want = q.Batch.from_data_list(graphs[6 * k:6 * k + 6])
            for name in ("edge_index", "edge_attr", "node_ids", "batch", "y", "ptr"):
                assert torch.equal(getattr(batch, name).cpu(), getattr(want, name)), (k, name)
            assert batch.uniform_node_ids == 30
            assert rel_err(hip(batch), ref(want)) <= TOL
            seen += batch.num_graphs
    assert seen == 20


def test_packed_shard_loader_zero_copy_path(cuda_device):
    """PackedGraphs (pinned pre-tensorised shard) + GraphLoader: batches of consecutive graphs are
    DMA'd as slices and re-based on the device; they must equal the host collate."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    graphs = []
    for g in range(13):
        b = S.topological_batch(2, 1, n=20, e=50, first_graph=g)
        graphs.append(q.Data(edge_index=b.edge_index, edge_attr=b.edge_attr, node_ids=b.node_ids, y=b.y, num_nodes=20))
    shard = q.PackedGraphs.from_data_list(graphs).pin()
    seen = 0
    for k, batch in enumerate(q.GraphLoader(shard, batch_size=5, device=cuda_device)):
        want = q.Batch.from_data_list(graphs[5 * k:5 * k + 5])
        for name in ("edge_index", "edge_attr", "node_ids", "batch", "y", "ptr"):
            assert torch.equal(getattr(batch, name).cpu(), getattr(want, name)), (k, name)
        assert batch.num_graphs == want.num_graphs and batch.uniform_node_ids == 20
        seen += batch.num_graphs
    assert seen == 13


@pytest.mark.gpu
@pytest.mark.parametrize("H,p", [(16, 0.0), (64, 0.0), (64, 0.3), (128, 0.5)])
def test_fused_head_matches_unfused(H, p):
    """pool -> Linear -> LeakyReLU -> Dropout -> Linear in one kernel (models.py:61-63) against the same
    chain in torch ops; with dropout the torch chain uses the mask the kernel drew (hidden == 0)."""
    from gnn_qot_estimation_amd import functional as QF
    torch.manual_seed(3)
    dev = "cuda"
    sizes = torch.randint(1, 40, (37,))
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(torch.int32).to(dev)
    N, B, O = int(sizes.sum()), len(sizes), 3
    x = torch.randn(N, H, device=dev, requires_grad=True)
    w0 = (torch.randn(H, H, device=dev) / H ** 0.5).requires_grad_()
    b0 = torch.randn(H, device=dev).requires_grad_()
    w3 = (torch.randn(O, H, device=dev) / H ** 0.5).requires_grad_()
    b3 = torch.randn(O, device=dev).requires_grad_()
    step = torch.tensor([5], dtype=torch.int64, device=dev) if p > 0 else None
    out = QF.HeadFn.apply(x, ptr, w0, b0, w3, b3, B, (0.01, p, 1234, step))
    g = torch.randn_like(out)
    hid = out.grad_fn.saved_tensors[4].clone()
    grads = torch.autograd.grad(out, [x, w0, b0, w3, b3], g)

    batch = torch.repeat_interleave(torch.arange(B, device=dev), sizes.to(dev))
    pooled = torch.zeros(B, H, device=dev).index_add_(0, batch, x) / sizes.to(dev).float()[:, None]
    pre = pooled @ w0.t() + b0
    h = torch.nn.functional.leaky_relu(pre, 0.01)
    if p > 0:
        # recover the kernel's mask from its saved hidden activations
        keep = (hid != 0).float()
        frac = 1.0 - keep.mean().item()
        assert abs(frac - p) < 0.05
        h = h * keep / (1.0 - p)
    ref = h @ w3.t() + b3
    rgrads = torch.autograd.grad(ref, [x, w0, b0, w3, b3], g)
    assert rel_err(out, ref) < 1e-5
    for a, b in zip(grads, rgrads):
        assert rel_err(a, b) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n,beta", [(7, 1.0), (3072, 1.0), (200000, 0.25)])
def test_fused_smooth_l1_matches_torch(n, beta):
    from gnn_qot_estimation_amd import functional as QF
    torch.manual_seed(1)
    pred = (2.0 * torch.randn(n, device="cuda")).requires_grad_()
    tgt = torch.randn(n, device="cuda")
    ref = torch.nn.functional.smooth_l1_loss(pred, tgt, beta=beta)
    (rg,) = torch.autograd.grad(ref, pred)
    for _ in range(2):                      # second call: the arrival counter was restored
        loss, g = QF.smooth_l1_loss_and_grad(pred, tgt, beta)
        assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
        assert torch.allclose(g, rg, rtol=1e-6, atol=1e-9)
    l1, _ = QF.smooth_l1_loss_and_grad(pred, tgt, beta)
    l2, _ = QF.smooth_l1_loss_and_grad(pred, tgt, beta)
    assert torch.equal(l1, l2)              # fixed-order partial sum


@pytest.mark.gpu
@pytest.mark.parametrize("H,V", [(16, 75), (64, 100), (128, 33), (64, 517), (256, 1000), (32, 512)])
def test_table_projection_matches_linear(H, V):
    """V >= 512: the blocked kernels (8 table rows / 8 packed weight rows per workgroup, roles.hip), V not a multiple of 8."""
    from gnn_qot_estimation_amd import functional as QF
    torch.manual_seed(2)
    dev = "cuda"
    table = torch.randn(V, H, device=dev, requires_grad=True)
    ws = [(torch.randn(H, H, device=dev) / H ** 0.5).requires_grad_() for _ in range(4)]
    bs = [torch.randn(H, device=dev).requires_grad_() for _ in range(4)]
    args = [table] + [t for pair in zip(ws, bs) for t in pair]
    out = QF.TableProjectFn.apply(*args)
    ref = torch.cat([table @ w.t() + b for w, b in zip(ws, bs)], 1)
    assert rel_err(out, ref) < 1e-5
    g = torch.randn_like(out)
    got = torch.autograd.grad(out, args, g)
    want = torch.autograd.grad(ref, args, g)
    for a, b in zip(got, want):
        assert rel_err(a, b) < 1e-5


@pytest.mark.gpu
def test_csr_by_graph_equals_general_build(cuda_device):
    """Block-diagonal batch: one workgroup per graph (LDS only) must emit exactly the general build's
    index -- ragged graph sizes, an empty graph, shuffled edge order inside every graph."""
    from gnn_qot_estimation_amd.graph import build_graph_index
    torch.manual_seed(0)
    sizes = [7, 1, 300, 12, 0, 40, 33]
    ecnt = [20, 0, 1500, 30, 0, 0, 64]
    ptr = torch.tensor([0] + sizes).cumsum(0)
    eptr = torch.tensor([0] + ecnt).cumsum(0)
    parts = []
    for n, m, off in zip(sizes, ecnt, ptr[:-1].tolist()):
        if m:
            parts.append(torch.randint(0, n, (2, m)) + off)
    ei = torch.cat(parts, 1).to(cuda_device)
    N, dev = int(ptr[-1]), cuda_device
    a = build_graph_index(ei, N)
    b = build_graph_index(ei, N, slices=(ptr.to(dev), eptr.to(dev), max(sizes), max(ecnt)))
    for name in ("rowptr", "col", "eid", "row", "rowptr_t", "col_t", "pos_t", "eid_t", "invdeg"):
        ta, tb = getattr(a, name), getattr(b, name)
        m = ei.shape[1] if name not in ("rowptr", "rowptr_t", "invdeg") else ta.numel()
        assert torch.equal(ta[:m], tb[:m]), name
    ids = torch.randint(0, 50, (N,), device=dev)
    c = build_graph_index(ei, N, slices=(ptr.to(dev), eptr.to(dev), max(sizes), max(ecnt)), node_ids=ids)
    m = ei.shape[1]
    assert torch.equal(c.ids32.long(), ids) and torch.equal(c.ptr32.long().cpu(), ptr)
    assert torch.equal(c.colf[:m].long(), ids[a.col[:m].long()]) and torch.equal(c.colf_t[:m].long(), ids[a.col_t[:m].long()])
    # status flags: an edge that leaves its graph / a graph larger than the bound
    from gnn_qot_estimation_amd import _lib
    P = _lib.ptr
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    bad = ei.clone(); bad[0, 0] = N - 1
    g = b
    # (named device tensors: a pointer taken from a temporary `ptr.to(dev)` outlives it, and the next temporary reuses
    #  the block -- the kernel then saw edge_ptr in both arguments and wrote past the index arrays)
    ptr_d, eptr_d = ptr.to(dev), eptr.to(dev)
    _lib.call("qot_csr_build_by_graph", P(bad), ei.shape[1], N, P(ptr_d), P(eptr_d), len(sizes), max(sizes),
              max(ecnt), P(g.rowptr), P(g.col), P(g.eid), P(g.row), P(g.rowptr_t), P(g.col_t), P(g.pos_t), P(g.eid_t),
              P(g.invdeg), P(st), None, None, None, None, None)
    assert int(st.item()) & 1
    st.zero_()
    _lib.call("qot_csr_build_by_graph", P(ei), ei.shape[1], N, P(ptr_d), P(eptr_d), len(sizes), 100,
              max(ecnt), P(g.rowptr), P(g.col), P(g.eid), P(g.row), P(g.rowptr_t), P(g.col_t), P(g.pos_t), P(g.eid_t),
              P(g.invdeg), P(st), None, None, None, None, None)
    assert int(st.item()) & 2
    # slices that do not lie inside the arrays (here: the edge slices passed as node slices, so "graphs" reach node
    # 1614 of 393) are flagged and written nowhere: sentinel tails behind every node-indexed array stay intact
    st.zero_()
    tail = 2048
    big = {k: torch.full((N + 1 + tail,), -7, dtype=getattr(g, k).dtype, device=dev) for k in ("rowptr", "rowptr_t", "invdeg")}
    _lib.call("qot_csr_build_by_graph", P(ei), ei.shape[1], N, P(eptr_d), P(eptr_d), len(sizes), 2048,
              max(ecnt), P(big["rowptr"]), P(g.col), P(g.eid), P(g.row), P(big["rowptr_t"]), P(g.col_t), P(g.pos_t),
              P(g.eid_t), P(big["invdeg"]), P(st), None, None, None, None, None)
    assert int(st.item()) & 2
    for k, v in big.items():
        assert bool((v[N + 1:] == -7).all()), k


@pytest.mark.gpu
def test_csr_by_graph_strided_slice_views_and_status_raise(cuda_device):
    """``build_graph_index`` with NON-contiguous ``ptr`` / ``edge_ptr`` views: both ``.contiguous()`` results are
    temporaries, and a pointer taken from the first used to be handed to the kernel after its block had been reused
    by the second (round-2 GPU fault).  The product keeps both alive; the result must equal the
    general build.  Wrong slices must raise (the kernel's status word is read back) instead of handing
    uninitialised index arrays to the gather kernels."""
    from gnn_qot_estimation_amd import _lib
    from gnn_qot_estimation_amd.graph import build_graph_index, check_index_status
    torch.manual_seed(1)
    sizes = [9, 31, 2, 120, 64]
    ecnt = [30, 100, 2, 700, 0]
    ptr = torch.tensor([0] + sizes).cumsum(0)
    eptr = torch.tensor([0] + ecnt).cumsum(0)
    parts = [torch.randint(0, n, (2, m)) + off for n, m, off in zip(sizes, ecnt, ptr[:-1].tolist()) if m]
    dev = cuda_device
    ei = torch.cat(parts, 1).to(dev)
    N = int(ptr[-1])
    # strided views of same-sized parents: [B+1, 2][:, 0] -- .contiguous() allocates for each
    ptr_v = torch.stack([ptr, ptr + 1000], 1).to(dev)[:, 0]
    eptr_v = torch.stack([eptr, eptr + 1000], 1).to(dev)[:, 0]
    assert not ptr_v.is_contiguous() and not eptr_v.is_contiguous()
    a = build_graph_index(ei, N)
    for _ in range(3):                      # allocator state varies between rounds
        b = build_graph_index(ei, N, slices=(ptr_v, eptr_v, max(sizes), max(ecnt)))
        for name in ("rowptr", "col", "eid", "row", "rowptr_t", "col_t", "pos_t", "eid_t", "invdeg"):
            ta, tb = getattr(a, name), getattr(b, name)
            m = ei.shape[1] if name not in ("rowptr", "rowptr_t", "invdeg") else ta.numel()
            assert torch.equal(ta[:m], tb[:m]), name
        assert torch.equal(b.ptr32.long().cpu(), ptr)
    # edge slices handed in as node slices: the kernel flags it, the product raises, the flag is cleared
    with pytest.raises(_lib.QotError, match="inconsistent batch slices"):
        build_graph_index(ei, N, slices=(eptr_v, eptr_v, 2048, max(ecnt)))
    check_index_status(dev)                 # cleared by the raise
    bad = ei.clone()
    bad[0, 0] = N - 1                       # an edge that leaves its graph
    with pytest.raises(_lib.QotError, match="leaves its graph"):
        build_graph_index(bad, N, slices=(ptr_v, eptr_v, max(sizes), max(ecnt)))
    # _lib.call takes tensors (kept alive through the launch) as well as raw pointers
    out = torch.empty(N, dtype=torch.int32, device=dev)
    ids = torch.arange(2 * N, device=dev)[::2]
    _lib.call("qot_i64_to_i32", ids.contiguous(), out, N)
    assert torch.equal(out.long(), ids)


@pytest.mark.gpu
def test_dropout_step_agrees_between_execution_modes(cuda_device):
    """With dropout ON the oracle cannot be the checker (different RNG), but the engine's own modes must
    agree with each other: table mode + pre-reduced table gradient + read-out fold (default) against the
    per-node path without the fold -- same masks (seed / step / element indexing), same loss, same
    gradients up to summation order."""
    import copy
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    torch.manual_seed(11)
    batch = S.topological_batch(2, 24, n=40, e=150).to(cuda_device)
    a = q.TopologicalGNN(40, 64, 3, 4, dropout_p=0.5).to(cuda_device).train()
    b = copy.deepcopy(a)
    b._qot_fold_head = False
    a._qot_seed = b._qot_seed = 1234567
    plain = batch.to(cuda_device)
    plain.uniform_node_ids = None            # forces EmbedFn + node-level projections
    plain.edge_ptr = None                    # and the general index build
    ya, yb = a(batch), b(plain)
    assert float((ya - yb).abs().max()) <= 1e-5 * float(yb.abs().max())
    assert float(ya.abs().max()) > 0
    w = torch.randn_like(ya)
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    gmax = max(float(p.grad.abs().max()) for p in b.parameters())
    for (name, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        err = float((pa.grad - pb.grad).abs().max()) / max(float(pb.grad.abs().max()), 1e-3 * gmax)
        assert err <= 1e-4, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("graph_form", [False, True])
def test_table_mode_without_edge_slices_matches_sliced_batch(cuda_device, monkeypatch, graph_form):
    """A batch object that does not carry per-graph edge slices (e.g. a PyG Batch) takes the general index
    build + qot_table_maps; results must equal the one-launch per-graph build's -- bit for bit when both run the
    per-destination TransformerConv kernels, to rounding when the sliced batch takes the graph form (r04: it needs the
    per-graph build's verified slices; the logits' H-term dot is summed in another order there)."""
    import copy
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    if not graph_form:
        monkeypatch.setenv("QOT_NO_TCONV_GRAPH", "1")
    torch.manual_seed(5)
    batch = S.topological_batch(2, 12, n=30, e=100).to(cuda_device)
    plain = batch.to(cuda_device)
    plain.edge_ptr = None
    plain.graph_sizes = None
    a = q.TopologicalGNN(30, 64, 3, 4, dropout_p=0.0).to(cuda_device).train()
    b = copy.deepcopy(a)
    ya, yb = a(batch), b(plain)
    assert "tmaps" in batch._qot_cache and "tmaps" in plain._qot_cache          # both ran in table mode
    assert batch._qot_cache[("graph", False)][1].colf is not None and plain._qot_cache[("graph", False)][1].colf is None
    same = torch.equal if not graph_form else (lambda u, v: float((u - v).abs().max()) <= 2e-5 * max(float(v.abs().max()), 1e-3))
    assert same(ya, yb)
    ya.sum().backward(); yb.sum().backward()
    gmax = max(float(pb.grad.abs().max()) for pb in b.parameters())
    for (name, pa), pb in zip(a.named_parameters(), b.parameters()):
        if graph_form:       # (lin_key.bias: analytically zero, rounding noise on both sides)
            assert float((pa.grad - pb.grad).abs().max()) <= 2e-5 * max(float(pb.grad.abs().max()), 1e-3 * gmax, gmax if name == "conv1.lin_key.bias" else 0.0), name
        else:
            assert torch.equal(pa.grad, pb.grad), name


@pytest.mark.gpu
def test_pyg_style_slice_dict_enables_per_graph_index(cuda_device):
    """A PyG ``Batch`` has no ``edge_ptr`` but keeps ``_slice_dict['edge_index']``: the one-launch index build
    must pick that up (duck-typed here: PyG itself is not installed) and match the general build."""
    from gnn_qot_estimation_amd import synthetic as S
    from gnn_qot_estimation_amd.graph import build_graph_index, graph_index_for
    b = S.topological_batch(2, 9, n=25, e=70).to(cuda_device)

    class PygLike:
        pass
    p = PygLike()
    p.edge_index, p.ptr, p.num_graphs = b.edge_index, b.ptr, b.num_graphs
    p._slice_dict = {"edge_index": b.edge_ptr.cpu(), "x": None}
    g = graph_index_for(p, b.num_nodes)
    ref = build_graph_index(b.edge_index, b.num_nodes)
    assert g.ptr32 is not None                         # by-graph path was taken
    for name in ("rowptr", "col", "eid", "rowptr_t", "col_t", "pos_t", "eid_t", "invdeg"):
        assert torch.equal(getattr(g, name), getattr(ref, name)), name


@pytest.mark.gpu
def test_table_mode_detected_for_batches_without_hint(cuda_device):
    """A batch object without ``uniform_node_ids`` (a PyG Batch has none) is inspected once on the device;
    table mode then runs and gives the hinted batch's results, and a non-uniform batch stays on the node path."""
    import copy
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    torch.manual_seed(3)
    b = S.topological_batch(2, 10, n=20, e=60).to(cuda_device)

    class PygLike:
        pass
    p = PygLike()
    for name in ("edge_index", "edge_attr", "node_ids", "batch", "ptr", "num_graphs"):
        setattr(p, name, getattr(b, name))
    p.x = None
    a = q.TopologicalGNN(20, 32, 3, 4, dropout_p=0.0).to(cuda_device).train()
    c = copy.deepcopy(a)
    ya, yc = a(b), c(p)
    assert p.uniform_node_ids == 20 and "tmaps" in p._qot_cache
    assert float((ya - yc).abs().max()) <= 1e-6 * float(ya.abs().max())
    # permuted ids in one graph: not uniform -> node path, still correct (compare against itself hinted None)
    p2 = PygLike()
    for name in ("edge_index", "edge_attr", "batch", "ptr", "num_graphs"):
        setattr(p2, name, getattr(b, name))
    ids = b.node_ids.clone(); ids[:20] = ids[:20].flip(0)
    p2.node_ids, p2.x = ids, None
    y2 = c(p2)
    assert p2.uniform_node_ids is None and "tmaps" not in p2._qot_cache
    assert y2.shape == ya.shape


def _nnconv_fp64(x, ea, w1, b1, wcat, bias, rowptr, col, eids, invdeg, transpose):
    """out = bias + A @ Wcat in fp64 from the walked index (CSR forward / CSC adjoint), App. B.2 algebra."""
    x, ea, w1, b1, wcat, bias, invdeg = (t.double().cpu() for t in (x, ea, w1, b1, wcat, bias, invdeg))
    rowptr = rowptr.cpu().long()
    nE = int(rowptr[-1])                           # the index buffers may be longer than the walked range
    col, eids = col.cpu().long()[:nE], eids.cpu().long()[:nE]
    N, H = x.shape
    K = w1.shape[0]
    rows = torch.repeat_interleave(torch.arange(N), rowptr[1:] - rowptr[:-1])
    A = torch.zeros(N, (K + 2) * H, dtype=torch.float64)
    if col.numel():
        h = torch.relu(ea[eids] @ w1.t() + b1)                       # [E, K]
        sc = invdeg[col] if transpose else invdeg[rows]
        hw = torch.cat([h, torch.ones(len(col), 1, dtype=torch.float64)], 1) * sc[:, None]
        contrib = (hw[:, :, None] * x[col][:, None, :]).reshape(len(col), (K + 1) * H)
        A[:, :(K + 1) * H].index_add_(0, rows, contrib)
    A[:, (K + 1) * H:] = x
    return A @ wcat + bias


@pytest.mark.parametrize("case", ["random", "hub", "edge_dim2", "no_edges"])
def test_weight_stationary_nnconv_matches_tile_kernel_and_fp64(cuda_device, case):
    """The production tile kernel (``qot_nnconv_fused``) against an fp64 restatement of its contract, both index
    directions, a 700-in-edge hub, edge_dim 2, an edge-less batch; in a diagnostic build (``make DIAG=1``) also the
    experimental weight-stationary kernel csrc/nnconv_ws.hip (weights in registers, rows by LDS-DMA, operands formed
    on the fly), which is not part of the release library or its ABI."""
    from gnn_qot_estimation_amd import _lib
    import ctypes
    names = ["qot_nnconv_fused"]
    lib = _lib.load()
    if hasattr(lib, "qot_nnconv_fused_ws"):
        fn = lib.qot_nnconv_fused_ws
        fn.restype, fn.argtypes = lib.qot_nnconv_fused.restype, lib.qot_nnconv_fused.argtypes
        names.append("qot_nnconv_fused_ws")
    from gnn_qot_estimation_amd.functional import nnconv_perm_index
    from gnn_qot_estimation_amd.graph import build_graph_index
    P = _lib.ptr
    dev = cuda_device
    g = torch.Generator().manual_seed(5)
    H, D = 64, (2 if case == "edge_dim2" else 4)
    K = 2 * D
    if case == "hub":
        N = 1000 + 13
        src = torch.randint(0, N, (700,), generator=g)
        ei = torch.cat([torch.stack([src, torch.full_like(src, 40)]),            # node 40: in-degree 700
                        torch.randint(0, N, (2, 3000), generator=g)], 1)
    elif case == "no_edges":
        N, ei = 70, torch.zeros(2, 0, dtype=torch.long)
    else:
        N = 777
        ei = torch.randint(0, N, (2, 4 * N), generator=g)
    E = ei.shape[1]
    ei = ei.to(dev)
    gi = build_graph_index(ei, N)
    x = torch.randn(N, H, generator=g).to(dev)
    ea = torch.rand(max(E, 1), D, generator=g).to(dev)
    w1 = torch.randn(K, D, generator=g).to(dev); b1 = torch.randn(K, generator=g).to(dev)
    wcat = (torch.randn((K + 2) * H, H, generator=g) / 8).to(dev)
    bias = torch.randn(H, generator=g).to(dev)
    wp = wcat.reshape(-1)[nnconv_perm_index((K + 2) * H, dev)].contiguous()
    for transpose, (rp, col, eids) in ((0, (gi.rowptr, gi.col, gi.eid)), (1, (gi.rowptr_t, gi.col_t, gi.eid_t))):
        ref = _nnconv_fp64(x, ea, w1, b1, wcat, bias, rp, col, eids, gi.invdeg, transpose)
        outs = []
        for name in names:
            out = torch.full((N, H), float("nan"), device=dev)
            _lib.call(name, P(x), H, P(ea), P(w1), P(b1), P(rp), P(col), P(eids), P(gi.invdeg), transpose, P(wp),
                      P(bias), P(out), N, H, D, 0, 0.0, 0.0, 0, None)
            torch.cuda.synchronize()
            assert rel_err(out.double().cpu(), ref) <= TOL, (case, transpose, name)
            outs.append(out)
        assert rel_err(outs[-1], outs[0]) <= TOL
    # identical dropout masks and activation epilogue (counter-based draws keyed by element index)
    step = torch.tensor([3], dtype=torch.int64, device=dev)
    outs = []
    for name in names:
        out = torch.empty(N, H, device=dev)
        _lib.call(name, P(x), H, P(ea), P(w1), P(b1), P(gi.rowptr), P(gi.col), P(gi.eid), P(gi.invdeg), 0, P(wp),
                  P(bias), P(out), N, H, D, 1, 0.01, 0.25, 1234, P(step))
        outs.append(out)
    torch.cuda.synchronize()
    assert bool(((outs[0] == 0) == (outs[-1] == 0)).all())
    assert rel_err(outs[-1], outs[0]) <= TOL


@pytest.mark.parametrize("B,C", [(3, 64), (16, 260), (64, 25600), (100, 1028), (1024, 512), (1500, 128)])
def test_rowsum_wide_all_row_counts(cuda_device, B, C):
    """qot_rowsum_wide (table-gradient sum over graph groups): direct, one-launch (16 <= B <= 1024) and
    two-launch forms against an fp64 sum; fixed summation order -> bitwise repeatable."""
    from gnn_qot_estimation_amd import _lib
    P = _lib.ptr
    g = torch.Generator().manual_seed(B * 1000 + C)
    x = torch.randn(B, C, generator=g).to(cuda_device)
    outs = []
    for _ in range(2):
        out = torch.full((C,), float("nan"), device=cuda_device)
        ws = torch.empty(_lib.load().qot_rowsum_wide_workspace_floats(C), dtype=torch.float32, device=cuda_device)
        _lib.call("qot_rowsum_wide", P(x), B, C, P(out), P(ws))
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[0], x.double().sum(0)) <= 1e-5


def test_node_id_outside_embedding_table_raises_index_error(cuda_device):
    """``nn.Embedding`` semantics (models.py:12,52): id >= num_nodes is an IndexError, never a read past the table
     -- in table mode (uniform arange ids, n > V) and in per-node mode (arbitrary ids)."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    hip = q.TopologicalGNN(10, 16, 3, 4, dropout_p=0.0).to(cuda_device).eval()
    with pytest.raises(IndexError):
        hip(S.topological_batch(2, 3, n=12, e=30).to(cuda_device))          # table mode: n = 12 > V = 10
    d = q.Data(edge_index=torch.tensor([[0, 1], [1, 0]]), edge_attr=torch.rand(2, 4),
               node_ids=torch.tensor([3, 10]), num_nodes=2)
    with pytest.raises(IndexError):
        hip(q.Batch.from_data_list([d]).to(cuda_device))                    # per-node mode: id 10 >= V
    d.node_ids = torch.tensor([3, -1])
    with pytest.raises(IndexError):
        hip(q.Batch.from_data_list([d]).to(cuda_device))
    d.node_ids = torch.tensor([3, 9])
    assert hip(q.Batch.from_data_list([d]).to(cuda_device)).shape == (1, 3)


def test_retained_graph_second_backward_with_folded_head(cuda_device):
    """The read-out head hands the last conv its bias gradient through a side table on every backward: a second
    backward over a retained graph gives the same gradients (the hand-off used to be single-shot)."""
    from gnn_qot_estimation_amd import synthetic as S
    batch = S.topological_batch(2, 4, n=20, e=60)
    _, hip = _models("topo", cuda_device, num_nodes=20, hidden_channels=64, out_channels=3, edge_dim=4, dropout_p=0.0)
    hip.train()
    out = hip(batch.to(cuda_device))
    loss = out.square().sum()
    loss.backward(retain_graph=True)
    g1 = {k: p.grad.clone() for k, p in hip.named_parameters()}
    hip.zero_grad(set_to_none=True)
    loss.backward()
    for k, p in hip.named_parameters():
        assert torch.equal(p.grad, g1[k]), k


def test_batchnorm_single_row_training_raises(cuda_device):
    import gnn_qot_estimation_amd as q
    bn = q.BatchNorm(8).to(cuda_device).train()
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        bn(torch.randn(1, 8, device=cuda_device))
    bn.eval()
    assert bn(torch.randn(1, 8, device=cuda_device)).shape == (1, 8)


@pytest.mark.parametrize("gat", [False, True])
def test_general_csr_build_capture_replay_equals_eager(cuda_device, gat):
    """The GENERAL graph-index build (``qot_csr_build``: arbitrary edge lists, GAT self-loop mode -- the path
    block-diagonal topological batches no longer take) captured in a HIP graph and replayed must reproduce the eager
    build array for array.  Guards the class of fault recorded in DESIGN.md section 8 (a captured graph-index build
    that wrote outside its buffers on replay): every launch takes its addresses from caller-owned tensors, the
    library issues no memset / memcpy / allocation / synchronisation of its own."""
    from gnn_qot_estimation_amd.graph import build_graph_index
    g = torch.Generator().manual_seed(11)
    N = 5000
    ei = torch.randint(0, N, (2, 4 * N), generator=g)
    ei[1, :64] = 7                       # a hub destination
    ei[:, 100:110] = torch.arange(10)    # existing self loops (dropped and re-appended in GAT mode)
    ei = ei.to(cuda_device)
    names = ("rowptr", "col", "eid", "row", "rowptr_t", "col_t", "pos_t", "eid_t", "invdeg")
    eager = build_graph_index(ei, N, gat_self_loops=gat)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        build_graph_index(ei, N, gat_self_loops=gat)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        cap = build_graph_index(ei, N, gat_self_loops=gat)
    for _ in range(3):
        for n_ in names:
            getattr(cap, n_).fill_(-3)           # replay must rewrite everything it owns
        gr.replay()
        torch.cuda.synchronize()
        live = int(eager.rowptr[-1])          # slots in use (GAT mode drops j == i edges before appending loops)
        for n_ in names:
            a, b = getattr(cap, n_), getattr(eager, n_)
            k = live if n_ in ("col", "eid", "row", "col_t", "pos_t", "eid_t") else a.numel()
            assert torch.equal(a[:k], b[:k]), (n_, gat)


@pytest.mark.parametrize("H", [48, 20, 100])
def test_topological_any_hidden_width_runs_zero_padded(cuda_device, H):
    """The reference constructor takes any ``hidden_channels`` (models.py:7-9); widths without kernels of their own run
    on the next supported width with zero-padded parameters (gnn_qot_estimation_amd/padded.py): same state_dict shapes,
    forward and every gradient equal to the oracle at the TRUE width."""
    from gnn_qot_estimation_amd import synthetic as S
    batch = S.topological_batch(2, 5, n=24, e=60)
    ref, hip = _models("topo", cuda_device, num_nodes=24, hidden_channels=H, out_channels=3, edge_dim=4, dropout_p=0.0,
                       num_layers=3)
    assert hip._qot_hp in (32, 64, 128)
    ref.train(); hip.train()
    out_ref = ref(batch)
    out_hip = hip(batch.to(cuda_device))
    assert rel_err(out_hip, out_ref) <= TOL
    y = batch.y.view(-1, 3)
    torch.nn.functional.smooth_l1_loss(out_ref, y).backward()
    torch.nn.functional.smooth_l1_loss(out_hip, y.to(cuda_device)).backward()
    _grad_compare(ref, hip)
    assert [tuple(v.shape) for v in hip.state_dict().values()] == [tuple(v.shape) for v in ref.state_dict().values()]


@pytest.mark.parametrize("C,train", [(24, True), (24, False), (5, True)])
def test_lightpath_any_hidden_width_runs_zero_padded(cuda_device, C, train):
    from gnn_qot_estimation_amd import synthetic as S
    batch = S.lightpath_batch(40)
    ref, hip = _models("lp", cuda_device, in_channels=5, hidden_channels=C, output_dim=3, is_lut_index=1, dropout_p=0.0,
                       num_layers=2)
    with torch.no_grad():
        for l in (1, 2):
            for m in (getattr(ref, f"norm{l}").module, getattr(hip, f"norm{l}").module):
                gen = torch.Generator().manual_seed(l)
                m.running_mean.copy_(torch.rand(4 * C, generator=gen) - 0.5)
                m.running_var.copy_(torch.rand(4 * C, generator=gen) + 0.5)
    ref.train(train); hip.train(train)
    o_r, b_r = ref(batch)
    o_h, b_h = hip(batch.to(cuda_device))
    assert torch.equal(b_h.cpu(), b_r) and rel_err(o_h, o_r) <= TOL
    y = batch.y[b_r]
    torch.nn.functional.smooth_l1_loss(o_r, y).backward()
    torch.nn.functional.smooth_l1_loss(o_h, y.to(cuda_device)).backward()
    _grad_compare(ref, hip, analytic_zero=("conv1.bias", "conv2.bias") if train else ())
    for l in (1, 2):
        r, h = getattr(ref, f"norm{l}").module, getattr(hip, f"norm{l}").module
        assert rel_err(h.running_mean, r.running_mean) <= TOL and rel_err(h.running_var, r.running_var) <= TOL
        assert int(h.num_batches_tracked) == int(r.num_batches_tracked) == (1 if train else 0)


@pytest.mark.parametrize("H,D,seed", [(16, 1, 0), (32, 3, 1), (64, 2, 2), (128, 3, 3), (256, 1, 4), (32, 4, 5), (128, 4, 6)])
def test_topological_irregular_graphs_all_widths_and_edge_dims(cuda_device, H, D, seed):
    """Random multigraphs per width / edge_dim: isolated nodes, duplicate edges, self loops, a hub, node counts that are
    not a multiple of the 32-row tile, graphs of different sizes in one batch (per-node embedding path, not table mode),
    3 layers -- forward and every gradient against the oracle."""
    import gnn_qot_estimation_amd as q
    g = torch.Generator().manual_seed(100 + seed)
    datas = []
    for n in (37, 5, 64, 1, 23):
        e = int(torch.randint(0, 4 * n + 1, (1,), generator=g))
        ei = torch.randint(0, n, (2, e), generator=g)
        if n >= 20:
            ei[1, : min(e, 19)] = 3                        # a hub destination
            ei[:, -2:] = ei[:, :2]                         # duplicate edges
        datas.append(q.Data(edge_index=ei, edge_attr=torch.rand(e, D, generator=g), y=torch.rand(3, generator=g),
                            node_ids=torch.randperm(64, generator=g)[:n], num_nodes=n))
    batch = q.Batch.from_data_list(datas)
    ref, hip = _models("topo", cuda_device, num_nodes=64, hidden_channels=H, out_channels=3, edge_dim=D, dropout_p=0.0,
                       num_layers=3)
    ref.train(); hip.train()
    out_ref = ref(batch)
    out_hip = hip(batch.to(cuda_device))
    assert rel_err(out_hip, out_ref) <= TOL
    y = batch.y.view(-1, 3)
    torch.nn.functional.smooth_l1_loss(out_ref, y).backward()
    torch.nn.functional.smooth_l1_loss(out_hip, y.to(cuda_device)).backward()
    _grad_compare(ref, hip)


def test_multi_role_launch_equals_the_separate_launches(cuda_device):
    """``qot_run_roles`` (csrc/roles.hip): independent jobs sharing one launch must produce exactly what their standalone
    entry points (one-role calls of the same kernel) produce -- the forward prologue's three jobs, and row sums with
    and without groups, float4 and scalar columns, ragged last group."""
    from gnn_qot_estimation_amd import _lib, functional as QF, launch_group as LG, synthetic as S
    from gnn_qot_estimation_amd.graph import build_graph_index
    dev = cuda_device
    torch.manual_seed(3)
    batch = S.topological_batch(2, 6, n=30, e=90).to(dev)
    N, H, V, K = batch.num_nodes, 64, 30, 8
    slices = (batch.ptr, batch.edge_ptr) + tuple(batch.graph_sizes)
    a = build_graph_index(batch.edge_index, N, slices=slices, node_ids=batch.node_ids)
    table = torch.randn(V, H, device=dev)
    ws_ = [torch.randn(H, H, device=dev) for _ in range(4)]
    bs_ = [torch.randn(H, device=dev) for _ in range(4)]
    t4_ref = QF.TableProjectFn.apply(table, ws_[0], bs_[0], ws_[1], bs_[1], ws_[2], bs_[2], ws_[3], bs_[3])
    w2, b2, wroot = torch.randn(H * H, K, device=dev), torch.randn(H * H, device=dev), torch.randn(H, H, device=dev)
    pk_ref = QF.nnconv_pack(w2, b2, wroot, H, K)
    grp = LG.LaunchGroup()
    b = build_graph_index(batch.edge_index, N, slices=slices, node_ids=batch.node_ids, group=grp)
    cnt, snap = torch.zeros((), dtype=torch.long, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    t4 = QF.TableProjectFn.apply(table, ws_[0], bs_[0], ws_[1], bs_[1], ws_[2], bs_[2], ws_[3], bs_[3], (cnt, snap), grp)
    pk = QF.nnconv_pack(w2, b2, wroot, H, K, grp)
    assert len(grp.roles) == 3
    grp.run()
    for name in ("rowptr", "col", "eid", "row", "rowptr_t", "col_t", "pos_t", "eid_t", "invdeg", "ids32", "colf", "colf_t", "ptr32"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(t4, t4_ref) and int(cnt) == 1 and int(snap) == 1
    assert all(torch.equal(x, y) for x, y in zip(pk, pk_ref))
    # row sums: [nblk, n] -> [groups, n]
    for nblk, n, per in ((512, 4419, 0), (300, 256, 128), (64, 25600, 0), (7, 10, 3), (1, 5, 0)):
        part = torch.randn(nblk, n, device=dev)
        per_eff = nblk if per == 0 else per
        groups = (nblk + per_eff - 1) // per_eff
        out = torch.full((groups, n), float("nan"), device=dev)
        out2 = torch.full((groups, n), float("nan"), device=dev)
        _lib.run_roles([_lib.make_role(_lib.ROLE_SUM_ROWS, (part, out), (nblk, n, per)),
                        _lib.make_role(_lib.ROLE_SUM_ROWS, (part, out2), (nblk, n, per))])
        ref = torch.stack([part[g * per_eff:(g + 1) * per_eff].double().sum(0) for g in range(groups)])
        assert torch.equal(out, out2)                                        # fixed order: bitwise reproducible
        assert float((out.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), (nblk, n, per)
    # a bad role is refused before anything is launched
    with pytest.raises(_lib.QotError):
        _lib.run_roles([_lib.make_role(_lib.ROLE_SUM_ROWS, (None, out), (4, 4, 0))])
    with pytest.raises(_lib.QotError):
        _lib.run_roles([_lib.make_role(99, (), ())])


def test_launch_groups_off_equals_on(cuda_device, monkeypatch):
    """``QOT_NO_LAUNCH_GROUPS=1`` (one launch per job) and the grouped step give bit-identical outputs and the same
    gradients up to the order of the second-stage sums: grouping moves launches, not arithmetic.  Also covers a second backward right after the first (the epilogue queue is
    re-armed per backward pass) and a non-table batch (the end-of-backward callback flushes instead of TableProjectFn)."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    torch.manual_seed(5)
    batch = S.topological_batch(2, 24, n=40, e=150).to(cuda_device)
    plain = batch.to(cuda_device)
    plain.uniform_node_ids = None
    plain.edge_ptr = None
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("QOT_NO_LAUNCH_GROUPS", mode)
        torch.manual_seed(7)
        m = q.TopologicalGNN(40, 64, 3, 4, dropout_p=0.5).to(cuda_device).train()
        m._qot_seed = 4242
        outs = []
        for data in (batch, plain, batch):
            batch._qot_cache = {}
            m.zero_grad(set_to_none=True)
            out = m(data)
            torch.nn.functional.smooth_l1_loss(out, data.y.view(-1, 3)).backward()
            outs.append((out.detach().clone(), [p.grad.detach().clone() for p in m.parameters()]))
        res[mode] = outs
    for (oa, ga), (ob, gb_) in zip(res["1"], res["0"]):
        assert torch.equal(oa, ob)
        gmax = max(float(y.abs().max()) for y in gb_)
        for x, y in zip(ga, gb_):         # second-stage sums run in a different (fixed) order in the grouped launch
            assert float((x - y).abs().max()) <= 2e-6 * max(float(y.abs().max()), 1e-3 * gmax)


@pytest.mark.parametrize("H,p,n,e", [(64, 0.0, 30, 100), (64, 0.5, 30, 100), (32, 0.0, 30, 100), (256, 0.0, 30, 100),
                                     (128, 0.5, 300, 900), (16, 0.5, 150, 400)])
def test_forward_loss_equals_forward_then_criterion(cuda_device, H, p, n, e):
    """``TopologicalGNN.forward_loss`` (criterion folded into the read-out head's kernel, loss value summed by the
    backward epilogue) against ``forward`` + ``smooth_l1_loss_and_grad`` + the oracle's ``F.smooth_l1_loss``: same
    output, loss, d loss / d out and parameter gradients.  H = 256 has no fused head: the fallback path.  With a target the
    read-out runs forward, criterion and backward in ONE kernel (``qot_head_train``: a graph's rows stay in LDS up to 128
    of them -- the 300- and 150-node cases take the re-read path); passing any other gradient than the one returned
    must raise."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import functional as QF, synthetic as S
    torch.manual_seed(2)
    batch = S.topological_batch(2, 12, n=n, e=e).to(cuda_device)
    y = batch.y.view(-1, 3) * 4.0 - 1.5                   # both branches of the Huber function
    m = q.TopologicalGNN(n, H, 3, 4, dropout_p=p).to(cuda_device).train()
    m._qot_seed = 77
    res = []
    for fused in (True, False):
        m.zero_grad(set_to_none=True)
        m._qot_step.zero_()
        if fused:
            out, loss, g = m.forward_loss(batch, y, beta=1.0)
        else:
            out = m(batch)
            loss, g = QF.smooth_l1_loss_and_grad(out, y, 1.0)
        out.backward(g)
        torch.cuda.synchronize()
        res.append((out.detach().clone(), float(loss), g.clone(), [p_.grad.clone() for p_ in m.parameters()]))
    (oa, la, ga, pa), (ob, lb, gb_, pb) = res
    assert float((oa - ob).abs().max()) <= 2e-6 * float(ob.abs().max())
    assert float((ga - gb_).abs().max()) <= 2e-6 * float(gb_.abs().max())
    assert abs(la - lb) <= 1e-6 * max(1.0, abs(lb))
    assert abs(lb - float(torch.nn.functional.smooth_l1_loss(ob, y))) <= 1e-6
    gmax = max(float(y_.abs().max()) for y_ in pb)
    for x_, y_ in zip(pa, pb):          # the one-kernel read-out rounds dropout(leaky_relu(.)) in another order: 1 ulp
        assert float((x_ - y_).abs().max()) <= 2e-6 * max(float(y_.abs().max()), 1e-3 * gmax)
    if H != 256:
        out, loss, g = m.forward_loss(batch, y)
        with pytest.raises(RuntimeError, match="forward_loss"):
            out.backward(torch.ones_like(g))


def test_batchnorm_statistics_from_gat_partials_survive_large_offsets(cuda_device):
    """the statistics BatchNorm takes from the GATConv epilogue's partials (no second pass over ``[N, 4C]``) must
    not lose the variance of a channel whose mean is far from the shift (the conv bias): channels with mean ~ 1e2 and
    std ~ 1e-1 (behind a ReLU, or large activations).  Per-lane sums are now taken relative to a data value and merged
    with Chan's formula; compared with the two-pass kernel (``qot_bn_stats``) and with fp64."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import functional as QF, synthetic as S
    from gnn_qot_estimation_amd.graph import graph_index_for
    torch.manual_seed(0)
    dev = cuda_device
    batch = S.lightpath_batch(300).to(dev)
    N, heads, C = batch.num_nodes, 4, 32
    z = torch.randn(N, heads * C, device=dev) * 0.1
    z[:, 0::7] += 100.0                      # every 7th channel: mean 1e2, std 1e-1
    z[:, 3::11] -= 40.0
    conv = q.GATConv(5, C, heads=4).to(dev)
    graph = graph_index_for(batch, N, gat_self_loops=True)
    out, part = QF.GatFn.apply(z, conv.att_src, conv.att_dst, conv.bias, graph, 0.2, True)
    w, b = torch.ones(heads * C, device=dev), torch.zeros(heads * C, device=dev)
    res = {}
    for name, partials in (("partials", (part, conv.bias)), ("two_pass", None)):
        rm, rv = torch.zeros(heads * C, device=dev), torch.ones(heads * C, device=dev)
        y = QF.BnFn.apply(out.detach(), w, b, rm, rv, True, 0.1, 1e-5, False, False, partials)
        res[name] = (y, rm, rv)
    o64 = out.detach().double()
    mean64, var64 = o64.mean(0), o64.var(0, unbiased=False)
    y64 = (o64 - mean64) / torch.sqrt(var64 + 1e-5)
    rv64 = 0.9 + 0.1 * o64.var(0, unbiased=True)
    for name, (y, rm, rv) in res.items():
        assert float((rm.double() - 0.1 * mean64).abs().max()) <= 1e-5 * float(mean64.abs().max()), name
        assert float(((rv.double() - rv64) / rv64).abs().max()) <= 1e-4, (name, float(((rv.double() - rv64) / rv64).abs().max()))
        assert float((y.double() - y64).abs().max()) <= 2e-3, (name, float((y.double() - y64).abs().max()))


def test_tile_form_forward_is_taken_only_for_small_graphs(cuda_device, monkeypatch):
    """The tile form of the TransformerConv forward pays ``n`` dots per workgroup for its row of ``T_q T_k^T``: right at
    cfg2 (n = 100, 16 destinations x ~4 in-edges per workgroup), 8x the kernel's time at 1000-node graphs (cfg4 / cfg5) --
    both give the same results, so only the choice of entry point can be tested."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import _lib, synthetic as S
    calls = []
    real = _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    monkeypatch.setenv("QOT_NO_TCONV_GRAPH", "1")       # r04: graphs of <= 128 nodes take the graph form first (test_gpu_tconv_graph.py)
    for n, e, H, want_tile in ((100, 400, 64, True), (1000, 4000, 128, False), (75, 60, 16, True)):
        calls.clear()
        batch = S.topological_batch(2, 4, n=n, e=e).to(cuda_device)
        m = q.TopologicalGNN(n, H, 3, 4, dropout_p=0.0).to(cuda_device).eval()
        with torch.no_grad():
            m(batch)
        assert ("qot_tconv_fwd_tile" in calls) == want_tile, (n, H, calls)
        # large tables: logits looked up in T_q T_k^T, a workgroup per table row (qot_tconv_fwd_rows; N >= 4 V here)
        assert ("qot_tconv_fwd_rows" in calls) != want_tile and "qot_tconv_fwd" not in calls
