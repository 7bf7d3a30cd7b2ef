"""GPU parity of TransformerConv's GRAPH form (csrc/tconv_graph.hip; topological_training/models.py:51-55 when
``node_ids == arange(n)`` in every graph): against the CPU oracle, against an fp64 restatement of the table-level
quantities, and against the per-destination kernels it replaces (``QOT_NO_TCONV_GRAPH=1``).

Tolerance: <= 1e-4 relative (BASELINE.json's bar) against the oracle; the two HIP forms agree to rounding (the logits'
H-term dot is summed in another order)."""
import pytest
import torch

from helpers import TOL, rel_err

pytestmark = pytest.mark.gpu


def _models(device, **kw):
    import gnn_qot_estimation_amd as q
    from oracle import sparse as O
    torch.manual_seed(0)
    ref = O.TopologicalGNN(**kw)
    hip = q.TopologicalGNN(**kw)
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() == 1 and p.abs().max() == 0:
                p.uniform_(-0.1, 0.1)
    hip.load_state_dict(ref.state_dict(), strict=True)
    return ref, hip.to(device)


def _step(model, batch, y):
    for p in model.parameters():
        p.grad = None
    out = model(batch)
    torch.nn.functional.smooth_l1_loss(out, y).backward()
    return out, {k: p.grad.detach().clone() for k, p in model.named_parameters()}


def _calls(monkeypatch):
    from gnn_qot_estimation_amd import _lib
    calls = []
    real = _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    return calls


@pytest.mark.parametrize("H,V,n,D", [(16, 75, 75, 4), (64, 100, 100, 4), (32, 40, 33, 3), (128, 128, 128, 4), (256, 20, 9, 1),
                                     (64, 7, 5, 6)])
def test_table_scores_match_fp64(cuda_device, H, V, n, D):
    """M = T_q T_k^T / sqrt(H), P = T_q W_e / sqrt(H) from the parameters (the kernel never forms T_k)."""
    from gnn_qot_estimation_amd import _lib
    torch.manual_seed(H + n)
    f = lambda *s: torch.randn(*s)
    table, wq, bq, wk, bk, we = f(V, H), f(H, H) / H ** 0.5, f(H), f(H, H) / H ** 0.5, f(H), f(H, D)
    tq = table.double() @ wq.double().t() + bq.double()
    tk = table.double() @ wk.double().t() + bk.double()
    M64 = (tq[:n] @ tk[:n].t()) / H ** 0.5
    P64 = (tq[:n] @ we.double()) / H ** 0.5
    dev = cuda_device
    ldm = _lib.load().qot_tconv_graph_ldm(n)
    M = torch.full((n, ldm), float("nan"), device=dev)
    Pm = torch.full((n, D), float("nan"), device=dev)
    args = [t.to(dev) for t in (table, wq, bq, wk, bk, we)]
    _lib.call("qot_table_scores", *args, M, Pm, n, H, D)
    torch.cuda.synchronize()
    assert rel_err(M[:, :n], M64) <= 1e-5
    assert rel_err(Pm, P64) <= 1e-5
    assert float(M[:, n:].abs().sum()) == 0.0


@pytest.mark.parametrize("B,n,e,H,V,D,p", [
    (8, 100, 400, 64, 100, 4, 0.0),        # cfg2's shape
    (6, 14, 42, 32, 75, 4, 0.0),           # V > n
    (5, 128, 300, 16, 128, 4, 0.0),        # the largest graph the form takes
    (3, 30, 80, 128, 30, 3, 0.0),
    (2, 12, 30, 256, 12, 2, 0.0),
    (4, 40, 0, 64, 40, 4, 0.0),            # trees only (e = 0 asks for no extra links)
    (1, 1, 0, 64, 3, 4, 0.0),              # single-node graph
    (7, 25, 90, 64, 25, 6, 0.0),           # edge_dim > 4
])
def test_graph_form_matches_oracle_and_the_per_destination_kernels(cuda_device, monkeypatch, B, n, e, H, V, D, p):
    from gnn_qot_estimation_amd import synthetic as S
    batch = S.topological_batch(2, B, n=n, e=e, edge_dim=D)
    ref, hip = _models(cuda_device, num_nodes=V, hidden_channels=H, out_channels=3, edge_dim=D, dropout_p=p)
    ref.train(); hip.train()
    y = batch.y.view(-1, 3)
    dbatch = batch.to(cuda_device)
    calls = _calls(monkeypatch)
    out, grads = _step(hip, dbatch, y.to(cuda_device))
    assert "qot_tconv_fwd_graph" in calls and "qot_tconv_bwd_graph" in calls, calls
    assert "qot_tconv_bwd_dst" not in calls
    out_ref, grads_ref = _step(ref, batch, y)
    assert rel_err(out, out_ref) <= TOL
    gmax = max(float(g.abs().max()) for g in grads_ref.values())
    for k, g in grads.items():
        floor = gmax if k == "conv1.lin_key.bias" else 1e-3 * gmax        # softmax shift invariance: analytically zero
        err = float((g.double().cpu() - grads_ref[k].double()).abs().max() / max(float(grads_ref[k].abs().max()), floor))
        assert err <= TOL, (k, err)
    if V > n:
        assert float(grads["node_embeddings.weight"][n:].abs().max()) == 0.0
    # the kernels it replaces
    monkeypatch.setenv("QOT_NO_TCONV_GRAPH", "1")
    dbatch._qot_cache = {}
    calls.clear()
    out2, grads2 = _step(hip, dbatch, y.to(cuda_device))
    assert "qot_tconv_fwd_graph" not in calls and "qot_tconv_bwd_dst" in calls
    assert rel_err(out, out2) <= 2e-5
    for k, g in grads.items():
        floor = gmax if k == "conv1.lin_key.bias" else 1e-3 * gmax
        err = float((g - grads2[k]).abs().max() / max(float(grads2[k].abs().max()), floor))
        assert err <= 2e-5, (k, err)


def test_graph_form_irregular_degrees_duplicates_and_isolated_nodes(cuda_device, monkeypatch):
    """Equal-size graphs with a hub (in-degree 40), duplicate edges, self loops and isolated nodes (zero in-degree rows
    yield exactly skip_i: SURVEY App. B.1)."""
    import gnn_qot_estimation_amd as q
    n, H, D = 24, 64, 4
    gen = torch.Generator().manual_seed(5)
    graphs = []
    for b in range(9):
        src = torch.randint(0, n - 4, (60,), generator=gen)
        dst = torch.randint(0, n - 4, (60,), generator=gen)
        if b % 3 == 0:
            dst[:40] = 3                                   # hub
        if b % 3 == 1:
            src[10:20], dst[10:20] = src[:10], dst[:10]    # duplicate edges
            src[20:24] = dst[20:24]                        # self loops
        ei = torch.stack([src, dst])                        # nodes n-4 .. n-1 stay isolated
        graphs.append(q.Data(edge_index=ei, edge_attr=torch.rand(60, D, generator=gen), node_ids=torch.arange(n),
                             y=torch.rand(3, generator=gen), num_nodes=n))
    batch = q.Batch.from_data_list(graphs)
    assert batch.uniform_node_ids == n
    ref, hip = _models(cuda_device, num_nodes=n, hidden_channels=H, out_channels=3, edge_dim=D, dropout_p=0.0)
    y = batch.y.view(-1, 3)
    calls = _calls(monkeypatch)
    out, grads = _step(hip, batch.to(cuda_device), y.to(cuda_device))
    assert "qot_tconv_fwd_graph" in calls
    out_ref, grads_ref = _step(ref, batch, y)
    assert rel_err(out, out_ref) <= TOL
    gmax = max(float(g.abs().max()) for g in grads_ref.values())
    for k, g in grads.items():
        floor = gmax if k == "conv1.lin_key.bias" else 1e-3 * gmax
        err = float((g.double().cpu() - grads_ref[k].double()).abs().max() / max(float(grads_ref[k].abs().max()), floor))
        assert err <= TOL, (k, err)


@pytest.mark.parametrize("H", [64, 16, 32])
def test_graph_form_many_graphs_is_bitwise_reproducible_and_position_independent(cuda_device, monkeypatch, H):
    """More graphs than one pass of the persistent workgroups takes (forward: 4 x 256, backward: 256): a batch tiling 5
    distinct graphs 300 times gives every copy the same output bits, two runs agree bit for bit (outputs and every
    gradient), and the tiled batch's parameter gradients equal the base batch's (mean loss over identical copies)."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    n, e = 20, 60
    base = S.topological_batch(2, 5, n=n, e=e)
    parts = []
    ptr, eptr = base.ptr.tolist(), base.edge_ptr.tolist()
    for b in range(5):
        lo, hi = eptr[b], eptr[b + 1]
        parts.append(q.Data(edge_index=base.edge_index[:, lo:hi] - ptr[b], edge_attr=base.edge_attr[lo:hi],
                            node_ids=torch.arange(n), y=base.y.view(-1, 3)[b], num_nodes=n))
    reps = 300
    big = q.Batch.from_data_list(parts * reps)
    torch.manual_seed(0)
    hip = q.TopologicalGNN(n, H, 3, 4, dropout_p=0.0).to(cuda_device).train()
    calls = _calls(monkeypatch)
    yb, ys = big.y.view(-1, 3).to(cuda_device), base.y.view(-1, 3).to(cuda_device)
    dbig = big.to(cuda_device)
    out1, g1 = _step(hip, dbig, yb)
    assert "qot_tconv_fwd_graph" in calls
    dbig._qot_cache = {}
    out2, g2 = _step(hip, dbig, yb)
    assert torch.equal(out1, out2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    o = out1.view(reps, 5, 3)
    assert torch.equal(o, o[:1].expand_as(o))
    outs, gs = _step(hip, base.to(cuda_device), ys)
    assert rel_err(o[0], outs) <= 1e-6
    gmax = max(float(g.abs().max()) for g in gs.values())
    for k in g1:
        err = float((g1[k] - gs[k]).abs().max() / max(float(gs[k].abs().max()), 1e-3 * gmax))
        assert err <= 2e-5, (k, err)


def test_graph_form_dropout_mask_is_the_per_destination_kernels_mask(cuda_device, monkeypatch):
    """The fused leaky_relu + dropout epilogue draws from (seed, step counter, element): the graph form and the kernels it
    replaces drop the SAME elements, forward and backward."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    batch = S.topological_batch(2, 6, n=50, e=160).to(cuda_device)
    torch.manual_seed(0)
    hip = q.TopologicalGNN(50, 64, 3, 4, dropout_p=0.5).to(cuda_device).train()
    y = batch.y.view(-1, 3)
    state = hip._qot_step.clone()
    out1, g1 = _step(hip, batch, y)
    hip._qot_step.copy_(state)
    monkeypatch.setenv("QOT_NO_TCONV_GRAPH", "1")
    batch._qot_cache = {}
    out2, g2 = _step(hip, batch, y)
    assert rel_err(out1, out2) <= 2e-5
    gmax = max(float(g.abs().max()) for g in g2.values())
    for k in g1:
        floor = gmax if k == "conv1.lin_key.bias" else 1e-3 * gmax
        err = float((g1[k] - g2[k]).abs().max() / max(float(g2[k].abs().max()), floor))
        assert err <= 5e-5, (k, err)


def test_graph_form_is_left_for_large_or_ragged_batches(cuda_device, monkeypatch):
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    calls = _calls(monkeypatch)
    for n, e, H, want in ((100, 400, 64, True), (129, 300, 64, False), (1000, 4000, 128, False), (75, 60, 16, True)):
        calls.clear()
        batch = S.topological_batch(2, 3, n=n, e=e).to(cuda_device)
        m = q.TopologicalGNN(n, H, 3, 4, dropout_p=0.0).to(cuda_device).eval()
        with torch.no_grad():
            m(batch)
        assert ("qot_tconv_fwd_graph" in calls) == want, (n, H, calls)


def test_graph_form_edgeless_batch(cuda_device, monkeypatch):
    """No edge anywhere: every row is its skip row; every attention-side gradient is exactly zero."""
    import gnn_qot_estimation_amd as q
    n, H = 10, 64
    graphs = [q.Data(edge_index=torch.zeros(2, 0, dtype=torch.long), edge_attr=torch.zeros(0, 4), node_ids=torch.arange(n),
                     y=torch.rand(3), num_nodes=n) for _ in range(3)]
    batch = q.Batch.from_data_list(graphs)
    ref, hip = _models(cuda_device, num_nodes=n, hidden_channels=H, out_channels=3, edge_dim=4, dropout_p=0.0)
    y = batch.y.view(-1, 3)
    calls = _calls(monkeypatch)
    out, grads = _step(hip, batch.to(cuda_device), y.to(cuda_device))
    assert "qot_tconv_fwd_graph" in calls
    out_ref, grads_ref = _step(ref, batch, y)
    assert rel_err(out, out_ref) <= TOL
    for k in ("conv1.lin_query.weight", "conv1.lin_key.weight", "conv1.lin_value.weight", "conv1.lin_edge.weight"):
        assert float(grads[k].abs().max()) == 0.0, k
    assert rel_err(grads["conv1.lin_skip.weight"], grads_ref["conv1.lin_skip.weight"]) <= TOL
    assert rel_err(grads["node_embeddings.weight"], grads_ref["node_embeddings.weight"]) <= TOL
