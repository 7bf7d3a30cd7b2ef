"""GATConv's walk kernels (csrc/gat.hip, r04) on graphs that take every branch of the walk: rows served from the slot-private
LDS image, the prefetched row, plain loads; destinations with more in-edges than one batch; nodes whose only in-edge is the
self loop; every per-head width (one destination per wave down to sixteen per wave).  Oracle: oracle/sparse.py's GATConv
(fp32 torch, gradient-checked in fp64 by tests/test_oracle_dual.py), same weights; forward and every gradient to the <= 1e-4
relative bar of BASELINE.json's north_star.  "parity unpinned" by the reference's own tests (it has none for this operator)."""
import pytest
import torch

from helpers import TOL, rel_err

pytestmark = pytest.mark.gpu


def _graph(kind, n, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "chain":                       # LightpathGNN's numbering: sources i - 1, i, i + 1 (cache + prefetch path)
        a = torch.arange(n - 1)
        return torch.cat([torch.stack([a, a + 1]), torch.stack([a + 1, a])], dim=1)
    if kind == "shuffled_chain":              # the same chains, numbered at random: every source is a plain load
        perm = torch.randperm(n, generator=g)
        a = torch.arange(n - 1)
        ei = torch.cat([torch.stack([a, a + 1]), torch.stack([a + 1, a])], dim=1)
        return perm[ei]
    if kind == "hubs":                        # a few destinations with 40+ in-edges (several batches), many with none
        m = 6 * n
        src = torch.randint(0, n, (m,), generator=g)
        dst = torch.randint(0, max(n // 8, 1), (m,), generator=g)
        extra = torch.randint(0, n, (2, n // 2), generator=g)          # some ordinary edges, some existing self loops
        loops = torch.arange(0, n, 7)
        return torch.cat([torch.stack([src, dst]), extra, torch.stack([loops, loops])], dim=1)
    if kind == "band":                        # neighbours at distance <= 3: partial overlap between consecutive destinations
        a = torch.arange(n)
        cols = [torch.stack([(a + d) % n, a]) for d in (-3, -1, 1, 2)]
        return torch.cat(cols, dim=1)
    raise ValueError(kind)


# small cases: every width on every kind of graph (one destination per lane group: the index pipeline's clamps and the
# first-destination path); large cases: more rows than one resident round of workgroups has lane groups (1024 workgroups x
# 256 / min(64, C) of them), so that every lane group walks a run of several destinations and the cache / prefetch paths run
_SMALL = [(k, n, C) for C in (4, 16, 64, 128, 256)
          for k, n in (("chain", 257), ("shuffled_chain", 131), ("hubs", 203), ("band", 96))]
_LARGE = [("chain", 70001, 4), ("band", 70001, 16), ("hubs", 70001, 4), ("chain", 20001, 64), ("chain", 9001, 128),
          ("hubs", 9001, 128), ("shuffled_chain", 9001, 128), ("band", 9001, 128), ("band", 9001, 256), ("chain", 13001, 256)]


@pytest.mark.parametrize("thin", [False, True])
@pytest.mark.parametrize("kind,n,C", _SMALL + _LARGE)
def test_gatconv_walk_matches_oracle(cuda_device, monkeypatch, kind, n, C, thin):
    """``thin``: the first layer's form (``GatThinFn``: the projection of the 5 input features formed inside the attention
    kernels, z never materialised) at every size; otherwise the dense kernels behind the skinny projection."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import _lib
    from oracle import sparse as O
    if thin:
        if not _lib.load().qot_gat_thin_supported(4, C, 5):
            pytest.skip("W^T does not fit next to the walk's LDS image at this width")
        monkeypatch.setenv("QOT_GAT_THIN_MIN_ROWS", "1")
    else:
        monkeypatch.setenv("QOT_NO_GAT_THIN", "1")
    torch.manual_seed(1)
    ref = O.GATConv(5, C, heads=4)
    hip = q.GATConv(5, C, heads=4)
    with torch.no_grad():
        ref.bias.uniform_(-0.2, 0.2)
    hip.load_state_dict(ref.state_dict(), strict=True)
    hip.to(cuda_device)
    ei = _graph(kind, n, seed=3)
    x = torch.randn(n, 5)
    w = torch.randn(n, 4 * C)                 # a fixed cotangent: every output column and row carries gradient
    xr = x.clone().requires_grad_(True)
    out_ref = ref(xr, ei)
    (out_ref * w).sum().backward()
    xh = x.clone().to(cuda_device).requires_grad_(True)
    out = hip(xh, ei.to(cuda_device))
    (out * w.to(cuda_device)).sum().backward()
    assert hip.thin_ok(x.to(cuda_device)) == thin
    assert rel_err(out, out_ref) <= TOL, (kind, C)
    assert rel_err(xh.grad, xr.grad) <= TOL, (kind, C)
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters())
    for (name, p), (_, pr) in zip(hip.named_parameters(), ref.named_parameters()):
        a, b = p.grad.detach().double().cpu(), pr.grad.detach().double()
        e = float((a - b).abs().max() / max(float(b.abs().max()), 1e-3 * gmax))
        assert e <= TOL, (kind, C, name, e)


def test_gatconv_walk_is_reproducible(cuda_device):
    """Two runs give the same bits: fixed summation orders everywhere, the bias gradient's per-workgroup partials included."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd.graph import build_graph_index
    torch.manual_seed(2)
    n, C = 13000, 128
    conv = q.GATConv(5, C, heads=4).to(cuda_device)
    ei = _graph("chain", n, 0).to(cuda_device)
    graph = build_graph_index(ei, n, gat_self_loops=True)
    x = torch.randn(n, 5, device=cuda_device)
    w = torch.randn(n, 4 * C, device=cuda_device)

    def run():
        for p in conv.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        out = conv(xx, ei, graph=graph)
        (out * w).sum().backward()
        return [out.detach().clone(), xx.grad.clone()] + [p.grad.clone() for p in conv.parameters()]

    a, b = run(), run()
    for u, v in zip(a, b):
        assert torch.equal(u, v)
