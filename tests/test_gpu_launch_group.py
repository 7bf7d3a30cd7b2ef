"""Launch-group robustness (round-3 advisor findings): a deferred gradient must never be read before the epilogue has
filled it, and a failed forward prologue must leave nothing behind in the batch object's cache."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _step(model, batch, y):
    for p in model.parameters():
        p.grad = None
    out = model(batch)
    torch.nn.functional.smooth_l1_loss(out, y).backward()
    return {k: p.grad.detach().clone() for k, p in model.named_parameters()}


@pytest.mark.parametrize("what", ["tensor_hook", "post_accumulate_hook", "existing_grad", "tied"])
def test_deferred_gradients_with_hooks_equal_the_immediate_launches(cuda_device, monkeypatch, what):
    """A clamp hook on a parameter (runs on the gradient as soon as autograd has it), a post-accumulate hook, an existing
    ``.grad`` (accumulated into at once): the default (grouped epilogue) must give what ``QOT_NO_LAUNCH_GROUPS=1`` gives."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    batch = S.topological_batch(2, 6, n=30, e=100).to(cuda_device)
    y = batch.y.view(-1, 3)
    torch.manual_seed(0)
    base = q.TopologicalGNN(30, 64, 3, 4, dropout_p=0.0).to(cuda_device).train()
    seen = []

    def prepare(m):
        names = ("conv2.nn.2.weight", "mlp.0.weight", "conv1.lin_edge.weight", "conv1.lin_query.weight")
        ps = dict(m.named_parameters())
        if what == "tensor_hook":
            for k in names:
                ps[k].register_hook(lambda g: g.clamp(-1e-3, 1e-3))
        elif what == "post_accumulate_hook":
            for k in names:
                ps[k].register_post_accumulate_grad_hook(lambda p: seen.append(float(p.grad.abs().sum())))
        elif what == "tied":
            m.conv1.lin_key.weight = m.conv1.lin_query.weight          # one Parameter, two uses

    a, b = copy.deepcopy(base), copy.deepcopy(base)
    prepare(a)
    prepare(b)

    def run(m):
        if what == "existing_grad":
            for p in m.parameters():
                p.grad = torch.ones_like(p)
            out = m(batch)
            torch.nn.functional.smooth_l1_loss(out, y).backward()
            return {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        return _step(m, batch, y)

    ga = run(a)
    n_a = list(seen)
    seen.clear()
    monkeypatch.setenv("QOT_NO_LAUNCH_GROUPS", "1")
    batch._qot_cache = {}
    gb = run(b)
    gmax = max(float(g.abs().max()) for g in gb.values())
    for k in ga:
        err = float((ga[k] - gb[k]).abs().max()) / max(float(gb[k].abs().max()), 1e-3 * gmax)
        assert err <= 2e-5, (what, k, err)
    if what == "post_accumulate_hook":
        assert len(n_a) == len(seen) == 4
        for u, v in zip(sorted(n_a), sorted(seen)):
            assert abs(u - v) <= 2e-5 * max(abs(v), 1e-6), (u, v)       # the hook saw FILLED gradients


def test_inconsistent_slices_raise_every_time_and_cache_nothing(cuda_device):
    """A batch whose ``edge_ptr`` does not describe ``edge_index``: the grouped prologue raises ``QotError`` -- and the NEXT
    forward on the same batch object raises again instead of gathering through the ``torch.empty`` index arrays the failed
    build left behind (round 3 cached them before the launch)."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import _lib, synthetic as S
    batch = S.topological_batch(2, 5, n=20, e=60).to(cuda_device)
    bad = batch.edge_ptr.clone()
    bad[2] = bad[2] + 7                      # graph 1 claims edges of graph 2
    batch.edge_ptr = bad
    m = q.TopologicalGNN(20, 64, 3, 4, dropout_p=0.0).to(cuda_device).eval()
    for _ in range(2):
        with pytest.raises(_lib.QotError):
            with torch.no_grad():
                m(batch)
        c = getattr(batch, "_qot_cache", {})
        assert ("graph", False) not in c and "tmaps" not in c and "ptr32" not in c, list(c)
    # a forward that raises BEFORE the launch (id outside the embedding table) caches nothing either
    good = S.topological_batch(2, 5, n=20, e=60).to(cuda_device)
    small = q.TopologicalGNN(10, 64, 3, 4, dropout_p=0.0).to(cuda_device).eval()
    with pytest.raises(IndexError):
        with torch.no_grad():
            small(good)
    c = getattr(good, "_qot_cache", {})
    assert ("graph", False) not in c and "tmaps" not in c, list(c)
    with torch.no_grad():
        out = m(good)                                             # and the same object still works with a fitting model
    assert torch.isfinite(out).all()
