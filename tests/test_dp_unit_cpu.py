"""``dp.FlatModel.all_reduce_grads``: which collective is issued (AVG inside RCCL vs SUM + divide) is decided once
per group, collectively, and both branches produce the mean (the AVG branch had no coverage and a
per-call try/except could issue mismatched collectives).  ``torch.distributed`` is replaced by a recording fake of a
2-rank group whose peer holds the same values."""
import types

import pytest
import torch

from gnn_qot_estimation_amd import dp


class _FakeDist:
    ReduceOp = types.SimpleNamespace(SUM="sum", AVG="avg", MIN="min", MAX="max")

    def __init__(self, backend, peer_flag=1):
        self.backend, self.calls, self.peer_flag = backend, [], peer_flag

    def is_available(self): return True
    def is_initialized(self): return True
    def get_world_size(self, group=None): return 2
    def get_backend(self, group=None): return self.backend

    def all_reduce(self, t, op=None, group=None):
        self.calls.append(op)
        if op == "sum":
            t.mul_(2)                      # the peer holds identical values
        elif op == "min":
            t.fill_(min(int(t.item()), self.peer_flag))
        # avg of two identical tensors / max: unchanged


@pytest.mark.parametrize("probe,peer,want", [(1, 1, "avg"), (0, 1, "sum"), (1, 0, "sum")])
def test_reduce_op_is_chosen_once_and_gives_the_mean(monkeypatch, probe, peer, want):
    fake = _FakeDist("nccl", peer_flag=peer)
    monkeypatch.setattr(dp, "dist", fake)
    monkeypatch.setattr(dp, "_avg_capable", lambda like, group=None: probe)
    monkeypatch.setattr(dp, "_REDUCE_MODE", {})
    m = torch.nn.Linear(3, 2)
    flat = dp.FlatModel(m)
    flat.flat_grad.copy_(torch.arange(flat.numel, dtype=torch.float32))
    before = flat.flat_grad.clone()
    flat.all_reduce_grads()
    flat.all_reduce_grads()
    assert torch.equal(flat.flat_grad, before)            # mean over two identical ranks, both branches
    assert fake.calls.count("min") == 1                   # decided once, by agreement (a peer's veto wins)
    assert [c for c in fake.calls if c != "min"] == [want, want]
    assert dp.reduce_mode(flat.flat_grad) == want


def test_gloo_never_probes_avg(monkeypatch):
    fake = _FakeDist("gloo")
    monkeypatch.setattr(dp, "dist", fake)
    monkeypatch.setattr(dp, "_REDUCE_MODE", {})
    assert dp.reduce_mode(torch.zeros(4)) == "sum" and "avg" not in fake.calls


def test_capability_flag_is_local_and_issues_no_collective(monkeypatch):
    """the decision must not ride on a data collective that a rejecting rank never joins -- the capability
    test is a local predicate (backend, device), only the MIN agreement is exchanged."""
    fake = _FakeDist("nccl")
    monkeypatch.setattr(dp, "dist", fake)
    assert dp._avg_capable(torch.zeros(4)) == 0           # CPU tensor: never AVG
    assert fake.calls == []
    monkeypatch.setattr(dp, "_REDUCE_MODE", {})
    assert dp.reduce_mode(torch.zeros(4)) == "sum"
    assert fake.calls == ["min"]


def test_force_issues_the_collective_in_a_single_rank_group(monkeypatch):
    fake = _FakeDist("gloo")
    fake.get_world_size = lambda group=None: 1
    monkeypatch.setattr(dp, "dist", fake)
    monkeypatch.setattr(dp, "_REDUCE_MODE", {})
    flat = dp.FlatModel(torch.nn.Linear(3, 2))
    flat.all_reduce_grads()
    assert fake.calls == []                               # early return at world == 1
    flat.all_reduce_grads(force=True)
    assert "sum" in fake.calls


def test_balanced_ranges_never_hands_out_an_empty_share_when_items_suffice():
    """weights [100, 1, 1, 1] over 4 ranks used to give (0,0), (0,1), (1,1), (1,4) -- two ranks empty, and
    ``harness.run_epoch`` skips a lightpath global batch when ANY rank's share is empty."""
    from gnn_qot_estimation_amd.batch import balanced_ranges
    assert balanced_ranges([100, 1, 1, 1], 4) == [(0, 1), (1, 2), (2, 3), (3, 4)]
    assert balanced_ranges([1, 1, 1, 100], 4) == [(0, 1), (1, 2), (2, 3), (3, 4)]
    for w, world in (([100, 1, 1, 1, 1, 1], 3), ([5, 5, 5, 5, 5, 5, 5, 5], 4), ([1, 9, 1, 9, 1, 9, 1], 3), ([0, 0, 7, 0, 0], 5), ([3] * 17, 8)):
        r = balanced_ranges(w, world)
        assert r[0][0] == 0 and r[-1][1] == len(w)
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert all(hi > lo for lo, hi in r), (w, world, r)
    # fewer items than ranks: contiguous cover, the surplus ranks are empty
    r = balanced_ranges([4, 4], 3)
    assert r[0][0] == 0 and r[-1][1] == 2 and sum(hi - lo for lo, hi in r) == 2
    assert balanced_ranges([], 2) == [(0, 0), (0, 0)]
    # even weights still split evenly
    assert balanced_ranges([1] * 8, 2) == [(0, 4), (4, 8)]
