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
    monkeypatch.setattr(dp, "_probe_avg", lambda like, group=None: probe)
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
