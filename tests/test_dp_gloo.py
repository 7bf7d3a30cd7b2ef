"""Data-parallel path on CPU with gloo, world_size 2: sharding by
contiguous graph ranges + ONE flat-gradient all-reduce reproduces the single-process
gradient.  The model here is the CPU oracle (the HIP modules have no CPU path); what is
under test is gnn_qot_estimation_amd.dp / batch.shard_graphs, which are device-agnostic."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, kind, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from gnn_qot_estimation_amd import synthetic as S
        from gnn_qot_estimation_amd.batch import shard_graphs
        from gnn_qot_estimation_amd.dp import FlatModel
        from oracle import sparse as O
        torch.manual_seed(0)
        if kind == "topo":
            full = S.topological_batch(2, 8, n=12, e=30)
            model = O.TopologicalGNN(12, 8, 3, 4, dropout_p=0.0)
        else:
            full = S.lightpath_batch(9)          # uneven LUT counts per shard (4 vs 5 graphs)
            model = O.LightpathGNN(5, 8, 3, 1, dropout_p=0.0).eval()    # eval: BN stats not sharded
        if rank == 1:                            # params must come from rank 0
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        flat = FlatModel(model)
        flat.broadcast_params()
        shard = shard_graphs(full, rank, world)
        flat.zero_grad()
        if kind == "topo":
            loss = F.smooth_l1_loss(model(shard), shard.y.view(-1, 3))
            loss.backward()
            flat.all_reduce_grads()
        else:
            out, lb = model(shard)
            loss = F.smooth_l1_loss(out, shard.y[lb])
            loss.backward()
            flat.all_reduce_grads(weight=torch.tensor(float(out.shape[0])))
        if rank == 0:
            ret["grad"] = flat.flat_grad.clone()
            ret["param"] = flat.flat_param.clone()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["topo", "lightpath"])
def test_two_rank_gradient_matches_single_process(kind):
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), kind, ret), nprocs=world, join=True)
        g2, p2 = ret["grad"], ret["param"]
    from gnn_qot_estimation_amd import synthetic as S
    from gnn_qot_estimation_amd.dp import FlatModel
    from oracle import sparse as O
    torch.manual_seed(0)
    if kind == "topo":
        full = S.topological_batch(2, 8, n=12, e=30)
        model = O.TopologicalGNN(12, 8, 3, 4, dropout_p=0.0)
        flat = FlatModel(model)
        F.smooth_l1_loss(model(full), full.y.view(-1, 3)).backward()
    else:
        full = S.lightpath_batch(9)
        model = O.LightpathGNN(5, 8, 3, 1, dropout_p=0.0).eval()
        flat = FlatModel(model)
        out, lb = model(full)
        F.smooth_l1_loss(out, full.y[lb]).backward()
    assert torch.allclose(p2, flat.flat_param)
    assert torch.allclose(g2, flat.flat_grad, rtol=1e-4, atol=1e-7), (g2 - flat.flat_grad).abs().max()


def test_flat_model_keeps_state_dict_and_single_leaf():
    from gnn_qot_estimation_amd.dp import FlatModel
    from oracle import sparse as O
    m = O.TopologicalGNN(10, 8, 3, 4)
    keys = list(m.state_dict().keys())
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = FlatModel(m)
    assert list(m.state_dict().keys()) == keys
    assert all(torch.equal(before[k], v) for k, v in m.state_dict().items())
    assert flat.numel == sum(p.numel() for p in m.parameters())
    opt = torch.optim.SGD([flat.leaf], lr=0.5)
    flat.flat_grad.fill_(1.0)
    opt.step()
    assert torch.allclose(m.conv2.bias, before["conv2.bias"] - 0.5)   # views follow the fused update


# --------------------------------------------------------------------------- harness.run_epoch with world > 1
class _TorchSGD:
    """``run_epoch`` only calls ``opt.step()``; on the CPU the fused HIP update is replaced by torch's."""

    def __init__(self, flat, lr, momentum):
        self.inner = torch.optim.SGD([flat.leaf], lr=lr, momentum=momentum)

    def step(self):
        self.inner.step()


def _epoch_setup(total):
    from gnn_qot_estimation_amd import synthetic as S
    from oracle import sparse as O
    import gnn_qot_estimation_amd as q
    torch.manual_seed(0)
    graphs = []
    for g in range(total):
        b = S.topological_batch(2, 1, n=10, e=24, first_graph=g)
        graphs.append(q.Data(edge_index=b.edge_index, edge_attr=b.edge_attr, node_ids=b.node_ids, num_nodes=10, y=b.y))
    model = O.TopologicalGNN(10, 8, 3, 4, dropout_p=0.0)
    return graphs, model


def _epoch_worker(rank, world, port, total, n_idx, bs, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from gnn_qot_estimation_amd import harness as Hn
        from gnn_qot_estimation_amd.dp import FlatModel
        graphs, model = _epoch_setup(total)
        flat = FlatModel(model)
        flat.broadcast_params()
        opt = _TorchSGD(flat, 0.1, 0.9)
        res = Hn.run_epoch(model, graphs, range(n_idx), kind="topological", batch_size=bs, out_dim=3, device="cpu",
                           criterion=torch.nn.SmoothL1Loss(), flat=flat, opt=opt)
        ev = Hn.run_epoch(model, graphs, range(n_idx), kind="topological", batch_size=bs, out_dim=3, device="cpu",
                          criterion=torch.nn.SmoothL1Loss())
        ret[rank] = dict(param=flat.flat_param.clone(), loss=res["avg_loss"], n=res["n"], r2=res["r2"], ev=ev["avg_loss"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_idx,bs", [(2, 21, 5), (3, 23, 7)])
def test_run_epoch_ranks_stay_in_step_and_match_single_process(world, n_idx, bs):
    """ADVICE r1 (harness.py:241): len(indices) % batch_size in {1, world-1}, batch_size not divisible by the
    world size, a trailing batch smaller than the world size -- no hang, and the epoch equals the 1-rank epoch."""
    assert n_idx % bs in (1, world - 1) and bs % world != 0
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_epoch_worker, args=(world, _free_port(), 24, n_idx, bs, ret), nprocs=world, join=True)
        got = dict(ret)
    from gnn_qot_estimation_amd import harness as Hn
    from gnn_qot_estimation_amd.dp import FlatModel
    graphs, model = _epoch_setup(24)
    flat = FlatModel(model)
    opt = _TorchSGD(flat, 0.1, 0.9)
    one = Hn.run_epoch(model, graphs, range(n_idx), kind="topological", batch_size=bs, out_dim=3, device="cpu",
                       criterion=torch.nn.SmoothL1Loss(), flat=flat, opt=opt)
    ev = Hn.run_epoch(model, graphs, range(n_idx), kind="topological", batch_size=bs, out_dim=3, device="cpu",
                      criterion=torch.nn.SmoothL1Loss())
    for r in range(world):
        assert torch.allclose(got[r]["param"], flat.flat_param, rtol=1e-4, atol=1e-6), r
        assert got[r]["n"] == one["n"] == n_idx
        # the per-rank loss sums weight every rank's local mean by its row count -> the global sum
        assert abs(got[r]["loss"] - one["avg_loss"]) < 1e-5 and abs(got[r]["ev"] - ev["avg_loss"]) < 1e-5
        assert abs(got[r]["r2"] - one["r2"]) < 1e-4


# --------------------------------------------------------------------------- work-balanced sharding of skewed batches
def _skewed_batch():
    """One power-law graph (60 nodes, hubs) in front of twelve small ones: equal graph counts per rank would give
    rank 0 almost all the edges."""
    from gnn_qot_estimation_amd import synthetic as S
    import gnn_qot_estimation_amd as q
    big = S.topological_batch(5, 1, n=60)
    parts = [q.Data(edge_index=big.edge_index, edge_attr=big.edge_attr, node_ids=big.node_ids, num_nodes=60, y=big.y)]
    for g in range(12):
        b = S.topological_batch(2, 1, n=12, e=22, first_graph=g)
        parts.append(q.Data(edge_index=b.edge_index, edge_attr=b.edge_attr, node_ids=b.node_ids, num_nodes=12, y=b.y))
    return q.Batch.from_data_list(parts)


def _skew_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from gnn_qot_estimation_amd.batch import shard_graphs
        from gnn_qot_estimation_amd.dp import FlatModel, loss_scale
        from oracle import sparse as O
        torch.manual_seed(0)
        full = _skewed_batch()
        model = O.TopologicalGNN(60, 8, 3, 4, dropout_p=0.0)
        flat = FlatModel(model)
        flat.broadcast_params()
        shard = shard_graphs(full, rank, world, balance="edges")
        flat.zero_grad()
        y = shard.y.view(-1, 3)
        loss = F.smooth_l1_loss(model(shard), y)
        (loss * loss_scale(y.shape[0], torch.device("cpu"))).backward()     # shards hold different graph counts
        flat.all_reduce_grads()
        ret[rank] = dict(grad=flat.flat_grad.clone(), edges=shard.num_edges, graphs=shard.num_graphs)
    finally:
        dist.destroy_process_group()


def test_edge_balanced_two_rank_split_of_a_skewed_batch():
    """SURVEY 8(e) 'balanced by sum e': the 2-rank split evens out the work within 10 % and the averaged gradient is
    still the single-process gradient."""
    from gnn_qot_estimation_amd.batch import balanced_ranges, graph_costs, shard_graphs
    from gnn_qot_estimation_amd.dp import FlatModel, graph_range
    from oracle import sparse as O
    full = _skewed_batch()
    costs = graph_costs(full)
    (a0, a1), (b0, b1) = balanced_ranges(costs, 2)
    wa, wb = float(costs[a0:a1].sum()), float(costs[b0:b1].sum())
    assert a0 == 0 and a1 == b0 and b1 == 13 and abs(wa - wb) <= 0.10 * max(wa, wb)
    assert graph_range(13, 1, 2, costs=costs) == (b0, b1)
    by_count = [shard_graphs(full, r, 2).num_edges for r in range(2)]
    assert max(by_count) > 2 * min(by_count)                 # what the count split would have done
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_skew_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        got = dict(ret)
    assert got[0]["graphs"] + got[1]["graphs"] == 13 and got[0]["graphs"] != got[1]["graphs"]
    assert abs(got[0]["edges"] - got[1]["edges"]) <= 0.10 * max(got[0]["edges"], got[1]["edges"])
    torch.manual_seed(0)
    model = O.TopologicalGNN(60, 8, 3, 4, dropout_p=0.0)
    flat = FlatModel(model)
    F.smooth_l1_loss(model(full), full.y.view(-1, 3)).backward()
    for r in range(2):
        assert torch.allclose(got[r]["grad"], flat.flat_grad, rtol=1e-4, atol=1e-7), (got[r]["grad"] - flat.flat_grad).abs().max()
