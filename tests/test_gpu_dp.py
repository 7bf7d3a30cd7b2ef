"""Data-parallel path with the HIP models: two processes (gloo rendezvous, both on cuda:0) each take a
contiguous graph range; gradients after the flat all-reduce -- and, for LightpathGNN, the synchronised
BatchNorm statistics -- must equal the single-process step on the whole batch.
RCCL itself only runs on the multi-GPU node; everything else of the N > 1 path is exercised here."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from helpers import rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _build(kind, dev):
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    torch.manual_seed(0)
    if kind == "topo":
        full = S.topological_batch(2, 8, n=20, e=60)
        model = q.TopologicalGNN(20, 64, 3, 4, dropout_p=0.0)
    else:
        full = S.lightpath_batch(9)              # 4 + 5 graphs: unequal LUT counts per shard
        model = q.LightpathGNN(5, 8, 3, 1, dropout_p=0.0)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1 and p.abs().max() == 0:
                p.uniform_(-0.1, 0.1)
    return full, model.to(dev).train()


def _worker(rank, world, port, kind, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnn_qot_estimation_amd.batch import shard_graphs
        from gnn_qot_estimation_amd.dp import FlatModel, loss_scale
        dev = torch.device("cuda:0")
        full, model = _build(kind, dev)
        if rank == 1:                            # parameters must come from rank 0
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        flat = FlatModel(model)
        flat.broadcast_params()
        shard = shard_graphs(full, rank, world).to(dev)
        flat.zero_grad()
        if kind == "topo":
            y = shard.y.view(-1, 3)
            loss = F.smooth_l1_loss(model(shard), y)
        else:
            out, lb = model(shard)
            y = shard.y[lb]
            loss = F.smooth_l1_loss(out, y)
        (loss * loss_scale(y.shape[0], dev)).backward()
        flat.all_reduce_grads()
        torch.cuda.synchronize()
        if rank == 0:
            ret["grad"] = flat.flat_grad.cpu()
            ret["param"] = flat.flat_param.cpu()
            if kind == "lightpath":
                ret["rm"] = model.norm1.module.running_mean.cpu()
                ret["rv"] = model.norm1.module.running_var.cpu()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["topo", "lightpath"])
def test_two_rank_hip_step_matches_single_process(kind):
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), kind, ret), nprocs=world, join=True)
        got = dict(ret)
    from gnn_qot_estimation_amd.dp import FlatModel
    dev = torch.device("cuda:0")
    full, model = _build(kind, dev)
    flat = FlatModel(model)
    assert torch.equal(got["param"], flat.flat_param.cpu())       # broadcast from rank 0
    full = full.to(dev)
    flat.zero_grad()
    if kind == "topo":
        F.smooth_l1_loss(model(full), full.y.view(-1, 3)).backward()
    else:
        out, lb = model(full)
        F.smooth_l1_loss(out, full.y[lb]).backward()
        assert rel_err(got["rm"], model.norm1.module.running_mean.cpu()) < 1e-5
        assert rel_err(got["rv"], model.norm1.module.running_var.cpu()) < 1e-5
    assert rel_err(got["grad"], flat.flat_grad.cpu()) < 2e-5


# --------------------------------------------------------------------------- harness.run_epoch, two ranks, LightpathGNN
def _lp_dataset():
    """24 lightpath graphs; graphs 4..7 and 20..23 carry NO LUT node: with batch 8 and two ranks, rank 1's share of the
    first batch (graphs 4..7) is LUT-less while the global batch is not, and the last batch's second half likewise."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    graphs = []
    for g in range(24):
        b = S.lightpath_batch(1, first_graph=g, lut=not (4 <= g < 8 or 20 <= g < 24))
        graphs.append(q.Data(x=b.x, edge_index=b.edge_index, y=b.y, num_nodes=b.num_nodes))
    return graphs


def _lp_epoch(dev, graphs):
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import harness as Hn
    from gnn_qot_estimation_amd.dp import FlatModel, FusedSGD
    torch.manual_seed(0)
    model = q.LightpathGNN(5, 8, 3, 1, dropout_p=0.0).to(dev)
    flat = FlatModel(model)
    flat.broadcast_params()
    opt = FusedSGD(flat, lr=0.1, momentum=0.9)
    res = Hn.run_epoch(model, graphs, range(24), kind="lightpath", batch_size=8, out_dim=3, device=dev,
                       criterion=torch.nn.SmoothL1Loss(), flat=flat, opt=opt)
    torch.cuda.synchronize()
    return model, flat, res


def _lp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        model, flat, res = _lp_epoch(dev, _lp_dataset())
        if rank == 0:
            ret["param"] = flat.flat_param.cpu()
            ret["rm"], ret["rv"] = model.norm1.module.running_mean.cpu(), model.norm1.module.running_var.cpu()
            ret["nbt"] = int(model.norm1.module.num_batches_tracked)
            ret["n"], ret["loss"], ret["skipped"] = res["n"], res["avg_loss"], res["skipped"]
    finally:
        dist.destroy_process_group()


def test_two_rank_lightpath_epoch_with_lut_less_shards_matches_single_process():
    """harness.run_epoch under data parallelism: whether a batch is skipped is a property of the GLOBAL batch (a rank
    whose share holds no LUT node contributes zero loss rows but joins the BatchNorm exchange), every rank walks the same
    global batches, and the epoch equals the single-process epoch: parameters, BatchNorm running statistics, row counts."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_lp_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        got = dict(ret)
    dev = torch.device("cuda:0")
    model, flat, res = _lp_epoch(dev, _lp_dataset())
    assert got["n"] == res["n"] == 16 and got["skipped"] == res["skipped"] == 0
    assert got["nbt"] == int(model.norm1.module.num_batches_tracked) == 3
    assert abs(got["loss"] - res["avg_loss"]) <= 1e-5
    assert rel_err(got["rm"], model.norm1.module.running_mean.cpu()) < 2e-5
    assert rel_err(got["rv"], model.norm1.module.running_var.cpu()) < 2e-5
    assert rel_err(got["param"], flat.flat_param.cpu()) < 5e-5


# --------------------------------------------------------------------------- RCCL itself, as far as one GPU allows
def _rccl_world1_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        from gnn_qot_estimation_amd import dp
        dev = torch.device("cuda:0")
        full, model = _build("topo", dev)
        flat = dp.FlatModel(model)
        flat.broadcast_params()
        flat.zero_grad()
        F.smooth_l1_loss(model(full.to(dev)), full.y.view(-1, 3).to(dev)).backward()
        before = flat.flat_grad.clone()
        flat.all_reduce_grads()
        scale = dp.loss_scale(5, dev)
        torch.cuda.synchronize()
        ret["mode"] = dp.reduce_mode(flat.flat_grad)
        ret["same"] = bool(torch.equal(before, flat.flat_grad))
        ret["scale"] = float(scale)
        ret["backend"] = dist.get_backend()
    finally:
        dist.destroy_process_group()


def test_rccl_single_rank_group_runs_the_avg_all_reduce():
    """The nccl (= RCCL) backend at world size 1 on the one GPU this box has: the library loads, a communicator is
    created, ``ReduceOp.AVG`` is accepted (``dp.reduce_mode`` == "avg"), the flat-gradient all-reduce and the loss-scale
    exchange run on the device and leave a single rank's values unchanged.  What stays unexercised until a multi-GPU node
    runs it is the exchange between ranks."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rccl_world1_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
        got = dict(ret)
    assert got["backend"] == "nccl" and got["mode"] == "avg"
    assert got["same"] and abs(got["scale"] - 1.0) < 1e-6
