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


def _lp_eval(model, dev):
    """evaluate() over 24 graphs in batches of 4: global batch 1 (graphs 4..7) and 5 (20..23) hold no LUT node at all
    (skipped as a whole, 8 graphs), every other batch is LUT-carrying on both ranks' shares."""
    from gnn_qot_estimation_amd import harness as Hn
    # second dataset: graphs 2..3 LUT-less as well, so that with batch 4 and two ranks rank 1's share of batch 0 is
    # LUT-less while the GLOBAL batch is not -- per-shard counting (round 2) reported those 2 graphs as skipped
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    graphs = []
    for g in range(24):
        b = S.lightpath_batch(1, first_graph=g, lut=not (2 <= g < 8 or 20 <= g < 24))
        graphs.append(q.Data(x=b.x, edge_index=b.edge_index, y=b.y, num_nodes=b.num_nodes))
    metrics, y_true, y_pred, skipped = Hn.evaluate(model, graphs, range(24), kind="lightpath", batch_size=4, device=dev,
                                                   return_predictions=True)
    return dict(skipped=int(skipped), rows=int(y_true.shape[0]), y_pred=y_pred.clone(),
                r2=[metrics[k]["R2"] for k in ("OSNR", "SNR", "BER")])


def _lp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        model, flat, res = _lp_epoch(dev, _lp_dataset())
        ev = _lp_eval(model, dev)
        if rank == 0:
            ret["eval"] = ev
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
    # evaluate(): the LUT skip is a property of the GLOBAL batch under data parallelism too
    ev = _lp_eval(model, dev)
    assert got["eval"]["skipped"] == ev["skipped"] == 8
    assert got["eval"]["rows"] == ev["rows"] == 14
    assert rel_err(got["eval"]["y_pred"], ev["y_pred"]) < 5e-5
    assert all(abs(a - b) <= 1e-4 * max(1.0, abs(b)) for a, b in zip(got["eval"]["r2"], ev["r2"]))


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
        assert float(before.abs().max()) > 0
        ret["mode"] = dp.reduce_mode(flat.flat_grad)
        flat.all_reduce_grads(force=True)          # world == 1 returns early without `force`: issue the collective
        torch.cuda.synchronize()
        ret["same"] = bool(torch.equal(before, flat.flat_grad))
        # the same op straight through torch.distributed, on the whole flat gradient (68 K floats at cfg2's width)
        direct = before.clone()
        work = dist.all_reduce(direct, op=dist.ReduceOp.AVG, async_op=True)
        work.wait()
        torch.cuda.synchronize()
        ret["direct_same"] = bool(torch.equal(before, direct))
        ret["numel"] = int(direct.numel())
        # the weighted form (LightpathGNN shards with different n_lut): sum of (w*g, w), then divide
        flat.all_reduce_grads(weight=torch.tensor(5.0, device=dev), force=True)
        torch.cuda.synchronize()
        ret["weighted_close"] = float((flat.flat_grad - before).abs().max()) <= 1e-6 * float(before.abs().max())
        scale = dp.loss_scale(5, dev)
        ret["scale"] = float(scale)
        ret["backend"] = dist.get_backend()
    finally:
        dist.destroy_process_group()


def test_rccl_single_rank_group_runs_the_avg_all_reduce():
    """The nccl (= RCCL) backend at world size 1 on the one GPU this box has: the library loads, a communicator is
    created, ``dp.reduce_mode`` == "avg", and the flat-gradient ``all_reduce(AVG)`` IS ISSUED (``force=True``: without it
    ``all_reduce_grads`` returns early at world size 1) through ``FlatModel`` and directly, bit-equal to its input on a
    single rank; the weighted (sum, divide) form likewise.  What stays unexercised until a multi-GPU node
    runs it is the exchange between ranks."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rccl_world1_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
        got = dict(ret)
    assert got["backend"] == "nccl" and got["mode"] == "avg"
    assert got["same"] and got["direct_same"] and got["weighted_close"] and got["numel"] > 1000
    assert abs(got["scale"] - 1.0) < 1e-6


def _rccl_capture_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import copy
        import gnn_qot_estimation_amd as q
        from gnn_qot_estimation_amd import dp, harness as Hn, synthetic as S
        from gnn_qot_estimation_amd.loader import GraphLoader
        # (1) the flat-gradient all-reduce of FlatModel inside a captured HIP graph, replayed three times
        full, model = _build("topo", dev)
        flat = dp.FlatModel(model)
        flat.flat_grad.copy_(torch.arange(flat.numel, device=dev, dtype=torch.float32) / 7.0)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            flat.all_reduce_grads(force=True)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            flat.all_reduce_grads(force=True)
        same = []
        for k in range(3):
            flat.flat_grad.mul_(1.5)
            want = flat.flat_grad.clone()
            g.replay()
            torch.cuda.synchronize()
            same.append(bool(torch.equal(flat.flat_grad, want)))
        ret["captured_all_reduce"] = same
        # (2) replayed training with the collective INSIDE every captured step == the same training without it
        torch.manual_seed(0)
        b = S.topological_batch(2, 96, n=12, e=30)
        shard = q.PackedGraphs.from_batch(b).to_device(dev)
        base = q.TopologicalGNN(12, 16, 3, 4, dropout_p=0.0).to(dev)
        out = {}
        for coll in (False, True):
            m = copy.deepcopy(base)
            fl = dp.FlatModel(m)
            opt = dp.FusedSGD(fl, lr=0.05, momentum=0.9, device_lr=True)
            opt.lr = 0.05
            rp = Hn.StepReplayer(m, "topological", 3, dev, fl, opt, collective=coll)
            for epoch in range(4):
                Hn.run_epoch(m, shard, range(96), kind="topological", batch_size=32, out_dim=3, device=dev,
                             criterion=torch.nn.SmoothL1Loss(), flat=fl, opt=opt, replayer=rp)
            torch.cuda.synchronize()
            out[coll] = (fl.flat_param.clone(), len(rp.graphs))
        ret["graphs"] = (out[False][1], out[True][1])
        ret["params_equal"] = bool(torch.equal(out[False][0], out[True][0]))
        ret["params_moved"] = float((out[True][0] - dp.FlatModel(copy.deepcopy(base)).flat_param).abs().max())
    finally:
        dist.destroy_process_group()


def test_rccl_all_reduce_inside_captured_steps():
    """RCCL takes part in stream capture (r04): ``FlatModel.all_reduce_grads(force=True)`` on the single-rank nccl group
    captured in a HIP graph and replayed three times is bit-equal to its input each time; and the replayed train steps of
    ``harness.StepReplayer(collective=True)`` -- forward, backward, pack, all-reduce, update in ONE graph -- leave exactly the
    parameters the collective-free replay leaves (AVG over one rank is the identity).  The child is a fresh process of a
    parent that has not touched the GPU."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rccl_capture_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
        got = dict(ret)
    assert got["captured_all_reduce"] == [True, True, True], got
    assert got["graphs"][0] >= 3 and got["graphs"][1] == got["graphs"][0], got
    assert got["params_equal"] and got["params_moved"] > 1e-4, got


def test_bench_forced_data_parallel_step_holds_the_collective_in_its_graph():
    """``BENCH_FORCE_DP=1 python bench.py``: the N > 1 step (pack, flat-gradient all-reduce, plain update) on a
    single-rank nccl group -- the line must say that the collective sits inside the captured multi-step graph."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_FORCE_DP="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline",
                          "--no-lightpath", "--no-reference-scale"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    cfg = line["config"]
    assert cfg["collective_in_graph"] is True and cfg["collective_backend"] == "nccl", cfg
    assert cfg["steps_per_graph"] == 4 and cfg["all_reduce_us"] > 0
    assert cfg["final_loss"] == cfg["final_loss"]


def test_bench_two_ranks_on_one_gpu_over_gloo_reports_the_collective():
    """``python bench.py --gpus 2`` (self-spawned ranks, SURVEY 8(e) / the driver's launch contract) rehearsed on the
    one-GPU box: ``BENCH_BACKEND=gloo BENCH_SHARE_GPU=1`` puts both ranks on cuda:0.  The children are fresh processes
    of a parent (bench.py) that never touches the GPU.  Checks the JSON line: two ranks took part in a real exchange
    (``collective_world_size``), the flat-gradient all-reduce was timed, the loss is finite, weak scaling doubles the
    graphs per step.  No scaling number is expected from this."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["collective_world_size"] == 2 and cfg["collective_backend"] == "gloo"
    assert cfg["global_batch"] == 2 * cfg["graphs_per_gpu"] and line["scaling"] == "weak"
    assert cfg["all_reduce_us"] is not None and cfg["all_reduce_us"] > 0 and cfg["all_reduce_floats"] > 1000
    assert cfg["collective_in_graph"] is False and "gloo" in cfg["collective_capture_error"]     # gloo: the two-graph fallback
    assert cfg["final_loss"] == cfg["final_loss"] and abs(cfg["final_loss"]) < 1e3
    assert line["value"] > 0 and line["steps"] == 3
