"""Train / evaluate loops with the behaviour of the reference scripts, driving the HIP models.

Counterpart of ``topological_training/train.py``, ``lightpath_training/train.py`` and the two
``test.py``.  What is kept from the reference: the 70/15/15 split without
shuffling (train.py:29-36), one 10 % chunk of the training range per epoch (train.py:79-90),
SGD(lr=0.1, momentum=0.9) with StepLR(step_size=10, gamma=0.5) (train.py:66-67), SmoothL1 loss
(train.py:69), sample-weighted epoch loss and uniform-average R2 (train.py:119-131), early stopping on
validation R2 with patience 10 and a best-model file (train.py:168-178), the checkpoint dictionary
(train.py:196-209), LUT target selection and batch skipping for LightpathGNN
(lightpath_training/train.py:112-135), min-max descaling and per-output R2 / MSE at test time
(topological_training/test.py:88-108).

What is different, because the device is an MI355X and not a laptop CPU: batches come from
``GraphLoader`` (pinned, prefetched), nothing in the batch loop reads a value back to the host -- the
loss sum and the R2 sufficient statistics accumulate on the device and are fetched once per epoch --
and the optimizer is one fused kernel over the flat parameter buffer.  With ``torch.distributed``
initialised every rank takes its share of each batch range and the statistics are summed over ranks.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .dp import FlatModel, FusedSGD, graph_range, loss_scale
from .loader import GraphLoader, PackedGraphs

# data constants of the reference's label scaling (constants.py:8-12), used at test.py:95-99
TARGET_RANGES = {
    "osnr": {"min": 12.47, "max": 33.49},
    "snr": {"min": 8.96, "max": 29.98},
    "ber": {"min": 1.70e-12, "max": 1.98e-2},
}


def split_ranges(total: int) -> Tuple[range, range, range]:
    """70 / 15 / 15 split in dataset order (train.py:29-36)."""
    tr = int(total * 0.7)
    va = int(total * 0.15)
    return range(0, tr), range(tr, tr + va), range(tr + va, total)


def epoch_chunk(epoch: int, train_len: int, fraction: float = 0.10) -> range:
    """The slice of the training range epoch ``epoch`` sees (train.py:79-90)."""
    num_chunks = int(1 / fraction)
    size = train_len // num_chunks
    start = (epoch % num_chunks) * size
    return range(start, min(start + size, train_len))


def step_lr(base_lr: float, epoch: int, step_size: int = 10, gamma: float = 0.5) -> float:
    """Learning rate in effect during ``epoch`` under StepLR stepped once per epoch (train.py:67,181)."""
    return base_lr * gamma ** (epoch // step_size)


class RegressionStats:
    """Streaming sufficient statistics for R2 and MSE, held on the device.

    ``sklearn.metrics.r2_score(..., multioutput=...)`` (train.py:127, test.py:106) needs only
    n, sum(y), sum(y^2) and sum((y - yhat)^2) per output; fp64 accumulators, one host read at the end.
    """

    def __init__(self, outputs: int, device):
        self.buf = torch.zeros(3 * outputs + 2, dtype=torch.float64, device=device)   # sy | syy | sse | n | loss*n
        self.o = outputs

    def update(self, y: torch.Tensor, yhat: torch.Tensor, loss: Optional[torch.Tensor] = None):
        y = y.detach().to(torch.float64)
        d = y - yhat.detach().to(torch.float64)
        n = y.shape[0]
        o = self.o
        self.buf[:o] += y.sum(0)
        self.buf[o:2 * o] += (y * y).sum(0)
        self.buf[2 * o:3 * o] += (d * d).sum(0)
        self.buf[3 * o] += n
        if loss is not None:
            self.buf[3 * o + 1] += loss.detach().to(torch.float64) * n

    def all_reduce(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)

    def result(self, scale: Optional[torch.Tensor] = None) -> Dict[str, object]:
        """``scale`` (per-output max-min) reports MSE in descaled units (test.py:93-108); R2 is invariant."""
        b = self.buf.cpu()
        o = self.o
        n = float(b[3 * o])
        if n == 0:
            nan = float("nan")
            return {"n": 0, "r2_raw": [nan] * o, "r2": nan, "mse_raw": [nan] * o, "loss_sum": 0.0}
        sy, syy, sse = b[:o], b[o:2 * o], b[2 * o:3 * o]
        sst = syy - sy * sy / n
        # sklearn: constant target -> 1.0 if perfect else 0.0
        r2 = torch.where(sst > 0, 1.0 - sse / sst.clamp_min(1e-300),
                         torch.where(sse == 0, torch.ones_like(sse), torch.zeros_like(sse)))
        mse = sse / n
        if scale is not None:
            mse = mse * scale.to(torch.float64).cpu() ** 2
        return {"n": int(n), "r2_raw": r2.tolist(), "r2": float(r2.mean()), "mse_raw": mse.tolist(),
                "loss_sum": float(b[3 * o + 1])}


def _targets_topological(model, data, out_dim):
    out = model(data)
    return out, data.y.view(-1, out_dim)


def _targets_lightpath(model, data, out_dim):
    out, lut_batch = model(data)            # ValueError when the batch has no LUT node (models.py:35-36)
    return out, data.y.view(-1, out_dim)[lut_batch]


_KINDS = {"topological": _targets_topological, "lightpath": _targets_lightpath}


@dataclass
class History:
    loss: List[float] = field(default_factory=list)
    val_loss: List[float] = field(default_factory=list)
    r2: List[float] = field(default_factory=list)
    val_r2: List[float] = field(default_factory=list)
    skipped_graphs: int = 0
    stopped_early: bool = False
    best_val_r2: float = float("-inf")
    epochs_run: int = 0

    def dump(self, directory: str):
        """loss_history.json etc. as the reference writes them (train.py:213-226)."""
        os.makedirs(directory, exist_ok=True)
        for name, vals in (("loss_history", self.loss), ("val_loss_history", self.val_loss),
                           ("r2_history", self.r2), ("val_r2_history", self.val_r2)):
            with open(os.path.join(directory, name + ".json"), "w") as f:
                json.dump(vals, f)


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _graph_costs(dataset, idx: Sequence[int]):
    """Per-graph edge counts of ``idx`` when the dataset is a pre-tensorised shard (else None: split by count)."""
    if isinstance(dataset, PackedGraphs) and len(idx):
        e = (dataset.edge_ptr[1:] - dataset.edge_ptr[:-1])
        return e[torch.as_tensor(list(idx), dtype=torch.long)].to(torch.float64)
    return None


def _local_batches(idx: Sequence[int], batch_size: int, rank: int, world: int, costs=None) -> List[List[int]]:
    """One entry per GLOBAL batch of ``batch_size`` graphs: this rank's contiguous share of it (possibly empty
    when a trailing batch holds fewer graphs than there are ranks).  Every rank gets the same number of
    entries, so every rank issues the same sequence of collectives, and a step is the single-process step
    over the same global batch.  ``costs`` (per-graph edge counts, aligned with ``idx``): the cut inside each global
    batch evens out edges instead of graph counts."""
    out: List[List[int]] = []
    for b0 in range(0, len(idx), batch_size):
        n = min(batch_size, len(idx) - b0)
        lo, hi = graph_range(n, rank, world, costs=None if costs is None or world == 1 else costs[b0:b0 + n])
        out.append(list(idx[b0 + lo:b0 + hi]))
    return out


class StepReplayer:
    """HIP-graph replay of whole steps over the cached batches of an HBM-resident shard.

    The reference revisits the same unshuffled chunks for 35 epochs (``train.py:79-95``).  With the shard in
    HBM and the batch objects cached, a step's inputs are the SAME device tensors on every visit, so the
    whole step (forward, loss, backward, optimizer update, statistics) is captured once per batch and
    replayed afterwards: no Python, no launches, no allocation on later visits.  First visit of a batch
    runs eagerly (it also builds and caches the graph index), the second is captured, later ones replay.
    All captures share one memory pool (steps never overlap), the learning rate lives in device memory
    (``FusedSGD(device_lr=True)``), dropout draws come from the device-side counter.  Batches whose
    forward raises (``ValueError``: no LUT node) are remembered and skipped.
    Data parallel (r04): RCCL takes part in stream capture, so with ``collective=True`` (set by ``fit`` for a TopologicalGNN
    run on more than one rank over the nccl backend) the captured step holds forward, backward, the pack of the gradients,
    ONE all-reduce of the flat gradient and the update -- the N > 1 step is the N = 1 step plus that exchange.  Every rank
    must visit the same sequence of batches (``_local_batches`` guarantees it).  LightpathGNN under data parallelism keeps
    the eager loop: its BatchNorm exchange and the LUT skip are decided per global batch on the host.
    """

    def __init__(self, model, kind: str, out_dim: int, device, flat: Optional[FlatModel], opt: Optional[FusedSGD],
                 collective: bool = False):
        self.model, self.kind, self.out_dim, self.device = model, kind, out_dim, device
        self.flat, self.opt = flat, opt
        self.collective = bool(collective)
        self._scale: Dict[int, torch.Tensor] = {}      # per batch object: this rank's share of the global mean loss
        self.pool = torch.cuda.graph_pool_handle()
        self.graphs: Dict[Tuple[int, bool], object] = {}
        self.visits: Dict[Tuple[int, bool], int] = {}
        self.skip: set = set()
        self.stats = {True: RegressionStats(out_dim, device), False: RegressionStats(out_dim, device)}
        self._loss = torch.zeros((), dtype=torch.float32, device=device)
        self._keep: List[object] = []          # batch objects whose ids key the tables

    def _step(self, data, training: bool):
        from . import functional as QF
        fwd = _KINDS[self.kind]
        if training:
            # The step differentiates with respect to fresh leaves that alias the parameters
            # (functional_call), not the Parameters themselves: a Parameter's gradient accumulator has the
            # stream affinity of whoever created it, and one that user code keeps alive (any earlier
            # forward whose output is still referenced) would drag a cross-stream sync into the capture
            # (observed as a crash in capture_end).
            names = [n for n, p in self.model.named_parameters() if p.requires_grad]
            leaves = {n: p.detach().requires_grad_(True) for n, p in zip(names, self.flat.params)}
            functional = lambda d: torch.func.functional_call(self.model, leaves, (d,))
            out, y = fwd(functional, data, self.out_dim)
            _, g = QF.smooth_l1_loss_and_grad(out, y, loss_out=self._loss)
            if self.collective:
                # this rank's share of the global mean loss (dp.loss_scale: one host-built scalar per batch object, made
                # on the batch's first -- eager -- visit; a capture must not copy from pageable host memory)
                sc = self._scale.get(id(data))
                if sc is None:
                    sc = self._scale[id(data)] = loss_scale(y.shape[0], self.device)
                g = g * sc
            grads = torch.autograd.grad(out, list(leaves.values()), g, allow_unused=True)
            if self.collective:
                torch.cat([(gr if gr is not None else torch.zeros_like(p)).reshape(-1)
                           for gr, p in zip(grads, self.flat.params)], out=self.flat.flat_grad)
                self.flat.all_reduce_grads(force=True)
                self.opt.step()
            else:
                self.opt.step(grads=list(grads))     # the pack into the flat gradient rides in the update kernel
        else:
            with torch.no_grad():
                out, y = fwd(self.model, data, self.out_dim)
                QF.smooth_l1_loss_and_grad(out, y, loss_out=self._loss)
        self.stats[training].update(y, out, self._loss)

    def run(self, data, training: bool) -> bool:
        """One step on ``data``; returns False when the batch is (remembered as) skipped."""
        key = (id(data), training)
        if key in self.skip:
            return False
        g = self.graphs.get(key)
        if g is not None:
            g.replay()
            return True
        seen = self.visits.get(key, 0)
        self.model.train(training)
        if seen == 0 or (training and self.opt.steps == 0):
            try:
                self._step(data, training)
            except ValueError:
                self.skip.add(key)
                self._keep.append(data)
                return False
            self.visits[key] = 1
            self._keep.append(data)
            return True
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        # (a captured collective: thread-local capture mode -- the process group's watchdog thread polls its events meanwhile)
        mode = dict(capture_error_mode="thread_local") if self.collective else {}
        with torch.cuda.graph(g, pool=self.pool, **mode):
            self._step(data, training)
        self.graphs[key] = g
        g.replay()                      # capture records, it does not execute
        return True


def run_epoch(model, dataset, indices: range, *, kind: str, batch_size: int, out_dim: int, device,
              criterion, flat: Optional[FlatModel] = None, opt: Optional[FusedSGD] = None,
              replayer: Optional[StepReplayer] = None) -> Dict[str, object]:
    """One pass over ``indices``; trains when ``opt`` is given, else evaluates under ``no_grad``."""
    fwd = _KINDS[kind]
    rank, world = _rank_world()
    # an HBM-resident shard keeps its batch objects (and the graph index the model attaches to them):
    # the chunks repeat every few epochs, so later visits do no graph preparation at all
    resident = isinstance(dataset, PackedGraphs) and dataset.device is not None
    loader = GraphLoader(dataset, batch_size, shuffle=False, device=device, cache_batches=resident,
                         batches=_local_batches(indices, batch_size, rank, world, _graph_costs(dataset, indices)))
    stats = RegressionStats(out_dim, device)
    skipped = 0
    training = opt is not None
    model.train(training)
    shares_ok = True
    if world > 1 and replayer is not None and replayer.collective:
        # a replayed step holds the collective: every rank must hold graphs of every global batch (the split is a pure
        # function of the indices, so every rank reaches the same verdict without talking)
        costs = _graph_costs(dataset, indices)
        for b0 in range(0, len(indices), batch_size):
            nb = min(batch_size, len(indices) - b0)
            for r in range(world):
                lo, hi = graph_range(nb, r, world, costs=None if costs is None else costs[b0:b0 + nb])
                shares_ok = shares_ok and hi > lo
    if replayer is not None and resident and (world == 1 or (replayer.collective and training and shares_ok)):
        st = replayer.stats[training]
        st.buf.zero_()
        for data in loader:
            if not replayer.run(data, training):
                skipped += data.num_graphs
        res = st.result()
        res["avg_loss"] = res["loss_sum"] / max(len(indices), 1)
        res["skipped"] = skipped
        return res
    coupled = world > 1 and training          # ranks meet in collectives inside every step
    with torch.set_grad_enabled(training):
        for data in loader:
            if training:
                flat.zero_grad()
            if data is None and not coupled:
                continue
            if coupled and kind == "lightpath":
                # Ranks are coupled inside forward/backward (global BatchNorm statistics), and whether a batch
                # is skipped is a property of the GLOBAL batch, as in the single-process reference: skipped only
                # when NO rank holds a LUT node.  One flag all-reduce (one host read) per batch.
                has = 0
                if data is not None:
                    has = int(bool((data.x[:, getattr(model, "is_lut_index", None)] == 1.0).any()))
                flags = torch.tensor([has, int(data is None)], dtype=torch.int32, device=device)
                dist.all_reduce(flags, op=dist.ReduceOp.MAX)
                any_lut, any_empty = (int(v) for v in flags.tolist())
                if any_empty:
                    # fewer graphs than ranks in a trailing batch: a rank without nodes cannot take part in the
                    # BatchNorm exchange; all ranks skip it (documented deviation, at most world-1 graphs per epoch)
                    skipped += 0 if data is None else data.num_graphs
                    continue
                model.allow_empty_lut = True
                try:
                    if not any_lut:
                        # the reference raises AFTER conv1/norm1 ran (models.py:30-36): running statistics move
                        with torch.no_grad():
                            fwd(model, data, out_dim)
                        skipped += data.num_graphs
                        continue
                    out, y = fwd(model, data, out_dim)
                finally:
                    model.allow_empty_lut = False
                loss = criterion(out, y) if y.shape[0] else out.sum() * 0.0
                (loss * loss_scale(y.shape[0], device)).backward()
                flat.all_reduce_grads()
                opt.step()
                if y.shape[0]:
                    stats.update(y, out, loss)
                continue
            if data is None:
                # empty share of a trailing batch: zero contribution, same collectives as the other ranks
                loss_scale(0, device)
                flat.all_reduce_grads()
                opt.step()
                continue
            try:
                out, y = fwd(model, data, out_dim)
            except ValueError:
                skipped += data.num_graphs          # lightpath_training/train.py:118-121
                continue
            loss = criterion(out, y)
            if training:
                if world > 1:
                    # this rank's share of the global mean loss, then a plain average of gradients
                    (loss * loss_scale(y.shape[0], device)).backward()
                    flat.all_reduce_grads()
                else:
                    loss.backward()
                opt.step()
            stats.update(y, out, loss)
    stats.all_reduce()
    res = stats.result()
    # the reference divides by len(loader.dataset), skipped graphs included (train.py:119)
    res["avg_loss"] = res["loss_sum"] / max(len(indices), 1)
    if world > 1:
        sk = torch.tensor([skipped], dtype=torch.int64, device=device)
        dist.all_reduce(sk, op=dist.ReduceOp.SUM)
        skipped = int(sk.item())
    res["skipped"] = skipped
    return res


def fit(model, dataset, *, kind: str = "topological", batch_size: int = 512, num_epochs: int = 35,
        patience: int = 10, lr: float = 0.1, momentum: float = 0.9, step_size: int = 10, gamma: float = 0.5,
        chunk_fraction: float = 0.10, output_dim: int = 3, device="cuda", best_path: Optional[str] = None,
        log: Callable[[str], None] = print, replay: Optional[bool] = None) -> History:
    """The training script's main loop (train.py:24-182) on a dataset object indexable by graph.

    ``replay`` (default: on for an HBM-resident shard in a single process): steps over cached batches are
    captured as HIP graphs on their second visit and replayed afterwards (``StepReplayer``)."""
    device = torch.device(device)
    model.to(device)
    tr, va, _ = split_ranges(len(dataset))
    flat = FlatModel(model)
    flat.broadcast_params()
    rank, world = _rank_world()
    resident = isinstance(dataset, PackedGraphs) and dataset.device is not None
    # more than one rank: replayed steps hold the RCCL all-reduce (nccl backend, TopologicalGNN; StepReplayer docstring)
    dp_ok = world == 1 or (kind == "topological" and dist.get_backend() == "nccl")
    use_replay = (resident and dp_ok) if replay is None else (bool(replay) and resident and dp_ok)
    opt = FusedSGD(flat, lr=lr, momentum=momentum, device_lr=use_replay)
    replayer = StepReplayer(model, kind, output_dim, device, flat, opt, collective=world > 1) if use_replay else None
    criterion = torch.nn.SmoothL1Loss()
    hist = History()
    counter = 0
    for epoch in range(num_epochs):
        chunk = epoch_chunk(epoch, len(tr), chunk_fraction)
        if epoch == 0:
            log(f"Training model with {len(chunk)} samples and validating with {len(va)} samples.")
        opt.lr = step_lr(lr, epoch, step_size, gamma)
        t = run_epoch(model, dataset, range(tr[0] + chunk[0], tr[0] + chunk[-1] + 1) if len(chunk) else range(0),
                      kind=kind, batch_size=batch_size, out_dim=output_dim, device=device, criterion=criterion,
                      flat=flat, opt=opt, replayer=replayer)
        v = run_epoch(model, dataset, va, kind=kind, batch_size=batch_size, out_dim=output_dim, device=device,
                      criterion=criterion, replayer=replayer)
        hist.loss.append(t["avg_loss"]); hist.r2.append(t["r2"])
        hist.val_loss.append(v["avg_loss"]); hist.val_r2.append(v["r2"])
        hist.skipped_graphs += t["skipped"]
        hist.epochs_run = epoch + 1
        log(f"Epoch {epoch + 1}, Loss: {t['avg_loss']:.4f}, R2 Score: {t['r2']:.4f}, "
            f"Val Loss: {v['avg_loss']:.4f}, Val R2 Score: {v['r2']:.4f}")
        if v["r2"] > hist.best_val_r2:
            hist.best_val_r2 = v["r2"]
            counter = 0
            if best_path and rank == 0:
                torch.save(model.state_dict(), best_path)
        else:
            counter += 1
            if counter >= patience:
                log("Early stopping triggered.")
                hist.stopped_early = True
                break
    return hist


def evaluate(model, dataset, indices: Optional[Sequence[int]] = None, *, kind: str = "topological",
             batch_size: int = 512, output_dim: int = 3, device="cuda",
             output_keys: Sequence[str] = ("osnr", "snr", "ber"),
             target_ranges: Dict[str, Dict[str, float]] = TARGET_RANGES, return_predictions: bool = False):
    """test.py's metric block: per-output R2 and MSE on min-max descaled values (test.py:76-121).

    ``return_predictions``: also return ``(y_true_descaled, y_pred_descaled, skipped_graphs)`` -- the arrays test.py
    writes to ``y_true_descaled.json`` / ``y_pred_descaled.json`` (test.py:92-103,129-136), in dataset order; they stay
    on the device until the loop is over (one host copy)."""
    device = torch.device(device)
    model.to(device)
    idx = range(len(dataset)) if indices is None else indices
    fwd = _KINDS[kind]
    stats = RegressionStats(output_dim, device)
    model.eval()
    rank, world = _rank_world()
    loader = GraphLoader(dataset, batch_size, shuffle=False, device=device,
                         batches=_local_batches(idx, batch_size, rank, world, _graph_costs(dataset, idx)))
    kept, skipped = [], 0
    # Under data parallelism a rank sees only its share of every global batch.  The reference decides "no LUT node in the
    # batch" (lightpath_training/test.py:82-85) on the WHOLE batch, so a share without LUT rows contributes zero rows
    # (allow_empty_lut) and the skip is decided afterwards from the row counts of all ranks, per global batch.
    sharded_lut = world > 1 and kind == "lightpath" and hasattr(model, "allow_empty_lut")
    shares = []                                        # (global batch, rows this rank produced, graphs in its share)
    with torch.no_grad():
        for b, data in enumerate(loader):
            if data is None:
                continue
            if sharded_lut:
                model.allow_empty_lut = True
                try:
                    out, y = fwd(model, data, output_dim)
                finally:
                    model.allow_empty_lut = False
                shares.append((b, int(y.shape[0]), int(data.num_graphs)))
                if y.shape[0] == 0:
                    continue
                stats.update(y, out)
                if return_predictions:
                    kept.append((b, y.detach().clone(), out.detach().clone()))
                continue
            try:
                out, y = fwd(model, data, output_dim)
            except ValueError:
                skipped += data.num_graphs            # lightpath_training/test.py:82-85
                continue
            stats.update(y, out)
            if return_predictions:
                kept.append((b, y.detach().clone(), out.detach().clone()))
    if sharded_lut:
        everyone = [None] * world
        dist.all_gather_object(everyone, shares)
        rows, graphs = {}, {}
        for per_rank in everyone:
            for b, r, g in per_rank:
                rows[b] = rows.get(b, 0) + r
                graphs[b] = graphs.get(b, 0) + g
        dead = {b for b, r in rows.items() if r == 0}  # global batches without any LUT node: skipped as a whole
        skipped = sum(g for b, r, g in shares if b in dead)
    stats.all_reduce()
    keys = list(output_keys)[:output_dim]
    scale = torch.tensor([target_ranges[k]["max"] - target_ranges[k]["min"] for k in keys], dtype=torch.float64)
    res = stats.result(scale)
    metrics = {k.upper(): {"R2": res["r2_raw"][i], "Test_MSE": res["mse_raw"][i]} for i, k in enumerate(keys)}
    if not return_predictions:
        return metrics
    parts = [(b, rank, y.cpu(), o.cpu()) for b, y, o in kept]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, (parts, skipped))
        parts = sorted((p for ps, _ in gathered for p in ps), key=lambda t: (t[0], t[1]))
        skipped = sum(sk for _, sk in gathered)
    lo = torch.tensor([target_ranges[k]["min"] for k in keys], dtype=torch.float64)
    cat = lambda j: (torch.cat([p[j].double() for p in parts]) if parts else torch.zeros(0, output_dim, dtype=torch.float64))
    y_true = cat(2) * scale + lo                      # min_max_descale (test.py:12-13)
    y_pred = cat(3) * scale + lo
    return metrics, y_true, y_pred, skipped


def next_model_path(root_dir: str) -> Tuple[str, int]:
    """``model_<k>.pth`` with k = 1 + the largest index present (train.py:185-195)."""
    os.makedirs(root_dir, exist_ok=True)
    idx = [int(n.split("_")[1].split(".")[0]) for n in os.listdir(root_dir) if n.startswith("model_")]
    k = max(idx) + 1 if idx else 0
    return os.path.join(root_dir, f"model_{k}.pth"), k


def save_checkpoint(path: str, model, model_params: Dict[str, object]):
    """The dictionary the reference test scripts expect (train.py:196-209, test.py:47-70)."""
    torch.save({"model_state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                "model_params": dict(model_params)}, path)


def load_checkpoint(path: str):
    """Reads a checkpoint written by this module or by the reference (tensors + plain containers only)."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    return ck["model_state_dict"], ck["model_params"]
