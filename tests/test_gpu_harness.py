"""Train / eval harness end to end on the GPU: the loop of
topological_training/train.py and lightpath_training/train.py on small synthetic datasets."""
import numpy as np
import pytest
import torch

import gnn_qot_estimation_amd as q
from gnn_qot_estimation_amd import harness as Hn
from gnn_qot_estimation_amd import synthetic as S
from oracle import sparse as osp

pytestmark = pytest.mark.gpu


def _topological_dataset(k):
    out = []
    for g in range(k):
        b = S.topological_batch(2, 1, n=12, e=30, first_graph=g)
        y = b.edge_attr[:, :3].mean(0, keepdim=True)        # a target the model can learn
        out.append(q.Data(edge_index=b.edge_index, edge_attr=b.edge_attr, node_ids=b.node_ids, y=y, num_nodes=12))
    return out


def test_fit_topological_learns_and_writes_reference_artifacts(tmp_path):
    torch.manual_seed(0)
    ds = q.PackedGraphs.from_data_list(_topological_dataset(600)).pin()
    model = q.TopologicalGNN(num_nodes=12, hidden_channels=16, out_channels=3, edge_dim=4, dropout_p=0.0)
    logs = []
    best = str(tmp_path / "best_model.pth")
    hist = Hn.fit(model, ds, kind="topological", batch_size=32, num_epochs=12, patience=10, chunk_fraction=0.5,
                  lr=0.05, best_path=best, log=logs.append)
    assert hist.epochs_run == len(hist.loss) == len(hist.val_r2) >= 1
    assert logs[0].startswith("Training model with 210 samples and validating with 90 samples.")
    assert np.isfinite(hist.loss).all() and np.isfinite(hist.val_loss).all()
    assert hist.val_loss[-1] < 0.5 * hist.val_loss[0]
    # best-model file holds a plain state_dict that the oracle model loads strictly
    sd = torch.load(best, map_location="cpu", weights_only=True)
    ref = osp.TopologicalGNN(num_nodes=12, hidden_channels=16, out_channels=3, edge_dim=4, dropout_p=0.0)
    ref.load_state_dict(sd, strict=True)
    # test.py: checkpoint dictionary -> rebuilt model -> per-output metrics
    path, _ = Hn.next_model_path(str(tmp_path / "models"))
    params = {"num_nodes": 12, "hidden_channels": 16, "output_dim": 3, "edge_dim": 4, "FEATURES": ["f"] * 4}
    Hn.save_checkpoint(path, model, params)
    sd2, got = Hn.load_checkpoint(path)
    m2 = q.TopologicalGNN(num_nodes=got["num_nodes"], hidden_channels=got["hidden_channels"],
                          out_channels=got["output_dim"], edge_dim=got["edge_dim"], dropout_p=0.0)
    m2.load_state_dict(sd2)
    _, _, te = Hn.split_ranges(len(ds))
    res = Hn.evaluate(m2, ds, te, batch_size=32)
    assert set(res) == {"OSNR", "SNR", "BER"} and all(np.isfinite(v["R2"]) for v in res.values())
    # the same test range through the oracle model on the CPU gives the same metrics
    graphs = [ds[i] for i in te]
    b = q.Batch.from_data_list(graphs)
    ref.load_state_dict(sd2)
    ref.eval()
    with torch.no_grad():
        pred = ref(b)
    st = Hn.RegressionStats(3, "cpu")
    st.update(b.y.view(-1, 3), pred)
    want = st.result()["r2_raw"]
    for i, k in enumerate(("OSNR", "SNR", "BER")):
        assert abs(res[k]["R2"] - want[i]) < 1e-3


def test_fit_lightpath_skips_batches_without_lut():
    torch.manual_seed(0)
    lp = S.lightpath_batch(64)
    graphs = []
    for g in range(64):
        s = q.shard_graphs(lp, g, 64)
        x = s.x.clone()
        if 8 <= g < 12:
            x[:, 1] = 0.0            # one whole batch (batch_size 4) without a LUT node
        graphs.append(q.Data(x=x, edge_index=s.edge_index, y=s.y, num_nodes=s.num_nodes))
    model = q.LightpathGNN(in_channels=5, hidden_channels=8, output_dim=3, is_lut_index=1, dropout_p=0.5)
    hist = Hn.fit(model, graphs, kind="lightpath", batch_size=4, num_epochs=2, chunk_fraction=0.5, log=lambda s: None)
    assert hist.epochs_run == 2 and hist.skipped_graphs == 4          # epoch 0 covers graphs 0..21
    assert np.isfinite(hist.loss).all()
    res = Hn.evaluate(model, graphs, kind="lightpath", batch_size=4)
    assert set(res) == {"OSNR", "SNR", "BER"}


def test_hbm_resident_shard_batches_and_cache():
    """``PackedGraphs.to_device``: batches are views of the resident shard, equal to the collated ones;
    with ``cache_batches`` the same batch object (graph index attached) comes back on the next epoch."""
    from gnn_qot_estimation_amd.loader import GraphLoader
    graphs = _topological_dataset(50)
    shard = q.PackedGraphs.from_data_list(graphs).to_device("cuda")
    ld = GraphLoader(shard, batch_size=16, device="cuda", cache_batches=True)
    first = list(ld)
    assert [b.num_graphs for b in first] == [16, 16, 16, 2]
    for k, b in enumerate(first):
        ref = q.Batch.from_data_list(graphs[16 * k:16 * k + 16])
        for name in ("edge_index", "edge_attr", "node_ids", "y", "batch", "ptr"):
            assert torch.equal(getattr(b, name).cpu(), getattr(ref, name)), name
        assert b.uniform_node_ids == 12
    model = q.TopologicalGNN(num_nodes=12, hidden_channels=16, out_channels=3, edge_dim=4, dropout_p=0.0).cuda()
    out0 = model(first[1])
    again = list(ld)
    assert all(a is b for a, b in zip(first, again))
    assert "graph" in again[1]._qot_cache or len(again[1]._qot_cache) > 0      # index built once, kept
    assert torch.equal(model(again[1]), out0)
    # the training loop accepts the resident shard (fit -> run_epoch -> cached batches)
    hist = Hn.fit(model, shard, kind="topological", batch_size=8, num_epochs=3, chunk_fraction=0.5, log=lambda s: None)
    assert hist.epochs_run == 3 and np.isfinite(hist.loss).all()


def test_replayed_training_equals_eager_training():
    """HBM-resident shard: steps captured per cached batch and replayed must train exactly like the eager
    loop (same StepLR schedule through the device-side learning rate, same statistics), for both models;
    the lightpath run includes a batch without LUT nodes that stays skipped."""
    import copy
    torch.manual_seed(0)
    topo = q.PackedGraphs.from_data_list(_topological_dataset(240)).to_device("cuda")
    base = q.TopologicalGNN(num_nodes=12, hidden_channels=16, out_channels=3, edge_dim=4, dropout_p=0.0)
    runs = {}
    for replay in (False, True):
        m = copy.deepcopy(base)
        runs[replay] = Hn.fit(m, topo, kind="topological", batch_size=16, num_epochs=7, chunk_fraction=0.5, lr=0.05,
                              step_size=2, gamma=0.5, log=lambda s: None, replay=replay), m
    (h0, m0), (h1, m1) = runs[False], runs[True]
    np.testing.assert_allclose(h1.loss, h0.loss, rtol=2e-4)
    np.testing.assert_allclose(h1.val_loss, h0.val_loss, rtol=2e-4)
    np.testing.assert_allclose(h1.val_r2, h0.val_r2, rtol=1e-3, atol=1e-4)
    for a, b in zip(m0.parameters(), m1.parameters()):
        assert float((a - b).abs().max()) <= 2e-4 * max(float(b.abs().max()), 1e-3)

    lp = S.lightpath_batch(96)
    graphs = []
    for g in range(96):
        s = q.shard_graphs(lp, g, 96)
        x = s.x.clone()
        if 8 <= g < 12:
            x[:, 1] = 0.0
        graphs.append(q.Data(x=x, edge_index=s.edge_index, y=s.y, num_nodes=s.num_nodes))
    shard = q.PackedGraphs.from_data_list(graphs).to_device("cuda")
    lbase = q.LightpathGNN(in_channels=5, hidden_channels=8, output_dim=3, is_lut_index=1, dropout_p=0.0)
    lruns = {}
    for replay in (False, True):
        m = copy.deepcopy(lbase)
        lruns[replay] = Hn.fit(m, shard, kind="lightpath", batch_size=4, num_epochs=5, chunk_fraction=0.5,
                               log=lambda s: None, replay=replay)
    assert lruns[True].skipped_graphs == lruns[False].skipped_graphs > 0
    np.testing.assert_allclose(lruns[True].loss, lruns[False].loss, rtol=2e-4)
    np.testing.assert_allclose(lruns[True].val_loss, lruns[False].val_loss, rtol=2e-4)
