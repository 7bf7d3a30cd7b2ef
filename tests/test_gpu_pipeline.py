"""End to end over the rows either side of the hot path: network-status samples ->
``to_graph.store_graphs`` (.gpickle files) -> ``TopologicalDataset`` / ``LightpathDataset`` -> ``pack()`` ->
``save_shard`` / ``load_shard`` -> ``GraphLoader`` -> HIP model == CPU oracle on the same ``Data``; and the
``python -m gnn_qot_estimation_amd.train`` / ``.test`` entry points reproduce the reference scripts' artefacts
(``topological_training/train.py:183-227``, ``test.py:119-138``)."""
import json
import os

import pytest
import torch

import gnn_qot_estimation_amd as q
from gnn_qot_estimation_amd import dataset as DS
from gnn_qot_estimation_amd import to_graph as TG
from helpers import TOL, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["topological", "lightpath"])
def test_graph_files_to_shard_to_loader_to_hip_model_equals_oracle(tmp_path, cuda_device, kind):
    from oracle import sparse as O
    ns = TG.synthetic_network_status(12, seed=4)
    d = TG.store_graphs(ns, kind, str(tmp_path / kind))
    ds = DS.TopologicalDataset(d) if kind == "topological" else DS.LightpathDataset(d)
    assert len(ds) == 12
    shard_file = str(tmp_path / "shard.pt")
    DS.save_shard(shard_file, ds.pack(), {"kind": kind})
    shard, meta = DS.load_shard(shard_file)
    assert meta == {"kind": kind} and len(shard) == 12
    torch.manual_seed(0)
    if kind == "topological":
        assert ds.FEATURES == ["freq", "mod_order", "num_spans", "path_len"] and ds.edge_dim == 4
        ref = O.TopologicalGNN(75, 16, 3, 4, dropout_p=0.0).eval()
        hip = q.TopologicalGNN(75, 16, 3, 4, dropout_p=0.0)
    else:
        ref = O.LightpathGNN(5, 32, 3, ds.feature_indices["is_lut"], dropout_p=0.0).eval()
        hip = q.LightpathGNN(5, 32, 3, ds.feature_indices["is_lut"], dropout_p=0.0)
    hip.load_state_dict(ref.state_dict(), strict=True)
    hip.to(cuda_device).eval()
    cpu_batches = list(q.GraphLoader([ds[i] for i in range(12)], 5, device="cpu"))
    for mode in ("pinned", "resident"):
        src = shard.pin() if mode == "pinned" else shard.to_device(cuda_device)
        seen = 0
        for got, want in zip(q.GraphLoader(src, 5, device=cuda_device), cpu_batches):
            assert torch.equal(got.edge_index.cpu(), want.edge_index) and torch.equal(got.batch.cpu(), want.batch)
            with torch.no_grad():
                if kind == "topological":
                    assert torch.equal(got.edge_attr.cpu(), want.edge_attr) and torch.equal(got.y.cpu(), want.y)
                    assert rel_err(hip(got), ref(want)) <= TOL
                else:
                    assert torch.equal(got.x.cpu(), want.x)
                    (o_h, b_h), (o_r, b_r) = hip(got), ref(want)
                    assert torch.equal(b_h.cpu(), b_r) and rel_err(o_h, o_r) <= TOL
            seen += got.num_graphs
        assert seen == 12


@pytest.mark.parametrize("kind", ["topological", "lightpath"])
def test_train_and_test_entry_points_write_the_reference_artefacts(tmp_path, cuda_device, kind, capsys):
    from gnn_qot_estimation_amd import test as test_cli, train as train_cli
    from oracle import sparse as O
    ns = TG.synthetic_network_status(60, seed=8)
    data_dir = TG.store_graphs(ns, kind, str(tmp_path / f"networkx_graphs_{kind}"))
    root = str(tmp_path / f"{kind}_training")
    model_path = train_cli.main(["--kind", kind, "--data", data_dir, "--root", root, "--epochs", "3", "--batch-size", "16"])
    assert model_path.endswith("models/model_0.pth") and os.path.exists(os.path.join(root, "best_model.pth"))
    assert sorted(os.listdir(os.path.join(root, "loss_training_0"))) == [
        "loss_history.json", "r2_history.json", "val_loss_history.json", "val_r2_history.json"]
    assert len(json.load(open(os.path.join(root, "loss_training_0", "loss_history.json")))) == 3
    assert "Epoch 3, Loss:" in open(os.path.join(root, "model_logger.txt")).read()
    ck = torch.load(model_path, map_location="cpu", weights_only=True)
    want_keys = (["num_nodes", "hidden_channels", "output_dim", "edge_dim", "FEATURES"] if kind == "topological" else
                 ["in_channels", "hidden_channels", "output_dim", "NODE_FEATURES", "feature_indices"])
    assert list(ck["model_params"]) == want_keys
    # a second run takes the next index, as the reference does
    assert train_cli.main(["--kind", kind, "--data", data_dir, "--root", root, "--epochs", "1", "--batch-size", "16"]).endswith("model_1.pth")
    os.remove(os.path.join(root, "models", "model_1.pth"))
    folder = test_cli.main(["--kind", kind, "--data", data_dir, "--root", root, "--batch-size", "16"])
    name = "results_metrics.json" if kind == "topological" else "results.json"
    assert sorted(os.listdir(folder)) == sorted([name, "y_pred_descaled.json", "y_true_descaled.json"])
    assert ("_model_0" in folder) == (kind == "topological")
    metrics = json.load(open(os.path.join(folder, name)))
    assert list(metrics) == ["OSNR", "SNR", "BER"] and set(metrics["OSNR"]) == {"R2", "Test_MSE"}
    y_true = torch.tensor(json.load(open(os.path.join(folder, "y_true_descaled.json"))))
    y_pred = torch.tensor(json.load(open(os.path.join(folder, "y_pred_descaled.json"))))
    # the checkpoint loads into the oracle model; its predictions on the test split are the ones written (descaled)
    ds = DS.TopologicalDataset(data_dir) if kind == "topological" else DS.LightpathDataset(data_dir)
    p = ck["model_params"]
    if kind == "topological":
        ref = O.TopologicalGNN(p["num_nodes"], p["hidden_channels"], p["output_dim"], p["edge_dim"], dropout_p=0.0)
    else:
        ref = O.LightpathGNN(p["in_channels"], p["hidden_channels"], p["output_dim"], ds.feature_indices["is_lut"], dropout_p=0.0)
    ref.load_state_dict(ck["model_state_dict"], strict=True)
    ref.eval()
    test_idx = range(int(60 * 0.7) + int(60 * 0.15), 60)
    batch = q.Batch.from_data_list([ds[i] for i in test_idx])
    with torch.no_grad():
        out = ref(batch)
    lo = torch.tensor([12.47, 8.96, 1.70e-12]); span = torch.tensor([33.49 - 12.47, 29.98 - 8.96, 1.98e-2 - 1.70e-12])
    if kind == "topological":
        want_pred, want_true = out * span + lo, batch.y.view(-1, 3) * span + lo
    else:
        want_pred, want_true = out[0] * span + lo, batch.y[out[1]] * span + lo
    assert y_true.shape == want_true.shape == (len(test_idx), 3)
    assert rel_err(y_true, want_true) <= 1e-6 and rel_err(y_pred, want_pred) <= 1e-4
