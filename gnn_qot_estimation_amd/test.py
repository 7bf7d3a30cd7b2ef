"""``python -m gnn_qot_estimation_amd.test --kind topological|lightpath`` -- the reference's evaluation scripts
(``topological_training/test.py:19-140``, ``lightpath_training/test.py:19-151``) on the HIP models.

Loads the latest ``<root>/models/model_<k>.pth`` (ours or the reference's: same dictionary), rebuilds the model from
``model_params`` with ``dropout_p=0.0`` (test.py:58-64), evaluates the last 15 % of the dataset, prints per-output R2 /
MSE on descaled values and writes the reference's three JSON files: ``results_metrics.json`` (topological,
test.py:126-127; ``results.json`` for lightpath, lightpath test.py:137), ``y_true_descaled.json``,
``y_pred_descaled.json`` into ``<root>/results/results_<timestamp>[_model_<k>]``.
"""
from __future__ import annotations

import argparse
import json
import os
from datetime import datetime

import torch

from . import harness
from .train import open_dataset


def latest_model(models_dir: str):
    files = [f for f in os.listdir(models_dir) if f.startswith("model_") and f.endswith(".pth")] if os.path.isdir(models_dir) else []
    if not files:
        raise FileNotFoundError("No saved models found in the 'models' directory.")      # test.py:43-44
    k = max(int(f.split("_")[1].split(".")[0]) for f in files)
    return os.path.join(models_dir, f"model_{k}.pth"), k


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--kind", choices=["topological", "lightpath"], required=True)
    ap.add_argument("--data", default=None)
    ap.add_argument("--root", default=None)
    ap.add_argument("--batch-size", type=int, default=512)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--is-lut-index", type=int, default=None, help="default: from the dataset (lightpath test.py:59)")
    args = ap.parse_args(argv)

    from . import LightpathGNN, TopologicalGNN
    kind = args.kind
    root = args.root or f"{kind}_training"
    data = args.data or f"networkx_graphs_{kind}"
    device = torch.device(args.device)
    dataset, meta = open_dataset(kind, data, True, device)
    path, k = latest_model(os.path.join(root, "models"))
    print(f"Loading model from {path}")
    state, params = harness.load_checkpoint(path)
    if kind == "topological":
        model = TopologicalGNN(num_nodes=params["num_nodes"], hidden_channels=params["hidden_channels"],
                               out_channels=params["output_dim"], edge_dim=params["edge_dim"], dropout_p=0.0)
    else:
        lut = args.is_lut_index
        if lut is None:
            lut = (meta.get("feature_indices") or params.get("feature_indices") or {"is_lut": 1})["is_lut"]
        model = LightpathGNN(in_channels=params["in_channels"], hidden_channels=params["hidden_channels"],
                             output_dim=params["output_dim"], is_lut_index=lut, dropout_p=0.0)
    model.load_state_dict(state, strict=True)
    print(f"Model loaded from {path}")
    _, _, test_idx = harness.split_ranges(len(dataset))
    metrics, y_true, y_pred, skipped = harness.evaluate(model, dataset, test_idx, kind=kind, batch_size=args.batch_size,
                                                        output_dim=params["output_dim"], device=device,
                                                        return_predictions=True)
    print(f"Test R2 Score per output: {[m['R2'] for m in metrics.values()]}")
    print(f"Test MSE per output: {[m['Test_MSE'] for m in metrics.values()]}")
    stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
    if kind == "topological":
        folder, name = os.path.join(root, "results", f"results_{stamp}_model_{k}"), "results_metrics.json"
    else:
        print(f"Total skipped graphs during testing: {skipped}")
        folder, name = os.path.join(root, "results", f"results_{stamp}"), "results.json"
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, name), "w") as f:
        json.dump(metrics, f, indent=4)
    with open(os.path.join(folder, "y_true_descaled.json"), "w") as f:
        json.dump(y_true.tolist(), f)
    with open(os.path.join(folder, "y_pred_descaled.json"), "w") as f:
        json.dump(y_pred.tolist(), f)
    print(f"Results saved to {folder}")
    return folder


if __name__ == "__main__":
    main()
