"""``python -m gnn_qot_estimation_amd.train --kind topological|lightpath`` -- the reference's training scripts end to end
(``topological_training/train.py:23-227``, ``lightpath_training/train.py:24-256``) on the HIP models.

dataset (directory of ``.gpickle`` graphs written by ``store_graphs`` / ``to_graph.store_graphs``, or a shard file
written by ``dataset.save_shard``) -> 70/15/15 split, 10 % chunk per epoch, SGD(0.1, 0.9) + StepLR(10, 0.5), SmoothL1,
early stopping on validation R2 (``harness.fit``) -> ``<root>/models/model_<k>.pth`` = ``{"model_state_dict",
"model_params"}`` with the reference's keys, ``best_model.pth``, ``<root>/loss_training_<k>/*.json``,
``<root>/model_logger.txt``.  ``<root>`` defaults to ``topological_training`` / ``lightpath_training``: the same paths the
reference writes, so its ``test.py`` / ``plot_*.py`` find them.  Hyper-parameters default to the reference's literals
(``train.py:38-52``: batch 512, 35 epochs, patience 10, hidden 16 / 32, num_nodes 75).
"""
from __future__ import annotations

import argparse
import os
import time

import torch

from . import harness
from .dataset import LightpathDataset, TopologicalDataset, load_shard


def make_logger(root: str):
    os.makedirs(root, exist_ok=True)
    path = os.path.join(root, "model_logger.txt")

    def log_message(*args):                     # train.py:12-21
        full = f"{time.strftime('%Y-%m-%d %H:%M:%S')} - {' '.join(map(str, args))}"
        print(full)
        with open(path, "a") as f:
            f.write(full + "\n")
    return log_message


def open_dataset(kind: str, data: str, resident: bool, device):
    """(dataset object, metadata dict): a graph directory is converted ONCE into a pre-tensorised shard."""
    if os.path.isdir(data):
        ds = TopologicalDataset(data) if kind == "topological" else LightpathDataset(data)
        meta = ({"FEATURES": list(ds.FEATURES), "edge_dim": ds.edge_dim} if kind == "topological" else
                {"NODE_FEATURES": list(ds.node_features), "feature_indices": dict(ds.feature_indices)})
        shard = ds.pack()
    else:
        shard, meta = load_shard(data)
    shard = shard.to_device(device) if resident else shard.pin()
    return shard, meta


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--kind", choices=["topological", "lightpath"], required=True)
    ap.add_argument("--data", default=None, help="graph directory or shard file (default: networkx_graphs_<kind>)")
    ap.add_argument("--root", default=None, help="output root (default: <kind>_training, as the reference)")
    ap.add_argument("--epochs", type=int, default=35)
    ap.add_argument("--batch-size", type=int, default=512)
    ap.add_argument("--patience", type=int, default=10)
    ap.add_argument("--hidden", type=int, default=None, help="hidden_channels (default 16 topological / 32 lightpath)")
    ap.add_argument("--num-nodes", type=int, default=75)
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--host-shard", action="store_true", help="keep the shard in pinned host memory instead of HBM")
    args = ap.parse_args(argv)

    from . import LightpathGNN, TopologicalGNN
    kind = args.kind
    root = args.root or f"{kind}_training"
    data = args.data or f"networkx_graphs_{kind}"
    device = torch.device(args.device)
    log = make_logger(root)
    dataset, meta = open_dataset(kind, data, not args.host_shard, device)
    hidden = args.hidden or (16 if kind == "topological" else 32)
    if kind == "topological":
        edge_dim = int(meta.get("edge_dim", dataset.edge_attr.shape[1]))
        model = TopologicalGNN(num_nodes=args.num_nodes, hidden_channels=hidden, out_channels=3, edge_dim=edge_dim,
                               dropout_p=args.dropout)
        params = {"num_nodes": args.num_nodes, "hidden_channels": hidden, "output_dim": 3, "edge_dim": edge_dim,
                  "FEATURES": meta.get("FEATURES")}                                   # train.py:200-206
    else:
        fidx = meta.get("feature_indices") or {"freq": 0, "is_lut": 1, "mod_order": 2, "num_spans": 3, "path_len": 4}
        num_features = dataset.x.shape[1]
        model = LightpathGNN(in_channels=num_features, hidden_channels=hidden, output_dim=3,
                             is_lut_index=fidx["is_lut"], dropout_p=args.dropout)
        params = {"in_channels": num_features, "hidden_channels": hidden, "output_dim": 3,
                  "NODE_FEATURES": meta.get("NODE_FEATURES"), "feature_indices": fidx}  # lightpath train.py:227-233
    hist = harness.fit(model, dataset, kind=kind, batch_size=args.batch_size, num_epochs=args.epochs,
                       patience=args.patience, device=device, best_path=os.path.join(root, "best_model.pth"), log=log)
    if kind == "lightpath":
        log(f"Total skipped {hist.skipped_graphs} graphs due to missing LUT nodes.")
    path, k = harness.next_model_path(os.path.join(root, "models"))
    harness.save_checkpoint(path, model, params)
    log("Model saved to", path)
    loss_dir = os.path.join(root, f"loss_training_{k}")
    hist.dump(loss_dir)
    log(f"Loss and metrics saved to {loss_dir}")
    return path


if __name__ == "__main__":
    main()
