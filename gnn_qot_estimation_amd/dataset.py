"""Dataset side of the hot path: graph files -> ``Data`` -> pre-tensorised shards.

Counterpart of ``topological_training/dataset.py`` and ``lightpath_training/dataset.py``
 without the PyG dependency (``from_networkx``).  The two dataset classes keep
the reference's constructor arguments, attributes (``FEATURES`` / ``edge_dim``; ``node_features`` /
``feature_indices``) and per-item conversion rules:

* nodes are relabelled 0..n-1 in node order (dataset.py:57) and an undirected graph contributes both
  directions of every link, grouped by source node in adjacency order -- the order PyG's
  ``from_networkx`` emits, which fixes the row order of ``edge_attr``;
* link (topological) or node (lightpath) attributes named in ``FEATURE_RANGES`` are min-max scaled,
  others are passed through, ``is_lut`` is kept as 0/1 (dataset.py:64-72; lightpath dataset.py:72-83);
* attribute rows follow the sorted feature list, a missing attribute is 0.0 (dataset.py:83-104);
* ``y`` = min-max scaled (osnr, snr, ber): shape [3] for topological graphs (a missing or unparsable
  label reads 0.0 before / after scaling as at dataset.py:108-121), [1,3] for lightpath graphs.

What is new: ``pack()`` converts a whole directory once into a ``PackedGraphs`` shard (flat tensors +
offsets) that ``save_shard`` / ``load_shard`` keep as a plain tensor file, so an epoch streams pinned
slices instead of unpickling and walking one networkx object per graph per epoch.
The graph files are the user's own (written by the reference's ``store_graphs.py``); none ship with
the reference.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, Iterable, List, Optional, Sequence

import torch

from .batch import Data
from .loader import PackedGraphs

# label / feature scaling tables of the reference (constants.py:1-12) -- data, not code
FEATURE_RANGES = {
    "mod_order": {"min": 0, "max": 64},
    "path_len": {"min": 24214, "max": 7834746},
    "num_spans": {"min": 1, "max": 106},
    "freq": {"min": 192.2, "max": 195.8},
}
TARGET_RANGES = {
    "osnr": {"min": 12.47, "max": 33.49},
    "snr": {"min": 8.96, "max": 29.98},
    "ber": {"min": 1.70e-12, "max": 1.98e-2},
}
TARGET_KEYS = ("osnr", "snr", "ber")


def min_max_scale(value, min_value, max_value):
    return (value - min_value) / (max_value - min_value)


def _scaled(key, value, ranges):
    """Scaled float when ``key`` has a range and ``value`` parses, else the value untouched."""
    try:
        r = ranges[key]
        return min_max_scale(float(value), r["min"], r["max"])
    except (ValueError, KeyError):
        return value


def _numeric_row(attr: Dict[str, object], keys: Sequence[str]) -> List[float]:
    row = [attr.get(k, 0.0) for k in keys if isinstance(attr.get(k, 0.0), (int, float))]
    if len(row) != len(keys):
        # the reference would hand torch.tensor a ragged list here and fail the same way
        raise ValueError(f"non-numeric attribute among {list(keys)}: {attr}")
    return [float(v) for v in row]


def _directed_links(G):
    """(index of node, relabelled) and the directed link list in ``from_networkx`` order."""
    index = {node: i for i, node in enumerate(G.nodes())}
    # G.adj of an undirected graph lists every link from both ends; of a directed one, successors only
    links = [(index[u], index[v]) for u in G.nodes() for v in G.adj[u]]
    return index, links


def _labels(G, target_ranges, tolerant: bool) -> List[float]:
    labels = G.graph.get("labels", {})
    out = []
    for key in TARGET_KEYS:
        if tolerant:
            try:
                r = target_ranges[key]
                out.append(min_max_scale(float(labels.get(key, 0.0)), r["min"], r["max"]))
            except (ValueError, KeyError):
                out.append(0.0)
        else:
            r = target_ranges[key]
            out.append(min_max_scale(float(labels.get(key, 0.0)), r["min"], r["max"]))
    return out


def topological_data_from_graph(G, features: Sequence[str], feature_ranges=FEATURE_RANGES,
                                target_ranges=TARGET_RANGES) -> Data:
    """One networkx-like graph -> ``Data`` (``TopologicalDataset.__getitem__``, dataset.py:45-123)."""
    index, links = _directed_links(G)
    rows = {}
    for u, v, attr in G.edges(data=True):
        scaled = {k: _scaled(k, val, feature_ranges) for k, val in attr.items()}
        rows[(index[u], index[v])] = _numeric_row(scaled, features)
    zero = [0.0] * len(features)
    ea = [rows.get((u, v)) or rows.get((v, u)) or zero for (u, v) in links]
    n = len(index)
    ei = torch.tensor(links, dtype=torch.long).t().contiguous() if links else torch.zeros(2, 0, dtype=torch.long)
    return Data(edge_index=ei, edge_attr=torch.tensor(ea, dtype=torch.float).reshape(len(links), len(features)),
                node_ids=torch.arange(n), y=torch.tensor(_labels(G, target_ranges, True), dtype=torch.float),
                num_nodes=n)


def lightpath_data_from_graph(G, node_features: Sequence[str], feature_ranges=FEATURE_RANGES,
                              target_ranges=TARGET_RANGES) -> Data:
    """``LightpathDataset.__getitem__`` (lightpath_training/dataset.py:54-123)."""
    index, links = _directed_links(G)
    x = []
    for node, attr in G.nodes(data=True):
        scaled = {k: (float(v) if k == "is_lut" else _scaled(k, v, feature_ranges)) for k, v in attr.items()}
        x.append(_numeric_row(scaled, node_features))
    n = len(index)
    ei = torch.tensor(links, dtype=torch.long).t().contiguous() if links else torch.zeros(2, 0, dtype=torch.long)
    return Data(x=torch.tensor(x, dtype=torch.float).reshape(n, len(node_features)), edge_index=ei,
                y=torch.tensor(_labels(G, target_ranges, False), dtype=torch.float).unsqueeze(0), num_nodes=n)


class _GraphDirectory:
    suffix = ".gpickle"

    def __init__(self, directory: str):
        self.directory = directory
        self.file_list = sorted(f for f in os.listdir(directory) if f.endswith(self.suffix))
        self.N = len(self.file_list)

    def __len__(self) -> int:
        return self.N

    def _read(self, idx: int):
        with open(os.path.join(self.directory, self.file_list[idx]), "rb") as f:
            return pickle.load(f)           # the user's own graph files (store_graphs.py output)

    def _read_or_next(self, idx: int):
        """A file that does not load is replaced by its successor (dataset.py:49-54)."""
        for k in range(max(self.N, 1)):
            j = (idx + k) % self.N
            try:
                return self._read(j)
            except (pickle.UnpicklingError, EOFError, FileNotFoundError) as e:
                print(f"Error loading {os.path.join(self.directory, self.file_list[j])}: {e}")
        raise RuntimeError(f"no loadable graph file in {self.directory}")

    def pack(self, indices: Optional[Iterable[int]] = None) -> PackedGraphs:
        """Convert (a range of) the directory once into a pre-tensorised shard."""
        idx = range(self.N) if indices is None else indices
        return PackedGraphs.from_data_list([self[i] for i in idx])


class TopologicalDataset(_GraphDirectory):
    """``TopologicalDataset(directory, features=None)`` (topological_training/dataset.py:14-42)."""

    def __init__(self, directory: str = "networkx_graphs_topological", features: Optional[Sequence[str]] = None):
        super().__init__(directory)
        self.FEATURES = features
        if self.FEATURES is None:
            found = set()
            for i in range(min(self.N, 100)):           # first 100 files (dataset.py:30)
                for _, _, attr in self._read(i).edges(data=True):
                    found.update(attr.keys())
            self.FEATURES = sorted(found)
        self.edge_dim = len(self.FEATURES)

    def __getitem__(self, idx: int) -> Data:
        return topological_data_from_graph(self._read_or_next(idx), self.FEATURES)


class LightpathDataset(_GraphDirectory):
    """``LightpathDataset(directory, node_features=None, feature_indices=None)``
    (lightpath_training/dataset.py:14-52)."""

    def __init__(self, directory: str = "networkx_graphs_lightpath", node_features: Optional[Sequence[str]] = None,
                 feature_indices: Optional[Dict[str, int]] = None):
        super().__init__(directory)
        self.node_features, self.feature_indices = node_features, feature_indices
        if self.node_features is None or self.feature_indices is None:
            found = set()
            for i in range(min(self.N, 1000)):          # first 1000 files (dataset.py:36)
                for _, attr in self._read(i).nodes(data=True):
                    found.update(attr.keys())
            self.node_features = sorted(found)
            self.feature_indices = {k: i for i, k in enumerate(self.node_features)}

    def __getitem__(self, idx: int) -> Data:
        return lightpath_data_from_graph(self._read_or_next(idx), self.node_features)


_SHARD_FIELDS = ("node_ptr", "edge_ptr", "edge_index", "edge_attr", "node_ids", "x", "y")


def save_shard(path: str, shard: PackedGraphs, meta: Optional[Dict[str, object]] = None):
    """A shard file is a dict of tensors (+ plain metadata): loadable with ``weights_only=True``."""
    blob = {k: getattr(shard, k) for k in _SHARD_FIELDS if getattr(shard, k, None) is not None}
    blob = {k: v.detach().cpu().contiguous() for k, v in blob.items()}
    blob["uniform_node_ids"] = -1 if shard.uniform_node_ids is None else int(shard.uniform_node_ids)
    blob["meta"] = dict(meta or {})
    torch.save(blob, path)


def load_shard(path: str):
    blob = torch.load(path, map_location="cpu", weights_only=True)
    meta = blob.pop("meta", {})
    uni = blob.pop("uniform_node_ids", -1)
    return PackedGraphs(**{k: blob.get(k) for k in _SHARD_FIELDS}, uniform_node_ids=None if uni < 0 else uni), meta
