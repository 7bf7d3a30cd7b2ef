"""Graph construction: one network-status sample -> the reference's two graph representations.

Counterpart of ``to_graph.py`` and the loop of ``store_graphs.py:64-79``.  The reference reads
an xarray ``.nc`` file -- and re-opens it for every sample (``to_graph.py:124-129, 216-219``); xarray and the ``.nc``
data are not available here, and the rules only need plain arrays, so a sample source is a ``NetworkStatus``: the same
variables under the same names (``data [sample, lp_feat, link, freq]``, ``target [sample, metric]``, coordinate arrays
``lp_feat``, ``metric``, ``link``, ``freq``), held in memory or in an ``.npz`` file; an ``.nc`` path is accepted when
xarray is importable.

Rules restated (graph-for-graph, including node order, attribute values and adjacency insertion order, which is what
fixes the column order of ``edge_index`` downstream):

``create_topological_graph`` (``to_graph.py:62-182``): 75 featureless nodes ``1..75``; an occupied (link, freq)
channel is one whose feature vector is not all zero; lightpaths are the unique ``conn_id`` values (first occupied
channel in (link, freq) scan order supplies the features); one undirected edge ``src_id -- dst_id`` per lightpath, added
in ascending ``conn_id`` order, carrying the requested features -- parallel lightpaths between the same node pair
collapse to one edge whose attributes are the LAST one's (``nx.Graph.add_edge`` updates), its position the first one's.

``create_lightpath_graph`` (``to_graph.py:187-312``): one node ``"lightpath_<conn_id>"`` per lightpath in first-seen
order with the requested features + ``is_lut`` (1 iff osnr == snr == ber == -1 on the first-seen channel,
``:247-251``); two lightpaths are linked when, on some shared link, they occupy frequencies with
``0 < |f1 - f2| < freq_threshold`` (``:279-310``; a lightpath with two such slots on one link gets a self loop, as in
the reference).  The pair test is one vectorised comparison per link.

Both return ``networkx.Graph`` objects that pickle into the ``.gpickle`` files ``dataset.TopologicalDataset`` /
``LightpathDataset`` (and the reference's own dataset classes) read; ``store_graphs`` writes them, ``build_shard``
skips the files and packs ``Data`` objects straight into a ``PackedGraphs`` shard.
"""
from __future__ import annotations

import os
import pickle
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Union

import numpy as np

DEFAULT_FEATURES = ["mod_order", "path_len", "num_spans", "freq"]        # store_graphs.py:51-56
NUM_TOPOLOGY_NODES = 75                                                   # to_graph.py:134


@dataclass
class NetworkStatus:
    """The variables ``to_graph.py`` reads from the ``.nc`` dataset, as arrays (``to_graph.py:23-38,124-129``)."""
    data: np.ndarray        # [sample, lp_feat, link, freq]
    target: np.ndarray      # [sample, metric]
    lp_feat: Sequence[str]  # names along axis 1 of data
    metric: Sequence[str]   # names along axis 1 of target
    link: np.ndarray        # [link] identifiers
    freq: np.ndarray        # [freq] centre frequencies (THz)

    def __post_init__(self):
        self.lp_feat = [str(v) for v in self.lp_feat]
        self.metric = [str(v) for v in self.metric]
        self.feature_indexes = {f: i for i, f in enumerate(self.lp_feat)}       # to_graph.py:27
        if self.data.ndim != 4 or self.data.shape[1] != len(self.lp_feat):
            raise ValueError("data must be [sample, lp_feat, link, freq]")
        if self.data.shape[2] != len(self.link) or self.data.shape[3] != len(self.freq):
            raise ValueError("link / freq coordinates do not match data")

    def __len__(self) -> int:
        return self.data.shape[0]

    def save(self, path: str):
        np.savez_compressed(path, data=self.data, target=self.target, lp_feat=np.array(self.lp_feat),
                            metric=np.array(self.metric), link=np.asarray(self.link), freq=np.asarray(self.freq))

    @classmethod
    def load(cls, path: str) -> "NetworkStatus":
        if path.endswith(".nc"):
            try:
                import xarray as xr
            except ImportError as e:
                raise ImportError("reading .nc needs xarray; convert once to .npz with NetworkStatus.save") from e
            ds = xr.open_dataset(path)
            try:
                return cls(ds["data"].values, ds["target"].values, ds["lp_feat"].values, ds["metric"].values,
                           ds["link"].values, ds["freq"].values)
            finally:
                ds.close()
        z = np.load(path, allow_pickle=False)
        return cls(z["data"], z["target"], z["lp_feat"], z["metric"], z["link"], z["freq"])


Source = Union[NetworkStatus, str]
_loaded: Dict[str, NetworkStatus] = {}


def _source(dataset: Source) -> NetworkStatus:
    if isinstance(dataset, NetworkStatus):
        return dataset
    if dataset not in _loaded:            # the reference caches only metadata and re-opens per sample
        _loaded[dataset] = NetworkStatus.load(dataset)
    return _loaded[dataset]


def _occupied(sample: np.ndarray):
    """(link index, freq index) of occupied channels in scan order and their feature vectors [lp_feat, n]
    (``to_graph.py:141-146,229-232``)."""
    occ = np.any(sample != 0, axis=0)
    idx = np.argwhere(occ)
    return idx[:, 0], idx[:, 1], sample[:, occ]


def create_topological_graph(sample_index: int, features_to_consider: Sequence[str], dataset: Source):
    import networkx as nx
    ds = _source(dataset)
    fi = ds.feature_indexes
    sample = np.asarray(ds.data[sample_index])
    G = nx.Graph()
    G.add_nodes_from(range(1, NUM_TOPOLOGY_NODES + 1))
    _, _, vec = _occupied(sample)
    conn = vec[fi["conn_id"], :].astype(int)
    src = vec[fi["src_id"], :].astype(int)
    dst = vec[fi["dst_id"], :].astype(int)
    _, first = np.unique(conn, return_index=True)            # ascending conn_id, first occupied channel of each
    rows = [fi[f] for f in features_to_consider]
    feats = vec[rows][:, first]                               # [feature, lightpath]
    for n, i in enumerate(first):
        G.add_edge(src[i], dst[i], **{f: feats[r, n] for r, f in enumerate(features_to_consider)})
    G.graph["labels"] = dict(zip(ds.metric, np.asarray(ds.target[sample_index])))
    return G


def lightpath_pairs(link_idx: np.ndarray, freq_val: np.ndarray, conn: np.ndarray, freq_threshold: float = 0.05):
    """All (conn_a, conn_b) with a shared link and ``0 < |f_a - f_b| < freq_threshold`` on it, as the reference finds
    them (``to_graph.py:279-310``) but with one vectorised comparison per link; returned in the reference's insertion
    order (links ascending; inside a link the (i, j) order of its frequency table).  The per-link tables are built from
    Python sets in the reference -- lightpaths per link and frequencies per (lightpath, link) -- and their iteration
    order decides the table order, so the same sets are built here."""
    out: List[tuple] = []
    order = np.argsort(link_idx, kind="stable")
    bounds = np.flatnonzero(np.diff(link_idx[order])) + 1
    for seg in np.split(order, bounds):
        if seg.size < 2:
            continue
        lps: set = set()
        per_lp: Dict[int, set] = {}
        for t in seg:                                   # scan order inside the link
            c = int(conn[t])
            lps.add(c)
            per_lp.setdefault(c, set()).add(freq_val[t])
        if len(lps) < 2:
            continue
        ids, fr = [], []
        for c in list(lps):
            for f in per_lp[c]:
                fr.append(f)
                ids.append(c)
        fr, ids = np.asarray(fr), np.asarray(ids)
        diff = np.abs(fr[:, None] - fr[None, :])
        ii, jj = np.where((diff < freq_threshold) & (diff > 0))
        seen = set()
        for a, b in zip(ids[ii].tolist(), ids[jj].tolist()):
            key = (a, b) if a <= b else (b, a)
            if key not in seen:
                seen.add(key)
                out.append((a, b))
    return out


def create_lightpath_graph(sample_index: int, features_to_consider: Sequence[str], dataset: Source,
                           freq_threshold: float = 0.05):
    import networkx as nx
    ds = _source(dataset)
    fi = ds.feature_indexes
    sample = np.asarray(ds.data[sample_index])
    G = nx.Graph()
    G.graph["labels"] = dict(zip(ds.metric, np.asarray(ds.target[sample_index])))
    link_idx, freq_idx, vec = _occupied(sample)
    conn = vec[fi["conn_id"], :].astype(np.int64)        # int(...) truncation, as the per-channel loop does
    uniq, first = np.unique(conn, return_index=True)
    seen_order = np.sort(first)                           # lightpaths in first-seen order (dict insertion order)
    is_lut = ((vec[fi["osnr"], :] == -1) & (vec[fi["snr"], :] == -1) & (vec[fi["ber"], :] == -1)).astype(int)
    rows = [fi[f] for f in features_to_consider]
    for i in seen_order:
        attrs = {f: vec[r, i] for r, f in zip(rows, features_to_consider)}
        attrs["is_lut"] = int(is_lut[i])
        G.add_node(f"lightpath_{int(conn[i])}", **attrs)
    freq_val = np.asarray(ds.freq)[freq_idx]
    for a, b in lightpath_pairs(link_idx, freq_val, conn, freq_threshold):
        G.add_edge(f"lightpath_{a}", f"lightpath_{b}")
    return G


# ------------------------------------------------------------------------------------------ store_graphs.py counterpart
def store_graphs(dataset: Source, representation: str = "lightpath", directory: Optional[str] = None,
                 features_to_consider: Sequence[str] = DEFAULT_FEATURES, storage_type: str = "pickle",
                 samples: Optional[Sequence[int]] = None) -> str:
    """The loop of ``store_graphs.py:58-79``: clear the directory, write ``graph_<i>.gpickle`` (or ``.gexf``)."""
    if representation not in ("lightpath", "topological"):
        raise ValueError("representation must be 'lightpath' or 'topological'")
    ds = _source(dataset)
    directory = directory or ("networkx_graphs_lightpath" if representation == "lightpath" else "networkx_graphs_topological")
    if os.path.exists(directory):
        for name in os.listdir(directory):
            os.remove(os.path.join(directory, name))
    os.makedirs(directory, exist_ok=True)
    make = create_lightpath_graph if representation == "lightpath" else create_topological_graph
    for i in (range(len(ds)) if samples is None else samples):
        G = make(i, list(features_to_consider), ds)
        if storage_type == "pickle":
            with open(os.path.join(directory, f"graph_{i}.gpickle"), "wb") as f:
                pickle.dump(G, f, protocol=pickle.HIGHEST_PROTOCOL)
        elif storage_type == "gexf":
            import networkx as nx
            nx.write_gexf(G, os.path.join(directory, f"graph_{i}.gexf"))
        else:
            raise ValueError("storage_type must be 'pickle' or 'gexf'")
    return directory


def build_shard(dataset: Source, representation: str = "lightpath", features_to_consider: Sequence[str] = DEFAULT_FEATURES,
                samples: Optional[Sequence[int]] = None):
    """Samples -> graphs -> ``Data`` -> one ``PackedGraphs`` shard, without the per-graph files in between; the
    conversion rules are those of the dataset classes (``dataset.topological_data_from_graph`` / ``lightpath_...``)."""
    from .dataset import lightpath_data_from_graph, topological_data_from_graph
    from .loader import PackedGraphs
    ds = _source(dataset)
    feats = sorted(features_to_consider)
    out = []
    for i in (range(len(ds)) if samples is None else samples):
        if representation == "lightpath":
            out.append(lightpath_data_from_graph(create_lightpath_graph(i, list(features_to_consider), ds),
                                                 sorted(feats + ["is_lut"])))
        else:
            out.append(topological_data_from_graph(create_topological_graph(i, list(features_to_consider), ds), feats))
    return PackedGraphs.from_data_list(out)


# ------------------------------------------------------------------------------------------ .nc-free sample generator
LP_FEAT = ["conn_id", "src_id", "dst_id", "mod_order", "path_len", "num_spans", "freq", "osnr", "snr", "ber"]
METRICS = ["osnr", "snr", "ber", "class"]


def synthetic_network_status(num_samples: int, num_links: int = 60, num_freqs: int = 72, max_lightpaths: int = 24,
                             seed: int = 0) -> NetworkStatus:
    """Network-status samples with the dataset's structure (the real ``.nc`` is not shipped, ``.gitignore:2``): per
    sample up to ``max_lightpaths`` lightpaths, each routed over 1-6 links on one 50 GHz slot (occasionally two
    adjacent slots), feature values inside ``constants.FEATURE_RANGES``, established lightpaths carrying measured
    osnr/snr/ber and exactly one lightpath under test (osnr = snr = ber = -1)."""
    rng = np.random.default_rng(seed)
    freq = np.round(192.2 + 0.05 * np.arange(num_freqs), 6)
    data = np.zeros((num_samples, len(LP_FEAT), num_links, num_freqs), dtype=np.float64)
    target = np.zeros((num_samples, len(METRICS)), dtype=np.float64)
    fi = {f: i for i, f in enumerate(LP_FEAT)}
    for s in range(num_samples):
        n_lp = int(rng.integers(2, max_lightpaths + 1))
        lut = int(rng.integers(0, n_lp))
        conn_ids = rng.choice(np.arange(1, 10 * max_lightpaths), size=n_lp, replace=False)
        for k in range(n_lp):
            src, dst = rng.choice(np.arange(1, NUM_TOPOLOGY_NODES + 1), size=2, replace=False)
            links = rng.choice(num_links, size=int(rng.integers(1, 7)), replace=False)
            f0 = int(rng.integers(0, num_freqs - 1))
            slots = [f0] if rng.random() < 0.85 else [f0, f0 + 1]
            spans = int(rng.integers(1, 107))
            vec = np.zeros(len(LP_FEAT))
            vec[fi["conn_id"]], vec[fi["src_id"]], vec[fi["dst_id"]] = conn_ids[k], src, dst
            vec[fi["mod_order"]] = float(rng.choice([4, 8, 16, 32, 64]))
            vec[fi["num_spans"]] = spans
            vec[fi["path_len"]] = float(rng.integers(24214, 7834746))
            vec[fi["freq"]] = freq[f0]
            if k == lut:
                vec[fi["osnr"]] = vec[fi["snr"]] = vec[fi["ber"]] = -1.0
            else:
                vec[fi["osnr"]] = rng.uniform(12.47, 33.49)
                vec[fi["snr"]] = rng.uniform(8.96, 29.98)
                vec[fi["ber"]] = rng.uniform(1.7e-12, 1.98e-2)
            for l in links:
                for fslot in slots:
                    if not data[s, :, l, fslot].any():        # a slot on a link carries one lightpath
                        data[s, :, l, fslot] = vec
        target[s] = [rng.uniform(12.47, 33.49), rng.uniform(8.96, 29.98), rng.uniform(1.7e-12, 1.98e-2), float(rng.integers(0, 2))]
    return NetworkStatus(data, target, LP_FEAT, METRICS, np.arange(num_links), freq)


def main(argv=None):
    """``python -m gnn_qot_estimation_amd.to_graph --dataset x.npz --representation lightpath`` (store_graphs.py:8-36)."""
    import argparse
    ap = argparse.ArgumentParser(description="Store networkx graphs in a directory.")
    ap.add_argument("--dataset", required=True, help=".npz written by NetworkStatus.save (or .nc with xarray installed)")
    ap.add_argument("--representation", default="lightpath", choices=["lightpath", "topological"])
    ap.add_argument("--storage_type", default="pickle", choices=["pickle", "gexf"])
    ap.add_argument("--directory", default=None)
    args = ap.parse_args(argv)
    d = store_graphs(args.dataset, args.representation, args.directory, storage_type=args.storage_type)
    print(f"Graphs stored in {d}")


if __name__ == "__main__":
    main()
