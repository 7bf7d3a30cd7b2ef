"""Per-channel loop restatement of the reference's graph construction rules (TEST INFRASTRUCTURE).

Follows ``to_graph.py:131-182`` (topological) and ``to_graph.py:221-312`` (lightpath) step by step over plain arrays --
one Python iteration per occupied (link, freq) channel, dictionaries and sets exactly where the reference uses them --
so that ``gnn_qot_estimation_amd.to_graph`` (vectorised scans, one comparison per link) can be checked graph-for-graph:
node order, attributes, edge set, adjacency insertion order.  xarray is absent here, so the reference module itself
cannot be imported (ordinary ImportError); the inputs are the same variables as arrays (``to_graph.NetworkStatus``).
PARITY PIN STATUS: unpinned by the reference (it holds no sample data and no expected graphs); pinned by the
hand-derived cases in ``tests/test_to_graph_cpu.py``.
"""
import networkx as nx
import numpy as np


def topological(ns, s, features):
    fi = ns.feature_indexes
    sample = np.asarray(ns.data[s])
    G = nx.Graph()
    G.add_nodes_from(range(1, 76))                                        # :133-135
    first_seen = {}                                                       # conn_id -> first occupied channel (:152-158)
    for l in range(sample.shape[1]):
        for f in range(sample.shape[2]):
            v = sample[:, l, f]
            if not np.any(v != 0):                                        # :141
                continue
            cid = int(v[fi["conn_id"]])
            if cid not in first_seen:
                first_seen[cid] = v
    for cid in sorted(first_seen):                                        # np.unique sorts (:152)
        v = first_seen[cid]
        G.add_edge(int(v[fi["src_id"]]), int(v[fi["dst_id"]]), **{k: v[fi[k]] for k in features})   # :161-174
    G.graph["labels"] = dict(zip(ns.metric, np.asarray(ns.target[s])))   # :177-178
    return G


def lightpath(ns, s, features, thr=0.05):
    fi = ns.feature_indexes
    sample = np.asarray(ns.data[s])
    G = nx.Graph()
    G.graph["labels"] = dict(zip(ns.metric, np.asarray(ns.target[s])))   # :222
    paths, on_link = {}, {}
    for l in range(sample.shape[1]):
        for f in range(sample.shape[2]):
            v = sample[:, l, f]
            if not np.any(v != 0):
                continue
            cid = int(v[fi["conn_id"]])                                   # :241
            if cid not in paths:
                lut = int(v[fi["osnr"]] == -1 and v[fi["snr"]] == -1 and v[fi["ber"]] == -1)      # :245-251
                attrs = {k: v[fi[k]] for k in features}
                attrs["is_lut"] = lut
                paths[cid] = (attrs, {})
            paths[cid][1].setdefault(l, set()).add(ns.freq[f])           # :266-267
            on_link.setdefault(l, set()).add(cid)                         # :270
    for cid, (attrs, _) in paths.items():
        G.add_node(f"lightpath_{cid}", **attrs)                           # :273-275
    for l, cids in on_link.items():
        cids = list(cids)
        if len(cids) < 2:
            continue
        table = [(fr, c) for c in cids for fr in paths[c][1][l]]          # :284-291
        done = set()
        for i, (fa, a) in enumerate(table):
            for j, (fb, b) in enumerate(table):
                d = abs(fa - fb)
                if d < thr and d > 0:                                     # :300
                    key = tuple(sorted((a, b)))
                    if key not in done:
                        G.add_edge(f"lightpath_{a}", f"lightpath_{b}")
                        done.add(key)
    return G
