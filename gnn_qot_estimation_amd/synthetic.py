"""Seeded synthetic optical-topology batches (BASELINE.json configs).

There is no dataset in the reference tree (``.gitignore:1-2``) and no network, so benchmark
and parity inputs are generated: per graph ``numpy.random.default_rng(1234 + 1000*cfg + g)``.
Layout follows Appendix C: both directions of every link, identical ``edge_attr`` for
(u,v)/(v,u), ``node_ids = arange(n)`` not offset, ``x = None`` (topological), flat ``y``.
"""
from __future__ import annotations

import numpy as np
import torch

from .batch import Batch, Data

# 14-node / 21-link NSFNET backbone (cfg 1)
NSFNET_LINKS = [(0, 1), (0, 2), (0, 7), (1, 2), (1, 3), (2, 5), (3, 4), (3, 10), (4, 5), (4, 6), (5, 9),
                (5, 13), (6, 7), (7, 8), (8, 9), (8, 11), (8, 12), (10, 11), (10, 12), (11, 13), (12, 13)]


def _links_to_graph(n, links, rng, edge_dim, with_y=True):
    links = np.asarray(links, dtype=np.int64).reshape(-1, 2)
    attr = rng.random((links.shape[0], edge_dim), dtype=np.float32)
    src = np.concatenate([links[:, 0], links[:, 1]])
    dst = np.concatenate([links[:, 1], links[:, 0]])
    ea = np.concatenate([attr, attr])
    order = np.lexsort((dst, src))                       # from_networkx order: by source, then neighbour
    d = Data(edge_index=torch.from_numpy(np.stack([src[order], dst[order]])),
             edge_attr=torch.from_numpy(ea[order]),
             node_ids=torch.arange(n, dtype=torch.long), num_nodes=n)
    if with_y:
        d.y = torch.from_numpy(rng.random(3, dtype=np.float32))
    return d


def random_links(n, num_links, rng):
    """Random spanning tree (n-1 links) + uniformly random extra links, no self loops/duplicates."""
    perm = rng.permutation(n)
    parents = np.array([perm[rng.integers(0, i)] for i in range(1, n)], dtype=np.int64) if n > 1 else np.zeros(0, np.int64)
    u = perm[1:]
    lo, hi = np.minimum(u, parents), np.maximum(u, parents)
    have = set((lo * n + hi).tolist())
    max_links = n * (n - 1) // 2
    num_links = min(num_links, max_links)
    while len(have) < num_links:
        need = num_links - len(have)
        a = rng.integers(0, n, size=2 * need + 8)
        b = rng.integers(0, n, size=2 * need + 8)
        for x, y in zip(a.tolist(), b.tolist()):
            if x == y:
                continue
            key = min(x, y) * n + max(x, y)
            if key not in have:
                have.add(key)
                if len(have) == num_links:
                    break
    keys = np.fromiter(sorted(have), dtype=np.int64)
    return np.stack([keys // n, keys % n], axis=1)


def powerlaw_links(n, rng, m=2, max_degree=64):
    """Barabasi-Albert (m links per new node) with the degree capped by rejection (cfg 5)."""
    deg = np.zeros(n, dtype=np.int64)
    links = []
    targets = list(range(m))
    repeated = []
    for new in range(m, n):
        chosen = set()
        tries = 0
        while len(chosen) < m and tries < 64:
            tries += 1
            t = targets[rng.integers(0, len(targets))] if not repeated else (
                repeated[rng.integers(0, len(repeated))] if rng.random() < 0.9 else int(rng.integers(0, new)))
            if t != new and t not in chosen and deg[t] < max_degree:
                chosen.add(t)
        for t in chosen:
            links.append((t, new))
            deg[t] += 1
            deg[new] += 1
            repeated.extend([t, new])
    return np.asarray(links, dtype=np.int64)


def topological_batch(cfg: int, num_graphs: int, n: int = None, e: int = None, edge_dim: int = 4,
                      first_graph: int = 0) -> Batch:
    """cfg 1: NSFNET x B; cfg 2/4: tree + random links (n nodes, e directed edges); cfg 5: power law."""
    graphs = []
    for g in range(first_graph, first_graph + num_graphs):
        rng = np.random.default_rng(1234 + 1000 * cfg + g)
        if cfg == 1:
            graphs.append(_links_to_graph(14, NSFNET_LINKS, rng, edge_dim))
        elif cfg == 5:
            graphs.append(_links_to_graph(n, powerlaw_links(n, rng), rng, edge_dim))
        else:
            graphs.append(_links_to_graph(n, random_links(n, e // 2, rng), rng, edge_dim))
    return Batch.from_data_list(graphs)


LIGHTPATH_FEATURES = ["freq", "is_lut", "mod_order", "num_spans", "path_len"]  # sorted, is_lut -> column 1


def lightpath_batch(num_graphs: int, cfg: int = 3, min_nodes: int = 2, max_nodes: int = 20,
                    first_graph: int = 0, lut: bool = True) -> Batch:
    """Chain line-graphs, n_g ~ U{min..max}; node 0 is the LUT (column 1 == 1.0); y [B,3]."""
    graphs = []
    for g in range(first_graph, first_graph + num_graphs):
        rng = np.random.default_rng(1234 + 1000 * cfg + g)
        n = int(rng.integers(min_nodes, max_nodes + 1))
        x = rng.random((n, 5), dtype=np.float32)
        x[:, 1] = 0.0
        if lut:
            x[0, 1] = 1.0
        a = np.arange(n - 1, dtype=np.int64)
        src = np.concatenate([a, a + 1])
        dst = np.concatenate([a + 1, a])
        order = np.lexsort((dst, src))
        d = Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(np.stack([src[order], dst[order]])),
                 y=torch.from_numpy(rng.random((1, 3), dtype=np.float32)), num_nodes=n)
        graphs.append(d)
    return Batch.from_data_list(graphs)


def tile_batch(base: Batch, times: int) -> Batch:
    """Concatenate ``times`` copies of ``base`` (distinct graphs are generated once, then
    tiled, when a benchmark needs more graphs than are worth generating on the host)."""
    if times == 1:
        return base
    n, b = base.num_nodes, base.num_graphs
    out = Batch()
    out.num_graphs = b * times
    out.uniform_node_ids = getattr(base, "uniform_node_ids", None)
    out._num_nodes = n * times
    out.ptr = torch.cat([base.ptr[:-1] + k * n for k in range(times)] + [torch.tensor([n * times])])
    out.batch = torch.cat([base.batch + k * b for k in range(times)])
    out.edge_index = torch.cat([base.edge_index + k * n for k in range(times)], dim=1)
    rep = lambda t: None if t is None else torch.cat([t] * times, dim=0)
    out.x, out.edge_attr, out.node_ids, out.y = rep(base.x), rep(base.edge_attr), rep(base.node_ids), rep(base.y)
    if getattr(base, "edge_ptr", None) is not None:
        e = base.num_edges
        out.edge_ptr = torch.cat([base.edge_ptr[:-1] + k * e for k in range(times)] + [torch.tensor([e * times])])
        out.graph_sizes = base.graph_sizes
        out.has_self_loops = getattr(base, "has_self_loops", None)
    return out
