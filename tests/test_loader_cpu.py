"""Batch packer / loader on the CPU: collate layout, ordering, ragged
last batch, staging-buffer reuse."""
import torch

import gnn_qot_estimation_amd as q
from gnn_qot_estimation_amd import synthetic as S
from gnn_qot_estimation_amd.loader import GraphLoader, _Staging, collate_into


def _graphs(k):
    out = []
    for g in range(k):
        b = S.topological_batch(2, 1, n=10 + (g % 3), e=24, first_graph=g)
        out.append(q.Data(edge_index=b.edge_index, edge_attr=b.edge_attr, node_ids=b.node_ids, y=b.y,
                          num_nodes=b.num_nodes))
    return out


def _same(a, b):
    for name in ("x", "edge_index", "edge_attr", "y", "node_ids", "batch", "ptr"):
        ta, tb = getattr(a, name), getattr(b, name)
        assert (ta is None) == (tb is None), name
        if ta is not None:
            assert torch.equal(ta, tb), name
    assert a.num_graphs == b.num_graphs and a.num_nodes == b.num_nodes
    assert a.uniform_node_ids == b.uniform_node_ids


def test_collate_into_staging_matches_from_data_list():
    gs = _graphs(7)
    st = _Staging(torch.device("cpu"), pin=False)
    _same(collate_into(gs, st), q.Batch.from_data_list(gs))
    # staging buffers are reused (and must not leak stale tails) for a smaller batch
    _same(collate_into(gs[:2], st), q.Batch.from_data_list(gs[:2]))
    uni = [g for g in gs if g.num_nodes == 10]
    b = collate_into(uni, st)
    assert b.uniform_node_ids == 10
    _same(b, q.Batch.from_data_list(uni))


def test_graph_loader_cpu_order_and_ragged_tail():
    gs = _graphs(10)
    ld = GraphLoader(gs, batch_size=4, device="cpu")
    batches = list(ld)
    assert len(ld) == 3 and [b.num_graphs for b in batches] == [4, 4, 2]
    _same(batches[1], q.Batch.from_data_list(gs[4:8]))
    assert len(list(GraphLoader(gs, batch_size=4, device="cpu", drop_last=True))) == 2
    gen = torch.Generator().manual_seed(3)
    shuffled = list(GraphLoader(gs, batch_size=10, device="cpu", shuffle=True, generator=gen))
    order = torch.randperm(10, generator=torch.Generator().manual_seed(3)).tolist()
    _same(shuffled[0], q.Batch.from_data_list([gs[i] for i in order]))


def test_lightpath_graphs_through_loader():
    lp = S.lightpath_batch(6)
    gs = [q.shard_graphs(lp, g, 6) for g in range(6)]
    data = [q.Data(x=s.x, edge_index=s.edge_index, y=s.y, num_nodes=s.num_nodes) for s in gs]
    b = next(iter(GraphLoader(data, batch_size=6, device="cpu")))
    assert torch.equal(b.x, lp.x) and torch.equal(b.edge_index, lp.edge_index) and torch.equal(b.batch, lp.batch)


def test_packed_graphs_roundtrip_and_dataset_protocol():
    from gnn_qot_estimation_amd.loader import PackedGraphs
    gs = _graphs(9)
    pk = PackedGraphs.from_data_list(gs)
    assert len(pk) == 9
    for g in (0, 4, 8):
        a, b = pk[g], gs[g]
        assert torch.equal(a.edge_index, b.edge_index) and torch.equal(a.edge_attr, b.edge_attr)
        assert torch.equal(a.node_ids, b.node_ids) and torch.equal(a.y, b.y) and a.num_nodes == b.num_nodes
    # as a dataset for the (CPU) loader it collates to the same batches as the list of graphs
    for x, y in zip(GraphLoader(pk, batch_size=4, device="cpu"), GraphLoader(gs, batch_size=4, device="cpu")):
        _same(x, y)
