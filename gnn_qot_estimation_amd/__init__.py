"""MI355X-native message-passing engine behind the QoT-GNN model interface.

Drop-in for the model layer of santiagolmedo/gnn_qot_estimation
(``topological_training/models.py``, ``lightpath_training/models.py``): identical
constructors, ``forward(data)`` contracts and ``state_dict`` keys; the PyG operator layer
underneath is replaced by hand-written gfx950 HIP kernels in ``libqot_gnn.so``
(C ABI: ``include/qot_gnn.h``).
"""
from .batch import Batch, Data, shard_graphs
from . import dataset, harness
from .lightpath import LightpathGNN
from .loader import GraphLoader, PackedGraphs
from .nn import BatchNorm, GATConv, NNConv, TransformerConv, global_mean_pool
from .topological import TopologicalGNN

__all__ = ["Batch", "Data", "shard_graphs", "TopologicalGNN", "LightpathGNN", "TransformerConv", "NNConv",
           "GATConv", "BatchNorm", "global_mean_pool", "GraphLoader", "PackedGraphs", "harness", "dataset"]
