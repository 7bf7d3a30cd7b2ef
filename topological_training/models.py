"""Import-path shim: the reference harness does ``from topological_training.models import
TopologicalGNN`` (topological_training/train.py:9, test.py).  Re-exports the HIP-backed class."""
from gnn_qot_estimation_amd.topological import TopologicalGNN  # noqa: F401
