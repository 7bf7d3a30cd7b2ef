"""Import-path shim: the reference harness does ``from lightpath_training.models import
LightpathGNN`` (lightpath_training/train.py:9, test.py).  Re-exports the HIP-backed class."""
from gnn_qot_estimation_amd.lightpath import LightpathGNN  # noqa: F401
