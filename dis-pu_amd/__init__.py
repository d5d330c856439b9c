"""dis-pu_amd: MI355X-native (gfx950) implementation of the Dis-PU point-sampling / grouping /
distance hot path, behind the reference's own Python op signatures.

Host side (this package) = shape/dtype validation + ctypes calls into csrc/ -> lib/libdispu_hip.so.
PyTorch-ROCm is used for device memory, streams and torch.distributed only.

Modules named like the reference's op wrappers:
  tf_sampling, tf_grouping, tf_interpolate, tf_nndistance, tf_approxmatch  (tf_ops/*/tf_*.py)
  nearest_neighbors                                                          (libs/nearest_neighbors/knn.pyx)
"""
from . import _lib  # noqa: F401  (does not load the .so until first use)

__all__ = ["tf_sampling", "tf_grouping", "tf_interpolate", "tf_nndistance", "tf_approxmatch",
           "nearest_neighbors"]
__version__ = "0.2.0"
