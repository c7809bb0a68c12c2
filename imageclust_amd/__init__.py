"""imageclust_amd -- MI355X-native embed + Ward-cluster engine behind imageclust's own function names.

Host-side mirror of the two reference packages the engine replaces:
  imageclust_amd.embeddings  <->  /root/reference/internal/embeddings/embeddings.go
  imageclust_amd.clustering  <->  /root/reference/internal/clustering/clustering.go
Both call the C-ABI of libimageclust_hip.so (include/imageclust.h) and nothing else.
"""
from . import _lib  # noqa: F401
from ._lib import Context, ICLError  # noqa: F401

__all__ = ["Context", "ICLError", "clustering", "embeddings"]
