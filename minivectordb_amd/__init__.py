"""minivectordb_amd -- MI355X-native query engine for MiniVectorDB's two-stage
retrieval path (search()/searchMany()): batched HNSW traversal over int8 rows,
exact int8 scan, float32 rerank + top-k, as hand-written HIP kernels for gfx950
behind the C ABI of include/mvdb_hip.h.

Only the hot path lives here; there is no CPU fallback.
"""
from ._native import LIB_PATH, MvdbError  # noqa: F401
from .bridge import HipBridge  # noqa: F401
from .config import PRESETS, derive_search_params, resolve_preset  # noqa: F401
from .db import MiniVectorDB  # noqa: F401

__all__ = ["HipBridge", "MiniVectorDB", "MvdbError", "PRESETS", "derive_search_params", "resolve_preset",
           "LIB_PATH"]
