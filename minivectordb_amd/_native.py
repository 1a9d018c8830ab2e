"""ctypes binding of libmvdb_hip.so (the C ABI declared in include/mvdb_hip.h).

The library is the product: there is no Python or CPU fallback.  A missing
library raises at import of this module; a missing GPU raises at the first
compute call (status 0 from the C ABI -> MvdbError with the library's message).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVDB_HIP_LIB selects a tuning variant built by tools/ (same ABI, same kernels, other constants)
LIB_PATH = os.environ.get("MVDB_HIP_LIB") or os.path.join(_HERE, "lib", "libmvdb_hip.so")


class MvdbError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("dim", C.c_int32), ("m", C.c_int32), ("ef_construction", C.c_int32),
                ("ef_search", C.c_int32), ("capacity", C.c_int32), ("results_cap", C.c_int32),
                ("seed", C.c_uint32), ("device", C.c_int32)]


# every symbol include/mvdb_hip.h declares: (name, restype, argtypes)
_vp, _i, _sz, _f, _d, _i64 = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_double, C.c_int64
SYMBOLS = [
    ("mvdb_create", _i, [C.POINTER(Config), C.POINTER(_vp)]),
    ("mvdb_destroy", None, [_vp]),
    ("mvdb_last_error", C.c_char_p, [_vp]),
    ("mvdb_device_available", _i, []),
    ("mvdb_get_max_ef", _i, []),
    ("mvdb_set_ef_search", None, [_vp, _i]),
    ("mvdb_get_ef_search", _i, [_vp]),
    ("mvdb_set_results_cap", None, [_vp, _i]),
    ("mvdb_get_results_cap", _i, [_vp]),
    ("mvdb_has_node", _i, [_vp, _i]),
    ("mvdb_get_max_elements", _i, [_vp]),
    ("mvdb_get_dim", _i, [_vp]),
    ("mvdb_element_count", _i, [_vp]),
    ("mvdb_was_ef_clamped", _i, [_vp]),
    ("mvdb_clear_ef_clamped", None, [_vp]),
    ("mvdb_load_dump", _i, [_vp, _vp, _sz]),
    ("mvdb_dump_size", _sz, [_vp]),
    ("mvdb_save_dump", _sz, [_vp, _vp]),
    ("mvdb_insert_many", _i, [_vp, _vp, _vp, _i]),
    ("mvdb_update_vector", _i, [_vp, _i, _vp]),
    ("mvdb_build_parallel", _i, [_vp, _vp, _i, _i]),
    ("mvdb_build_gpu", _i, [_vp, _vp, _i, _i, _vp]),
("mvdb_append_gpu", _i, [_vp, _vp, _i, _i, _vp]),
    ("mvdb_attach_f32", _i, [_vp, _vp, _i64, _i]),
    ("mvdb_prepare_queries", _i, [_vp, _vp, _i, _vp, _vp]),
    ("mvdb_prepare_queries_dev", _i, [_vp, _vp, _i, _vp, _vp, _vp]),
    ("mvdb_search_many", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    ("mvdb_search_many_dev", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    ("mvdb_upload", _i, [_vp]),
    ("mvdb_search_many_dev_noalloc", _i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    ("mvdb_search", _i, [_vp, _vp, _i, _vp, _vp]),
    ("mvdb_rerank_many", _i, [_vp, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    ("mvdb_rerank_many_dev", _i, [_vp, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    ("mvdb_query_many_dev", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mvdb_query_many", _i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    ("mvdb_query_many_submit", _i, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    ("mvdb_query_many_wait", _i, [_vp, _i, _vp, _vp, _vp]),
    ("mvdb_query_many_slot_shape", _i, [_vp, _i, _vp, _vp]),
    ("mvdb_derive_params", None, [_i, _i, _i, _i, _vp, _vp, _vp]),
    ("mvdb_ensure_results_cap_at_least", None, [_vp, _i]),
    ("mvdb_to_score", _d, [_d, _i]),
    ("mvdb_dist_many", _i, [_vp, _vp, _i, _vp, _i, _i, _vp]),
    ("mvdb_scan_many", _i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    ("mvdb_scan_many_dev", _i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    ("mvdb_merge_topk_dev", _i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    ("mvdb_topk_block_bytes", _sz, [_i, _i]),
    ("mvdb_merge_topk_packed_dev", _i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    ("mvdb_set_tuning", _i, [_vp, _i, _i]),
    ("mvdb_check_errors", _i, [_vp]),
    ("mvdb_upload_counters", None, [_vp, _vp]),
    ("mvdb_debug_inject_abort", None, [_vp, _i]),
    ("mvdb_debug_inject_build_failure", None, [_vp, _i]),
    ("mvdb_set_relabel", None, [_vp, _i]),
    ("mvdb_relabelled", _i, [_vp]),
    ("mvdb_lean_fallbacks", None, [_vp, _vp]),
    ("mvdb_lean_replays", None, [_vp, _vp]),
    ("mvdb_debug_set_lean_log_cap", None, [_vp, _i]),
    ("mvdb_release_stream", _i, [_vp, _vp]),
    ("mvdb_enable_timing", None, [_vp, _i]),
    ("mvdb_last_stage_ms", _f, [_vp, _i]),
    ("mvdb_stage_stats", _i, [_vp, _i, _vp, _vp, _i]),
]

_lib = None


def lib() -> C.CDLL:
    """Load the shared library (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MvdbError(f"{LIB_PATH} is missing: run `python -m minivectordb_amd.build` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError here = ABI drift, keep it loud
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
