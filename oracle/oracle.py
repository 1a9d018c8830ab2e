"""ctypes wrapper over oracle/libmvdb_oracle.so -- the CPU ORACLE.

TEST INFRASTRUCTURE ONLY (see oracle/mvdb_oracle.h).  Import this from tests/,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py -- never
from the product package ``minivectordb_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmvdb_oracle.so")

MAX_EF = 4096


class Counters(C.Structure):
    _fields_ = [("n_dist", C.c_int64), ("n_l0lists", C.c_int64),
                ("n_upperlists", C.c_int64), ("n_found", C.c_int64)]


COUNTER_DTYPE = np.dtype([("n_dist", "<i8"), ("n_l0lists", "<i8"),
                          ("n_upperlists", "<i8"), ("n_found", "<i8")])


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "mvdb_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    vp, i32, u32, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t
    L.mvo_create.restype = vp
    L.mvo_create.argtypes = [i32, i32, i32, i32, i32, u32, i32]
    L.mvo_destroy.argtypes = [vp]
    for name in ("mvo_set_ef_search", "mvo_set_results_cap"):
        getattr(L, name).argtypes = [vp, i32]
    for name in ("mvo_get_ef_search", "mvo_get_results_cap", "mvo_was_ef_clamped",
                 "mvo_get_max_elements", "mvo_element_count", "mvo_entry_point",
                 "mvo_max_level"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = i32
    L.mvo_clear_ef_clamped.argtypes = [vp]
    L.mvo_get_max_ef.restype = i32
    L.mvo_has_node.argtypes = [vp, i32]
    L.mvo_has_node.restype = i32
    L.mvo_node_level.argtypes = [vp, i32]
    L.mvo_node_level.restype = i32
    L.mvo_seed_rng.argtypes = [vp, u32]
    for name in ("mvo_dist_l2_i8", "mvo_dot_i8"):
        getattr(L, name).argtypes = [vp, vp, i32]
        getattr(L, name).restype = C.c_int32
    for name in ("mvo_dist_l2_sq", "mvo_dist_dot"):
        getattr(L, name).argtypes = [vp, vp, i32]
        getattr(L, name).restype = C.c_float
    L.mvo_insert.argtypes = [vp, i32, vp]
    L.mvo_update_vector.argtypes = [vp, i32, vp]
    L.mvo_insert_many.argtypes = [vp, vp, vp, i32]
    L.mvo_build_parallel.argtypes = [vp, vp, i32, i32]
    L.mvo_search.argtypes = [vp, vp, i32, vp, vp, vp]
    L.mvo_search.restype = i32
    L.mvo_search_many.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, i32]
    L.mvo_dump_size.argtypes = [vp]
    L.mvo_dump_size.restype = sz
    L.mvo_save.argtypes = [vp, vp]
    L.mvo_save.restype = sz
    L.mvo_load.argtypes = [vp, vp, sz]
    L.mvo_load.restype = i32
    L.mvo_normalize_f32.argtypes = [vp, i32]
    L.mvo_quantize_i8.argtypes = [vp, i32, vp]
    L.mvo_l2sq_f32.argtypes = [vp, vp, i32]
    L.mvo_l2sq_f32.restype = C.c_double
    L.mvo_to_score.argtypes = [C.c_double, i32]
    L.mvo_to_score.restype = C.c_double
    L.mvo_derive_search_params.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp]
    L.mvo_next_results_cap.argtypes = [i32, i32, i32]
    L.mvo_next_results_cap.restype = i32
    L.mvo_rerank_many.argtypes = [vp, i32, i32, vp, C.c_int64, vp, vp, i32, i32, vp, i32,
                                  vp, vp, vp, i32]
    L.mvo_scan_many.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, i32]
    L.mvo_exact_f32_topk.argtypes = [vp, i32, i32, vp, C.c_int64, i32, vp, vp, i32]
    _lib = L
    return L


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _i8(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.int8)
    return a


# presets, src/config.ts:23-54
PRESETS = {
    "fast": dict(m=12, ef_construction=150, baseEfSearch=80, rerankMultiplier=40,
                 maxAnnK=2048, resultsCap=2048),
    "balanced": dict(m=16, ef_construction=300, baseEfSearch=140, rerankMultiplier=120,
                     maxAnnK=4096, resultsCap=4096),
    "accurate": dict(m=24, ef_construction=600, baseEfSearch=200, rerankMultiplier=250,
                     maxAnnK=4096, resultsCap=4096),
}


class OracleIndex:
    """The WASM index + WasmBridge method surface (src/core/wasm-bridge.ts:24)."""

    def __init__(self, dim, m, ef_construction, capacity, ef_search=50, results_cap=1000, seed=42):
        self.L = lib()
        self.dim, self.m, self.efc = int(dim), int(m), int(ef_construction)
        self.h = self.L.mvo_create(dim, m, ef_construction, ef_search, results_cap, seed, capacity)
        if not self.h:
            raise ValueError("invalid config (update_config would trap, assembly/config.ts:95-100)")

    def close(self):
        if self.h:
            self.L.mvo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- bridge getters / setters
    def get_max_ef(self):
        return self.L.mvo_get_max_ef()

    def set_ef_search(self, ef):
        self.L.mvo_set_ef_search(self.h, int(ef))

    def get_ef_search(self):
        return self.L.mvo_get_ef_search(self.h)

    def set_results_cap(self, cap):
        cap = int(cap)  # clampResultsCap, wasm-bridge.ts:105-110
        if cap <= 0:
            cap = 1
        self.L.mvo_set_results_cap(self.h, min(cap, MAX_EF))

    def get_results_cap(self):
        return self.L.mvo_get_results_cap(self.h)

    def has_node(self, i):
        return bool(self.L.mvo_has_node(self.h, int(i)))

    def node_level(self, i):
        return self.L.mvo_node_level(self.h, int(i))

    @property
    def max_elements(self):
        return self.L.mvo_get_max_elements(self.h)

    @property
    def element_count(self):
        return self.L.mvo_element_count(self.h)

    @property
    def entry_point(self):
        return self.L.mvo_entry_point(self.h)

    @property
    def max_level(self):
        return self.L.mvo_max_level(self.h)

    def was_ef_clamped(self):
        return bool(self.L.mvo_was_ef_clamped(self.h))

    # --- write path
    def insert(self, i, vec_i8):
        v = _i8(vec_i8)
        assert v.size == self.dim
        self.L.mvo_insert(self.h, int(i), _p(v))

    def insert_many(self, ids, packed_i8):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        p = _i8(packed_i8).reshape(-1)
        if p.size != ids.size * self.dim:  # wasm-bridge.ts:274-278
            raise ValueError(f"insertManyPacked length mismatch: packed={p.size}, ids={ids.size}, dim={self.dim}")
        self.L.mvo_insert_many(self.h, _p(ids), _p(p), ids.size)

    def build_parallel(self, packed_i8, threads):
        p = _i8(packed_i8).reshape(-1, self.dim)
        self.L.mvo_build_parallel(self.h, _p(p), p.shape[0], int(threads))

    # --- search
    def search(self, q_i8, k, with_counters=False):
        q = _i8(q_i8)
        kk = max(int(k), 0)
        ids = np.empty(max(kk, 1), np.int32)
        dist = np.empty(max(kk, 1), np.float32)
        ctr = Counters()
        n = self.L.mvo_search(self.h, _p(q), int(k), _p(ids), _p(dist), C.byref(ctr))
        if with_counters:
            return ids[:n].copy(), dist[:n].copy(), ctr
        return ids[:n].copy(), dist[:n].copy()

    def search_many(self, qs_i8, k, threads=1):
        q = _i8(qs_i8).reshape(-1, self.dim)
        nq, kk = q.shape[0], max(int(k), 0)
        ids = np.full((nq, max(kk, 1)), -1, np.int32)
        dist = np.zeros((nq, max(kk, 1)), np.float32)
        cnt = np.zeros(nq, np.int32)
        ctr = np.zeros(nq, COUNTER_DTYPE)
        self.L.mvo_search_many(self.h, _p(q), nq, int(k), _p(ids), _p(dist), _p(cnt), _p(ctr), int(threads))
        return ids, dist, cnt, ctr

    def scan_many(self, qs_i8, k, metric="l2", threads=1):
        q = _i8(qs_i8).reshape(-1, self.dim)
        nq = q.shape[0]
        ids = np.full((nq, max(k, 1)), -1, np.int32)
        val = np.zeros((nq, max(k, 1)), np.int32)
        cnt = np.zeros(nq, np.int32)
        self.L.mvo_scan_many(self.h, _p(q), nq, int(k), 0 if metric == "l2" else 1, _p(ids), _p(val), _p(cnt),
                             int(threads))
        return ids, val, cnt

    # --- dump
    def save_bytes(self) -> bytes:
        n = self.L.mvo_dump_size(self.h)
        buf = np.empty(n, np.uint8)
        used = self.L.mvo_save(self.h, _p(buf))
        return buf[:used].tobytes()

    def load_bytes(self, b: bytes) -> bool:
        buf = np.frombuffer(b, np.uint8)
        return bool(self.L.mvo_load(self.h, _p(buf), buf.size))


# ---- host-side functions of MiniVectorDB ----
def normalize_f32(v):
    v = np.array(v, dtype=np.float32).reshape(-1).copy()  # resolveToF32 copies, :1221-1223
    lib().mvo_normalize_f32(_p(v), v.size)
    return v


def normalize_rows_f32(a):
    a = np.array(a, dtype=np.float32, order="C").copy()
    L = lib()
    for r in a:
        L.mvo_normalize_f32(_p(r), r.size)
    return a


def quantize_i8(v):
    v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
    out = np.empty(v.size, np.int8)
    lib().mvo_quantize_i8(_p(v), v.size, _p(out))
    return out


def quantize_rows_i8(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty(a.shape, np.int8)
    lib().mvo_quantize_i8(_p(a), a.size, _p(out))  # element-wise, rows irrelevant
    return out


def l2sq_f32(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return lib().mvo_l2sq_f32(_p(a), _p(b), a.size)


def to_score(l2, mode="l2"):
    return lib().mvo_to_score(float(l2), {"l2": 0, "cosine": 1, "sim": 2}.get(mode, 2))


def dist_l2_i8(a, b):
    a, b = _i8(a), _i8(b)
    return lib().mvo_dist_l2_i8(_p(a), _p(b), a.size)


def dot_i8(a, b):
    a, b = _i8(a), _i8(b)
    return lib().mvo_dot_i8(_p(a), _p(b), a.size)


def derive_search_params(topk, preset="balanced", wasm_max_ef=MAX_EF, **over):
    p = dict(PRESETS[preset])
    p.update(over)
    tk, ef, ak = C.c_int(), C.c_int(), C.c_int()
    lib().mvo_derive_search_params(int(topk), p["baseEfSearch"], p["rerankMultiplier"], p["maxAnnK"],
                                   wasm_max_ef, C.byref(tk), C.byref(ef), C.byref(ak))
    return tk.value, ef.value, ak.value


def next_results_cap(cur, n, wasm_max_ef=MAX_EF):
    return lib().mvo_next_results_cap(int(cur), int(n), int(wasm_max_ef))


def rerank_many(q_f32, base_f32, cand, cand_count, ann_k, topk, allow_bitmap=None, threads=1):
    q = np.ascontiguousarray(q_f32, dtype=np.float32)
    base = np.ascontiguousarray(base_f32, dtype=np.float32)
    cand = np.ascontiguousarray(cand, dtype=np.int32)
    cc = np.ascontiguousarray(cand_count, dtype=np.int32)
    nq, dim = q.shape
    out_ids = np.full((nq, topk), -1, np.int32)
    out_l2 = np.zeros((nq, topk), np.float64)
    out_n = np.zeros(nq, np.int32)
    allow = None
    if allow_bitmap is not None:
        allow = np.ascontiguousarray(allow_bitmap, dtype=np.uint32)
    lib().mvo_rerank_many(_p(q), nq, dim, _p(base), base.shape[0], _p(cand), _p(cc), cand.shape[1], int(ann_k),
                          _p(allow) if allow is not None else None, int(topk), _p(out_ids), _p(out_l2), _p(out_n),
                          int(threads))
    return out_ids, out_l2, out_n


def exact_f32_topk(q_f32, base_f32, topk, threads=1):
    q = np.ascontiguousarray(q_f32, dtype=np.float32)
    base = np.ascontiguousarray(base_f32, dtype=np.float32)
    nq, dim = q.shape
    ids = np.empty((nq, topk), np.int32)
    l2 = np.empty((nq, topk), np.float64)
    lib().mvo_exact_f32_topk(_p(q), nq, dim, _p(base), base.shape[0], int(topk), _p(ids), _p(l2), int(threads))
    return ids, l2


def search_many_db(ix: OracleIndex, queries_f32, base_f32, topk=10, preset="balanced", allow_bitmap=None,
                   threads=1, score="l2"):
    """MiniVectorDB.searchMany (src/mini-vector-db.ts:560-655) over raw vectors,
    metadata reduced to an allow bitmap.  Returns (ids, l2, n, ann_ids, ann_cnt)."""
    tk, ef, ann_k = derive_search_params(topk, preset, ix.get_max_ef())
    ix.set_ef_search(ef)                                              # :572
    cur = ix.get_results_cap()                                        # :577
    nxt = next_results_cap(cur, ann_k, ix.get_max_ef())
    if nxt != cur:
        ix.set_results_cap(nxt)
    qf = normalize_rows_f32(queries_f32)                              # :583
    qi = quantize_rows_i8(qf)                                         # :590
    ids, dist, cnt, ctr = ix.search_many(qi, ann_k, threads=threads)  # :593
    out_ids, out_l2, out_n = rerank_many(qf, base_f32, ids, cnt, ann_k, tk, allow_bitmap, threads)
    return out_ids, out_l2, out_n, ids, cnt, ctr
