"""HipBridge -- the ANN-index interface of MiniVectorDB on an MI355X.

Mirror of ``class WasmBridge`` (reference src/core/wasm-bridge.ts:24): same method
names, argument meaning, return shapes and error behaviour, backed by the C ABI
of libmvdb_hip.so instead of the AssemblyScript/WASM module.  The reference's
methods are async behind a promise-chain mutex (:39-50); here they are plain
synchronous calls guarded by one lock per bridge.

``searchMany`` returns the reference's ``{id, dist}[][]``; ``searchManyArrays``
returns the same data as numpy arrays (ids, dist, count) without building
nq x k Python objects.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import time
from typing import List, Optional, Sequence

import numpy as np

from . import _native
from ._native import Config, MvdbError

LOAD_ERROR = "WASM load_index failed (format mismatch / config mismatch / corrupt dump)."  # wasm-bridge.ts:457-459


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HipBridge:
    def __init__(self, device: int = -1):
        self._L = _native.lib()
        self._h = C.c_void_p()
        self._lock = threading.RLock()   # withLock, wasm-bridge.ts:39-50
        self._cfg = None                 # currentConfig, wasm-bridge.ts:28
        self._device = device
        self._max_ef = 0

    # ---- lifecycle -------------------------------------------------------
    def init(self, config: dict):
        """WasmBridge.init (wasm-bridge.ts:52-103); config keys: dim, m, ef,
        efSearch?, capacity, seed?, resultsCap?"""
        with self._lock:
            self.close()
            cfg = dict(config)
            seed = cfg.get("seed")
            if seed is None:  # wasm-bridge.ts:89-93
                seed = (int(time.time() * 1000) ^ (os.getpid() << 16)) & 0xFFFFFFFF
            c = Config(dim=int(cfg["dim"]), m=int(cfg["m"]), ef_construction=int(cfg["ef"]),
                       ef_search=int(cfg.get("efSearch") or 50), capacity=int(cfg["capacity"]),
                       results_cap=int(cfg.get("resultsCap") if cfg.get("resultsCap") is not None else 1000),
                       seed=int(seed) & 0xFFFFFFFF, device=int(self._device))
            if not self._L.mvdb_create(C.byref(c), C.byref(self._h)):
                self._h = C.c_void_p()
                raise MvdbError(self._L.mvdb_last_error(None).decode())
            self._cfg = cfg
            self._max_ef = self._L.mvdb_get_max_ef()

    def close(self):
        with self._lock:
            if self._h:
                self._L.mvdb_destroy(self._h)
                self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise MvdbError("WasmBridge not initialized")  # wasm-bridge.ts:202
        return self._h

    def _check(self, ok):
        if not ok:
            raise MvdbError(self._L.mvdb_last_error(self._h).decode())

    # ---- getters / setters ---------------------------------------------------
    def getMaxEf(self) -> int:                      # :112-121
        return self._max_ef or self._L.mvdb_get_max_ef()

    def setEfSearch(self, efSearch: int) -> None:   # :123-131 (ef <= 0 ignored)
        with self._lock:
            self._L.mvdb_set_ef_search(self.handle, int(efSearch))

    def _clampResultsCap(self, cap: int) -> int:    # :105-110
        x = int(cap)
        if x <= 0:
            return 1
        return min(x, self._max_ef) if self._max_ef > 0 else x

    def getResultsCap(self) -> int:                 # :170-177
        with self._lock:
            return self._L.mvdb_get_results_cap(self.handle)

    def setResultsCap(self, cap: int) -> None:      # :179-185
        with self._lock:
            self._L.mvdb_set_results_cap(self.handle, self._clampResultsCap(cap))

    def hasNode(self, id: int) -> bool:             # :187-193
        with self._lock:
            return bool(self._L.mvdb_has_node(self.handle, int(id)))

    def getMaxElements(self) -> int:
        return self._L.mvdb_get_max_elements(self.handle)

    def elementCount(self) -> int:
        return self._L.mvdb_element_count(self.handle)

    def setTuning(self, traverse_waves: int = 0, rerank_waves: int = 0) -> None:
        """Kernel variants of the timed path (mvdb_set_tuning); 0 = the engine's default."""
        with self._lock:
            self._check(self._L.mvdb_set_tuning(self.handle, int(traverse_waves), int(rerank_waves)))

    def checkErrors(self) -> None:
        """After synchronising `_dev` work: raise if a kernel of this handle reported a failure."""
        with self._lock:
            self._check(self._L.mvdb_check_errors(self.handle))

    def uploadCounters(self) -> dict:
        """Device-mirror traffic of this handle: full uploads, incremental patches, nodes patched."""
        out = (C.c_longlong * 3)()
        self._L.mvdb_upload_counters(self.handle, out)
        return dict(full_uploads=out[0], patches=out[1], nodes_patched=out[2])

    def setRelabel(self, mode):
        """The traversal's copy of the index in breadth-first label order (mvdb_set_relabel): True / 1 = kept when it
        makes lists local (the default), 2 = always, False / 0 = never."""
        self._L.mvdb_set_relabel(self.handle, int(mode))

    def relabelled(self) -> bool:
        return bool(self._L.mvdb_relabelled(self.handle))

    def leanFallbacks(self) -> dict:
        """Queries the lean traversal handed to the one-wave kernel, by reason (mvdb_lean_fallbacks)."""
        out = (C.c_longlong * 3)()
        self._L.mvdb_lean_fallbacks(self.handle, out)
        return dict(big_key=out[0], log_full=out[1], mismatch=out[2])

    def leanReplays(self) -> dict:
        """Queries whose result heap the replay kernel took from the admission log directly / rebuilt (mvdb_lean_replays)."""
        out = (C.c_longlong * 2)()
        self._L.mvdb_lean_replays(self.handle, out)
        return dict(short_cuts=out[0], full_replays=out[1])

    def _dim(self) -> int:
        return int(self._cfg["dim"])

    def _rows(self, a, dtype, what: str) -> np.ndarray:
        """nq x dim array of `dtype`; a wrong row length is an error, never a reinterpretation."""
        a = np.ascontiguousarray(a, dtype)
        dim = self._dim()
        if a.ndim == 1:
            if a.size % dim:
                raise MvdbError(f"{what}: length {a.size} is not a multiple of dim={dim}")
            a = a.reshape(-1, dim)
        if a.ndim != 2 or a.shape[1] != dim:
            raise MvdbError(f"{what}: expected rows of dim={dim}, got shape {a.shape}")
        return a

    def _allow(self, allow_bitmap):
        if allow_bitmap is None:
            return None
        allow = np.ascontiguousarray(allow_bitmap, np.uint32).reshape(-1)
        need = (max(self.getMaxElements(), 1) + 31) // 32
        if allow.size < need:
            raise MvdbError(f"allow bitmap too short: {allow.size} words for {self.getMaxElements()} ids ({need} needed)")
        return allow

    def reinitIndex(self, capacity: int) -> None:   # :199-240
        with self._lock:
            if self._cfg is None:
                raise MvdbError("WasmBridge not initialized")
            cfg = dict(self._cfg)
            cfg["capacity"] = int(capacity)
            self.init(cfg)

    # ---- write path (host graph; device mirror re-uploaded lazily) ---------------
    def insert(self, id: int, vectorI8) -> None:    # :242-247
        self.insertManyPacked(np.array([id], np.int32), np.asarray(vectorI8, np.int8).reshape(-1),
                              len(vectorI8))

    def insertMany(self, pairs: Sequence[dict]) -> None:  # :253-262
        if not pairs:
            return
        ids = np.array([p["id"] for p in pairs], np.int32)
        packed = np.concatenate([np.asarray(p["vectorI8"], np.int8).reshape(-1) for p in pairs])
        self.insertManyPacked(ids, packed, packed.size // len(pairs))

    def insertManyPacked(self, ids, packed, dim: int) -> None:  # :268-288
        ids = np.ascontiguousarray(ids, np.int32)
        packed = np.ascontiguousarray(packed, np.int8).reshape(-1)
        if ids.size == 0:
            return
        if packed.size != ids.size * dim:
            raise MvdbError(f"insertManyPacked length mismatch: packed={packed.size}, ids={ids.size}, dim={dim}")
        with self._lock:
            self._check(self._L.mvdb_insert_many(self.handle, _ptr(ids), _ptr(packed), ids.size))

    def updateVector(self, id: int, vectorI8) -> None:  # :290-299 update_and_reconnect: absent id = silent no-op
        with self._lock:
            if self.hasNode(id):
                self.insert(id, vectorI8)

    def buildParallel(self, packed, threads: int) -> None:
        """Benchmark-data builder (not in the reference): same insert algorithm,
        `threads` workers, ids 0..n-1."""
        packed = np.ascontiguousarray(packed, np.int8)
        n = packed.shape[0]
        with self._lock:
            self._check(self._L.mvdb_build_parallel(self.handle, _ptr(packed), n, int(threads)))

    def buildGpu(self, packed, max_block: int = 0) -> dict:
        """Bulk builder on the GPU (not in the reference; mvdb_build_gpu): blocks of new nodes linked with
        the reference's per-insert algorithm.  ids 0..n-1 into an empty index.  Returns the build's counters."""
        packed = self._rows(packed, np.int8, "buildGpu")
        st = np.zeros(8, np.float64)
        with self._lock:
            self._check(self._L.mvdb_build_gpu(self.handle, _ptr(packed), packed.shape[0], int(max_block), _ptr(st)))
        keys = ("seconds", "host_prefix_seconds", "n_dist", "n_expanded", "evictions", "removed_back_edges", "passes",
                "host_linked")
        return dict(zip(keys, st.tolist()))

    def appendGpu(self, packed, max_block: int = 0) -> dict:
        """Bulk insert into the live index (mvdb_append_gpu): ids elementCount .. elementCount + n - 1, dense ids
        required; the same block-parallel builder as buildGpu.  Returns the counters of the call."""
        packed = self._rows(packed, np.int8, "appendGpu")
        st = np.zeros(8, np.float64)
        with self._lock:
            self._check(self._L.mvdb_append_gpu(self.handle, _ptr(packed), packed.shape[0], int(max_block), _ptr(st)))
        keys = ("seconds", "host_prefix_seconds", "n_dist", "n_expanded", "evictions", "removed_back_edges", "passes",
                "host_linked")
        return dict(zip(keys, st.tolist()))

    # ---- search ---------------------------------------------------------------
    def searchManyArrays(self, vectorsI8, k: int, with_counters: bool = False):
        q = self._rows(vectorsI8, np.int8, "searchMany")
        nq, kk = q.shape[0], max(int(k), 1)
        ids = np.full((nq, kk), -1, np.int32)
        dist = np.zeros((nq, kk), np.float32)
        cnt = np.zeros(nq, np.int32)
        ctr = np.zeros((nq, 4), np.int32) if with_counters else None
        with self._lock:
            self._check(self._L.mvdb_search_many(self.handle, _ptr(q), nq, int(k), _ptr(ids), _ptr(dist),
                                                 _ptr(cnt), _ptr(ctr)))
        if with_counters:
            return ids, dist, cnt, ctr
        return ids, dist, cnt

    def searchMany(self, vectorsI8, k: int) -> List[List[dict]]:  # :326-355
        ids, dist, cnt = self.searchManyArrays(vectorsI8, k)
        return [[{"id": int(ids[q, i]), "dist": float(dist[q, i])} for i in range(cnt[q])]
                for q in range(ids.shape[0])]

    def search(self, vectorI8, k: int) -> List[dict]:  # :301-321
        v = np.asarray(vectorI8, np.int8).reshape(-1)
        if v.size != self._dim():
            raise MvdbError(f"search: vector length {v.size} != dim {self._dim()}")
        return self.searchMany(v.reshape(1, -1), k)[0]

    # ---- dump -----------------------------------------------------------------
    def save(self, filePath: str) -> None:          # :357-375
        with self._lock:
            n = self._L.mvdb_dump_size(self.handle)
            buf = np.empty(n, np.uint8)
            used = self._L.mvdb_save_dump(self.handle, _ptr(buf))
            with open(filePath, "wb") as f:
                f.write(buf[:used].tobytes())

    def saveBytes(self) -> bytes:
        with self._lock:
            n = self._L.mvdb_dump_size(self.handle)
            buf = np.empty(n, np.uint8)
            used = self._L.mvdb_save_dump(self.handle, _ptr(buf))
            return buf[:used].tobytes()

    def load(self, filePath: str) -> None:          # :377-468
        if not os.path.exists(filePath):            # :379 silent no-op
            return
        with open(filePath, "rb") as f:
            self.loadBytes(f.read())

    def loadBytes(self, data: bytes) -> None:
        buf = np.frombuffer(data, np.uint8)
        with self._lock:
            ok = self._L.mvdb_load_dump(self.handle, _ptr(buf), buf.size)
            if not ok:  # the engine already re-initialised a fresh empty index (:463-466)
                raise MvdbError(LOAD_ERROR)

    # ---- stages beyond the WasmBridge surface (compute parts of MiniVectorDB) ----
    def attachF32(self, base_f32, resident: bool = True) -> None:
        base = self._rows(base_f32, np.float32, "attachF32")
        with self._lock:
            self._check(self._L.mvdb_attach_f32(self.handle, _ptr(base), base.shape[0], 1 if resident else 0))

    def prepareQueries(self, q_f32):
        q = self._rows(q_f32, np.float32, "prepareQueries")
        qn = np.empty_like(q)
        qi = np.empty(q.shape, np.int8)
        with self._lock:
            self._check(self._L.mvdb_prepare_queries(self.handle, _ptr(q), q.shape[0], _ptr(qn), _ptr(qi)))
        return qn, qi

    def rerankMany(self, q_f32, cand, cand_count, ann_k: int, topk: int, allow_bitmap=None):
        q = self._rows(q_f32, np.float32, "rerankMany")
        cand = np.ascontiguousarray(cand, np.int32)
        cc = np.ascontiguousarray(cand_count, np.int32).reshape(-1)
        nq = q.shape[0]
        if cand.ndim != 2 or cand.shape[0] != nq or cc.size != nq:
            raise MvdbError(f"rerankMany: {nq} queries but cand {cand.shape}, cand_count {cc.shape}")
        out_ids = np.full((nq, max(topk, 1)), -1, np.int32)
        out_l2 = np.zeros((nq, max(topk, 1)), np.float64)
        out_n = np.zeros(nq, np.int32)
        allow = self._allow(allow_bitmap)
        with self._lock:
            self._check(self._L.mvdb_rerank_many(self.handle, _ptr(q), nq, _ptr(cand), _ptr(cc), cand.shape[1],
                                                 int(ann_k), _ptr(allow), int(topk), _ptr(out_ids), _ptr(out_l2),
                                                 _ptr(out_n)))
        return out_ids, out_l2, out_n

    def queryMany(self, q_f32, topk: int, ef_search: int, ann_k: int, allow_bitmap=None, with_counters=False):
        q = self._rows(q_f32, np.float32, "queryMany")
        nq = q.shape[0]
        out_ids = np.full((nq, max(topk, 1)), -1, np.int32)
        out_l2 = np.zeros((nq, max(topk, 1)), np.float64)
        out_n = np.zeros(nq, np.int32)
        ctr = np.zeros((nq, 4), np.int32)
        allow = self._allow(allow_bitmap)
        with self._lock:
            self._check(self._L.mvdb_query_many(self.handle, _ptr(q), nq, int(topk), int(ef_search), int(ann_k),
                                                _ptr(allow), _ptr(out_ids), _ptr(out_l2), _ptr(out_n), _ptr(ctr)))
        return (out_ids, out_l2, out_n, ctr) if with_counters else (out_ids, out_l2, out_n)

    # ---- several batches in flight from host buffers (mvdb_query_many_submit / _wait; slots 0..3)
    def submitMany(self, slot: int, q_f32, topk: int, ef_search: int, ann_k: int, allow_bitmap=None):
        q = self._rows(q_f32, np.float32, "submitMany")
        allow = self._allow(allow_bitmap)
        with self._lock:
            self._check(self._L.mvdb_query_many_submit(self.handle, int(slot), _ptr(q), q.shape[0], int(topk),
                                                       int(ef_search), int(ann_k), _ptr(allow)))
        self._slot_shape = getattr(self, "_slot_shape", {})
        self._slot_shape[int(slot)] = (q.shape[0], int(topk))

    def waitMany(self, slot: int):
        shape = getattr(self, "_slot_shape", {}).pop(int(slot), None)
        if shape is None:
            raise MvdbError(f"waitMany: nothing was submitted on slot {slot}")
        nq, topk = shape
        out_ids = np.full((nq, max(topk, 1)), -1, np.int32)
        out_l2 = np.zeros((nq, max(topk, 1)), np.float64)
        out_n = np.zeros(nq, np.int32)
        with self._lock:
            self._check(self._L.mvdb_query_many_wait(self.handle, int(slot), _ptr(out_ids), _ptr(out_l2), _ptr(out_n)))
        return out_ids, out_l2, out_n

    def distMany(self, q_i8, ids, metric: str = "l2"):
        q = self._rows(q_i8, np.int8, "distMany")
        ids = np.ascontiguousarray(ids, np.int32)
        nq, per_q = ids.shape
        if q.shape[0] != nq:
            raise MvdbError(f"distMany: {q.shape[0]} queries but {nq} id rows")
        out = np.zeros((nq, per_q), np.int32)
        with self._lock:
            self._check(self._L.mvdb_dist_many(self.handle, _ptr(q), nq, _ptr(ids), per_q,
                                               0 if metric == "l2" else 1, _ptr(out)))
        return out

    def scanMany(self, q_i8, k: int, metric: str = "l2"):
        q = self._rows(q_i8, np.int8, "scanMany")
        nq = q.shape[0]
        ids = np.full((nq, max(k, 1)), -1, np.int32)
        val = np.zeros((nq, max(k, 1)), np.int32)
        cnt = np.zeros(nq, np.int32)
        with self._lock:
            self._check(self._L.mvdb_scan_many(self.handle, _ptr(q), nq, int(k), 0 if metric == "l2" else 1,
                                               _ptr(ids), _ptr(val), _ptr(cnt)))
        return ids, val, cnt
