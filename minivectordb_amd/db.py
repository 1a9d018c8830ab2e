"""MiniVectorDB -- the search()/searchMany() path of the reference facade on an MI355X.

Mirror of the hot-path half of ``class MiniVectorDB`` (reference
src/mini-vector-db.ts:41): same option names, result shapes, on-disk files
(``vectors.f32.bin``, ``dump.bin``, ``ann.oplog``) and ef/annK derivation.  The
compute of search()/searchMany() (:474-546, :560-655) -- query normalise+quantise,
HNSW traversal, candidate filtering, float32 rerank, top-k -- runs on the GPU
through one fused C-ABI call.

Out of scope: the embedder (vector inputs only), the LokiJS
metadata store (replaced by a minimal in-memory map persisted as JSON -- NOT
the reference's metadata.json format), JSONL import/export, compaction, HTTP.
"""
from __future__ import annotations

import json
import os
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

import numpy as np

from .bridge import HipBridge, MvdbError
from .config import derive_search_params, resolve_preset

ScoreMode = str  # "l2" | "cosine" | "sim"


def _to_score(l2: float, mode: ScoreMode) -> float:
    """toScore, src/mini-vector-db.ts:995-1001 (host side, f64)."""
    if mode == "l2":
        return l2
    cosine = 1 - l2 / 2
    if mode == "cosine":
        return max(-1.0, min(1.0, cosine))
    return max(0.0, min(1.0, 1 - l2 / 4))


class MiniVectorDB:
    def __init__(self, opts: Dict[str, Any]):
        o = dict(opts)
        self.storageDir = o.get("storageDir") or os.path.join(os.getcwd(), "data")
        collection = o.get("collection") or ""
        self.prefix = f"{collection}." if collection else ""
        self.mode = o.get("mode") or "balanced"
        preset = resolve_preset(self.mode)
        self.cfg = dict(
            dim=int(o.get("dim") or 384),
            capacity=int(o.get("capacity") or 1_200_000),          # src/config.ts:84
            m=int(o.get("m") or preset["m"]),
            ef_construction=int(o.get("ef_construction") or preset["ef_construction"]),
            baseEfSearch=preset["baseEfSearch"], rerankMultiplier=preset["rerankMultiplier"],
            maxAnnK=preset["maxAnnK"], resultsCap=preset["resultsCap"],
            preloadVectors=bool(o.get("preloadVectors", False)), seed=o.get("seed"),
            bulkInsert=o.get("bulkInsert"), bulkInsertMin=int(o.get("bulkInsertMin") or 4096),  # engine options
        )
        p = lambda name: os.path.join(self.storageDir, self.prefix + name)  # :146-150
        self.dumpPath, self.vecPath, self.oplogPath = p("dump.bin"), p("vectors.f32.bin"), p("ann.oplog")
        self.metaPath, self.statePath = p("metadata.json"), p("state.json")
        self.wasm = HipBridge(device=int(o.get("device", -1)))     # the ANN-index seam (:47)
        self.wasmMaxEf = 0
        self.isReady = False
        # minimal metadata store: external id <-> internal id, soft delete, metadata
        self._ext2int: Dict[str, int] = {}
        self._items: Dict[int, Dict[str, Any]] = {}
        self._next_id = 0
        self._vec = np.zeros((0, self.cfg["dim"]), np.float32)      # vecCache / file image
        self._f32_dirty = True

    # ---- open / init / load -------------------------------------------------------
    @classmethod
    def open(cls, opts: Dict[str, Any]) -> "MiniVectorDB":       # :121-133
        db = cls(opts)
        db.init()
        try:
            db.load()
        except Exception:                                         # :127-131 load errors are swallowed
            pass
        return db

    def init(self):                                               # :183-218
        os.makedirs(self.storageDir, exist_ok=True)
        c = self.cfg
        self.wasm.init(dict(dim=c["dim"], m=c["m"], ef=c["ef_construction"], efSearch=c["baseEfSearch"],
                            capacity=c["capacity"], seed=c["seed"], resultsCap=c["resultsCap"]))
        self.wasmMaxEf = self.wasm.getMaxEf()
        self.capacity = c["capacity"]
        self.isReady = True

    def load(self):                                               # :710-749
        if os.path.exists(self.metaPath):
            with open(self.metaPath) as f:
                st = json.load(f)
            self._items = {int(k): v for k, v in st["items"].items()}
            self._ext2int = {v["external_id"]: k for k, v in self._items.items()}
            self._next_id = int(st["next_id"])
        if os.path.exists(self.vecPath):
            raw = np.fromfile(self.vecPath, dtype="<f4")
            self._vec = raw[: raw.size // self.cfg["dim"] * self.cfg["dim"]].reshape(-1, self.cfg["dim"]).copy()
            self._f32_dirty = True
        loaded = False
        if os.path.exists(self.dumpPath):
            try:
                self.wasm.load(self.dumpPath)
                loaded = True
            except MvdbError:
                loaded = False
        active = [i for i, it in self._items.items() if not it["deleted"]]
        if not loaded and active:                                 # autoRebuildOnLoad :738-745
            self._reinsert(sorted(active))
        else:
            self._replay_oplog()                                  # :747, :1544-1587

    def _replay_oplog(self):
        if not os.path.exists(self.oplogPath):
            return
        ids = []
        with open(self.oplogPath) as f:
            for line in f:
                parts = line.split()
                if len(parts) == 2 and parts[0] == "U":          # `D` lines are ignored on replay (:1562-1563)
                    ids.append(int(parts[1]))
        alive = [i for i in dict.fromkeys(ids) if i in self._items and not self._items[i]["deleted"]
                 and i < self._vec.shape[0]]
        self._reinsert(alive)

    def _reinsert(self, ids: Sequence[int]):
        if not len(ids):
            return
        ids = np.asarray(ids, np.int32)
        _, q8 = self.wasm.prepareQueries(self._vec[ids])          # re-quantise from the f32 store (:1304-1312)
        self.wasm.insertManyPacked(ids, q8.reshape(-1), self.cfg["dim"])

    # ---- write path (host) -----------------------------------------------------------
    def insert(self, item: Dict[str, Any]):                       # :220-243
        self.insertMany([item])

    def insertMany(self, items: Sequence[Dict[str, Any]]):        # :245-384 (vector inputs only)
        if not items:
            return
        dim = self.cfg["dim"]
        vecs = np.zeros((len(items), dim), np.float32)
        for i, it in enumerate(items):
            v = np.asarray(it["input"], np.float32).reshape(-1)
            if v.size != dim:
                raise ValueError(f"Vector dimension mismatch. Expected {dim}, got {v.size}")
            vecs[i] = v
        qn, q8 = self.wasm.prepareQueries(vecs)                   # normalise + quantise, bit-exact (:286-319)
        ids = []
        for it in items:
            ext = str(it["id"])
            if ext in self._ext2int:
                iid = self._ext2int[ext]
            else:
                iid = self._next_id
                if iid < 0 or iid >= self.capacity:               # :1141-1158
                    raise MvdbError(f"Database capacity exceeded: internalId={iid}, capacity={self.capacity}. "
                                    f"Fix: call db.rebuild({{ capacity: {self.capacity * 2} }}) or open with "
                                    "larger capacity.")
                self._next_id += 1
                self._ext2int[ext] = iid
            self._items[iid] = dict(external_id=ext, metadata=it.get("metadata"), deleted=False)
            ids.append(iid)
        need = max(ids) + 1
        if need > self._vec.shape[0]:
            grown = np.zeros((need, dim), np.float32)
            grown[: self._vec.shape[0]] = self._vec
            self._vec = grown
        self._vec[ids] = qn
        self._f32_dirty = True
        with open(self.vecPath, "r+b" if os.path.exists(self.vecPath) else "w+b") as f:  # :1038-1070
            for iid, row in zip(ids, qn):
                f.seek(iid * dim * 4)
                f.write(row.astype("<f4").tobytes())
        ids_a = np.asarray(ids, np.int32)
        first = self.wasm.elementCount()
        # engine option (not in the reference): a large batch of fresh consecutive ids goes through the GPU builder
        # (mvdb_append_gpu: same per-insert algorithm, block-parallel edge order) instead of one insert at a time
        bulk = (self.cfg.get("bulkInsert") == "gpu" and len(ids) >= int(self.cfg.get("bulkInsertMin", 4096))
                and np.array_equal(ids_a, np.arange(first, first + len(ids), dtype=np.int32)))
        if bulk:
            try:
                self.wasm.appendGpu(q8)
            except MvdbError:                                     # ids not dense (holes below): the ordinary path
                bulk = False
        if not bulk:
            self.wasm.insertManyPacked(ids_a, q8.reshape(-1), dim)                       # :356
        with open(self.oplogPath, "a") as f:                      # :358-361
            f.write("".join(f"U {i}\n" for i in ids))

    def removeMany(self, ext_ids: Sequence[str]) -> int:          # :386-434 (soft delete)
        n = 0
        lines = []
        for ext in ext_ids:
            iid = self._ext2int.get(str(ext))
            if iid is None or self._items[iid]["deleted"]:
                continue
            self._items[iid]["deleted"] = True
            lines.append(f"D {iid}\n")
            n += 1
        if lines:
            with open(self.oplogPath, "a") as f:
                f.write("".join(lines))
        return n

    def remove(self, ext_id: str) -> bool:
        return self.removeMany([ext_id]) > 0

    # ---- the hot path --------------------------------------------------------------------
    def _allow_bitmap(self, flt) -> Optional[np.ndarray]:
        """Inter-stage filter of :595-621 as a bitmap over internal ids: bit set iff the item
        exists, is not deleted, passes the object filter and the predicate."""
        n_bits = max(self.wasm.getMaxElements(), 1)
        bits = np.zeros((n_bits + 31) // 32, np.uint32)
        obj = flt if isinstance(flt, dict) else None
        pred = flt if callable(flt) else None
        for iid, it in self._items.items():
            if it["deleted"] or iid >= n_bits:
                continue
            md = it.get("metadata") or {}
            if obj is not None and any(md.get(k) != v for k, v in obj.items()):
                continue
            if pred is not None and not pred(it.get("metadata")):
                continue
            bits[iid >> 5] |= np.uint32(1 << (iid & 31))
        return bits

    def _sync_f32(self):
        if self._f32_dirty:
            self.wasm.attachF32(self._vec if self._vec.shape[0] else np.zeros((1, self.cfg["dim"]), np.float32),
                                resident=self.cfg["preloadVectors"])
            self._f32_dirty = False

    def searchMany(self, queries, opts: Optional[Dict[str, Any]] = None) -> List[List[Dict[str, Any]]]:  # :560-655
        opts = opts or {}
        if not len(queries):
            return []
        dim = self.cfg["dim"]
        topK, efSearch, annK = derive_search_params(opts.get("topK", 10), self.cfg, self.wasmMaxEf)  # :568-576
        scoreMode = opts.get("score") or "l2"
        qs = np.zeros((len(queries), dim), np.float32)
        for i, q in enumerate(queries):
            v = np.asarray(q, np.float32).reshape(-1)
            if v.size != dim:
                raise ValueError(f"Query[{i}] dim mismatch. Expected {dim}, got {v.size}")  # :584-588
            qs[i] = v
        self._sync_f32()
        allow = self._allow_bitmap(opts.get("filter"))
        ids, l2, n = self.wasm.queryMany(qs, topK, efSearch, annK, allow)
        out = []
        for qi in range(len(queries)):
            row = []
            for j in range(n[qi]):
                it = self._items.get(int(ids[qi, j]))
                if it is None or it["deleted"]:                   # :641-642 re-check
                    continue
                row.append(dict(id=it["external_id"], score=_to_score(float(l2[qi, j]), scoreMode),
                                metadata=it.get("metadata")))
            out.append(row)
        return out

    def search(self, query, opts: Optional[Dict[str, Any]] = None) -> List[Dict[str, Any]]:  # :474-546
        q = np.asarray(query, np.float32).reshape(-1)
        if q.size != self.cfg["dim"]:
            raise ValueError(f"Query vector dimension mismatch. Expected {self.cfg['dim']}, got {q.size}")
        return self.searchMany([q], opts)[0]

    # ---- persistence ---------------------------------------------------------------------------
    def _save_meta(self):
        with open(self.metaPath + ".tmp", "w") as f:
            json.dump(dict(items={str(k): v for k, v in self._items.items()}, next_id=self._next_id), f)
        os.replace(self.metaPath + ".tmp", self.metaPath)

    def save(self):                                               # :692, :1589-1632
        self._save_meta()
        self.wasm.save(self.dumpPath + ".tmp")
        os.replace(self.dumpPath + ".tmp", self.dumpPath)
        with open(self.statePath, "w") as f:
            json.dump(dict(dim=self.cfg["dim"], m=self.cfg["m"], ef_construction=self.cfg["ef_construction"],
                           capacity=self.capacity), f)
        open(self.oplogPath, "w").close()                          # oplog truncated after a snapshot (:1619-1631)

    def getStats(self) -> Dict[str, Any]:                         # :915-931
        active = sum(1 for it in self._items.values() if not it["deleted"])
        return dict(dim=self.cfg["dim"], mode=self.mode, capacity=self.capacity, items=active,
                    deleted=len(self._items) - active, m=self.cfg["m"], ef_construction=self.cfg["ef_construction"])

    def close(self):                                              # :943-957 (the metadata store is flushed)
        if self.isReady:
            self._save_meta()
        self.wasm.close()
        self.isReady = False
