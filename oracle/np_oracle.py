"""Second, independent restatement of the hot path in numpy / pure Python.

TEST INFRASTRUCTURE ONLY.  Exists to cross-check oracle/mvdb_oracle.c: the two
were written separately against the reference lines cited below and must agree
bit for bit (tests/test_oracle_cross.py).  Pure-Python loops: small cases only.
"""
from __future__ import annotations

import math
import struct

import numpy as np

MAX_EF = 4096            # assembly/hnsw.ts:30
MAGIC = 0x57534E48       # assembly/hnsw.ts:1033
F32_EMPTY_WORST = float(np.float32(99999999.0))  # hnsw.ts:216-219 (f32 literal -> 1.0e8)


# ---- src/mini-vector-db.ts host math ---------------------------------------
def normalize_f32(v):
    """normalizeF32InPlace, src/mini-vector-db.ts:963-970 (f64 math, f32 store)."""
    v = np.array(v, dtype=np.float32)
    acc = 0.0
    for x in v.tolist():          # strictly sequential f64 accumulation
        acc += x * x
    if acc <= 0:
        return v
    inv = 1.0 / math.sqrt(acc)
    return (v.astype(np.float64) * inv).astype(np.float32)


def quantize_i8(v):
    """quantizeToI8, src/mini-vector-db.ts:972-984; Math.round ties go to +inf."""
    out = np.zeros(len(v), np.int8)
    for i, x in enumerate(np.asarray(v, dtype=np.float32).tolist()):
        if x > 1:
            x = 1.0
        elif x < -1:
            x = -1.0
        y = x * 127
        if y != y:
            q = 0                      # Int8Array store of NaN
        else:
            q = math.floor(y + 0.5)    # Math.round
            q = 127 if q > 127 else (-127 if q < -127 else q)
        out[i] = q
    return out


def l2sq_f32(a, b):
    """l2SqF32, src/mini-vector-db.ts:986-993 (sequential f64)."""
    s = 0.0
    for x, y in zip(np.asarray(a, np.float32).tolist(), np.asarray(b, np.float32).tolist()):
        d = x - y
        s += d * d
    return s


def to_score(l2, mode):
    """toScore, src/mini-vector-db.ts:995-1001."""
    if mode == "l2":
        return l2
    cosine = 1 - l2 / 2
    if mode == "cosine":
        return max(-1.0, min(1.0, cosine))
    return max(0.0, min(1.0, 1 - l2 / 4))


# ---- assembly/math.ts -------------------------------------------------------
def dist_l2_i8(a, b):
    d = np.asarray(a, np.int8).astype(np.int64) - np.asarray(b, np.int8).astype(np.int64)
    return int((d * d).sum())


def dot_i8(a, b):
    return int((np.asarray(a, np.int8).astype(np.int64) * np.asarray(b, np.int8).astype(np.int64)).sum())


def dist_l2_sq(a, b):
    return float(np.float32(dist_l2_i8(a, b)))   # math.ts:126 <f32> cast


# ---- dump V3 parser: assembly/hnsw.ts:1129-1249 ----------------------------
class Graph:
    pass


def parse_dump(buf: bytes, dim, m, ef_construction, max_layers=4):
    """Returns a Graph or None (= load_index returned 0)."""
    m_max0 = 2 * m
    if len(buf) < 48:
        return None
    hdr = struct.unpack_from("<12i", buf, 0)
    magic, ver, d_dim, d_m, d_m0, d_ef, d_layers, max_el, present, entry, max_level, rcap = hdr
    if magic != MAGIC or ver != 3:
        return None
    if (d_dim, d_m, d_m0, d_ef, d_layers) != (dim, m, m_max0, ef_construction, max_layers):
        return None
    if max_el <= 0 or present < 0 or present > max_el:
        return None
    if entry != -1 and not (0 <= entry < max_el):
        return None
    if max_level < -1 or max_level >= max_layers:
        return None
    if present > 0 and entry == -1:
        return None
    g = Graph()
    g.dim, g.m, g.m_max0, g.max_elements = dim, m, m_max0, max_el
    g.entry, g.max_level, g.results_cap = entry, max_level, rcap
    g.level = {}
    g.vec = {}
    g.lists = {}      # id -> [ (count, [cap ids]) per layer ]
    off = 48
    for _ in range(present):
        if off + 8 > len(buf):
            return None
        nid, lvl = struct.unpack_from("<2i", buf, off)
        off += 8
        if nid < 0 or nid >= max_el or lvl < 0 or lvl >= max_layers:
            return None
        if off + dim > len(buf):
            return None
        g.vec[nid] = np.frombuffer(buf, np.int8, dim, off).copy()
        off += dim
        ll = []
        for l in range(lvl + 1):
            cap = m_max0 if l == 0 else m
            if off + 4 + 4 * cap > len(buf):
                return None
            cnt = struct.unpack_from("<i", buf, off)[0]
            off += 4
            if cnt < 0 or cnt > cap:
                return None
            ids = list(struct.unpack_from("<%di" % cap, buf, off))
            off += 4 * cap
            ids = [x if 0 <= x < max_el else -1 for x in ids]   # :1222-1229
            ll.append((cnt, ids))
        g.level[nid] = lvl
        g.lists[nid] = ll
    if entry != -1 and entry not in g.lists:
        return None
    return g


# ---- heaps: assembly/hnsw.ts:141-325 ----------------------------------------
def _cand_push(ids, ds, nid, d):                    # :153-168
    ids.append(nid)
    ds.append(d)
    i = len(ds) - 1
    while i > 0:
        p = (i - 1) >> 1
        if ds[p] <= ds[i]:
            break
        ids[i], ids[p] = ids[p], ids[i]
        ds[i], ds[p] = ds[p], ds[i]
        i = p


def _cand_pop(ids, ds):                             # :173-213
    top = (ids[0], ds[0])
    lid, ld = ids.pop(), ds.pop()
    n = len(ds)
    if n == 0:
        return top
    ids[0], ds[0] = lid, ld
    i = 0
    while True:
        l = 2 * i + 1
        if l >= n:
            break
        r = l + 1
        sm = l
        if r < n and ds[r] < ds[l]:
            sm = r
        if ds[i] <= ds[sm]:
            break
        ids[i], ids[sm] = ids[sm], ids[i]
        ds[i], ds[sm] = ds[sm], ds[i]
        i = sm
    return top


def _res_push(ids, ds, nid, d):                     # :233-248
    ids.append(nid)
    ds.append(d)
    i = len(ds) - 1
    while i > 0:
        p = (i - 1) >> 1
        if ds[p] >= ds[i]:
            break
        ids[i], ids[p] = ids[p], ids[i]
        ds[i], ds[p] = ds[p], ds[i]
        i = p


def _res_down(ids, ds):                             # :268-287 / :295-314
    n = len(ds)
    i = 0
    while True:
        l = 2 * i + 1
        if l >= n:
            break
        r = l + 1
        lg = l
        if r < n and ds[r] > ds[l]:
            lg = r
        if ds[i] >= ds[lg]:
            break
        ids[i], ids[lg] = ids[lg], ids[i]
        ds[i], ds[lg] = ds[lg], ds[i]
        i = lg


def _res_pop(ids, ds):                              # :250-289
    top = (ids[0], ds[0])
    lid, ld = ids.pop(), ds.pop()
    if ds:
        ids[0], ds[0] = lid, ld
        _res_down(ids, ds)
    return top


# ---- search: assembly/hnsw.ts:423-523, :960-1027 ------------------------------
def _layer_list(g, nid, layer):
    ll = g.lists.get(nid)
    if ll is None or layer >= len(ll):
        return None
    return ll[layer]


def search_layer0(g, q, entry, ef, counters):
    if ef <= 0:
        return [], []
    ef = min(ef, MAX_EF)
    if entry < 0 or entry >= g.max_elements or entry not in g.lists:
        return [], []
    visited = set()
    cand_cap = max(min(ef * 2 + 32, MAX_EF), ef)
    c_ids, c_ds, r_ids, r_ds = [], [], [], []
    d0 = dist_l2_sq(q, g.vec[entry])
    counters["n_dist"] += 1
    visited.add(entry)
    _cand_push(c_ids, c_ds, entry, d0)
    _res_push(r_ids, r_ds, entry, d0)
    while c_ds:
        cur, cur_d = _cand_pop(c_ids, c_ds)
        if cur < 0:
            continue
        worst = r_ds[0] if r_ds else F32_EMPTY_WORST
        if len(r_ds) >= ef and cur_d > worst:
            break
        lst = _layer_list(g, cur, 0)
        if lst is None:
            continue
        cnt, nbrs = lst
        if cnt < 0:
            continue
        cnt = min(cnt, g.m_max0)
        counters["n_l0lists"] += 1
        for nb in nbrs[:cnt]:
            if nb < 0 or nb >= g.max_elements or nb not in g.lists:
                continue
            if nb in visited:
                continue
            visited.add(nb)
            d = dist_l2_sq(q, g.vec[nb])
            counters["n_dist"] += 1
            worst = r_ds[0] if r_ds else F32_EMPTY_WORST
            if len(r_ds) < ef or d < worst:
                if len(c_ds) < cand_cap:
                    _cand_push(c_ids, c_ds, nb, d)
                if len(r_ds) < ef:
                    _res_push(r_ids, r_ds, nb, d)
                elif d < r_ds[0]:
                    r_ids[0], r_ds[0] = nb, d
                    _res_down(r_ids, r_ds)
    n = len(r_ds)
    out_i, out_d = [0] * n, [0.0] * n
    for i in range(n - 1, -1, -1):                  # res_to_sorted_full :317-325
        out_i[i], out_d[i] = _res_pop(r_ids, r_ds)
    return out_i, out_d


def search(g, q, k, ef_search, results_cap):
    """search(), hnsw.ts:960-1027.  Returns (ids, dists, counters)."""
    ctr = {"n_dist": 0, "n_l0lists": 0, "n_upperlists": 0}
    if g.entry == -1 or g.max_elements <= 0 or k <= 0:
        return [], [], ctr
    cur = g.entry
    if cur not in g.lists:
        return [], [], ctr
    cur_d = dist_l2_sq(q, g.vec[cur])
    ctr["n_dist"] += 1
    l = g.max_level
    while l > 0:
        changed = True
        while changed:
            changed = False
            lst = _layer_list(g, cur, l)
            cnt, nbrs = lst if lst is not None else (0, [])
            cnt = max(0, min(cnt, g.m))
            ctr["n_upperlists"] += 1
            for nb in nbrs[:cnt]:
                if nb < 0 or nb >= g.max_elements or nb not in g.lists:
                    continue
                d = dist_l2_sq(q, g.vec[nb])
                ctr["n_dist"] += 1
                if d < cur_d:
                    cur_d, cur, changed = d, nb, True
        l -= 1
    ef = min(max(ef_search, k), MAX_EF)
    ids, ds = search_layer0(g, q, cur, ef, ctr)
    n = min(len(ids), k, results_cap, MAX_EF)
    return ids[:n], ds[:n], ctr


def rerank(q_f32, base_f32, cand_ids, ann_k, topk, allowed=None):
    """searchMany :606-651 for one query."""
    kept = []
    for cid in cand_ids:
        if allowed is not None and cid not in allowed:
            continue
        kept.append(cid)
        if len(kept) >= ann_k:
            break
    items = []
    for cid in kept:
        if 0 <= cid < len(base_f32):
            items.append((l2sq_f32(q_f32, base_f32[cid]), cid))
    items.sort()
    return [c for _, c in items[:topk]], [l for l, _ in items[:topk]]
