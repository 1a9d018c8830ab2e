"""BASELINE.json's configurations at their real sizes (for the sharded ones: one shard at its real size), through the
C ABI: size-independent properties of the whole pipeline plus oracle equality on a sample of the same queries.
The graphs come from the engine's GPU builder (a sequential reference build of 1 M nodes takes hours), the oracle
loads the very dump the engine searched."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from minivectordb_amd import HipBridge, derive_search_params, resolve_preset
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _build(n, dim, preset, seed=12345):
    p = resolve_preset(preset)
    b = HipBridge()
    b.init(dict(dim=dim, m=p["m"], ef=p["ef_construction"], efSearch=p["baseEfSearch"], capacity=n, seed=42,
                resultsCap=4096))
    base = np.empty((n, dim), np.float32)
    b8 = np.empty((n, dim), np.int8)
    chunk = 131072

    def draw(ci):                                                  # one generator per chunk: a few host threads draw
        s0 = ci * chunk
        np.random.default_rng([seed, ci]).standard_normal((min(chunk, n - s0), dim), dtype=np.float32,
                                                          out=base[s0:s0 + chunk])
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(draw, range(-(-n // chunk))))
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        base[s:e], b8[s:e] = b.prepareQueries(base[s:e])
    stats = b.buildGpu(b8)
    return b, p, base, b8, stats


def _oracle_of(b, dim, p):
    ix = O.OracleIndex(dim, p["m"], p["ef_construction"], 8, seed=1)
    assert ix.load_bytes(b.saveBytes())
    return ix


def test_config2_one_million_balanced(gpu):
    """configs[1]: 1 M x 384, balanced (M 16, ef = annK = 1200), 1 k-query batch, topK 10; float32 store resident and
    (config 5's mode) in pinned host memory."""
    n, dim, nq = 1_000_000, 384, 1000
    b, p, base, b8, stats = _build(n, dim, "balanced")
    assert stats["seconds"] < 30
    tk, ef, ann_k = derive_search_params(10, p)
    q = np.random.default_rng(54321).standard_normal((nq, dim), dtype=np.float32)
    qn, q8 = b.prepareQueries(q)
    b.setEfSearch(ef)
    ids, dist, cnt, ctr = b.searchManyArrays(q8, ann_k, with_counters=True)
    assert np.all(cnt == ann_k) and np.all(np.diff(dist, axis=1) >= 0)
    assert all(len(set(r.tolist())) == ann_k for r in ids[::50])
    chk = b.distMany(q8[:32], ids[:32, :64])                      # listed distances are the true int8 L2
    assert np.array_equal(chk.astype(np.float32), dist[:32, :64])
    ids2, dist2, _ = b.searchManyArrays(q8, ann_k)                # idempotent: the visited sets are clean again
    assert np.array_equal(ids, ids2) and np.array_equal(dist, dist2)
    b.attachF32(base, resident=True)
    ti, tl, tn = b.queryMany(q, tk, ef, ann_k)
    assert np.all(tn == tk) and np.all(np.diff(tl, axis=1) >= 0)
    b.attachF32(base, resident=False)                             # the store stays on the host
    hi, hl, hn = b.queryMany(q[:200], tk, ef, ann_k)
    assert np.array_equal(hi, ti[:200]) and np.array_equal(hl.view(np.uint64), tl[:200].view(np.uint64))
    ix = _oracle_of(b, dim, p)                                    # oracle equality on a sample of the same batch
    sub = slice(0, 24)
    oi, ol, on, ann, acnt, octr = O.search_many_db(ix, q[sub], base, 10, "balanced", threads=8)
    assert np.array_equal(ann, ids[sub]) and np.array_equal(oi, ti[sub]) and np.array_equal(on, tn[sub])
    assert np.array_equal(ol.view(np.uint64), tl[sub].view(np.uint64))
    oi8, od8, oc8, octr8 = ix.search_many(q8[sub], ann_k, threads=8)
    assert np.array_equal(ctr[sub, 0], octr8["n_dist"]) and np.array_equal(ctr[sub, 1], octr8["n_l0lists"])
    ix.close()
    b.close()


def test_config4_shard_accurate(gpu):
    """configs[3]: one of the 8 shards of 10 M x 384 at its real size, 1.25 M nodes, preset accurate (M 24, M_MAX0 48,
    ef_construction 600, ef = annK = 2500: the candidate heap runs at its 4096-entry cap and drops pushes silently,
    hnsw.ts:500)."""
    n, dim, nq = 1_250_000, 384, 256
    b, p, base, b8, stats = _build(n, dim, "accurate", seed=777)
    assert (p["m"], p["ef_construction"]) == (24, 600)
    tk, ef, ann_k = derive_search_params(10, p)
    assert (ef, ann_k) == (200, 2500)
    q = np.random.default_rng(4321).standard_normal((nq, dim), dtype=np.float32)
    qn, q8 = b.prepareQueries(q)
    b.attachF32(base, resident=True)
    ti, tl, tn = b.queryMany(q, tk, ef, ann_k)
    assert np.all(tn == tk) and np.all(np.diff(tl, axis=1) >= 0)
    ix = _oracle_of(b, dim, p)
    sub = slice(0, 16)
    oi, ol, on, ann, acnt, _ = O.search_many_db(ix, q[sub], base, 10, "accurate", threads=8)
    assert np.array_equal(oi, ti[sub]) and np.array_equal(ol.view(np.uint64), tl[sub].view(np.uint64))
    b.setEfSearch(ef)
    ids, dist, cnt = b.searchManyArrays(q8[sub], ann_k)
    assert np.array_equal(ids, ann) and np.array_equal(cnt, acnt)
    assert b.leanFallbacks() == dict(big_key=0, log_full=0, mismatch=0)
    ix.close()
    b.close()


def test_config5_shard_balanced_host_f32(gpu):
    """configs[4]: one of the 8 shards of 100 M x 384 at its real size, 12.5 M nodes, preset balanced, float32 store
    in pinned host memory (preloadVectors = false: the rerank reads its rows over PCIe).  The int8 rows (4.8 GB) and
    the float32 store (19.2 GB) are far past 2 GiB, so every row offset needs 64-bit arithmetic; the visited sets of one
    256-query batch take 400 MB.  Oracle equality on 8 queries, properties on the rest; the same answers with the
    store in HBM."""
    n, dim, nq = 12_500_000, 384, 256
    b, p, base, b8, stats = _build(n, dim, "balanced", seed=99)
    tk, ef, ann_k = derive_search_params(10, p)
    q = np.random.default_rng(11).standard_normal((nq, dim), dtype=np.float32)
    q[7] = base[n - 1] * 3.0                                      # a stored vector from the far end of the store
    qn, q8 = b.prepareQueries(q)
    b.attachF32(base, resident=False)
    ti, tl, tn = b.queryMany(q, tk, ef, ann_k)
    assert np.all(tn == tk) and np.all(np.diff(tl, axis=1) >= 0)
    assert ti[7, 0] == n - 1 and tl[7, 0] < 1e-9
    assert ti.max() > (1 << 32) // dim                            # answers come from beyond the 4 GiB mark too
    t2 = b.queryMany(q, tk, ef, ann_k)
    assert np.array_equal(t2[0], ti) and np.array_equal(t2[1].view(np.uint64), tl.view(np.uint64))
    ix = _oracle_of(b, dim, p)
    sub = slice(0, 8)
    oi, ol, on, ann, acnt, _ = O.search_many_db(ix, q[sub], base, 10, "balanced", threads=8)
    assert np.array_equal(oi, ti[sub]) and np.array_equal(ol.view(np.uint64), tl[sub].view(np.uint64))
    ix.close()
    b.attachF32(base, resident=True)                              # the same rows from HBM
    t3 = b.queryMany(q[:64], tk, ef, ann_k)
    assert np.array_equal(t3[0], ti[:64]) and np.array_equal(t3[1].view(np.uint64), tl[:64].view(np.uint64))
    b.close()
