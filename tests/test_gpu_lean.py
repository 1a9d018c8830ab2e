"""The lean traversal (csrc/traverse_lean.hip.hpp: result KEYS in the traversal, the result heap replayed from the
admission log afterwards) against the oracle, on the corners of ITS machinery: the sorted key window and its
refills (dense data, sparse data, heavy ties inside 16 bits), the window running full, a log too short (the query
goes to the one-wave kernel), distances beyond 16 bits (likewise), and the counters that say which kernel ran."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _index(b8, dim, m, efc, seed=7):
    n = b8.shape[0]
    ix = O.OracleIndex(dim, m, efc, max(n, 1), seed=seed)
    ix.insert_many(np.arange(n, dtype=np.int32), b8)
    return ix


def _bridge(ix, dim, m, efc):
    from minivectordb_amd import HipBridge
    b = HipBridge()
    b.init(dict(dim=dim, m=m, ef=efc, capacity=8, seed=1))
    b.loadBytes(ix.save_bytes())
    b.setResultsCap(4096)
    ix.set_results_cap(4096)
    return b


def _check(b, ix, q8, k, ef, tag=""):
    b.setEfSearch(ef)
    ix.set_ef_search(ef)
    oi, od, oc, octr = ix.search_many(q8, k, threads=8)
    for rnd in range(2):
        ids, dist, cnt, ctr = b.searchManyArrays(q8, k, with_counters=True)
        t = f"{tag} k={k} ef={ef} round={rnd}"
        assert np.array_equal(cnt, oc), f"{t}: counts differ"
        bad = np.nonzero((ids != oi).any(axis=1))[0]
        assert bad.size == 0, f"{t}: ids differ in queries {bad[:8]} (first at col {np.nonzero(ids[bad[0]] != oi[bad[0]])[0][:4]})"
        assert np.array_equal(dist.view(np.uint32), od.view(np.uint32)), f"{t}: dists differ"
        assert np.array_equal(ctr[:, 0], octr["n_dist"]) and np.array_equal(ctr[:, 1], octr["n_l0lists"]), t
        assert np.array_equal(ctr[:, 2], octr["n_upperlists"]), t


def _unit_i8(n, dim, seed):
    x = np.random.default_rng(seed).standard_normal((n, dim), dtype=np.float32)
    return O.quantize_rows_i8(O.normalize_rows_f32(x))


def test_lean_is_the_default_and_runs(gpu):
    b8 = _unit_i8(5000, 64, 1)
    ix = _index(b8, 64, 16, 100)
    b = _bridge(ix, 64, 16, 100)
    q8 = _unit_i8(64, 64, 2)
    for k, ef in [(1, 1), (10, 10), (10, 64), (100, 100), (300, 300), (1200, 1200), (50, 700), (4096, 4096)]:
        _check(b, ix, q8, k, ef)
    assert b.leanFallbacks() == dict(big_key=0, log_full=0, mismatch=0)


def test_lean_sparse_and_clustered_keys(gpu):
    # Gaussian clusters: result keys spread over a wide range (sparse window refills) and most listed neighbours visited
    n, dim = 8000, 96
    rng = np.random.default_rng(5)
    centers = rng.standard_normal((40, dim), dtype=np.float32)
    x = centers[rng.integers(0, 40, n)] + np.float32(0.35) * rng.standard_normal((n, dim), dtype=np.float32)
    base = O.normalize_rows_f32(x)
    b8 = O.quantize_rows_i8(base)
    ix = _index(b8, dim, 16, 100)
    b = _bridge(ix, dim, 16, 100)
    q = base[rng.integers(0, n, 96)] + np.float32(0.05) * rng.standard_normal((96, dim), dtype=np.float32)
    q8 = O.quantize_rows_i8(O.normalize_rows_f32(q.astype(np.float32)))
    for k, ef in [(10, 40), (200, 200), (1200, 1200), (2500, 2500)]:
        _check(b, ix, q8, k, ef, "clustered")
    assert b.leanFallbacks() == dict(big_key=0, log_full=0, mismatch=0)


def test_lean_heavy_ties_within_16_bits(gpu):
    # few distinct coordinate values -> few distinct distances, hundreds of equal keys: window multiplicities,
    # boundary ties decided by the replayed heap
    rng = np.random.default_rng(7)
    b8 = (rng.integers(-1, 2, (3000, 16)) * 30).astype(np.int8)     # d <= 16 * 60^2 = 57600 < 65536
    ix = _index(b8, 16, 16, 60)
    b = _bridge(ix, 16, 16, 60)
    q8 = (rng.integers(-1, 2, (64, 16)) * 30).astype(np.int8)
    for k, ef in [(10, 10), (64, 64), (500, 500), (1200, 1200)]:
        _check(b, ix, q8, k, ef, "ties")
    assert b.leanFallbacks() == dict(big_key=0, log_full=0, mismatch=0)


def test_lean_many_distinct_keys_fill_the_window(gpu):
    # low dimension, wide spread of distances and small ef: admissions land INSIDE the window (ef < 64 keeps every key there)
    b8 = _unit_i8(6000, 8, 11)
    ix = _index(b8, 8, 8, 60)
    b = _bridge(ix, 8, 8, 60)
    q8 = _unit_i8(64, 8, 12)
    for k, ef in [(5, 5), (40, 40), (63, 63), (64, 64), (65, 65), (100, 100), (130, 130), (900, 900)]:
        _check(b, ix, q8, k, ef, "window")
    assert b.leanFallbacks() == dict(big_key=0, log_full=0, mismatch=0)


def test_lean_hands_over_what_it_cannot_hold(gpu):
    # (a) distances beyond 16 bits: the one-wave kernel answers, bit-exact, and the counter says so
    rng = np.random.default_rng(9)
    b8 = (rng.integers(-1, 2, (3000, 16)) * 64).astype(np.int8)     # d up to 16 * 128^2
    ix = _index(b8, 16, 16, 60)
    b = _bridge(ix, 16, 16, 60)
    q8 = (rng.integers(-1, 2, (48, 16)) * 64).astype(np.int8)
    _check(b, ix, q8, 200, 200, "big keys")
    f = b.leanFallbacks()
    assert f["big_key"] > 0 and f["mismatch"] == 0
    # (b) a log too short for some queries: those go to the one-wave kernel on a bitmap left clean
    b8 = _unit_i8(5000, 64, 1)
    ix = _index(b8, 64, 16, 100)
    b = _bridge(ix, 64, 16, 100)
    q8 = _unit_i8(64, 64, 2)
    b._L.mvdb_debug_set_lean_log_cap(b.handle, 350)
    _check(b, ix, q8, 300, 300, "short log")
    f = b.leanFallbacks()
    assert f["log_full"] > 0 and f["mismatch"] == 0
    b._L.mvdb_debug_set_lean_log_cap(b.handle, 0)
    _check(b, ix, q8, 300, 300, "log back to default")


def _fused_equals_oracle(b, ix, base_f32, q_f32, ef, tag):
    """The fused searchMany path with topK = annK returns EVERY candidate of the traversal (re-ranked): its id set is
    the candidate set the replay kernel produced -- from the log directly (short cut) or from the rebuilt heap."""
    qn = O.normalize_rows_f32(q_f32)
    q8 = O.quantize_rows_i8(qn)
    ix.set_ef_search(ef)
    oi, od, oc, _ = ix.search_many(q8, ef, threads=8)
    e_ids, e_l2, e_n = O.rerank_many(qn, base_f32, oi, oc, ef, ef, threads=8)
    for rnd in range(2):
        g_ids, g_l2, g_n = b.queryMany(q_f32, ef, ef, ef)
        assert np.array_equal(g_n, e_n), f"{tag} ef={ef} round={rnd}: counts differ"
        bad = np.nonzero((g_ids != e_ids).any(axis=1))[0]
        assert bad.size == 0, f"{tag} ef={ef} round={rnd}: candidate sets differ in queries {bad[:8]}"
        assert np.array_equal(g_l2.view(np.uint64), e_l2.view(np.uint64)), f"{tag} ef={ef}"


def test_fused_path_candidate_set_with_and_without_the_short_cut(gpu):
    # the final top-10 of a real query never shows a candidate of rank ~ef: ask for ALL of them (topK = annK = ef)
    rng = np.random.default_rng(21)
    # (a) quantised unit vectors: about half of the queries end with an evicted key equal to the threshold
    base = O.normalize_rows_f32(rng.standard_normal((6000, 64), dtype=np.float32))
    b8 = O.quantize_rows_i8(base)
    ix = _index(b8, 64, 16, 100)
    b = _bridge(ix, 64, 16, 100)
    b.attachF32(base, resident=True)
    q = rng.standard_normal((96, 64), dtype=np.float32)
    for ef in (10, 64, 300, 1200):
        _fused_equals_oracle(b, ix, base, q, ef, "unit")
    r = b.leanReplays()
    assert r["short_cuts"] > 0 and r["full_replays"] > 0, r
    # (b) a handful of distinct distances: ties at the threshold in nearly every query
    t8 = (rng.integers(-1, 2, (3000, 16)) * 30).astype(np.int8)
    tbase = t8.astype(np.float32)
    ix2 = _index(t8, 16, 16, 60)
    b2 = _bridge(ix2, 16, 16, 60)
    b2.attachF32(tbase, resident=True)
    tq = rng.integers(-1, 2, (64, 16)).astype(np.float32)
    tq[tq.sum(axis=1) == 0, 0] = 1.0
    for ef in (10, 64, 500, 1200):
        _fused_equals_oracle(b2, ix2, tbase, tq, ef, "ties")
    r2 = b2.leanReplays()
    assert r2["full_replays"] > 0, r2
    # (c) fewer admissions than ef: everything admitted survives
    s8 = O.quantize_rows_i8(O.normalize_rows_f32(rng.standard_normal((150, 32), dtype=np.float32)))
    sbase = s8.astype(np.float32)
    ix3 = _index(s8, 32, 8, 40)
    b3 = _bridge(ix3, 32, 8, 40)
    b3.attachF32(sbase, resident=True)
    _fused_equals_oracle(b3, ix3, sbase, rng.standard_normal((32, 32), dtype=np.float32), 400, "small")
