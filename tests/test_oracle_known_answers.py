"""Pins the CPU oracle with the known answers the reference's own tests hold for the path
(SURVEY.md 8c).  The reference tests are end-to-end (tests/search_many_test.ts, basic_test.ts,
rebuild_threshold_test.ts, persistence_test.ts, config_mismatch_test.ts); here their vectors and
assertions are re-enacted against the oracle's index + host math.  Derivable constants: a
normalised one-hot quantises to 127 in one slot, so int8 self-distance is 0 and two different
one-hots are 2*127^2 = 32258 apart; float32 L2^2 is 0 and 2.
"""
import numpy as np

from oracle import np_oracle as NP
from oracle import oracle as O


def one_hot(dim, i, val=1.0):
    v = np.zeros(dim, np.float32)
    v[i % dim] = val
    return v


def build_onehot_db(n, dim=384, preset="balanced", capacity=50000):
    p = O.PRESETS[preset]
    ix = O.OracleIndex(dim, p["m"], p["ef_construction"], capacity, ef_search=p["baseEfSearch"],
                       results_cap=p["resultsCap"], seed=42)
    base = np.stack([O.normalize_f32(one_hot(dim, i)) for i in range(n)])
    for i in range(n):                                   # db.insert one at a time, like the tests
        ix.insert(i, O.quantize_i8(base[i]))
    return ix, base


def test_quantised_one_hot_constants():
    v = O.normalize_f32(one_hot(384, 5, 3.0))            # always re-normalised (:1221-1223)
    assert v[5] == np.float32(1.0) and np.count_nonzero(v) == 1
    q = O.quantize_i8(v)
    assert q[5] == 127 and np.count_nonzero(q) == 1
    q2 = O.quantize_i8(O.normalize_f32(one_hot(384, 9)))
    assert O.dist_l2_i8(q, q) == 0
    assert O.dist_l2_i8(q, q2) == 2 * 127 * 127 == 32258
    assert O.dot_i8(q, q) == 127 * 127 and O.dot_i8(q, q2) == 0
    assert O.l2sq_f32(v, v) == 0.0 and O.l2sq_f32(v, O.normalize_f32(one_hot(384, 9))) == 2.0


def test_search_many_test_ts():
    """tests/search_many_test.ts:26-68: 128 one-hot docs, 64 one-hot queries, topK 3:
    result length 64, rows non-empty, top-1 == d-i for the first 10."""
    ix, base = build_onehot_db(128)
    queries = np.stack([one_hot(384, i) for i in range(64)])
    ids, l2, n, ann, cnt, _ = O.search_many_db(ix, queries, base, topk=3, preset="balanced")
    assert ids.shape[0] == 64
    for i in range(10):
        assert n[i] > 0 and ids[i, 0] == i
        assert l2[i, 0] == 0.0
        assert n[i] == 3 and l2[i, 1] == 2.0              # every other doc is exactly 2 away
        assert ids[i, 1] < ids[i, 2]                      # ties broken by internal id (:637)


def test_basic_test_ts_top1_filter_and_reload():
    """tests/basic_test.ts:57-124: two one-hot docs; top-1; object filter excludes the other doc;
    save -> reopen -> same top-1."""
    ix, base = build_onehot_db(2, capacity=1000)
    q = one_hot(384, 0)[None]
    ids, l2, n, *_ = O.search_many_db(ix, q, base, topk=1)
    assert n[0] == 1 and ids[0, 0] == 0
    allow = np.zeros(32, np.uint32)
    allow[0] = 1 << 1                                    # filter passes only doc 1
    ids, l2, n, *_ = O.search_many_db(ix, q, base, topk=5, allow_bitmap=allow)
    assert n[0] == 1 and ids[0, 0] == 1 and l2[0, 0] == 2.0
    dump = ix.save_bytes()
    p = O.PRESETS["balanced"]
    ix2 = O.OracleIndex(384, p["m"], p["ef_construction"], 1000, seed=7)
    assert ix2.load_bytes(dump)
    ids, l2, n, *_ = O.search_many_db(ix2, q, base, topk=1)
    assert n[0] == 1 and ids[0, 0] == 0


def test_rebuild_threshold_test_ts_deleted_never_returned():
    """tests/rebuild_threshold_test.ts:53-80: 200 one-hot docs, first 60 % soft-deleted: deleted ids are
    never returned (:53-63); after rebuild({capacity:25000}) (compact: survivors re-inserted with
    contiguous ids) the first 10 survivors are found in their own top-3 (:71-80).  (With all
    pairwise distances tied, late inserts get no incoming edge -- add_connection only replaces on a
    strictly smaller distance, hnsw.ts:634 -- which is why the reference asserts on early docs only.)"""
    n, ndel = 200, 120
    ix, base = build_onehot_db(n, capacity=20000)
    allow = np.zeros((20000 + 31) // 32, np.uint32)
    for i in range(ndel, n):
        allow[i >> 5] |= np.uint32(1 << (i & 31))
    queries = np.stack([one_hot(384, i) for i in range(10)])
    ids, l2, cnt, *_ = O.search_many_db(ix, queries, base, topk=10, allow_bitmap=allow)
    for i in range(10):
        assert i not in ids[i, :cnt[i]].tolist()
        assert all(x >= ndel for x in ids[i, :cnt[i]].tolist())
    # rebuild: survivors only, new contiguous internal ids 0..79 <-> doc-(120+j)
    p = O.PRESETS["balanced"]
    ix2 = O.OracleIndex(384, p["m"], p["ef_construction"], 25000, seed=42)
    base2 = base[ndel:]
    for j in range(n - ndel):
        ix2.insert(j, O.quantize_i8(base2[j]))
    queries = np.stack([one_hot(384, i) for i in range(ndel, ndel + 10)])
    ids, l2, cnt, *_ = O.search_many_db(ix2, queries, base2, topk=3)
    for j in range(10):
        assert cnt[j] > 0 and j in ids[j, :cnt[j]].tolist()


def test_config_mismatch_test_ts():
    """tests/config_mismatch_test.ts:57-140: a dump saved at dim 384 must not load at dim 512;
    the failed load leaves a fresh empty index."""
    ix, base = build_onehot_db(4, capacity=100)
    dump = ix.save_bytes()
    p = O.PRESETS["balanced"]
    other = O.OracleIndex(512, p["m"], p["ef_construction"], 100, seed=1)
    assert not other.load_bytes(dump)
    assert other.element_count == 0 and other.max_elements == 100 and other.entry_point == -1
    ids, dist = other.search(np.zeros(512, np.int8), 5)
    assert len(ids) == 0
    for kw in (dict(m=12), dict(ef_construction=299)):   # M and EF_CONSTRUCTION are compat keys too (:1151-1159)
        args = dict(m=p["m"], ef_construction=p["ef_construction"])
        args.update(kw)
        o = O.OracleIndex(384, args["m"], args["ef_construction"], 100, seed=1)
        assert not o.load_bytes(dump)


def test_capacity_and_bad_ids_are_silent():
    """insert/has_node guards (hnsw.ts:720, :135-138): out-of-range ids are ignored."""
    ix = O.OracleIndex(16, 4, 20, 3, seed=1)
    v = np.ones(16, np.int8)
    ix.insert(5, v)
    ix.insert(-1, v)
    assert ix.element_count == 0 and not ix.has_node(5) and not ix.has_node(-1)
    ix.insert(2, v)
    assert ix.element_count == 1 and ix.has_node(2)


def test_results_cap_and_ef_rules():
    ix, base = build_onehot_db(64, capacity=100)
    q = O.quantize_i8(O.normalize_f32(one_hot(384, 3)))
    ix.set_ef_search(10)
    ix.set_results_cap(7)
    ids, d = ix.search(q, 20)                            # out_count = min(found, k, results_cap) (:1015-1017)
    assert len(ids) == 7 and ids[0] == 3 and d[0] == 0.0
    ix.set_results_cap(0)                                # bridge clamps 0 -> 1 (wasm-bridge.ts:105-110)
    assert ix.get_results_cap() == 1
    ix.set_results_cap(99999)
    assert ix.get_results_cap() == 4096
    ix.set_ef_search(0)                                  # ignored (config.ts:128)
    assert ix.get_ef_search() == 10
    ids, d = ix.search(q, 0)                             # k <= 0 -> 0 (:963)
    assert len(ids) == 0
    ids, d = ix.search(q, 5000)                          # ef clamped to MAX_EF, flag raised
    assert ix.was_ef_clamped() and ids[0] == 3 and 1 <= len(ids) <= 64   # ragged: only the reachable component
    assert O.derive_search_params(10, "balanced") == (10, 140, 1200)
    assert O.derive_search_params(10, "fast") == (10, 80, 400)
    assert O.derive_search_params(10, "accurate") == (10, 200, 2500)
    assert O.derive_search_params(100, "accurate") == (100, 200, 4096)
    assert O.derive_search_params(0, "balanced") == (1, 140, 120)
    assert O.next_results_cap(1000, 1200) == 2000 and O.next_results_cap(4096, 1200) == 4096
    assert O.next_results_cap(0, 1200) == 2000


def test_quantize_round_half_up_and_clamps():
    # Math.round ties go toward +inf: 2.5 -> 3, -2.5 -> -2
    x = np.array([2.5 / 127, -2.5 / 127, 1.7, -1.7, 0.0, -0.0, 126.5 / 127, -126.5 / 127, np.nan], np.float32)
    got = O.quantize_i8(x)
    exp = NP.quantize_i8(x)
    assert np.array_equal(got, exp)
    assert got[2] == 127 and got[3] == -127 and got[8] == 0
    xs = np.linspace(-1.2, 1.2, 20001, dtype=np.float32)
    assert np.array_equal(O.quantize_i8(xs), NP.quantize_i8(xs))
    assert O.quantize_i8(xs).min() == -127


def test_to_score():
    for l2 in (0.0, 0.5, 2.0, 4.0, 5.0):
        for mode in ("l2", "cosine", "sim"):
            assert O.to_score(l2, mode) == NP.to_score(l2, mode)
    assert O.to_score(5.0, "cosine") == -1.0 and O.to_score(5.0, "sim") == 0.0 and O.to_score(0.0, "sim") == 1.0
