"""The C oracle and the independent numpy/pure-Python restatement must agree bit for bit."""
import numpy as np
import pytest

from oracle import np_oracle as NP
from oracle import oracle as O
from tests._data import built_index, gaussian_unit


@pytest.mark.parametrize("n,dim,preset,k,ef", [(600, 32, "fast", 50, 20), (500, 64, "balanced", 300, 140),
                                               (400, 16, "accurate", 10, 4)])
def test_search_lists_identical(n, dim, preset, k, ef):
    dump, base, b8 = built_index(n, dim, preset)
    p = O.PRESETS[preset]
    g = NP.parse_dump(dump, dim, p["m"], p["ef_construction"])
    assert g is not None
    ix = O.OracleIndex(dim, p["m"], p["ef_construction"], 4, seed=1)
    assert ix.load_bytes(dump)
    ix.set_ef_search(ef)
    ix.set_results_cap(4096)
    qs = O.quantize_rows_i8(gaussian_unit(12, dim, 54321))
    for q in qs:
        ids, dist, ctr = ix.search(q, k, with_counters=True)
        pi, pd, pc = NP.search(g, q, k, ef, 4096)
        assert ids.tolist() == pi
        assert dist.tolist() == pd
        assert (ctr.n_dist, ctr.n_l0lists, ctr.n_upperlists) == (pc["n_dist"], pc["n_l0lists"], pc["n_upperlists"])


def test_tie_heavy_graph_identical():
    """Tiny value range => ties everywhere: exercises the heap tie rules (hnsw.ts:163,201,206,243,278,283)."""
    rng = np.random.default_rng(0)
    n, dim = 700, 8
    b8 = rng.integers(-2, 3, (n, dim)).astype(np.int8)
    ix = O.OracleIndex(dim, 6, 40, n, seed=3)
    ix.insert_many(np.arange(n, dtype=np.int32), b8)
    dump = ix.save_bytes()
    g = NP.parse_dump(dump, dim, 6, 40)
    ix.set_results_cap(4096)
    for ef in (5, 64, 300):
        ix.set_ef_search(ef)
        for q in rng.integers(-2, 3, (10, dim)).astype(np.int8):
            ids, dist = ix.search(q, ef)
            pi, pd, _ = NP.search(g, q, ef, ef, 4096)
            assert ids.tolist() == pi and dist.tolist() == pd


def test_host_math_identical():
    rng = np.random.default_rng(9)
    for _ in range(20):
        v = (rng.standard_normal(384) * rng.uniform(1e-4, 100)).astype(np.float32)
        a = O.normalize_f32(v)
        b = NP.normalize_f32(v)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.array_equal(O.quantize_i8(a), NP.quantize_i8(a))
        w = rng.standard_normal(384).astype(np.float32)
        assert O.l2sq_f32(a, w) == NP.l2sq_f32(a, w)
    z = np.zeros(8, np.float32)
    assert np.array_equal(O.normalize_f32(z), z)          # zero vector passes through (:966)
    x = rng.integers(-127, 128, (50, 384)).astype(np.int8)
    for i in range(0, 50, 2):
        assert O.dist_l2_i8(x[i], x[i + 1]) == NP.dist_l2_i8(x[i], x[i + 1])
        assert O.dot_i8(x[i], x[i + 1]) == NP.dot_i8(x[i], x[i + 1])


def test_rerank_identical():
    dump, base, b8 = built_index(500, 64, "balanced")
    ix = O.OracleIndex(64, 16, 300, 4, seed=1)
    assert ix.load_bytes(dump)
    qf = gaussian_unit(6, 64, 5)
    ids, dist, cnt, _ = ix.search_many(O.quantize_rows_i8(qf), 200)
    allowed = set(range(0, 500, 3))
    allow = np.zeros(16, np.uint32)
    for i in allowed:
        allow[i >> 5] |= np.uint32(1 << (i & 31))
    oi, ol, on = O.rerank_many(qf, base, ids, cnt, 150, 7, allow)
    for qi in range(6):
        pi, pl = NP.rerank(qf[qi], base, ids[qi, :cnt[qi]].tolist(), 150, 7, allowed)
        assert oi[qi, :on[qi]].tolist() == pi and ol[qi, :on[qi]].tolist() == pl


def test_dump_rejects_corruption_like_load_index():
    dump, base, b8 = built_index(300, 16, "fast")
    p = O.PRESETS["fast"]
    def both(buf):
        ix = O.OracleIndex(16, p["m"], p["ef_construction"], 7, seed=1)
        a = ix.load_bytes(bytes(buf))
        b = NP.parse_dump(bytes(buf), 16, p["m"], p["ef_construction"]) is not None
        assert a == b
        if not a:                                          # fresh empty index of the configured capacity
            assert ix.max_elements == 7 and ix.element_count == 0
        return a
    assert both(dump)
    assert not both(dump[:40])                             # shorter than the header
    assert not both(dump[:-1])                             # truncated last record
    bad = bytearray(dump); bad[0] ^= 1; assert not both(bad)        # magic
    bad = bytearray(dump); bad[4] = 2; assert not both(bad)         # version
    bad = bytearray(dump); bad[36:40] = (10**6).to_bytes(4, "little"); assert not both(bad)   # entry out of range
    bad = bytearray(dump); bad[40:44] = (4).to_bytes(4, "little"); assert not both(bad)       # max_level >= MAX_LAYERS
    bad = bytearray(dump); bad[48 + 8 + 16:48 + 8 + 16 + 4] = (999).to_bytes(4, "little"); assert not both(bad)  # count > cap
    bad = bytearray(dump); bad[48 + 4:48 + 8] = (9).to_bytes(4, "little"); assert not both(bad)  # level >= MAX_LAYERS
