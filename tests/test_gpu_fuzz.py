"""Seeded random sweep of the traversal against the oracle: small random graphs of random shape (dimension, M, size, key
distribution) searched with random (k, ef) through the sorted-list path (mvdb_search_many: lean kernel + full replay
+ heap-sort) and through the fused path with topK = annK (every candidate: the replay's short cut or full replay),
twice in a row on the same buffers.  The corner tests elsewhere are chosen by hand; this one is not."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _data(rng, n, dim, style):
    if style == "unit":
        return O.normalize_rows_f32(rng.standard_normal((n, dim), dtype=np.float32))
    if style == "ties":                          # few distinct coordinate values: hundreds of equal distances
        x = rng.integers(-1, 2, (n, dim)).astype(np.float32)
        x[np.abs(x).sum(axis=1) == 0, 0] = 1.0
        return O.normalize_rows_f32(x)
    c = rng.standard_normal((max(2, n // 60), dim), dtype=np.float32)   # clusters: most listed neighbours already visited
    return O.normalize_rows_f32(c[rng.integers(0, c.shape[0], n)] + np.float32(0.3) * rng.standard_normal((n, dim), dtype=np.float32))


@pytest.mark.parametrize("seed", range(48))
def test_random_graphs_sorted_and_fused(gpu, seed):
    from minivectordb_amd import HipBridge
    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.choice([4, 8, 16, 32, 48, 64, 96, 128]))
    m = int(rng.choice([2, 4, 8, 16, 24, 32]))
    n = int(rng.choice([1, 5, 40, 300, 1500, 4000]))
    style = ["unit", "ties", "clusters"][seed % 3]
    base = _data(rng, n, dim, style)
    b8 = O.quantize_rows_i8(base)
    efc = int(rng.choice([8, 40, 100]))
    ix = O.OracleIndex(dim, m, efc, max(n, 1), seed=7)
    ix.insert_many(np.arange(n, dtype=np.int32), b8)
    ix.set_results_cap(4096)
    b = HipBridge()
    b.init(dict(dim=dim, m=m, ef=efc, capacity=8, seed=1))
    b.loadBytes(ix.save_bytes())
    b.setResultsCap(4096)
    b.attachF32(base, resident=True)
    q = _data(rng, 40, dim, style if seed % 2 else "unit")
    qn = O.normalize_rows_f32(q)
    q8 = O.quantize_rows_i8(qn)
    tag = f"seed={seed} dim={dim} m={m} n={n} {style}"
    for _ in range(4):
        ef = int(rng.choice([1, 3, 17, 64, 65, 200, 700, 1500]))
        k = int(rng.choice([1, ef, max(1, ef // 3), min(4096, 2 * ef)]))
        ix.set_ef_search(ef)
        b.setEfSearch(ef)
        oi, od, oc, octr = ix.search_many(q8, k, threads=8)
        for rnd in range(2):
            ids, dist, cnt, ctr = b.searchManyArrays(q8, k, with_counters=True)
            assert np.array_equal(cnt, oc) and np.array_equal(ids, oi), f"{tag} k={k} ef={ef} round={rnd}"
            assert np.array_equal(dist.view(np.uint32), od.view(np.uint32)), f"{tag} k={k} ef={ef}"
            assert np.array_equal(ctr[:, 0], octr["n_dist"]), f"{tag} k={k} ef={ef}"
        fi, fd, fc, _ = ix.search_many(q8, ef, threads=8)                # fused path, every candidate
        e_ids, e_l2, e_n = O.rerank_many(qn, base, fi, fc, ef, ef, threads=8)
        g_ids, g_l2, g_n = b.queryMany(q, ef, ef, ef)
        assert np.array_equal(g_n, e_n) and np.array_equal(g_ids, e_ids), f"{tag} fused ef={ef}"
        assert np.array_equal(g_l2.view(np.uint64), e_l2.view(np.uint64)), f"{tag} fused ef={ef}"
    f = b.leanFallbacks()
    assert f["mismatch"] == 0, f
    b.close()
    ix.close()
