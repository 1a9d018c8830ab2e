"""GPU parity of BOTH traversal paths (the lean kernel + result-heap replay, and the one-wave kernel with the result
heap inside) against the oracle on shapes chosen to reach their corners: result heap filling up inside a
list, candidate heap saturated (pushes dropped, hnsw.ts:500), lists of 64 neighbours (two row
passes), M_MAX0 > 64 (one-wave kernel only), tiny graphs, heavy distance ties."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _build(n, dim, m, efc, seed, coarse=False):
    rng = np.random.default_rng(seed)
    if coarse:  # few distinct coordinate values -> many equal int8 distances
        b8 = (rng.integers(-1, 2, (n, dim)) * 64).astype(np.int8)
    else:
        x = rng.standard_normal((n, dim), dtype=np.float32)
        b8 = O.quantize_rows_i8(O.normalize_rows_f32(x))
    ix = O.OracleIndex(dim, m, efc, max(n, 1), seed=seed)
    if n:
        ix.insert_many(np.arange(n, dtype=np.int32), b8)
    return ix, b8


def _bridge(ix, dim, m, efc):
    from minivectordb_amd import HipBridge
    b = HipBridge()
    b.init(dict(dim=dim, m=m, ef=efc, capacity=8, seed=1))
    b.loadBytes(ix.save_bytes())
    b.setResultsCap(4096)
    ix.set_results_cap(4096)
    return b


def _same(b, ix, q8, k, ef):
    b.setEfSearch(ef)
    ix.set_ef_search(ef)
    oi, od, oc, octr = ix.search_many(q8, k, threads=8)
    outs = {}
    # both kernels, each twice in a row on the same buffers (the second launch works on bitmaps the launcher cleared again)
    for waves in ("0", "1"):   # 0 = the lean path (the default), 1 = result heap inside the traversal
        b.setTuning(traverse_waves=int(waves))
        try:
            for rnd in range(2):
                ids, dist, cnt, ctr = b.searchManyArrays(q8, k, with_counters=True)
                tag = f"waves={waves} round={rnd} k={k} ef={ef}"
                assert np.array_equal(cnt, oc), f"{tag}: counts differ"
                assert np.array_equal(ids, oi), f"{tag}: ids differ"
                assert np.array_equal(dist.view(np.uint32), od.view(np.uint32)), f"{tag}: dists differ"
                assert np.array_equal(ctr[:, 0], octr["n_dist"]) and np.array_equal(ctr[:, 1], octr["n_l0lists"]), tag
                assert np.array_equal(ctr[:, 2], octr["n_upperlists"]), tag
        finally:
            b.setTuning()
        outs[waves] = ids
    return outs


@pytest.mark.parametrize("n", [1, 2, 3, 17, 70])
def test_tiny_graphs(gpu, n):
    ix, b8 = _build(n, 32, 8, 40, 3)
    b = _bridge(ix, 32, 8, 40)
    q8 = O.quantize_rows_i8(O.normalize_rows_f32(np.random.default_rng(9).standard_normal((24, 32), dtype=np.float32)))
    for k, ef in [(1, 1), (3, 2), (10, 7), (100, 16), (5, 300)]:
        _same(b, ix, q8, k, ef)


@pytest.mark.parametrize("m", [4, 16, 32, 40])
def test_list_widths(gpu, m):
    # m=32 -> 64-entry lists (two row passes in the two-wave kernel); m=40 -> 80 entries (one-wave kernel)
    ix, b8 = _build(2500, 64, m, 80, 5)
    b = _bridge(ix, 64, m, 80)
    q8 = O.quantize_rows_i8(O.normalize_rows_f32(np.random.default_rng(10).standard_normal((48, 64), dtype=np.float32)))
    for k, ef in [(10, 50), (300, 300), (1200, 140)]:
        _same(b, ix, q8, k, ef)


def test_saturated_candidate_heap(gpu):
    # ef = MAX_EF: cand_cap = ef, so pushes are dropped for most of the search (hnsw.ts:443-445, :500)
    ix, b8 = _build(9000, 32, 16, 60, 6)
    b = _bridge(ix, 32, 16, 60)
    q8 = O.quantize_rows_i8(O.normalize_rows_f32(np.random.default_rng(11).standard_normal((32, 32), dtype=np.float32)))
    _same(b, ix, q8, 4096, 4096)
    _same(b, ix, q8, 2100, 2100)  # cand_cap = min(2*ef+32, 4096) = 4096 < admitted


def test_heavy_ties(gpu):
    ix, b8 = _build(3000, 16, 16, 60, 7, coarse=True)
    b = _bridge(ix, 16, 16, 60)
    q8 = (np.random.default_rng(12).integers(-1, 2, (64, 16)) * 64).astype(np.int8)
    for k, ef in [(10, 10), (64, 64), (500, 500), (1200, 1200)]:
        _same(b, ix, q8, k, ef)


def test_fused_query_path_set_only_output(gpu):
    """mvdb_query_many lets the traversal skip its final heap sort when nothing is cut off (set_only).
    Both kernels, with and without an allow bitmap, must return the oracle pipeline's bits; with
    efSearch > annK the candidate list IS cut off (found > k) and the kernels must sort as the reference."""
    from tests._data import bridge_from_dump, built_index, oracle_from_dump
    dump, base, b8 = built_index(4000, 384, "balanced")
    b = bridge_from_dump(dump, 384, "balanced")
    ix = oracle_from_dump(dump, 384, "balanced")
    b.attachF32(base)
    rng = np.random.default_rng(77)
    q = rng.standard_normal((60, 384)).astype(np.float32)
    allow = rng.integers(0, 2**32, (4000 + 31) // 32, dtype=np.uint64).astype(np.uint32)
    qf = O.normalize_rows_f32(q)
    qi = O.quantize_rows_i8(qf)
    for ef, ann_k, bitmap in [(1200, 1200, None), (1200, 1200, allow), (300, 120, None), (300, 120, allow), (50, 400, None)]:
        ix.set_ef_search(ef)
        ix.set_results_cap(4096)
        ids, dist, cnt, _ = ix.search_many(qi, ann_k, threads=8)
        oi, ol, on = O.rerank_many(qf, base, ids, cnt, ann_k, 10, bitmap, 8)
        for waves in ("0", "1"):
            b.setTuning(traverse_waves=int(waves))
            try:
                gi, gl, gn = b.queryMany(q, 10, ef, ann_k, allow_bitmap=bitmap)
            finally:
                b.setTuning()
            assert np.array_equal(gn, on), (waves, ef, ann_k)
            assert np.array_equal(gi, oi), (waves, ef, ann_k)
            assert np.array_equal(gl.view(np.uint64), ol.view(np.uint64)), (waves, ef, ann_k)


def test_aborted_query_is_reported_not_returned_empty(gpu):
    """A traversal that aborts a query (no kernel does on its own; forced here through
    mvdb_debug_inject_abort) must fail the host calls and stay visible on the `_dev` path: n = -1 for that
    query after the rerank and mvdb_check_errors() == 0 -- never "0 results, status ok"."""
    import torch
    from minivectordb_amd import MvdbError
    from minivectordb_amd.pipeline import ResidentSearcher
    from tests._data import bridge_from_dump, built_index, oracle_from_dump
    dump, base, b8 = built_index(4000, 384, "balanced")
    b = bridge_from_dump(dump, 384, "balanced")
    ix = oracle_from_dump(dump, 384, "balanced")
    b.attachF32(base)
    q = np.random.default_rng(5).standard_normal((40, 384)).astype(np.float32)
    qi = O.quantize_rows_i8(O.normalize_rows_f32(q))
    oi, ol, on, _, _, _ = O.search_many_db(ix, q, base, 10, "balanced", threads=8)
    dev = torch.device("cuda", 0)
    s = ResidentSearcher(b, 40, 10, dev)
    q_dev = torch.from_numpy(q).to(dev)
    for waves in (0, 1):
        b.setTuning(traverse_waves=waves)
        b._L.mvdb_debug_inject_abort(b.handle, 17)
        with pytest.raises(MvdbError, match="aborted"):
            b.searchManyArrays(qi, 50)
        with pytest.raises(MvdbError, match="aborted"):
            b.queryMany(q, 10, 140, 1200)
        ids, l2, n = s.query_many(q_dev, 140, 1200)          # device-pointer path: nothing synchronises inside
        torch.cuda.synchronize(dev)
        n_h = n.cpu().numpy()
        assert n_h[17] == -1 and np.array_equal(np.delete(n_h, 17), np.delete(on, 17)), waves
        assert np.array_equal(np.delete(ids.cpu().numpy(), 17, 0), np.delete(oi, 17, 0))
        assert s.ctr[17, 3].item() == -1
        with pytest.raises(MvdbError, match="aborted"):
            s.check()
        s.check()                                            # the error word was cleared by the failed check
        b._L.mvdb_debug_inject_abort(b.handle, -1)
        gi, gl, gn = b.queryMany(q, 10, 140, 1200)           # and the handle keeps working
        assert np.array_equal(gn, on) and np.array_equal(gi, oi) and np.array_equal(gl.view(np.uint64), ol.view(np.uint64))
    b.setTuning()


def test_clustered_rows_mode_switch(gpu):
    # Gaussian clusters: early in a query nearly every listed neighbour is new (rows are gathered together with
    # the visited test), later most are visited already (rows after the visited words): the per-query switch
    # between the two orders (traverse_duo.hip.hpp, MVDB_DUO_SPEC_ADAPT) must not change a bit.
    n, dim, m, efc = 6000, 64, 16, 100
    rng = np.random.default_rng(77)
    centers = rng.standard_normal((48, dim), dtype=np.float32)
    x = centers[rng.integers(0, 48, n)] + np.float32(0.3) * rng.standard_normal((n, dim), dtype=np.float32)
    base = O.normalize_rows_f32(x)
    b8 = O.quantize_rows_i8(base)
    ix = O.OracleIndex(dim, m, efc, n, seed=7)
    ix.insert_many(np.arange(n, dtype=np.int32), b8)
    b = _bridge(ix, dim, m, efc)
    q = base[rng.integers(0, n, 64)] + np.float32(0.05) * rng.standard_normal((64, dim), dtype=np.float32)
    q8 = O.quantize_rows_i8(O.normalize_rows_f32(q.astype(np.float32)))
    for k, ef in [(10, 64), (300, 300), (1200, 1200)]:
        _same(b, ix, q8, k, ef)
