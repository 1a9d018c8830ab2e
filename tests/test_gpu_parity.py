"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bar: int8 distances and HNSW candidate (id, dist) lists bit-exact; float32 rerank L2 bit-exact in
f64 (stricter than the 1e-5 relative the north star allows) with identical topK ids.
"""
import numpy as np
import pytest

from minivectordb_amd import MvdbError
from oracle import oracle as O
from tests._data import bridge_from_dump, built_index, gaussian_unit, oracle_from_dump

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(gpu):
    dump, base, b8 = built_index(4000, 384, "balanced")
    return dump, base, b8


def _check_lists(b, ix, q8, k, ef):
    b.setEfSearch(ef)
    ix.set_ef_search(ef)
    ids, dist, cnt, ctr = b.searchManyArrays(q8, k, with_counters=True)
    oi, od, oc, octr = ix.search_many(q8, k, threads=8)
    assert np.array_equal(cnt, oc)
    for qi in range(len(cnt)):
        n = cnt[qi]
        assert np.array_equal(ids[qi, :n], oi[qi, :n]), f"query {qi}: candidate id list differs"
        assert np.array_equal(dist[qi, :n].view(np.uint32), od[qi, :n].view(np.uint32)), f"query {qi}: dist differs"
    assert np.array_equal(ctr[:, 0], octr["n_dist"])
    assert np.array_equal(ctr[:, 1], octr["n_l0lists"])
    assert np.array_equal(ctr[:, 2], octr["n_upperlists"])
    return ids, cnt


def test_search_many_bit_exact_balanced(small):
    dump, base, b8 = small
    b = bridge_from_dump(dump, 384, "balanced")
    ix = oracle_from_dump(dump, 384, "balanced")
    b.setResultsCap(4096)
    ix.set_results_cap(4096)
    q8 = O.quantize_rows_i8(gaussian_unit(96, 384, 54321))
    _check_lists(b, ix, q8, 1200, 140)      # balanced: ef = annK = 1200
    _check_lists(b, ix, q8[:32], 10, 50)    # small ef, k < ef
    _check_lists(b, ix, q8[:16], 4096, 20)  # clamp: ef = k = MAX_EF
    _check_lists(b, ix, q8[:8], 5000, 20)   # k > MAX_EF -> ef clamped, flag raised
    assert b._L.mvdb_was_ef_clamped(b.handle) == 1


def test_prepare_queries_bit_exact(gpu):
    from minivectordb_amd import HipBridge
    b = HipBridge()
    b.init(dict(dim=384, m=16, ef=300, capacity=4, seed=1))
    rng = np.random.default_rng(5)
    q = rng.standard_normal((257, 384)).astype(np.float32) * rng.uniform(1e-3, 50, (257, 1)).astype(np.float32)
    q[3] = 0                       # zero vector passes through un-normalised (:966)
    q[4, :] = 0; q[4, 7] = 1       # one-hot
    q[5] = np.float32(1e-30)       # underflowing norm
    q[6, 0] = np.float32(3e38)     # norm overflows to inf -> all zero after scaling
    qn, qi = b.prepareQueries(q)
    on = O.normalize_rows_f32(q)
    assert np.array_equal(qn.view(np.uint32), on.view(np.uint32))
    assert np.array_equal(qi, O.quantize_rows_i8(on))


def test_rerank_bit_exact(small):
    dump, base, b8 = small
    b = bridge_from_dump(dump, 384, "balanced")
    ix = oracle_from_dump(dump, 384, "balanced")
    b.setResultsCap(4096); ix.set_results_cap(4096)
    b.setEfSearch(140); ix.set_ef_search(140)
    qf = gaussian_unit(64, 384, 999)
    q8 = O.quantize_rows_i8(qf)
    ids, dist, cnt = b.searchManyArrays(q8, 1200)
    b.attachF32(base, resident=True)
    # no filter
    gi, gl, gn = b.rerankMany(qf, ids, cnt, 1200, 10)
    oi, ol, on = O.rerank_many(qf, base, ids, cnt, 1200, 10)
    assert np.array_equal(gn, on) and np.array_equal(gi, oi)
    assert np.array_equal(gl.view(np.uint64), ol.view(np.uint64))       # f64 bit-exact
    # allow bitmap (deleted / filter), ann_k cap smaller than the list, ragged counts, short store
    rng = np.random.default_rng(3)
    allow = rng.integers(0, 2**32, (4000 + 31) // 32, dtype=np.uint64).astype(np.uint32)
    cnt2 = cnt.copy(); cnt2[::3] = 7; cnt2[1] = 0
    gi, gl, gn = b.rerankMany(qf, ids, cnt2, 300, 17, allow)
    oi, ol, on = O.rerank_many(qf, base, ids, cnt2, 300, 17, allow)
    assert np.array_equal(gn, on) and np.array_equal(gi, oi)
    assert np.array_equal(gl.view(np.uint64), ol.view(np.uint64))
    b.attachF32(base[:2500], resident=True)     # preloaded store: ids beyond it are skipped (mini-vector-db.ts:1090)
    gi, gl, gn = b.rerankMany(qf, ids, cnt, 1200, 10)
    oi, ol, on = O.rerank_many(qf, base[:2500], ids, cnt, 1200, 10)
    assert np.array_equal(gn, on) and np.array_equal(gi, oi)
    assert np.array_equal(gl.view(np.uint64), ol.view(np.uint64))
    # file mode (store on the host): the reference's reader throws "Vector store short read" (:1120-1124)
    b.attachF32(base[:2500], resident=False)
    with pytest.raises(MvdbError, match="short read"):
        b.rerankMany(qf, ids, cnt, 1200, 10)
    keep = np.zeros((4000 + 31) // 32, np.uint32)                       # a filter that keeps only stored rows: no error
    keep[:2500 // 32] = 0xFFFFFFFF
    gi, gl, gn = b.rerankMany(qf, ids, cnt, 1200, 10, keep)
    oi, ol, on = O.rerank_many(qf, base[:2500], ids, cnt, 1200, 10, keep)
    assert np.array_equal(gn, on) and np.array_equal(gi, oi)
    assert np.array_equal(gl.view(np.uint64), ol.view(np.uint64))
    b.attachF32(base, resident=False)                                   # a complete host store answers like the resident one
    gi, gl, gn = b.rerankMany(qf, ids, cnt, 1200, 10)
    oi, ol, on = O.rerank_many(qf, base, ids, cnt, 1200, 10)
    assert np.array_equal(gn, on) and np.array_equal(gi, oi)


def test_query_many_matches_reference_pipeline(small):
    dump, base, b8 = small
    b = bridge_from_dump(dump, 384, "balanced")
    ix = oracle_from_dump(dump, 384, "balanced")
    b.attachF32(base)
    q = np.random.default_rng(11).standard_normal((80, 384)).astype(np.float32)   # un-normalised on purpose
    tk, ef, ann_k = O.derive_search_params(10, "balanced")
    gi, gl, gn, ctr = b.queryMany(q, tk, ef, ann_k, with_counters=True)
    oi, ol, on, ann, acnt, octr = O.search_many_db(ix, q, base, 10, "balanced", threads=8)
    assert np.array_equal(gn, on) and np.array_equal(gi, oi)
    assert np.array_equal(gl.view(np.uint64), ol.view(np.uint64))
    assert np.array_equal(ctr[:, 0], octr["n_dist"])


def test_dist_many_bit_exact(small):
    dump, base, b8 = small
    b = bridge_from_dump(dump, 384, "balanced")
    rng = np.random.default_rng(2)
    q8 = rng.integers(-127, 128, (33, 384)).astype(np.int8)
    ids = rng.integers(0, 4000, (33, 50)).astype(np.int32)
    ids[0, 0] = -1
    for metric, fn in (("l2", O.dist_l2_i8), ("dot", O.dot_i8)):
        out = b.distMany(q8, ids, metric)
        for qi in range(33):
            for j in range(50):
                exp = fn(q8[qi], b8[ids[qi, j]]) if ids[qi, j] >= 0 else -2**31
                assert out[qi, j] == exp


@pytest.mark.parametrize("dim,preset,n", [(512, "accurate", 1500), (64, "fast", 2000), (100, "fast", 700)])
def test_other_dims_and_presets(gpu, dim, preset, n):
    dump, base, b8 = built_index(n, dim, preset)
    b = bridge_from_dump(dump, dim, preset)
    ix = oracle_from_dump(dump, dim, preset)
    b.setResultsCap(4096); ix.set_results_cap(4096)
    q8 = O.quantize_rows_i8(gaussian_unit(40, dim, 54321))
    tk, ef, ann_k = O.derive_search_params(10, preset)
    _check_lists(b, ix, q8, ann_k, ef)


def test_scan_many_bit_exact(small):
    dump, base, b8 = small
    b = bridge_from_dump(dump, 384, "balanced")
    ix = oracle_from_dump(dump, 384, "balanced")
    q8 = O.quantize_rows_i8(gaussian_unit(300, 384, 77))
    for metric in ("l2", "dot"):
        for k in (1, 10, 32):
            gi, gv, gc = b.scanMany(q8, k, metric)
            oi, ov, oc = ix.scan_many(q8, k, metric, threads=8)
            assert np.array_equal(gc, oc)
            assert np.array_equal(gi, oi), (metric, k)
            assert np.array_equal(gv, ov), (metric, k)


def test_scan_many_large_with_gaps(gpu):
    """Large enough for several levels of the level-wise sweep and the cross-split bound exchange (>= 65536 slots),
    with absent ids (gaps) and heavy value ties (coarse coordinates), k up to the kernel's maximum."""
    from minivectordb_amd import HipBridge
    n_slots, dim = 90000, 64
    rng = np.random.default_rng(31)
    ids = np.sort(rng.choice(n_slots, size=80000, replace=False)).astype(np.int32)
    b8 = (rng.integers(-3, 4, (ids.size, dim)) * 40).astype(np.int8)  # 7 distinct values per coordinate: many ties
    ix = O.OracleIndex(dim, 2, 4, n_slots, seed=5)
    ix.insert_many(ids, b8)
    b = HipBridge()
    b.init(dict(dim=dim, m=2, ef=4, capacity=8, seed=1))
    b.loadBytes(ix.save_bytes())
    q8 = (rng.integers(-3, 4, (333, dim)) * 40).astype(np.int8)
    for metric in ("l2", "dot"):
        for k in (1, 10, 32):
            gi, gv, gc = b.scanMany(q8, k, metric)
            oi, ov, oc = ix.scan_many(q8, k, metric, threads=8)
            assert np.array_equal(gc, oc)
            assert np.array_equal(gi, oi), (metric, k)
            assert np.array_equal(gv, ov), (metric, k)


def test_scan_many_512d_bit_exact(gpu):
    """BASELINE config 3's dimension: the 512-d instantiation of the MFMA scan (8 k-steps per row), more than one
    256-query tile, k from 1 to the kernel's maximum, L2 and dot, against the oracle's exhaustive scan."""
    from minivectordb_amd import HipBridge
    n, dim = 24000, 512
    rng = np.random.default_rng(41)
    b8 = O.quantize_rows_i8(O.normalize_rows_f32(rng.standard_normal((n, dim), dtype=np.float32)))
    b8[100] = b8[7]; b8[2000] = b8[7]                       # exact duplicates: ties broken by id
    ix = O.OracleIndex(dim, 2, 4, n, seed=5)
    ix.insert_many(np.arange(n, dtype=np.int32), b8)
    b = HipBridge()
    b.init(dict(dim=dim, m=2, ef=4, capacity=8, seed=1))
    b.loadBytes(ix.save_bytes())
    q8 = O.quantize_rows_i8(O.normalize_rows_f32(rng.standard_normal((333, dim), dtype=np.float32)))
    q8[5] = b8[7]
    for metric in ("l2", "dot"):
        for k in (1, 10, 32):
            gi, gv, gc = b.scanMany(q8, k, metric)
            oi, ov, oc = ix.scan_many(q8, k, metric, threads=8)
            assert np.array_equal(gc, oc)
            assert np.array_equal(gi, oi), (metric, k)
            assert np.array_equal(gv, ov), (metric, k)
    with pytest.raises(Exception):
        b.scanMany(q8[:4], 33, "l2")                         # k beyond the documented maximum is an error, not a truncation


@pytest.mark.parametrize("dim,n", [(100, 5000), (200, 3001), (320, 2111), (16, 1500)])
def test_scan_many_partial_rows(gpu, dim, n):
    """Dimensions whose padded row is shorter than the kernel's k-step grid (zero-filled pieces of the LDS-DMA
    staging, 6- and 2-k-step instantiations), node counts that end inside a staged tile, ragged query count."""
    from minivectordb_amd import HipBridge
    rng = np.random.default_rng(dim)
    b8 = rng.integers(-127, 128, (n, dim), dtype=np.int16).astype(np.int8)
    ix = O.OracleIndex(dim, 2, 4, n, seed=5)
    ix.insert_many(np.arange(n, dtype=np.int32), b8)
    b = HipBridge()
    b.init(dict(dim=dim, m=2, ef=4, capacity=8, seed=1))
    b.loadBytes(ix.save_bytes())
    q8 = rng.integers(-127, 128, (77, dim), dtype=np.int16).astype(np.int8)
    for metric in ("l2", "dot"):
        for k in (3, 17):
            gi, gv, gc = b.scanMany(q8, k, metric)
            oi, ov, oc = ix.scan_many(q8, k, metric, threads=8)
            assert np.array_equal(gc, oc)
            assert np.array_equal(gi, oi), (metric, k)
            assert np.array_equal(gv, ov), (metric, k)


def _exact_topk(b8, q8, k, metric):
    """Exhaustive top-k in numpy with the scan's order: ascending L2 (or descending dot), ties to the lower id.
    float32 GEMM is exact here: every partial sum is an integer below 2^24 (|dot| <= 127^2 * 512 < 2^24)."""
    n = b8.shape[0]
    vf = b8.astype(np.float32)
    vn = (b8.astype(np.int64) ** 2).sum(1)
    ids = np.empty((q8.shape[0], k), np.int32)
    vals = np.empty((q8.shape[0], k), np.int32)
    for s in range(0, q8.shape[0], 256):
        qc = q8[s:s + 256]
        dot = (qc.astype(np.float32) @ vf.T).astype(np.int64)
        val = (qc.astype(np.int64) ** 2).sum(1)[:, None] + vn[None, :] - 2 * dot if metric == "l2" else -dot
        key = (val << 20) | np.arange(n, dtype=np.int64)[None, :]          # n < 2^20
        part = np.argpartition(key, k - 1, axis=1)[:, :k]
        kk = np.take_along_axis(key, part, 1)
        order = np.argsort(kk, axis=1)
        top = np.take_along_axis(part, order, 1)
        ids[s:s + 256] = top
        v = np.take_along_axis(val, top, 1)
        vals[s:s + 256] = v if metric == "l2" else -v
    return ids, vals


@pytest.mark.parametrize("dim,rows,slots,nq,k,metric,coarse", [(64, 235000, 240000, 3100, 10, "l2", True),
                                                                (320, 236000, 236000, 3072, 12, "dot", False),
                                                                (512, 240000, 240000, 3073, 1, "l2", False)])
def test_scan_many_wide_levels(gpu, dim, rows, slots, nq, k, metric, coarse):
    """Batches of >= 3072 queries with k <= 12 send every level of >= 100 000 nodes to the wide scan kernel (512
    queries per workgroup, A fragments in AccVGPRs): 2-, 8-(for a 6-step row) and 8-k-step instantiations, a ragged
    last query tile, absent slots behind the last row, value ties.  Checked against an exhaustive numpy top-k."""
    from minivectordb_amd import HipBridge
    rng = np.random.default_rng(dim + k)
    if coarse:
        b8 = (rng.integers(-3, 4, (rows, dim)) * 40).astype(np.int8)
        q8 = (rng.integers(-3, 4, (nq, dim)) * 40).astype(np.int8)
    else:
        b8 = rng.integers(-127, 128, (rows, dim), dtype=np.int16).astype(np.int8)
        q8 = rng.integers(-127, 128, (nq, dim), dtype=np.int16).astype(np.int8)
    b = HipBridge()
    b.init(dict(dim=dim, m=2, ef=4, capacity=slots, seed=3))
    b.buildParallel(b8, 16)
    gi, gv, gc = b.scanMany(q8, k, metric)
    oi, ov = _exact_topk(b8, q8, k, metric)
    assert np.all(gc == k)
    assert np.array_equal(gi, oi)
    assert np.array_equal(gv, ov)
    b.close()


def test_scans_on_two_streams_overlap(gpu):
    """mvdb_scan_many_dev keeps its partial lists and bounds per stream: two sweeps with different queries queued on two
    streams at once (several rounds, so that they do run side by side) each return the oracle's exhaustive result."""
    import ctypes as C
    import torch
    from minivectordb_amd import HipBridge
    n, dim, nq, k = 20000, 128, 300, 10
    rng = np.random.default_rng(77)
    b8 = O.quantize_rows_i8(O.normalize_rows_f32(rng.standard_normal((n, dim), dtype=np.float32)))
    ix = O.OracleIndex(dim, 2, 4, n, seed=5)
    ix.insert_many(np.arange(n, dtype=np.int32), b8)
    b = HipBridge()
    b.init(dict(dim=dim, m=2, ef=4, capacity=8, seed=1))
    b.loadBytes(ix.save_bytes())
    b.scanMany(b8[:4], k, "l2")                              # uploads the index
    dev = torch.device("cuda", 0)
    qs = [O.quantize_rows_i8(O.normalize_rows_f32(rng.standard_normal((nq, dim), dtype=np.float32))) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    qd = [torch.from_numpy(q).to(dev) for q in qs]
    outs = [(torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.int32, device=dev),
             torch.empty((nq,), dtype=torch.int32, device=dev)) for _ in range(2)]
    torch.cuda.synchronize(dev)
    vp = C.c_void_p
    for rnd in range(6):
        for j in range(2):
            b._check(b._L.mvdb_scan_many_dev(b.handle, vp(qd[j].data_ptr()), nq, k, 0, vp(outs[j][0].data_ptr()),
                                             vp(outs[j][1].data_ptr()), vp(outs[j][2].data_ptr()), vp(streams[j].cuda_stream)))
    torch.cuda.synchronize(dev)
    for j in range(2):
        oi, ov, oc = ix.scan_many(qs[j], k, "l2", threads=8)
        assert np.array_equal(outs[j][2].cpu().numpy(), oc)
        assert np.array_equal(outs[j][0].cpu().numpy(), oi) and np.array_equal(outs[j][1].cpu().numpy(), ov)
    b.close()
    ix.close()
