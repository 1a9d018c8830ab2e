"""The traversal's relabelled copy of the index (csrc/device_index.hpp: node labels in breadth-first order of the
layer-0 lists) must change nothing a caller can see: ids, distances and work counters equal the oracle's with the copy
and without it, on both kernels, on graphs with holes in the id space and several components, and an incremental edit
drops the copy (the edited mirror is walked under its own ids) until the next full upload."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _clustered(rng, n, dim):
    c = rng.standard_normal((max(2, n // 50), dim), dtype=np.float32)
    x = c[rng.integers(0, c.shape[0], n)] + np.float32(0.3) * rng.standard_normal((n, dim), dtype=np.float32)
    return O.normalize_rows_f32(x)


def _bridge(ix, dim, m, efc):
    from minivectordb_amd import HipBridge
    b = HipBridge()
    b.init(dict(dim=dim, m=m, ef=efc, capacity=8, seed=1))
    b.loadBytes(ix.save_bytes())
    b.setResultsCap(4096)
    ix.set_results_cap(4096)
    return b


def _check(b, ix, q8, k, ef, tag):
    b.setEfSearch(ef)
    ix.set_ef_search(ef)
    oi, od, oc, octr = ix.search_many(q8, k, threads=8)
    for waves in (0, 1):
        b.setTuning(traverse_waves=waves)
        try:
            ids, dist, cnt, ctr = b.searchManyArrays(q8, k, with_counters=True)
        finally:
            b.setTuning()
        assert np.array_equal(cnt, oc) and np.array_equal(ids, oi), f"{tag} waves={waves} k={k} ef={ef}"
        assert np.array_equal(dist.view(np.uint32), od.view(np.uint32)), f"{tag} waves={waves}"
        assert np.array_equal(ctr[:, 0], octr["n_dist"]) and np.array_equal(ctr[:, 1], octr["n_l0lists"]), f"{tag} waves={waves}"
        assert np.array_equal(ctr[:, 2], octr["n_upperlists"]), f"{tag} waves={waves}"


@pytest.mark.parametrize("style", ["clustered", "iid"])
def test_same_answers_with_and_without_the_relabelled_copy(gpu, style):
    rng = np.random.default_rng(21)
    n, dim, m, efc = 12000, 64, 16, 100
    base = _clustered(rng, n, dim) if style == "clustered" else O.normalize_rows_f32(rng.standard_normal((n, dim), dtype=np.float32))
    b8 = O.quantize_rows_i8(base)
    ix = O.OracleIndex(dim, m, efc, n, seed=5)
    ix.insert_many(np.arange(n, dtype=np.int32), b8)
    b = _bridge(ix, dim, m, efc)
    b.attachF32(base, resident=True)
    q = base[rng.integers(0, n, 64)] + np.float32(0.05) * rng.standard_normal((64, dim), dtype=np.float32)
    qn = O.normalize_rows_f32(q)
    q8 = O.quantize_rows_i8(qn)
    fused = {}
    for mode in (2, 0, 1):                        # always / never / kept only where lists become local (the default)
        b.setRelabel(mode)
        for k, ef in [(10, 40), (200, 200), (64, 700)]:
            _check(b, ix, q8, k, ef, f"{style} relabel={mode}")
        if mode != 1 or style == "clustered":     # (mode 1 keeps the copy where lists become local; 12 bitmap lines in all make any list "local")
            assert b.relabelled() == (mode != 0)
        fused[mode] = b.queryMany(q, 10, 300, 300)
    for other in (0, 1):
        for x, y in zip(fused[2], fused[other]):
            assert np.array_equal(x, y)
    fi, fd, fc, _ = (ix.set_ef_search(300), ix.search_many(q8, 300, threads=8))[1]
    e_ids, e_l2, e_n = O.rerank_many(qn, base, fi, fc, 300, 10, threads=8)
    assert np.array_equal(fused[2][0], e_ids) and np.array_equal(fused[2][1].view(np.uint64), e_l2.view(np.uint64))
    b.close()
    ix.close()


def test_holes_components_and_an_edit(gpu):
    """ids with gaps (absent nodes get labels too), two far-apart components (the breadth-first order restarts), then an
    insert: the patch drops the relabelled copy, a re-upload brings it back; the oracle agrees at every stage."""
    rng = np.random.default_rng(22)
    n, dim, m, efc = 3000, 32, 8, 60
    base = _clustered(rng, n, dim)
    base[n // 2:] = O.normalize_rows_f32(-np.abs(base[n // 2:]) - 3.0)    # a second blob far from the first
    b8 = O.quantize_rows_i8(base)
    ids = np.arange(n, dtype=np.int32) * 3 + 1                           # two of three ids stay absent
    cap = int(ids[-1]) + 400
    ix = O.OracleIndex(dim, m, efc, cap, seed=6)
    ix.insert_many(ids, b8)
    b = _bridge(ix, dim, m, efc)
    q8 = O.quantize_rows_i8(O.normalize_rows_f32(base[rng.integers(0, n, 48)] + np.float32(0.05) * rng.standard_normal((48, dim), dtype=np.float32)))
    b.setRelabel(2)
    _check(b, ix, q8, 50, 120, "holes")
    assert b.relabelled()
    extra = O.quantize_rows_i8(_clustered(rng, 40, dim))
    new_ids = np.arange(40, dtype=np.int32) + int(ids[-1]) + 5
    b.insertManyPacked(new_ids, extra.reshape(-1), dim)
    ix.close()
    ix = O.OracleIndex(dim, m, efc, 8, seed=1)                           # the oracle on the edited graph's own dump
    assert ix.load_bytes(b.saveBytes())
    ix.set_results_cap(4096)
    _check(b, ix, q8, 50, 120, "after the insert")
    up = b.uploadCounters()
    if up["patches"] >= 1:                                               # the mirror was edited in place
        assert not b.relabelled()
    b.setRelabel(0)
    b.setRelabel(2)                                                      # forces the next search to upload again
    _check(b, ix, q8, 50, 120, "after the re-upload")
    assert b.relabelled()
    b.close()
    ix.close()
