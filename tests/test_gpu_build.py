"""The GPU bulk builder (mvdb_build_gpu): it need not reproduce the sequential edge order (the
multi-threaded host builder does not either), but it must
  * emit a V3 dump the oracle's load_index accepts and the product re-serialises byte for byte,
  * keep the level assignment of a sequential build with the same seed (same RNG stream, hnsw.ts:416-420),
  * keep every list well-formed (count <= cap, dense prefix, ids present, no self edge, no duplicate),
  * reach the recall and the work per query of the reference's sequential insert on the same data,
  * be searched bit-exactly like any other dump (GPU lists == oracle lists on that dump)."""
import struct

import numpy as np
import pytest

from oracle import oracle as O
from tests import _dump as D
from tests._data import built_index, gaussian_unit

pytestmark = pytest.mark.gpu


def _recall(ids, gt, k=10):
    return float(np.mean([len(set(ids[i, :k].tolist()) & set(gt[i].tolist())) / k for i in range(ids.shape[0])]))


def _exact_topk(base, qn, k=10):
    d = 2.0 - 2.0 * (qn.astype(np.float64) @ base.astype(np.float64).T)
    return np.argsort(d, axis=1, kind="stable")[:, :k]


@pytest.mark.parametrize("n,dim,preset", [(20000, 64, "balanced"), (6000, 384, "fast")])
def test_gpu_build_quality_and_format(gpu, n, dim, preset):
    from minivectordb_amd import HipBridge
    p = O.PRESETS[preset]
    seq_dump, base, b8 = built_index(n, dim, preset)            # the reference's sequential insert (oracle), seed 42
    b = HipBridge()
    b.init(dict(dim=dim, m=p["m"], ef=p["ef_construction"], capacity=n, seed=42))
    st = b.buildGpu(b8)
    assert b.elementCount() == n and st["n_dist"] > 0 and st["passes"] >= 1
    dump = b.saveBytes()

    # format: the oracle loads it, both re-serialise it unchanged
    ix = O.OracleIndex(dim, p["m"], p["ef_construction"], 8, seed=1)
    assert ix.load_bytes(dump)
    assert ix.save_bytes() == dump
    b2 = HipBridge()
    b2.init(dict(dim=dim, m=p["m"], ef=p["ef_construction"], capacity=8, seed=1))
    b2.loadBytes(dump)
    assert b2.saveBytes() == dump

    # levels, rows and header equal the sequential build's; lists are well-formed
    recs, seq = D.walk(dump), D.walk(seq_dump)
    assert D.header(dump)[:11] == D.header(seq_dump)[:11]
    assert [(r[0], r[1]) for r in recs] == [(r[0], r[1]) for r in seq]
    m, m0 = p["m"], 2 * p["m"]
    deg0 = 0
    for nid, lvl, vec_off, lists in recs:
        assert dump[vec_off:vec_off + dim] == b8[nid].tobytes()
        for l, c_off, i_off, cap in lists:
            cnt = D.get_i32(dump, c_off)
            ids = np.frombuffer(dump, np.int32, cap, i_off)
            v = ids[:cnt]
            assert 0 <= cnt <= cap and (ids[cnt:] == -1).all()
            assert (v >= 0).all() and (v < n).all() and (v != nid).all() and len(set(v.tolist())) == cnt
            deg0 += cnt if l == 0 else 0
    assert deg0 / n > 0.8 * m0                                 # lists are back-filled to the cap (hnsw.ts:555-571)

    # quality: recall@10 and work per query of the two-stage pipeline, GPU-built vs sequentially built graph
    rng = np.random.default_rng(54321)
    q = rng.standard_normal((200, dim)).astype(np.float32)
    qn = O.normalize_rows_f32(q)
    gt = _exact_topk(base, qn)
    tk, ef, ann_k = O.derive_search_params(10, preset)
    b.attachF32(base)
    gi, gl, gn, gctr = b.queryMany(q, tk, ef, ann_k, with_counters=True)
    ix_seq = O.OracleIndex(dim, p["m"], p["ef_construction"], 8, seed=1)
    assert ix_seq.load_bytes(seq_dump)
    si, sl, sn, _, _, sctr = O.search_many_db(ix_seq, q, base, 10, preset, threads=8)
    r_gpu, r_seq = _recall(gi, gt), _recall(si, gt)
    assert r_gpu >= r_seq - 0.02, (r_gpu, r_seq)
    assert abs(gctr[:, 0].mean() / sctr["n_dist"].mean() - 1) < 0.12, (gctr[:, 0].mean(), sctr["n_dist"].mean())

    # and the GPU-built dump is searched bit-exactly, like any dump
    oi, ol, on, _, _, octr = O.search_many_db(ix, q, base, 10, preset, threads=8)
    assert np.array_equal(gn, on) and np.array_equal(gi, oi) and np.array_equal(gl.view(np.uint64), ol.view(np.uint64))
    assert np.array_equal(gctr[:, 0], octr["n_dist"])


def test_gpu_build_small_and_limits(gpu):
    """Fewer nodes than the sequential host prefix (nothing reaches the GPU), a non-empty index (refused),
    rows whose stride is padded (dim 100), blocks capped small (many passes)."""
    from minivectordb_amd import HipBridge, MvdbError
    b8 = O.quantize_rows_i8(gaussian_unit(300, 32, 5))
    b = HipBridge()
    b.init(dict(dim=32, m=8, ef=40, capacity=300, seed=7))
    st = b.buildGpu(b8)
    assert st["host_linked"] == 300 and st["passes"] == 0 and b.elementCount() == 300
    ix = O.OracleIndex(32, 8, 40, 300, seed=7)
    ix.insert_many(np.arange(300, dtype=np.int32), b8)
    assert b.saveBytes() == ix.save_bytes()                    # below the prefix it IS the sequential build
    with pytest.raises(MvdbError):
        b.buildGpu(b8)
    b8 = O.quantize_rows_i8(gaussian_unit(5000, 100, 6))
    b = HipBridge()
    b.init(dict(dim=100, m=12, ef=150, capacity=6000, seed=3))
    st = b.buildGpu(b8, max_block=100)
    assert st["passes"] >= 29 and b.elementCount() == 5000 and b.getMaxElements() == 6000
    ix = O.OracleIndex(100, 12, 150, 8, seed=1)
    assert ix.load_bytes(b.saveBytes())
    q8 = O.quantize_rows_i8(gaussian_unit(32, 100, 8))
    ix.set_ef_search(80); b.setEfSearch(80)
    oi, od, oc, _ = ix.search_many(q8, 50, threads=8)
    gi, gd, gc = b.searchManyArrays(q8, 50)
    assert np.array_equal(gc, oc) and np.array_equal(gi, oi)
    # self-hit: every stored row finds itself first
    gi, gd, gc = b.searchManyArrays(b8[:64], 1)
    assert (gd[:, 0] == 0).all()
    # max_block below the 64-node floor of a pass: the request buffers are sized for the floor
    b = HipBridge()
    b.init(dict(dim=100, m=12, ef=150, capacity=6000, seed=3))
    st = b.buildGpu(b8[:4000], max_block=8)
    assert b.elementCount() == 4000 and st["passes"] >= 30
    ix = O.OracleIndex(100, 12, 150, 8, seed=1)
    assert ix.load_bytes(b.saveBytes())
    ix.set_ef_search(80); b.setEfSearch(80)
    oi, od, oc, _ = ix.search_many(q8, 50, threads=8)
    gi, gd, gc = b.searchManyArrays(q8, 50)
    assert np.array_equal(gc, oc) and np.array_equal(gi, oi)


def test_gpu_build_failure_takes_the_unlinked_nodes_back(gpu):
    """A failure after the host prefix (device memory, a copy, a launch; forced through
    mvdb_debug_inject_build_failure) must leave the index holding exactly the nodes it reports: the rest absent
    again, the RNG stream where sequential inserts of the linked nodes would have left it -- so inserting the
    remaining ids through insert() gives the levels one build over all the data gives."""
    from minivectordb_amd import HipBridge, MvdbError
    dim, n = 48, 9000
    b8 = O.quantize_rows_i8(gaussian_unit(n, dim, 21))
    cfg = dict(dim=dim, m=8, ef=60, capacity=n, seed=5)
    whole = HipBridge()
    whole.init(cfg)
    whole.buildGpu(b8)
    b = HipBridge()
    b.init(cfg)
    b._L.mvdb_debug_inject_build_failure(b.handle, 3)
    with pytest.raises(MvdbError, match="injected"):
        b.buildGpu(b8)
    b._L.mvdb_debug_inject_build_failure(b.handle, -1)
    k = b.elementCount()
    assert 2048 <= k < n                                       # the host prefix stays, nothing beyond it
    recs = D.walk(b.saveBytes())
    assert [r[0] for r in recs] == list(range(k))
    ix = O.OracleIndex(dim, 8, 60, 8, seed=1)
    assert ix.load_bytes(b.saveBytes())                        # a consistent V3 dump
    b.appendGpu(b8[k:])                                        # the remaining ids go in by any path
    assert b.elementCount() == n
    ref = D.walk(whole.saveBytes())
    assert [(r[0], r[1]) for r in D.walk(b.saveBytes())] == [(r[0], r[1]) for r in ref]   # levels of ONE build
    gi, gd, gc = b.searchManyArrays(b8[:64], 1)
    assert (gd[:, 0] == 0).all()


def test_gpu_build_is_deterministic(gpu):
    from minivectordb_amd import HipBridge
    b8 = O.quantize_rows_i8(gaussian_unit(12000, 48, 9))
    dumps = []
    for _ in range(2):
        b = HipBridge()
        b.init(dict(dim=48, m=16, ef=100, capacity=12000, seed=11))
        b.buildGpu(b8)
        dumps.append(b.saveBytes())
        b.close()
    assert dumps[0] == dumps[1]


def test_gpu_append_into_a_live_index(gpu):
    """mvdb_append_gpu: a bulk insert of fresh consecutive ids into an index that is being searched.  Levels and rows
    equal those of one build over all the data (the RNG stream goes on in id order), the dump is a valid V3 file,
    searches after the append are bit-exact against the oracle on that dump, quality matches a single build, and
    the misuse cases are errors."""
    from minivectordb_amd import HipBridge, MvdbError
    n0, n1, dim, preset = 30000, 12000, 64, "balanced"
    p = O.PRESETS[preset]
    rng = np.random.default_rng(7)
    base = O.normalize_rows_f32(rng.standard_normal((n0 + n1, dim), dtype=np.float32))
    b8 = O.quantize_rows_i8(base)
    cfg = dict(dim=dim, m=p["m"], ef=p["ef_construction"], capacity=n0 + n1, seed=42, resultsCap=4096)
    whole = HipBridge()
    whole.init(cfg)
    whole.buildGpu(b8)
    b = HipBridge()
    b.init(cfg)
    b.buildGpu(b8[:n0])
    q = gaussian_unit(300, dim, 99)
    q8 = O.quantize_rows_i8(q)
    b.setEfSearch(p["baseEfSearch"])
    before = b.searchManyArrays(q8, 100)                         # the index is live: device mirror uploaded
    assert before[0].max() < n0
    st = b.appendGpu(b8[n0:])
    assert b.elementCount() == n0 + n1 and st["passes"] >= 1 and st["n_dist"] > 0
    dump = b.saveBytes()
    ix = O.OracleIndex(dim, p["m"], p["ef_construction"], 8, seed=1)
    assert ix.load_bytes(dump) and ix.save_bytes() == dump
    recs, ref = D.walk(dump), D.walk(whole.saveBytes())
    assert [(r[0], r[1]) for r in recs] == [(r[0], r[1]) for r in ref]      # same ids, same levels as one build
    for nid, lvl, vec_off, lists in recs[::997]:
        assert dump[vec_off:vec_off + dim] == b8[nid].tobytes()
    # bit-exact search on the appended index, new nodes are found
    ix.set_ef_search(p["baseEfSearch"]); ix.set_results_cap(4096)
    ids, dist, cnt, ctr = b.searchManyArrays(q8, 100, with_counters=True)
    oi, od, oc, octr = ix.search_many(q8, 100, threads=8)
    assert np.array_equal(cnt, oc) and np.array_equal(ids, oi) and np.array_equal(dist.view(np.uint32), od.view(np.uint32))
    assert np.array_equal(ctr[:, 0], octr["n_dist"])
    assert (ids >= n0).any()
    self_q = b.searchManyArrays(b8[n0 + 5:n0 + 6], 1)
    assert self_q[0][0, 0] == n0 + 5 and self_q[1][0, 0] == 0.0
    # quality: recall against the exact answer as good as one build over everything
    gt = _exact_topk(base, O.normalize_rows_f32(q))
    whole.setEfSearch(p["baseEfSearch"])
    r_app, r_whole = _recall(ids, gt), _recall(whole.searchManyArrays(q8, 100)[0], gt)
    assert r_app >= r_whole - 0.02, (r_app, r_whole)
    # misuse
    with pytest.raises(MvdbError, match="capacity"):
        b.appendGpu(b8[:10])
    holes = HipBridge()
    holes.init(cfg)
    holes.insertManyPacked(np.array([0, 1, 3], np.int32), b8[:3], dim)
    with pytest.raises(MvdbError, match="dense"):
        holes.appendGpu(b8[:10])
    # an append into an EMPTY index is a build; a small append stays on the host prefix path
    e = HipBridge()
    e.init(cfg)
    e.appendGpu(b8[:500])
    e.appendGpu(b8[500:1500])
    assert e.elementCount() == 1500
    ix2 = O.OracleIndex(dim, p["m"], p["ef_construction"], 8, seed=1)
    assert ix2.load_bytes(e.saveBytes())
    for x in (whole, b, holes, e):
        x.close()


def test_db_facade_bulk_insert_option(gpu):
    """MiniVectorDB.insertMany with the engine option bulkInsert="gpu": large batches of fresh ids go through
    mvdb_append_gpu, small ones and upserts through insert; search, save and re-open work as with the ordinary path."""
    import tempfile
    from minivectordb_amd.db import MiniVectorDB
    dim, n = 64, 9000
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        db = MiniVectorDB.open(dict(storageDir=d, dim=dim, mode="fast", capacity=20000, preloadVectors=True,
                                    bulkInsert="gpu", bulkInsertMin=2000))
        db.insertMany([dict(id=f"a-{i}", input=x[i], metadata=dict(i=i)) for i in range(100)])        # small: host path
        db.insertMany([dict(id=f"a-{i}", input=x[i], metadata=dict(i=i)) for i in range(100, 6000)])  # bulk: GPU
        assert db.wasm.uploadCounters()["full_uploads"] >= 0
        db.insertMany([dict(id=f"a-{i}", input=x[i], metadata=dict(i=i)) for i in range(5990, n)])    # upserts + new: host
        res = db.searchMany([x[5], x[3000], x[8999]], dict(topK=3))
        assert [r[0]["id"] for r in res] == ["a-5", "a-3000", "a-8999"]
        assert all(abs(r[0]["score"]) < 1e-6 for r in res)
        db.save()
        db.close()
        db2 = MiniVectorDB.open(dict(storageDir=d, dim=dim, mode="fast", capacity=20000, preloadVectors=True))
        res2 = db2.searchMany([x[5], x[3000], x[8999]], dict(topK=3))
        assert res2 == res
        db2.close()
