"""GPU: the HIP path against the committed golden vectors, the reference's end-to-end tests
re-enacted through the MiniVectorDB mirror, and size-independent properties at a larger size."""
import os
import tempfile

import numpy as np
import pytest

from minivectordb_amd import HipBridge, MiniVectorDB
from oracle import oracle as O
from tests._golden import NAMES, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", NAMES)
def test_hip_path_reproduces_golden(gpu, name):
    g = load(name)
    b = HipBridge()
    b.init(dict(dim=g["dim"], m=g["m"], ef=g["ef_construction"], capacity=3, seed=1, resultsCap=4096))
    b.loadBytes(g["dump"])
    b.attachF32(g["base_f32"])
    qn, q8 = b.prepareQueries(g["queries"])
    assert np.array_equal(qn.view(np.uint32), g["q_norm"].view(np.uint32))
    assert np.array_equal(q8, g["q_i8"])
    b.setEfSearch(g["ef_search"])
    ids, dist, cnt, ctr = b.searchManyArrays(q8, g["ann_k"], with_counters=True)
    assert np.array_equal(cnt, g["ann_cnt"])
    for qi in range(len(cnt)):
        n = cnt[qi]
        assert np.array_equal(ids[qi, :n], g["ann_ids"][qi, :n])
        assert np.array_equal(dist[qi, :n].view(np.uint32), g["ann_dist"][qi, :n].view(np.uint32))
    assert np.array_equal(ctr[:, 0], g["n_dist"]) and np.array_equal(ctr[:, 1], g["n_l0lists"])
    assert np.array_equal(ctr[:, 2], g["n_upperlists"])
    ti, tl, tn = b.queryMany(g["queries"], g["topk"], g["ef_search"], g["ann_k"])
    assert np.array_equal(tn, g["top_n"]) and np.array_equal(ti, g["top_ids"])
    assert np.array_equal(tl.view(np.uint64), g["top_l2"].view(np.uint64))
    fi, fl, fn = b.queryMany(g["queries"], g["topk"], g["ef_search"], g["ann_k"], g["allow"])
    assert np.array_equal(fn, g["f_n"]) and np.array_equal(fi, g["f_ids"])
    assert np.array_equal(fl.view(np.uint64), g["f_l2"].view(np.uint64))
    si, sv, sc = b.scanMany(q8, 10, "l2")
    assert np.array_equal(sc, g["scan_cnt"]) and np.array_equal(si, g["scan_ids"]) and np.array_equal(sv, g["scan_val"])


def one_hot(dim, i):
    v = [0.0] * dim
    v[i % dim] = 1.0
    return v


def test_reference_search_many_test(gpu):
    """reference tests/search_many_test.ts: 128 one-hot docs, 64 one-hot queries, topK 3."""
    with tempfile.TemporaryDirectory() as d:
        db = MiniVectorDB.open(dict(storageDir=d, dim=384, mode="balanced", capacity=50000, preloadVectors=False))
        dim = db.getStats()["dim"]
        for i in range(128):
            db.insert(dict(id=f"d-{i}", input=one_hot(dim, i), metadata=dict(i=i)))
        results = db.searchMany([one_hot(dim, i) for i in range(64)], dict(topK=3))
        assert isinstance(results, list) and len(results) == 64
        for i in range(10):
            assert len(results[i]) > 0 and results[i][0]["id"] == f"d-{i}"
            assert results[i][0]["score"] == 0.0
        db.close()


def test_reference_basic_and_persistence_tests(gpu):
    """reference tests/basic_test.ts (top-1, object filter, save/reopen) and persistence_test.ts
    (A saved in the dump, B only in ann.oplog: both found after reopen)."""
    with tempfile.TemporaryDirectory() as d:
        opts = dict(storageDir=d, dim=384, mode="balanced", capacity=1000, preloadVectors=True)
        db = MiniVectorDB.open(opts)
        db.insert(dict(id="a", input=one_hot(384, 0), metadata=dict(kind="x")))
        db.insert(dict(id="b", input=one_hot(384, 1), metadata=dict(kind="y")))
        r = db.search(one_hot(384, 0), dict(topK=1))
        assert r[0]["id"] == "a"
        r = db.search(one_hot(384, 0), dict(topK=5, filter=dict(kind="y")))
        assert [x["id"] for x in r] == ["b"] and r[0]["score"] == 2.0
        r = db.search(one_hot(384, 0), dict(topK=5, filter=lambda m: m["kind"] == "x", score="cosine"))
        assert [x["id"] for x in r] == ["a"] and r[0]["score"] == 1.0
        db.save()
        db.insert(dict(id="c", input=one_hot(384, 2), metadata=dict(kind="z")))   # not saved: oplog only
        db.close()
        for f in ("dump.bin", "vectors.f32.bin", "ann.oplog"):
            assert os.path.getsize(os.path.join(d, f)) > 0
        db2 = MiniVectorDB.open(opts)
        assert db2.search(one_hot(384, 0), dict(topK=1))[0]["id"] == "a"
        assert db2.search(one_hot(384, 2), dict(topK=1))[0]["id"] == "c"
        db2.close()


def test_reference_rebuild_threshold_and_capacity_tests(gpu):
    """reference tests/rebuild_threshold_test.ts (deleted ids never returned) and
    capacity_overflow_test.ts (insert beyond capacity raises with capacity/rebuild text)."""
    with tempfile.TemporaryDirectory() as d:
        db = MiniVectorDB.open(dict(storageDir=d, dim=384, mode="balanced", capacity=200))
        db.insertMany([dict(id=f"doc-{i}", input=one_hot(384, i), metadata=dict(i=i)) for i in range(200)])
        assert db.removeMany([f"doc-{i}" for i in range(120)]) == 120
        for i in range(10):
            res = db.search(one_hot(384, i), dict(topK=10))
            assert all(r["id"] != f"doc-{i}" for r in res)
            assert all(int(r["id"].split("-")[1]) >= 120 for r in res)
        with pytest.raises(Exception) as e:
            db.insert(dict(id="overflow", input=one_hot(384, 3)))
        assert "capacity" in str(e.value).lower() and "rebuild" in str(e.value).lower()
        db.close()


def test_config_mismatch_dump_is_not_surfaced(gpu):
    """reference tests/config_mismatch_test.ts: a dump saved at dim 384 reopened at dim 512."""
    with tempfile.TemporaryDirectory() as d:
        db = MiniVectorDB.open(dict(storageDir=d, dim=384, capacity=100))
        db.insert(dict(id="old", input=one_hot(384, 0)))
        db.save()
        db.close()
        os.remove(os.path.join(d, "metadata.json"))
        os.remove(os.path.join(d, "vectors.f32.bin"))
        db2 = MiniVectorDB.open(dict(storageDir=d, dim=512, capacity=100))
        assert db2.wasm.elementCount() == 0
        db2.insert(dict(id="new", input=one_hot(512, 0)))
        assert [r["id"] for r in db2.search(one_hot(512, 0), dict(topK=5))] == ["new"]
        db2.close()


def test_properties_at_scale(gpu):
    """120k x 384 balanced graph from the engine's parallel builder (too large for a committed
    fixture): size-independent properties of the full pipeline + oracle equality on a sample."""
    n, dim, nq = 120_000, 384, 1000
    rng = np.random.default_rng(12345)
    b = HipBridge()
    b.init(dict(dim=dim, m=16, ef=300, capacity=n, seed=42, resultsCap=4096))
    base, b8 = b.prepareQueries(rng.standard_normal((n, dim), dtype=np.float32))
    assert np.all(np.abs(b8.astype(np.int32)) <= 127)
    b.buildParallel(b8, 16)
    b.attachF32(base)
    dump = b.saveBytes()
    b2 = HipBridge()
    b2.init(dict(dim=dim, m=16, ef=300, capacity=3, seed=1, resultsCap=4096))
    b2.loadBytes(dump)
    assert b2.saveBytes() == dump                                    # dump round trip at size
    q = np.random.default_rng(54321).standard_normal((nq, dim), dtype=np.float32)
    qn, q8 = b.prepareQueries(q)
    b.setEfSearch(140)
    ids, dist, cnt = b.searchManyArrays(q8, 1200)
    assert np.all(cnt == 1200)
    assert np.all(np.diff(dist, axis=1) >= 0)                        # ascending candidate lists
    assert all(len(set(r.tolist())) == 1200 for r in ids[:200])      # no duplicate ids
    chk = b.distMany(q8[:50], ids[:50, :64])                         # listed distances are the true int8 L2
    assert np.array_equal(chk.astype(np.float32), dist[:50, :64])
    ids2, dist2, cnt2 = b.searchManyArrays(q8, 1200)                 # idempotent (visited scratch fully reset)
    assert np.array_equal(ids, ids2) and np.array_equal(dist, dist2)
    ti, tl, tn = b.queryMany(q, 10, 140, 1200)
    assert np.all(tn == 10) and np.all(np.diff(tl, axis=1) >= 0)
    for qi in range(0, nq, 97):                                       # final ids come from the candidate list
        assert set(ti[qi].tolist()) <= set(ids[qi].tolist())
    sub = slice(0, 48)                                                # oracle equality on a sample
    ix = O.OracleIndex(dim, 16, 300, 3, seed=1)
    assert ix.load_bytes(dump)
    oi, ol, on, ann, acnt, _ = O.search_many_db(ix, q[sub], base, 10, "balanced", threads=8)
    assert np.array_equal(ann, ids[sub]) and np.array_equal(oi, ti[sub])
    assert np.array_equal(ol.view(np.uint64), tl[sub].view(np.uint64))


def test_merge_topk_kernel_matches_host_merge(gpu):
    """Sharded path: the HIP merge of per-shard top-k lists == the (l2, id) host merge."""
    import torch

    from minivectordb_amd.distributed import merge_topk_host
    from minivectordb_amd.pipeline import ResidentSearcher
    b = HipBridge()
    b.init(dict(dim=16, m=4, ef=20, capacity=4, seed=1))
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8)
    parts, nq, topk = 8, 300, 10
    l2 = np.sort(rng.integers(0, 40, (parts, nq, topk)).astype(np.float64) / 8, axis=2)   # many ties across shards
    ids = rng.permutation(parts * nq * topk).astype(np.int32).reshape(parts, nq, topk)
    for p in range(parts):                                                               # ids ascending inside ties
        for q in range(nq):
            order = np.lexsort((ids[p, q], l2[p, q]))
            ids[p, q], l2[p, q] = ids[p, q][order], l2[p, q][order]
    n = rng.integers(0, topk + 1, (parts, nq)).astype(np.int32)
    s = ResidentSearcher(b, nq, topk, dev)
    oi, ol, on = s.merge_topk(torch.from_numpy(ids).to(dev), torch.from_numpy(l2).to(dev),
                              torch.from_numpy(n).to(dev), parts, nq)
    ei, el, en = merge_topk_host(ids, l2, n, topk)
    on = on.cpu().numpy()
    assert np.array_equal(on, en)
    for q in range(nq):
        assert np.array_equal(oi.cpu().numpy()[q, :en[q]], ei[q, :en[q]])
        assert np.array_equal(ol.cpu().numpy()[q, :en[q]], el[q, :en[q]])
    # the packed form (one block per shard, as ONE all-gather delivers them), with an aborted query on one shard
    from minivectordb_amd.distributed import TopkBlock
    n[3, 17] = -1
    full = torch.stack([TopkBlock(nq, topk).fill(ids[p], l2[p], n[p]).buf for p in range(parts)]).to(dev)
    blk = s.merge_blocks(full, nq)
    ei, el, en = merge_topk_host(ids, l2, n, topk)
    assert en[17] == -1 and np.array_equal(blk.n.cpu().numpy(), en)
    for q in range(nq):
        k = max(en[q], 0)
        assert np.array_equal(blk.ids.cpu().numpy()[q, :k], ei[q, :k])
        assert np.array_equal(blk.l2.cpu().numpy()[q, :k], el[q, :k])


def test_node_napi_bridge(gpu):
    """The N-API addon + node/hip-bridge.js (the WasmBridge drop-in for the Node.js host)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("node") is None or not os.path.exists(os.path.join(root, "node", "mvdb_napi.node")):
        pytest.skip("node or the built addon is not present on this box")
    r = subprocess.run(["node", os.path.join(root, "node", "test_bridge.js")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "node bridge test passed" in r.stdout


def test_pipelined_batches_equal_single_stream(gpu):
    """ResidentSearcher.submit (traversal and rerank of consecutive batches on two streams, small rerank
    workgroups) returns the same bits as query_many, batch after batch, with buffers reused."""
    import os
    import torch
    from minivectordb_amd.pipeline import ResidentSearcher
    from tests._data import bridge_from_dump, built_index, gaussian_unit
    dump, base, b8 = built_index(4000, 384, "balanced")
    b = bridge_from_dump(dump, 384, "balanced")
    b.attachF32(base, resident=True)
    dev = torch.device("cuda", 0)
    s = ResidentSearcher(b, 200, 10, dev)
    b.setTuning(rerank_waves=2)
    try:
        batches = [torch.from_numpy(gaussian_unit(200, 384, 100 + i)).to(dev) for i in range(5)]
        outs = []
        for i, q in enumerate(batches + batches + batches):
            if i < len(batches):
                slot, ids, l2, n = s.submit(q, 140, 1200)                      # rerank on a second stream
            elif i < 2 * len(batches):
                slot, ids, l2, n = s.submit_interleaved(q, 140, 1200)          # two whole batches in flight
            else:
                slot, ids, l2, n = s.submit_interleaved(q, 140, 1200, depth=4)  # four
            s.wait(slot)
            outs.append((ids.clone(), l2.clone(), n.clone()))  # on the current stream, after the wait
        s.drain()
        torch.cuda.synchronize(dev)
    finally:
        b.setTuning()
    b.checkErrors()
    for q, (ids, l2, n) in zip(batches + batches + batches, outs):
        r_ids, r_l2, r_n = s.query_many(q, 140, 1200)
        torch.cuda.synchronize(dev)
        assert torch.equal(ids, r_ids) and torch.equal(l2, r_l2) and torch.equal(n, r_n)


def test_host_slots_keep_batches_in_flight(gpu):
    """mvdb_query_many_submit / _wait (the path a Node or C host uses to keep several batches in flight): results
    equal the oracle's searchMany restatement and the one-at-a-time mvdb_query_many, bit for bit; ragged batch
    sizes, an allow bitmap on one slot, slots reused; misuse is an error, not undefined behaviour."""
    from minivectordb_amd import MvdbError, derive_search_params, resolve_preset
    from tests._data import bridge_from_dump, built_index, gaussian_unit
    dump, base, b8 = built_index(4000, 384, "balanced")
    b = bridge_from_dump(dump, 384, "balanced")
    b.attachF32(base, resident=True)
    tk, ef, ann_k = derive_search_params(10, resolve_preset("balanced"))
    allow = np.zeros((4000 + 31) // 32, np.uint32)
    allow[::2] = 0xFFFFFFFF
    sizes = [200, 37, 1, 128, 200, 64, 5]
    batches = [gaussian_unit(n, 384, 300 + i) for i, n in enumerate(sizes)]
    masks = [allow if i == 3 else None for i in range(len(sizes))]
    got, pending = [None] * len(sizes), []
    for i, q in enumerate(batches):
        if len(pending) == 3:
            j, slot = pending.pop(0)
            got[j] = b.waitMany(slot)
        b.submitMany(i % 3, q, tk, ef, ann_k, masks[i])
        pending.append((i, i % 3))
    for j, slot in pending:
        got[j] = b.waitMany(slot)
    for q, m, (ids, l2, n) in zip(batches, masks, got):
        r_ids, r_l2, r_n = b.queryMany(q, tk, ef, ann_k, m)
        assert np.array_equal(n, r_n)
        for qi in range(q.shape[0]):
            assert np.array_equal(ids[qi, :n[qi]], r_ids[qi, :n[qi]])
            assert np.array_equal(l2[qi, :n[qi]].view(np.uint64), r_l2[qi, :n[qi]].view(np.uint64))
    # misuse
    with pytest.raises(MvdbError):
        b.waitMany(1)                                  # nothing submitted
    b.submitMany(2, batches[1], tk, ef, ann_k)
    with pytest.raises(MvdbError):
        b.submitMany(2, batches[2], tk, ef, ann_k)     # slot busy
    with pytest.raises(MvdbError):
        b.submitMany(4, batches[2], tk, ef, ann_k)     # no such slot
    ids, l2, n = b.waitMany(2)
    assert np.array_equal(n, got[1][2])
    b.close()
