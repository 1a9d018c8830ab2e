"""The no-allocation entry point (include/mvdb_hip.h: mvdb_search_many_dev_noalloc; it has no counterpart in
the reference) against the entry point it is a variant of, mvdb_search_many_dev: candidate ids, distances,
counts and work counters bit-identical, through the C ABI, on two calls in a row -- the second with a
smaller input, so stale bits of the first call in the shared static scratch would show."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _call(fn, b, q8_dev, nq, k, torch, dev):
    ids = torch.full((nq, k), -7, dtype=torch.int32, device=dev)
    dist = torch.full((nq, k), -7.0, dtype=torch.float32, device=dev)
    cnt = torch.full((nq,), -7, dtype=torch.int32, device=dev)
    ctr = torch.full((nq, 4), -7, dtype=torch.int32, device=dev)
    vp = C.c_void_p
    stream = torch.cuda.current_stream(dev).cuda_stream
    ok = fn(b.handle, vp(q8_dev.data_ptr()), nq, k, vp(ids.data_ptr()), vp(dist.data_ptr()), vp(cnt.data_ptr()),
            vp(ctr.data_ptr()), vp(stream))
    torch.cuda.synchronize(dev)
    return ok, ids.cpu().numpy(), dist.cpu().numpy(), cnt.cpu().numpy(), ctr.cpu().numpy()


def test_noalloc_equals_search_many_dev(gpu):
    import torch
    from tests._data import bridge_from_dump, built_index, oracle_from_dump
    dev = torch.device("cuda", 0)
    dump, base, b8 = built_index(4000, 384, "balanced")
    b = bridge_from_dump(dump, 384, "balanced")
    ix = oracle_from_dump(dump, 384, "balanced")
    b.setResultsCap(4096); ix.set_results_cap(4096)
    b.setEfSearch(140); ix.set_ef_search(140)
    L = b._L
    rng = np.random.default_rng(123)
    q_big = O.quantize_rows_i8(O.normalize_rows_f32(rng.standard_normal((200, 384), dtype=np.float32)))
    q_small = O.quantize_rows_i8(O.normalize_rows_f32(rng.standard_normal((37, 384), dtype=np.float32)))

    # before mvdb_upload the call must refuse (it may not allocate the device mirror itself)
    qd = torch.from_numpy(q_big).to(dev)
    ok, *_ = _call(L.mvdb_search_many_dev_noalloc, b, qd, 200, 1200, torch, dev)
    assert ok == 0 and b"mvdb_upload" in L.mvdb_last_error(b.handle)
    assert L.mvdb_upload(b.handle) == 1

    for q8, k in ((q_big, 1200), (q_small, 300)):           # two calls in a row, the second smaller
        nq = q8.shape[0]
        qd = torch.from_numpy(q8).to(dev)
        ok_a, ids_a, dist_a, cnt_a, ctr_a = _call(L.mvdb_search_many_dev_noalloc, b, qd, nq, k, torch, dev)
        ok_b, ids_b, dist_b, cnt_b, ctr_b = _call(L.mvdb_search_many_dev, b, qd, nq, k, torch, dev)
        assert ok_a == 1 and ok_b == 1, L.mvdb_last_error(b.handle)
        assert np.array_equal(cnt_a, cnt_b) and np.array_equal(ids_a, ids_b)
        assert np.array_equal(dist_a.view(np.uint32), dist_b.view(np.uint32))
        assert np.array_equal(ctr_a, ctr_b)
        oi, od, oc, _ = ix.search_many(q8, k, threads=8)     # and both equal the oracle
        assert np.array_equal(cnt_a, oc) and np.array_equal(ids_a, oi)
        assert np.array_equal(dist_a.view(np.uint32), od.view(np.uint32))
    assert L.mvdb_check_errors(b.handle) == 1

    # The property itself, not only the output: under stream capture (error mode "global") a hipMalloc, hipFree or
    # synchronisation anywhere in the call fails the capture.  Timing is switched on for the handle, so an event
    # pair created per launch (StageTimer) would show as well.  The replayed graph must give the same bits.
    L.mvdb_enable_timing(b.handle, 1)
    nq, k = q_small.shape[0], 300
    qd = torch.from_numpy(q_small).to(dev)
    ids = torch.full((nq, k), -7, dtype=torch.int32, device=dev)
    dist = torch.full((nq, k), -7.0, dtype=torch.float32, device=dev)
    cnt = torch.full((nq,), -7, dtype=torch.int32, device=dev)
    ctr = torch.full((nq, 4), -7, dtype=torch.int32, device=dev)
    free_before = torch.cuda.mem_get_info(dev)[0]
    vp = C.c_void_p
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize(dev)
    with torch.cuda.graph(g):
        ok = L.mvdb_search_many_dev_noalloc(b.handle, vp(qd.data_ptr()), nq, k, vp(ids.data_ptr()), vp(dist.data_ptr()),
                                            vp(cnt.data_ptr()), vp(ctr.data_ptr()),
                                            vp(torch.cuda.current_stream(dev).cuda_stream))
    assert ok == 1, L.mvdb_last_error(b.handle)
    for _ in range(2):
        ids.fill_(-7); cnt.fill_(-7)
        g.replay()
        torch.cuda.synchronize(dev)
        assert np.array_equal(cnt.cpu().numpy(), cnt_a) and np.array_equal(ids.cpu().numpy(), ids_a)
        assert np.array_equal(dist.cpu().numpy().view(np.uint32), dist_a.view(np.uint32))
    assert torch.cuda.mem_get_info(dev)[0] >= free_before - (8 << 20)     # the graph object itself, nothing per call
    L.mvdb_enable_timing(b.handle, 0)
    del g

    # beyond the static scratch: 4000 nodes -> 512 B of bitmap per query -> 131072 queries fit, 140000 do not
    too_many = torch.zeros((140000, 384), dtype=torch.int8, device=dev)
    ok, *_ = _call(L.mvdb_search_many_dev_noalloc, b, too_many, 140000, 1, torch, dev)
    assert ok == 0 and b"64 MiB" in L.mvdb_last_error(b.handle)
    # a change of the index needs a new mvdb_upload
    b.insertManyPacked(np.array([5], np.int32), b8[6], 384)
    ok, *_ = _call(L.mvdb_search_many_dev_noalloc, b, qd, 37, 300, torch, dev)
    assert ok == 0 and b"mvdb_upload" in L.mvdb_last_error(b.handle)
