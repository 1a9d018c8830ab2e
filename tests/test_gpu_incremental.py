"""Incremental device update: after inserts / upserts / oplog replay only the rows of the touched
nodes are patched into the device mirror (DeviceIndex::patch).  The mirror must then answer exactly like a fresh
full upload of the same host graph -- candidate lists, distances, work counters, exact scan -- and the counters of
the handle must show that no second full upload happened."""
import numpy as np
import pytest

from oracle import oracle as O
from tests._data import gaussian_unit

pytestmark = pytest.mark.gpu


def _fresh(dump, dim, m, efc):
    from minivectordb_amd import HipBridge
    b = HipBridge()
    b.init(dict(dim=dim, m=m, ef=efc, capacity=8, seed=1))
    b.loadBytes(dump)
    b.setResultsCap(4096)
    return b


def _same_answers(b, ref, q8, k, ef):
    for x in (b, ref):
        x.setEfSearch(ef)
    a = b.searchManyArrays(q8, k, with_counters=True)
    r = ref.searchManyArrays(q8, k, with_counters=True)
    for x, y in zip(a, r):
        assert np.array_equal(x, y)
    for metric in ("l2", "dot"):
        sa, sr = b.scanMany(q8[:40], 10, metric), ref.scanMany(q8[:40], 10, metric)
        for x, y in zip(sa, sr):
            assert np.array_equal(x, y)


def test_patched_mirror_equals_full_upload(gpu):
    from minivectordb_amd import HipBridge
    dim, m, efc, n, cap = 64, 16, 100, 9000, 10000
    rng = np.random.default_rng(17)
    b8 = O.quantize_rows_i8(gaussian_unit(n, dim, 3))
    ix = O.OracleIndex(dim, m, efc, cap, seed=42)
    ix.insert_many(np.arange(n, dtype=np.int32), b8)
    b = HipBridge()
    b.init(dict(dim=dim, m=m, ef=efc, capacity=cap, seed=42))
    b.loadBytes(ix.save_bytes())
    ix.close()
    b.setResultsCap(4096)
    q8 = O.quantize_rows_i8(gaussian_unit(96, dim, 5))
    _same_answers(b, _fresh(b.saveBytes(), dim, m, efc), q8, 200, 200)
    assert b.uploadCounters() == dict(full_uploads=1, patches=0, nodes_patched=0)
    nxt = n
    for rnd in range(4):
        new_ids = np.arange(nxt, nxt + 20, dtype=np.int32)                       # inserts (new ids, new upper lists)
        nxt += 20
        new_vec = O.quantize_rows_i8(gaussian_unit(20, dim, 100 + rnd))
        b.insertManyPacked(new_ids, new_vec.reshape(-1), dim)
        ups = rng.integers(0, nxt, 10).astype(np.int32)                          # upserts: update_and_reconnect
        up_vec = O.quantize_rows_i8(gaussian_unit(10, dim, 200 + rnd))
        b.insertManyPacked(ups, up_vec.reshape(-1), dim)
        b.updateVector(int(ups[0]), up_vec[1])
        dump = b.saveBytes()
        _same_answers(b, _fresh(dump, dim, m, efc), q8, 200, 200)
        ox = O.OracleIndex(dim, m, efc, 8, seed=1)                               # and both == the oracle on that dump
        assert ox.load_bytes(dump)
        ox.set_ef_search(200); ox.set_results_cap(4096)
        oi, od, oc, octr = ox.search_many(q8, 200, threads=8)
        gi, gd, gc, gctr = b.searchManyArrays(q8, 200, with_counters=True)
        assert np.array_equal(gi, oi) and np.array_equal(gc, oc) and np.array_equal(gctr[:, 0], octr["n_dist"])
        c = b.uploadCounters()
        assert c["full_uploads"] == 1 and c["patches"] == rnd + 1 and 0 < c["nodes_patched"] < (rnd + 1) * 2500
    # the new rows really are searched: every inserted vector finds itself
    gi, gd, gc = b.searchManyArrays(new_vec, 1)
    assert (gd[:, 0] == 0).all()


def test_patch_falls_back_when_a_foreign_dump_names_absent_ids(gpu):
    """A dump whose lists name an in-range ABSENT id (tests/test_gpu_guards.py): inserting that id changes the
    canonical form of lists nobody edited, so the mirror must be rebuilt, not patched."""
    from minivectordb_amd import HipBridge
    from tests import _dump as D
    from tests.test_gpu_guards import _craft, _sparse_index
    dim, m, efc, n, cap = 32, 8, 60, 1500, 2000
    dump, ids = _sparse_index(n, cap, dim, m, efc, 13)
    crafted, done = _craft(dump, ids, cap, 5)
    b = HipBridge()
    b.init(dict(dim=dim, m=m, ef=efc, capacity=8, seed=1))
    b.loadBytes(crafted)
    b.setResultsCap(4096)
    q8 = O.quantize_rows_i8(gaussian_unit(48, dim, 5))
    b.searchManyArrays(q8, 50)
    absent = sorted(set(range(cap)) - set(ids.tolist()))
    buf = np.frombuffer(crafted, np.uint8)
    named = [a for a in absent if any(D.get_i32(crafted, i_off + 4 * j) == a for r in D.walk(crafted) for (l, c_off, i_off, capl) in r[3]
                                      for j in range(D.get_i32(crafted, c_off)))][:3]
    assert named, "the crafted dump names absent ids"
    vec = O.quantize_rows_i8(gaussian_unit(len(named), dim, 9))
    b.insertManyPacked(np.array(named, np.int32), vec.reshape(-1), dim)
    _same_answers(b, _fresh(b.saveBytes(), dim, m, efc), q8, 100, 100)
    assert b.uploadCounters()["full_uploads"] == 2


def test_patch_takes_the_full_upload_when_a_batch_outgrows_the_upper_slack(gpu):
    """The device keeps max(4096, lists/8) spare upper lists; a dirty batch whose SUMMED levels exceed what is left
    must take the full upload (which re-sizes the slack) -- not pass a per-node check and write past the array."""
    from minivectordb_amd import HipBridge
    dim, m, efc, n, cap = 32, 8, 40, 3000, 40000      # cap / 4 dirty nodes is the patch limit: 700 inserts stay below
    b8 = O.quantize_rows_i8(gaussian_unit(n, dim, 3))
    b = HipBridge()
    b.init(dict(dim=dim, m=m, ef=efc, capacity=cap, seed=42))
    b.insertManyPacked(np.arange(n, dtype=np.int32), b8.reshape(-1), dim)
    b.setResultsCap(4096)
    q8 = O.quantize_rows_i8(gaussian_unit(64, dim, 5))
    b.setEfSearch(100)
    b.searchManyArrays(q8, 100)
    assert b.uploadCounters()["full_uploads"] == 1
    nxt, seen_second = n, False
    for rnd in range(9):                      # ~0.875 upper lists per insert: 9 x 700 inserts cross 4096 lists of slack
        ids = np.arange(nxt, nxt + 700, dtype=np.int32)
        nxt += 700
        b.insertManyPacked(ids, O.quantize_rows_i8(gaussian_unit(700, dim, 300 + rnd)).reshape(-1), dim)
        _same_answers(b, _fresh(b.saveBytes(), dim, m, efc), q8, 100, 100)
        seen_second = seen_second or b.uploadCounters()["full_uploads"] >= 2
    c = b.uploadCounters()
    assert seen_second and c["full_uploads"] == 2 and c["patches"] >= 6, c
