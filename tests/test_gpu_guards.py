"""Graph guards of the traversal (reference assembly/hnsw.ts:480-492 layer 0, :986-994 descent)
on the GPU: dumps whose lists hold what a clean build never writes -- an in-range ABSENT id, a DUPLICATE id, a
SELF edge, a -1 hole inside the counted prefix, counts shorter than the stored ids, and an upper-layer edge to a
node whose level is lower than that layer.  load_index keeps all of these (it only rewrites ids outside
[0, max_elements), :1222-1229); the engine canonicalises lists at upload (device_index.hpp), so candidate
lists, distances AND the per-query work counters must still equal the oracle's for every kernel variant."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import _dump as D
from tests.test_gpu_traverse_modes import _same

pytestmark = pytest.mark.gpu


def _sparse_index(n, cap, dim, m, efc, seed):
    rng = np.random.default_rng(seed)
    ids = np.sort(rng.choice(cap, size=n, replace=False)).astype(np.int32)
    b8 = O.quantize_rows_i8(O.normalize_rows_f32(rng.standard_normal((n, dim), dtype=np.float32)))
    ix = O.OracleIndex(dim, m, efc, cap, seed=seed)
    ix.insert_many(ids, b8)
    dump = ix.save_bytes()
    ix.close()
    return dump, ids


def _craft(dump, ids, cap, seed, per_kind=60):
    """Patch list slots in place; returns (bytes, how many patches of each kind were applied)."""
    rng = np.random.default_rng(seed)
    buf = bytearray(dump)
    recs = D.walk(dump)
    present = set(ids.tolist())
    absent = np.array(sorted(set(range(cap)) - present), np.int32)
    level_of = {r[0]: r[1] for r in recs}
    done = dict(absent=0, dup=0, self=0, hole=0, short=0, low=0, swap=0)
    order = rng.permutation(len(recs))
    it = iter(order.tolist())

    def pick(pred):
        for i in it:
            if pred(recs[i]):
                return recs[i]
        return None

    for _ in range(per_kind):
        # in-range absent id in a counted slot (skipped by node_ptr == 0, :490-492 / :994)
        r = pick(lambda r: D.get_i32(buf, r[3][0][1]) >= 3)
        if r is None:
            break
        l, c_off, i_off, capl = r[3][int(rng.integers(0, len(r[3])))]
        cnt = D.get_i32(buf, c_off)
        if cnt > 0:
            D.put_i32(buf, i_off + 4 * int(rng.integers(0, cnt)), int(absent[rng.integers(0, absent.size)]))
            done["absent"] += 1
        # duplicate: slot j repeats slot 0 (second occurrence is visited already on layer 0; re-evaluated in the descent)
        r = pick(lambda r: D.get_i32(buf, r[3][0][1]) >= 3)
        l, c_off, i_off, capl = r[3][int(rng.integers(0, len(r[3])))]
        cnt = D.get_i32(buf, c_off)
        if cnt >= 2:
            D.put_i32(buf, i_off + 4 * int(rng.integers(1, cnt)), D.get_i32(buf, i_off))
            done["dup"] += 1
        # self edge
        r = pick(lambda r: D.get_i32(buf, r[3][0][1]) >= 2)
        l, c_off, i_off, capl = r[3][int(rng.integers(0, len(r[3])))]
        cnt = D.get_i32(buf, c_off)
        if cnt > 0:
            D.put_i32(buf, i_off + 4 * int(rng.integers(0, cnt)), r[0])
            done["self"] += 1
        # -1 hole inside the counted prefix (:488 skips it)
        r = pick(lambda r: D.get_i32(buf, r[3][0][1]) >= 3)
        l, c_off, i_off, capl = r[3][0]
        D.put_i32(buf, i_off + 4 * int(rng.integers(0, D.get_i32(buf, c_off))), -1)
        done["hole"] += 1
        # count shorter than the stored ids (ids beyond count are never read)
        r = pick(lambda r: D.get_i32(buf, r[3][0][1]) >= 4)
        l, c_off, i_off, capl = r[3][0]
        D.put_i32(buf, c_off, int(rng.integers(0, D.get_i32(buf, c_off))))
        done["short"] += 1
        # reorder: swap two counted slots (list order decides admission order, :485)
        r = pick(lambda r: D.get_i32(buf, r[3][0][1]) >= 4)
        l, c_off, i_off, capl = r[3][0]
        cnt = D.get_i32(buf, c_off)
        a, b = (int(x) for x in rng.choice(cnt, 2, replace=False))
        va, vb = D.get_i32(buf, i_off + 4 * a), D.get_i32(buf, i_off + 4 * b)
        D.put_i32(buf, i_off + 4 * a, vb)
        D.put_i32(buf, i_off + 4 * b, va)
        done["swap"] += 1
        # upper-layer edge to a node whose level is below that layer (documented: such a node has no list there)
        r = pick(lambda r: r[1] >= 2 and D.get_i32(buf, r[3][2][1]) >= 1)
        if r is not None:
            low = [i for i in ids.tolist() if level_of[i] < 2]
            l, c_off, i_off, capl = r[3][2]
            D.put_i32(buf, i_off, low[int(rng.integers(0, len(low)))])
            done["low"] += 1
    return bytes(buf), done


@pytest.mark.parametrize("dim,m,efc,n,cap", [(32, 8, 60, 2600, 3500), (384, 16, 100, 1500, 2000)])
def test_crafted_lists_match_oracle(gpu, dim, m, efc, n, cap):
    from minivectordb_amd import HipBridge
    dump, ids = _sparse_index(n, cap, dim, m, efc, 13)
    crafted, done = _craft(dump, ids, cap, 5)
    assert min(done[k] for k in ("absent", "dup", "self", "hole", "short", "swap")) >= 20 and done["low"] >= 5, done
    ix = O.OracleIndex(dim, m, efc, 8, seed=1)
    assert ix.load_bytes(crafted), "load_index accepts these dumps (only out-of-range ids are rewritten)"
    b = HipBridge()
    b.init(dict(dim=dim, m=m, ef=efc, capacity=8, seed=1))
    b.loadBytes(crafted)
    assert b.saveBytes() == ix.save_bytes() == crafted       # nothing is "repaired" on the host side
    b.setResultsCap(4096)
    ix.set_results_cap(4096)
    q8 = O.quantize_rows_i8(O.normalize_rows_f32(np.random.default_rng(3).standard_normal((64, dim), dtype=np.float32)))
    for k, ef in [(10, 16), (200, 200), (1200, 140)]:
        _same(b, ix, q8, k, ef)


def test_entry_point_with_emptied_lists(gpu):
    """Every list of the entry point emptied (count 0) and a node whose layer-0 list is only self/absent ids:
    searches return what is reachable, as the oracle does (ragged rows, :431-439, :459)."""
    from minivectordb_amd import HipBridge
    dim, m, efc, n, cap = 32, 8, 60, 400, 600
    dump, ids = _sparse_index(n, cap, dim, m, efc, 21)
    buf = bytearray(dump)
    recs = D.walk(dump)
    entry = D.header(dump)[9]
    for r in recs:
        if r[0] == entry:
            for l, c_off, i_off, capl in r[3]:
                D.put_i32(buf, c_off, 0)
    ix = O.OracleIndex(dim, m, efc, 8, seed=1)
    assert ix.load_bytes(bytes(buf))
    b = HipBridge()
    b.init(dict(dim=dim, m=m, ef=efc, capacity=8, seed=1))
    b.loadBytes(bytes(buf))
    q8 = O.quantize_rows_i8(O.normalize_rows_f32(np.random.default_rng(4).standard_normal((16, dim), dtype=np.float32)))
    outs = _same(b, ix, q8, 50, 50)
    assert all((o[:, 0] == entry).all() for o in outs.values())   # only the entry point is reachable
