"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol
include/mvdb_hip.h declares; dump.bin V3 parse/serialise and the host write path are byte-identical
to the oracle; bridge setters follow the WasmBridge rules.  No compute call needs a GPU here."""
import os
import re

import numpy as np
import pytest

from minivectordb_amd import HipBridge, MvdbError, _native, derive_search_params, resolve_preset
from oracle import oracle as O
from tests._data import built_index, gaussian_unit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mvdb_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mvdb_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    L = _native.lib()
    bound = {name for name, _, _ in _native.SYMBOLS}
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in mvdb_hip.h but not exported"
    assert declared == bound, f"python binding drift: {declared ^ bound}"


def test_create_rejects_what_update_config_traps():
    for bad in (dict(dim=0), dict(dim=6), dict(m=0), dict(ef=0)):     # assembly/config.ts:95-100
        cfg = dict(dim=16, m=4, ef=20, capacity=10, seed=1)
        cfg.update(bad)
        with pytest.raises(MvdbError):
            HipBridge().init(cfg)


def test_dump_round_trip_and_insert_parity_with_oracle():
    dump, base, b8 = built_index(1500, 64, "balanced")
    b = HipBridge()
    b.init(dict(dim=64, m=16, ef=300, capacity=9, seed=1))
    b.loadBytes(dump)
    assert b.saveBytes() == dump                       # parse -> serialise is byte-identical
    assert b.getMaxElements() == 1500 and b.elementCount() == 1500 and b.hasNode(1499) and not b.hasNode(1500)
    # same insert sequence (same seed) through the product's host graph == the oracle's dump
    b2 = HipBridge()
    b2.init(dict(dim=64, m=16, ef=300, capacity=1500, seed=42))
    b2.insertManyPacked(np.arange(1500, dtype=np.int32), b8.reshape(-1), 64)
    assert b2.saveBytes() == dump
    # upserts (update_and_reconnect) and sparse ids keep matching
    rng = np.random.default_rng(4)
    ix = O.OracleIndex(64, 16, 300, 4000, seed=9)
    b3 = HipBridge()
    b3.init(dict(dim=64, m=16, ef=300, capacity=4000, seed=9))
    ids = rng.permutation(4000)[:800].astype(np.int32)
    vec = O.quantize_rows_i8(gaussian_unit(800, 64, 6))
    ix.insert_many(ids, vec); b3.insertManyPacked(ids, vec.reshape(-1), 64)
    upd = ids[rng.integers(0, 800, 150)]
    vec2 = O.quantize_rows_i8(gaussian_unit(150, 64, 7))
    ix.insert_many(upd, vec2); b3.insertManyPacked(upd, vec2.reshape(-1), 64)
    b3.updateVector(3999 if 3999 not in set(ids.tolist()) else int(ids[0]), vec2[0])
    assert b3.saveBytes() == ix.save_bytes()


def test_load_failure_message_and_fresh_index():
    dump, _, _ = built_index(1500, 64, "balanced")
    b = HipBridge()
    b.init(dict(dim=128, m=16, ef=300, capacity=11, seed=1))
    with pytest.raises(MvdbError) as e:
        b.loadBytes(dump)
    msg = str(e.value)                                  # tests/config_mismatch_test.ts greps these words
    assert "mismatch" in msg and "config" in msg
    assert b.getMaxElements() == 11 and b.elementCount() == 0     # reinitFresh, wasm-bridge.ts:463-466
    b.load("/nonexistent/dump.bin")                     # missing file is a silent no-op (:379)
    for cut in (10, 47, len(dump) - 3):
        b2 = HipBridge()
        b2.init(dict(dim=64, m=16, ef=300, capacity=5, seed=1))
        with pytest.raises(MvdbError):
            b2.loadBytes(dump[:cut])
        assert b2.elementCount() == 0
    with pytest.raises(MvdbError):
        b.insertManyPacked(np.arange(3, dtype=np.int32), np.zeros(5, np.int8), 128)   # :274-278


def test_bridge_setters_follow_wasm_rules():
    b = HipBridge()
    b.init(dict(dim=16, m=4, ef=20, capacity=10, seed=1))
    assert b.getMaxEf() == 4096 and b.getResultsCap() == 1000
    b.setResultsCap(0); assert b.getResultsCap() == 1
    b.setResultsCap(10**6); assert b.getResultsCap() == 4096
    b.setEfSearch(77); b.setEfSearch(0); b.setEfSearch(-3)
    assert b._L.mvdb_get_ef_search(b.handle) == 77
    b2 = HipBridge()
    b2.init(dict(dim=16, m=4, ef=20, capacity=10, seed=1, resultsCap=2048, efSearch=80))
    assert b2.getResultsCap() == 2048 and b2._L.mvdb_get_ef_search(b2.handle) == 80
    b2.reinitIndex(20)
    assert b2.getMaxElements() == 20 and b2.getResultsCap() == 2048


def test_param_derivation_matches_oracle():
    for preset in ("fast", "balanced", "accurate", "unknown-mode"):
        for topk in (None, 0, 1, 3, 10, 17, 100):
            p = resolve_preset(preset)
            exp = O.derive_search_params(10 if topk is None else topk,
                                         preset if preset in O.PRESETS else "balanced")
            assert derive_search_params(topk, p) == exp
    import ctypes as C
    L = _native.lib()
    tk, ef, ak = C.c_int(), C.c_int(), C.c_int()
    L.mvdb_derive_params(10, 140, 120, 4096, C.byref(tk), C.byref(ef), C.byref(ak))
    assert (tk.value, ef.value, ak.value) == (10, 140, 1200)
    for l2 in (0.0, 1.0, 3.9, 7.0):
        for mode, name in ((0, "l2"), (1, "cosine"), (2, "sim")):
            assert L.mvdb_to_score(l2, mode) == O.to_score(l2, name)


def test_compute_without_gpu_fails_loudly():
    """No CPU fallback: on a box without a HIP device a search must raise, not return."""
    if _native.lib().mvdb_device_available():
        pytest.skip("a GPU is present")
    dump, _, b8 = built_index(1500, 64, "balanced")
    b = HipBridge()
    b.init(dict(dim=64, m=16, ef=300, capacity=5, seed=1))
    b.loadBytes(dump)
    with pytest.raises(MvdbError):
        b.searchManyArrays(b8[:2], 10)


def test_product_loader_accepts_and_rejects_exactly_like_load_index():
    """The corruption matrix of tests/test_oracle_cross.py (and more) against the PRODUCT's loader
    (HostGraph::load through HipBridge.loadBytes) beside the oracle's load_index (hnsw.ts:1129-1249):
    same accept/reject verdict for every mutation, the same fresh empty index after a reject
    (wasm-bridge.ts:463-466), and byte-identical re-serialisation after an accept (neighbour ids out of
    range are rewritten to -1 by both, :1222-1229)."""
    import struct
    from tests import _dump as D
    dump, base, b8 = built_index(300, 16, "fast")
    p = resolve_preset("fast")
    recs = D.walk(dump)
    r0, r5 = recs[0], recs[5]

    def both(buf, tag):
        buf = bytes(buf)
        ix = O.OracleIndex(16, p["m"], p["ef_construction"], 7, seed=1)
        a = ix.load_bytes(buf)
        b = HipBridge()
        b.init(dict(dim=16, m=p["m"], ef=p["ef_construction"], capacity=7, seed=1))
        try:
            b.loadBytes(buf)
            ok = True
        except MvdbError:
            ok = False
        assert ok == a, f"{tag}: product {'accepts' if ok else 'rejects'}, load_index {'accepts' if a else 'rejects'}"
        if a:
            assert b.saveBytes() == ix.save_bytes(), f"{tag}: re-serialised dumps differ"
            assert b.getResultsCap() == ix.get_results_cap(), tag
        else:
            assert b.getMaxElements() == 7 and b.elementCount() == 0, tag
        ix.close(); b.close()
        return a

    def hdr(field, value):
        bad = bytearray(dump)
        struct.pack_into("<i", bad, 4 * field, value)
        return bad

    assert both(dump, "clean")
    assert both(dump + b"\x00" * 9, "trailing bytes")                   # nothing reads past the last record
    for cut in (0, 10, 40, 47, 48, 49, 56, 60, r0[3][0][1] + 2, len(dump) // 2, len(dump) - 1):
        assert not both(dump[:cut], f"cut at {cut}")
    bad = bytearray(dump); bad[0] ^= 1; assert not both(bad, "magic")
    for field, values, name in [(1, (2, 4, 0), "version"), (2, (32, 0), "dim"), (3, (p["m"] + 1,), "m"),
                                (4, (2 * p["m"] + 1,), "m_max0"), (5, (p["ef_construction"] + 1,), "ef_construction"),
                                (6, (3, 5), "max_layers"), (7, (0, -5), "max_elements"), (8, (-1, 10**6), "present"),
                                (9, (-2, 10**6, 300), "entry"), (10, (-2, 4, 99), "max_level")]:
        for v in values:
            assert not both(hdr(field, v), f"{name}={v}")
    assert not both(hdr(9, -1), "entry -1 with present > 0")            # :1173
    both(hdr(7, 299), "max_elements one short of the highest id")        # record id out of range -> reject
    both(hdr(7, 5000), "max_elements larger than the data")              # accepted: a larger, sparse index
    both(hdr(8, 299), "present one short")                               # accepted: the last record is never read
    both(hdr(8, 301), "present one too many")                            # runs off the end -> reject
    both(hdr(10, 0), "max_level lower than the nodes' levels")           # header is trusted (:1170 range check only)
    for v in (0, -7, 1, 4096, 10**6):
        assert both(hdr(11, v), f"results_cap={v}")                      # :1176-1179 (<= 0 keeps the current cap)
    # record level
    rec_off = r5[2] - 8
    for v, name in [(-1, "id -1"), (300, "id == max_elements"), (10**6, "id huge")]:
        bad = bytearray(dump); D.put_i32(bad, rec_off, v); assert not both(bad, name)
    bad = bytearray(dump); D.put_i32(bad, rec_off, r0[0]); both(bad, "repeated id")
    for v in (-1, 4, 9):
        bad = bytearray(dump); D.put_i32(bad, rec_off + 4, v); assert not both(bad, f"level={v}")
    for l, c_off, i_off, cap in r5[3]:
        for v in (-1, cap + 1, 999):
            bad = bytearray(dump); D.put_i32(bad, c_off, v); assert not both(bad, f"layer {l} count={v}")
        bad = bytearray(dump); D.put_i32(bad, c_off, 0); assert both(bad, f"layer {l} count=0")
        for v in (-2, -10**6, 300, 10**9):                               # sanitised to -1 by both
            bad = bytearray(dump); D.put_i32(bad, i_off, v); assert both(bad, f"layer {l} neighbour={v}")
        bad = bytearray(dump); D.put_i32(bad, i_off, r5[0]); assert both(bad, f"layer {l} self edge")
    # the entry point's own record missing (present shrunk so that it is never read) -> entry must exist (:1241)
    entry = D.header(dump)[9]
    pos = [i for i, r in enumerate(recs) if r[0] == entry][0]
    bad = hdr(8, pos)
    assert not both(bad[:recs[pos][2] - 8] if pos else bad, "entry record absent")


def test_python_bridge_rejects_wrong_shapes():
    """row lengths and the allow bitmap are checked before any pointer reaches the C ABI."""
    dump, base, b8 = built_index(300, 16, "fast")
    p = resolve_preset("fast")
    b = HipBridge()
    b.init(dict(dim=16, m=p["m"], ef=p["ef_construction"], capacity=7, seed=1))
    b.loadBytes(dump)
    for call in (lambda: b.searchManyArrays(np.zeros((3, 15), np.int8), 5),
                 lambda: b.search(np.zeros(17, np.int8), 5),
                 lambda: b.queryMany(np.zeros((2, 12), np.float32), 5, 50, 50),
                 lambda: b.queryMany(np.zeros((2, 16), np.float32), 5, 50, 50, allow_bitmap=np.zeros(3, np.uint32)),
                 lambda: b.rerankMany(np.zeros((2, 16), np.float32), np.zeros((3, 8), np.int32), np.zeros(2, np.int32), 8, 5),
                 lambda: b.prepareQueries(np.zeros(33, np.float32)),
                 lambda: b.attachF32(np.zeros((4, 8), np.float32)),
                 lambda: b.distMany(np.zeros((2, 16), np.int8), np.zeros((3, 4), np.int32)),
                 lambda: b.setTuning(traverse_waves=7)):
        with pytest.raises(MvdbError):
            call()


def test_build_records_kernel_resources_and_the_lean_kernel_keeps_its_residency():
    """minivectordb_amd/build.py writes every kernel's registers / scratch / occupancy next to the library and refuses a
    build whose lean traversal lost its second wave per SIMD (a regression no functional test sees: one live value more
    took it from 206 to 257 VGPRs and the bench from 403 k to 325 k queries/s).  The record must be there and the guard must fire on a doctored record."""
    from minivectordb_amd import build as B
    assert os.path.exists(B.RESOURCES), "lib/kernel_resources.txt missing: build with python -m minivectordb_amd.build"
    rows = {l.split()[0]: l.split()[1:] for l in open(B.RESOURCES)}
    lean = "_ZN4mvdb20traverse_lean_kernelILi3ELb1EEEvNS_12TraverseArgsE"      # 384-byte rows, the benchmarked shape
    assert lean in rows and int(rows[lean][3]) >= 2 and int(rows[lean][2]) == 0, rows.get(lean)
    B.check_resources()
    good = B.RESOURCES
    try:
        bad = good + ".doctored"
        with open(bad, "w") as f:
            f.write(open(good).read().replace(lean + " " + " ".join(rows[lean]), lean + " 257 0 0 1"))
        B.RESOURCES = bad
        with pytest.raises(RuntimeError):
            B.check_resources()
    finally:
        B.RESOURCES = good
        if os.path.exists(good + ".doctored"):
            os.remove(good + ".doctored")
