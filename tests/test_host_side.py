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
