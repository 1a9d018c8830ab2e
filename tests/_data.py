"""Shared synthetic data for the tests (BASELINE.md section 2 recipe, scaled down)."""
import functools

import numpy as np

from oracle import oracle as O


def gaussian_unit(n, dim, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, dim), dtype=np.float32)
    return O.normalize_rows_f32(x)


@functools.lru_cache(maxsize=8)
def built_index(n, dim, preset, seed=42, capacity=None, parallel=0):
    """Oracle-built index (reference insert order, seed_rng(seed)) -> (dump bytes, f32 base, i8 base)."""
    p = O.PRESETS[preset]
    base = gaussian_unit(n, dim, 12345)
    b8 = O.quantize_rows_i8(base)
    ix = O.OracleIndex(dim, p["m"], p["ef_construction"], capacity or n, seed=seed)
    if parallel:
        ix.build_parallel(b8, parallel)
    else:
        ix.insert_many(np.arange(n, dtype=np.int32), b8)
    dump = ix.save_bytes()
    ix.close()
    return dump, base, b8


def oracle_from_dump(dump, dim, preset, capacity=8):
    p = O.PRESETS[preset]
    ix = O.OracleIndex(dim, p["m"], p["ef_construction"], capacity, seed=1)
    assert ix.load_bytes(dump)
    return ix


def bridge_from_dump(dump, dim, preset, capacity=8):
    from minivectordb_amd import HipBridge
    p = O.PRESETS[preset]
    b = HipBridge()
    b.init(dict(dim=dim, m=p["m"], ef=p["ef_construction"], capacity=capacity, seed=1))
    b.loadBytes(dump)
    return b
