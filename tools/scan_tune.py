"""Experiments only: level plan of the exact scan (MVDB_SCAN_TUNE variant library reads the two env vars per call).

    python tools/build_variants.py scantune:MVDB_SCAN_TUNE
    MVDB_HIP_LIB=.../libmvdb_hip_scantune.so python tools/scan_tune.py [rows] [dim] [nq] [k]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivectordb_amd import HipBridge  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
rng = np.random.default_rng(21)
b8 = rng.integers(-127, 128, (rows, dim), dtype=np.int16).astype(np.int8)
q8 = rng.integers(-127, 128, (nq, dim), dtype=np.int16).astype(np.int8)
b = HipBridge()
b.init({"dim": dim, "m": 2, "ef": 4, "capacity": rows, "seed": 3})
b.buildParallel(b8, 16)
L, h = b._L, b.handle
L.mvdb_enable_timing(h, 1)


def stats(stage):
    n, tot = C.c_longlong(0), C.c_double(0.0)
    L.mvdb_stage_stats(h, stage, C.byref(n), C.byref(tot), 1)
    return n.value, tot.value


ref = None
for level0, ratio in ((4096, 4), (2048, 4), (1024, 4), (512, 4), (4096, 8), (2048, 8), (1024, 8), (512, 8), (1024, 16), (2048, 16), (1024, 6), (2048, 6), (1024, 32)):
    os.environ["MVDB_SCAN_LEVEL0"] = str(level0)
    os.environ["MVDB_SCAN_RATIO"] = str(ratio)
    for _ in range(2):
        ids, val, cnt = b.scanMany(q8, k, "l2")
    stats(3)
    for _ in range(5):
        ids, val, cnt = b.scanMany(q8, k, "l2")
    n3, t3 = stats(3)
    if ref is None:
        ref = (ids.copy(), val.copy())
    same = bool(np.array_equal(ids, ref[0]) and np.array_equal(val, ref[1]))
    ms = t3 / max(n3, 1)
    print(f"level0 {level0:6d} ratio {ratio:8d}: {ms:.3f} ms  {2.0 * nq * rows * dim / (ms * 1e-3) / 1e12:.0f} TOP/s  same_as_first={same}", flush=True)
