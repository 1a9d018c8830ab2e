"""Experiments only: the wide scan kernel (512 queries per workgroup) against the standard one, same process, same data.

    python tools/build_variants.py scantune:MVDB_SCAN_TUNE
    MVDB_HIP_LIB=.../libmvdb_hip_scantune.so python tools/scan_wide_ab.py [rows] [dim] [nq] [k]

Prints the stage-3 (scan) time of both and whether ids / values / counts are identical; then a few ragged shapes.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivectordb_amd import HipBridge  # noqa: E402


def make(rows, dim, seed):
    rng = np.random.default_rng(seed)
    b8 = rng.integers(-127, 128, (rows, dim), dtype=np.int16).astype(np.int8)
    b = HipBridge()
    b.init({"dim": dim, "m": 2, "ef": 4, "capacity": rows, "seed": 3})
    b.buildParallel(b8, 16)
    b._L.mvdb_enable_timing(b.handle, 1)
    return b, rng


def stats(b, stage=3):
    n, tot = C.c_longlong(0), C.c_double(0.0)
    b._L.mvdb_stage_stats(b.handle, stage, C.byref(n), C.byref(tot), 1)
    return n.value, tot.value


def run(b, q8, k, metric, wide, reps):
    os.environ["MVDB_SCAN_WIDE"] = str(wide)
    for _ in range(2):
        out = b.scanMany(q8, k, metric)
    stats(b)
    for _ in range(reps):
        out = b.scanMany(q8, k, metric)
    n3, t3 = stats(b)
    return out, t3 / max(n3, 1)


if __name__ == "__main__":
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    ok = True
    shapes = ((5000, 100, 700, 10, "l2"), (20000, 200, 1030, 12, "ip"), (3000, 320, 513, 1, "l2"), (3000, 320, 513, 10, "l2"),
              (3000, 512, 513, 1, "l2"), (3000, 320, 512, 1, "l2"), (3000, 320, 512, 2, "ip"), (777, 512, 5, 10, "l2"), (40000, 512, 1024, 7, "ip"))
    if os.environ.get("WIDE_ONLY_BIG"):
        shapes = ()
    for (r, d, n, kk, metric) in shapes:
        b, rng = make(r, d, 5)
        q8 = rng.integers(-127, 128, (n, d), dtype=np.int16).astype(np.int8)
        o0, _ = run(b, q8, kk, metric, 0, 1)
        o1, _ = run(b, q8, kk, metric, 1, 1)  # 1: every level through the wide kernel
        same = all(np.array_equal(x, y) for x, y in zip(o0, o1))
        ok &= same
        bad = np.nonzero((o0[0] != o1[0]).any(axis=1) | (o0[1] != o1[1]).any(axis=1) | (o0[2] != o1[2]))[0]
        print(f"rows {r} dim {d} nq {n} k {kk} {metric}: same={same} bad_queries={bad[:8].tolist()} of {len(bad)}", flush=True)
        if len(bad):
            q = int(bad[0])
            print("   std ", o0[0][q].tolist(), o0[1][q].tolist(), int(o0[2][q]), "\n   wide", o1[0][q].tolist(), o1[1][q].tolist(), int(o1[2][q]), flush=True)
        b.close()
    b, rng = make(rows, dim, 21)
    q8 = rng.integers(-127, 128, (nq, dim), dtype=np.int16).astype(np.int8)
    o0, ms0 = run(b, q8, k, "l2", 0, 5)
    o1, ms1 = run(b, q8, k, "l2", int(os.environ.get("WIDE_MIN_LEVEL", "100000")), 5)
    o0b, ms0b = run(b, q8, k, "l2", 0, 5)
    same = all(np.array_equal(x, y) for x, y in zip(o0, o1))
    ok &= same
    top = lambda ms: 2.0 * nq * rows * dim / (ms * 1e-3) / 1e12  # noqa: E731
    print(f"{rows} x {dim}, {nq} queries, k {k}: standard {ms0:.3f} ms ({top(ms0):.0f} TOP/s), wide {ms1:.3f} ms ({top(ms1):.0f} TOP/s), "
          f"standard again {ms0b:.3f} ms, same={same}", flush=True)
    sys.exit(0 if ok else 1)
