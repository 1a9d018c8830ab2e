"""Experiments only (MVDB_SCAN_TUNE variant): level 0 through scan_seed_kernel (MVDB_SCAN_SEED=1, the default) against
scan_kernel (0): identical results?  time of the whole sweep.

    MVDB_HIP_LIB=.../libmvdb_hip_scantune.so python tools/scan_seed_ab.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scan_wide_ab import make, stats  # noqa: E402


def run(b, q8, k, metric, seed, reps):
    os.environ["MVDB_SCAN_SEED"] = str(seed)
    for _ in range(2):
        out = b.scanMany(q8, k, metric)
    stats(b)
    for _ in range(reps):
        out = b.scanMany(q8, k, metric)
    n3, t3 = stats(b)
    return out, t3 / max(n3, 1)


ok = True
for (r, d, n, kk, metric) in ((5000, 100, 700, 10, "l2"), (20000, 200, 1030, 32, "dot"), (3000, 320, 513, 1, "l2"), (700, 512, 5, 10, "l2"),
                              (1024, 64, 333, 17, "dot"), (1025, 384, 100, 10, "l2"), (9, 16, 40, 10, "l2"), (40000, 512, 1024, 7, "dot")):
    b, rng = make(r, d, 5)
    q8 = rng.integers(-127, 128, (n, d), dtype=np.int16).astype(np.int8)
    o0, _ = run(b, q8, kk, metric, 0, 1)
    o1, _ = run(b, q8, kk, metric, 1, 1)
    same = all(np.array_equal(x, y) for x, y in zip(o0, o1))
    ok &= same
    print(f"rows {r} dim {d} nq {n} k {kk} {metric}: same={same}", flush=True)
    b.close()
for (rows, dim, nq, k) in ((1_000_000, 512, 4096, 10), (1_000_000, 384, 1000, 10), (200_000, 128, 256, 32)):
    b, rng = make(rows, dim, 21)
    q8 = rng.integers(-127, 128, (nq, dim), dtype=np.int16).astype(np.int8)
    o0, ms0 = run(b, q8, k, "l2", 0, 5)
    o1, ms1 = run(b, q8, k, "l2", 1, 5)
    o0b, ms0b = run(b, q8, k, "l2", 0, 5)
    same = all(np.array_equal(x, y) for x, y in zip(o0, o1))
    ok &= same
    print(f"{rows} x {dim}, {nq} queries, k {k}: scan_kernel level 0 {ms0:.3f} ms, seed kernel {ms1:.3f} ms, scan_kernel again {ms0b:.3f} ms, same={same}", flush=True)
    b.close()
sys.exit(0 if ok else 1)
