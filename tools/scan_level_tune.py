"""Experiments only (MVDB_SCAN_TUNE variant): first-level size and ratio of the level plan with the wide kernel on.

    python tools/build_variants.py scantune:MVDB_SCAN_TUNE
    MVDB_HIP_LIB=.../libmvdb_hip_scantune.so python tools/scan_level_tune.py [rows] [dim] [nq] [k]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scan_wide_ab import make, run  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
b, rng = make(rows, dim, 21)
q8 = rng.integers(-127, 128, (nq, dim), dtype=np.int16).astype(np.int8)
ref = None
for level0, ratio in ((1024, 8), (128, 8), (256, 8), (512, 8), (2048, 8), (256, 4), (512, 4), (1024, 4), (1024, 16), (512, 16), (256, 16), (1024, 8)):
    os.environ["MVDB_SCAN_LEVEL0"] = str(level0)
    os.environ["MVDB_SCAN_RATIO"] = str(ratio)
    out, ms = run(b, q8, k, "l2", 100000, 5)
    if ref is None:
        ref = out
    same = all(np.array_equal(x, y) for x, y in zip(out, ref))
    print(f"level0 {level0:5d} ratio {ratio:3d}: {ms:.3f} ms  same_as_first={same}", flush=True)
