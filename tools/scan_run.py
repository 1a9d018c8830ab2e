"""Runs a few exact scans (1M x 512, 4096 queries by default) and nothing else: the program to put after
`rocprofv3 --pmc ... --` / `--kernel-trace` when looking at scan_kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivectordb_amd import HipBridge  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
rng = np.random.default_rng(21)
b8 = rng.integers(-127, 128, (rows, dim), dtype=np.int16).astype(np.int8)
q8 = rng.integers(-127, 128, (nq, dim), dtype=np.int16).astype(np.int8)
b = HipBridge()
b.init({"dim": dim, "m": 2, "ef": 4, "capacity": rows, "seed": 3})
b.buildParallel(b8, 16)
for _ in range(reps):
    b.scanMany(q8, k, "l2")
b.close()
print("done")
