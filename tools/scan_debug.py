"""Diagnostic: where the scan kernel's wave time goes (MVDB_SCAN_DEBUG variant library, never shipped).

    python tools/build_variants.py scandbg:MVDB_SCAN_DEBUG,MVDB_SCAN_TUNE && MVDB_HIP_LIB=.../libmvdb_hip_scandbg.so python tools/scan_debug.py

(scan_kernel on every level: MVDB_SCAN_WIDE=0, a knob of MVDB_SCAN_TUNE builds; tools/scan_wide_debug.py is the wide kernel's.)
"""
import ctypes as C
import os
import sys

import numpy as np

os.environ["MVDB_SCAN_WIDE"] = "0"

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
for _ in range(2):
    b.scanMany(q8, k, "l2")
out = np.zeros(16, np.uint64)
L = b._L
L.mvdb_debug_scan_counters.argtypes = [C.c_void_p, C.c_void_p]
assert L.mvdb_debug_scan_counters(b.handle, out.ctypes.data_as(C.c_void_p))
tiles, entries, iters, rows_flushed, slow_cyc, store_barrier_cyc, total_cyc, compact_cyc = (int(x) for x in out[:8])
print(f"compaction + refresh: {100 * compact_cyc / total_cyc:.1f}% of wave cycles, {compact_cyc / max(rows_flushed, 1):.0f} cycles per compacted row; "
      f"rest of slow path {(slow_cyc - compact_cyc) / max(entries, 1):.0f} cycles per entry")
print(f"wave-tiles {tiles}  slow-path entries {entries} ({entries / max(tiles, 1):.3f}/tile)  loop iterations {iters}")
print(f"row compactions {rows_flushed} ({rows_flushed / max(tiles, 1):.3f}/tile)")
print(f"wave cycles: total {total_cyc:.3g}  slow path {100 * slow_cyc / total_cyc:.1f}%  "
      f"stage store + barrier {100 * store_barrier_cyc / total_cyc:.1f}%  per tile {total_cyc / max(tiles, 1):.0f}")
if os.environ.get("MVDB_SCAN_ABLATE_STAMPS"):
    print(f"end of tile per wave-tile: wait for loads {iters / tiles:.0f}  LDS stores {rows_flushed / tiles:.0f}  barrier {compact_cyc / tiles:.0f} cycles")
print("end of tile, cycles per wave-pass: drain LDS reads %.0f  own DMA landed %.0f  barrier %.0f  prefetch + DMA issue %.0f" % tuple(int(x) / max(tiles, 1) for x in out[8:12]))
