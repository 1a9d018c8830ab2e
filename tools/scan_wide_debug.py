"""Diagnostic: where a wave of the wide scan kernel spends its time (MVDB_SCAN_DEBUG + MVDB_SCAN_TUNE variant, never shipped).

    python tools/build_variants.py widedbg:MVDB_SCAN_DEBUG,MVDB_SCAN_TUNE
    MVDB_SCAN_WIDE=100000 MVDB_HIP_LIB=.../libmvdb_hip_widedbg.so python tools/scan_wide_debug.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivectordb_amd import HipBridge  # noqa: E402

rows, dim, nq, k = 1_000_000, 512, 4096, 10
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
tiles, slow, comp, hand, wait, barr, total, first = (int(x) for x in out[:8])
print(f"wave-tiles {tiles}; per wave-tile (s_memtime ticks): compute {comp / tiles:.0f}  candidate handling {hand / tiles:.0f} "
      f"({slow / tiles:.2f} calls, {hand / max(slow, 1):.0f} per call incl. the empty ones)  first fragments {first / tiles:.0f}  "
      f"wait for own DMA {wait / tiles:.0f}  barrier {barr / tiles:.0f}  total {total / tiles:.0f}")
c8, c9, c10, c11, c12, c13 = (int(x) for x in out[8:14])
print(f"candidate path, ticks per call: finding the tiles {c8 / max(slow, 1):.0f}; per flagged tile ({c11 / max(slow, 1):.2f} per call): select + stage {c9 / max(c11, 1):.0f}, "
      f"append loop {c10 / max(c11, 1):.0f} ({c12 / max(c11, 1):.2f} rounds, {c13 / max(c11, 1):.3f} compaction rounds)")
