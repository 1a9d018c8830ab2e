"""How the traversal launch time depends on the number of queries resident on the chip.

    python tools/occupancy_probe.py [rows] [dim] [ef]

One index, the default (lean) traversal kernel, batches of 64 ... 5120 queries through mvdb_search_many
(sorted-list path, stage timer 1 = the traversal launch alone).  256 CUs x 8 workgroups = 2048 resident
queries: below that a launch lasts as long as its slowest query; the per-query figure
ms / ceil(nq / 2048) tells how much a query slows down when its CU fills up.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from minivectordb_amd import HipBridge  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
    ef = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
    rng = np.random.default_rng(11)
    base = rng.standard_normal((rows, dim), dtype=np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    b8 = np.clip(np.floor(base * 127.0 + 0.5), -127, 127).astype(np.int8)
    del base
    b = HipBridge()
    b.init({"dim": dim, "m": 16, "ef": 300, "efSearch": ef, "capacity": rows, "seed": 7, "resultsCap": ef})
    b.buildGpu(b8)
    L, h = b._L, b.handle
    L.mvdb_enable_timing(h, 1)

    def stats(stage):
        n, tot = C.c_longlong(0), C.c_double(0.0)
        L.mvdb_stage_stats(h, stage, C.byref(n), C.byref(tot), 1)
        return n.value, tot.value

    out = []
    for nq in (64, 128, 256, 512, 768, 1024, 1280, 2560, 5120):
        q = rng.standard_normal((nq, dim), dtype=np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        q8 = np.clip(np.floor(q * 127.0 + 0.5), -127, 127).astype(np.int8)
        for _ in range(2):
            r = b.searchManyArrays(q8, ef, with_counters=True)
        stats(1)
        for _ in range(5):
            r = b.searchManyArrays(q8, ef, with_counters=True)
        n1, t1 = stats(1)
        ms = t1 / max(n1, 1)
        nd = float(r[3][:, 0].mean())
        row = {"nq": nq, "launch_ms": round(ms, 4), "queries_per_s": round(nq / ms * 1e3), "mean_n_dist": round(nd, 1),
               "gbs_rows_only": round(nq * nd * dim / ms / 1e6, 1)}
        out.append(row)
        print(json.dumps(row), flush=True)
    b.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
