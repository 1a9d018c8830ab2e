"""Exact int8 scan kernel (brute-force int8 top-k) on the GPU box.

    python tools/scan_bench.py [rows] [dim] [nq] [k]

The graph is irrelevant to the scan, so the index is built with the cheapest legal parameters.
Prints kernel time (engine stage timer), int8 TOP/s against the dense i8 MFMA peak, the HBM
floor, and checks a sample of queries against the oracle's scan.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from minivectordb_amd import HipBridge  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker only)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    rng = np.random.default_rng(21)
    b8 = rng.integers(-127, 128, (rows, dim), dtype=np.int16).astype(np.int8)
    q8 = rng.integers(-127, 128, (nq, dim), dtype=np.int16).astype(np.int8)
    b = HipBridge()
    b.init({"dim": dim, "m": 2, "ef": 4, "capacity": rows, "seed": 3})
    t0 = time.time()
    b.buildParallel(b8, 16)
    build_s = time.time() - t0
    L, h = b._L, b.handle
    L.mvdb_enable_timing(h, 1)

    def stats(stage):
        n, tot = C.c_longlong(0), C.c_double(0.0)
        L.mvdb_stage_stats(h, stage, C.byref(n), C.byref(tot), 1)
        return n.value, tot.value

    ix = O.OracleIndex(dim, 2, 4, 8, seed=1)
    assert ix.load_bytes(b.saveBytes())
    out = {"rows": rows, "dim": dim, "nq": nq, "k": k, "build_s": round(build_s, 1)}
    for metric in ("l2", "dot"):
        for _ in range(2):
            ids, val, cnt = b.scanMany(q8, k, metric)
        stats(3)
        for _ in range(5):
            ids, val, cnt = b.scanMany(q8, k, metric)
        n3, t3 = stats(3)
        ms = t3 / max(n3, 1)
        ops = 2.0 * nq * rows * dim
        out[metric] = {"ms": ms, "TOPs": ops / (ms * 1e-3) / 1e12, "qps": nq / (ms * 1e-3),
                       "hbm_floor_ms": rows * dim / 8e12 * 1e3}
        ns = min(nq, 32)
        t0 = time.perf_counter()
        o_ids, o_val, o_cnt = ix.scan_many(q8[:ns], k, metric, threads=16)
        out[metric]["cpu_oracle_qps_16thr"] = ns / (time.perf_counter() - t0)
        out[metric]["parity_sample_ok"] = bool(np.array_equal(o_ids, ids[:ns]) and np.array_equal(o_val, val[:ns])
                                               and np.array_equal(o_cnt, cnt[:ns]))
    print(json.dumps(out))
    b.close()
    return 0 if all(out[m]["parity_sample_ok"] for m in ("l2", "dot")) else 1


if __name__ == "__main__":
    sys.exit(main())
