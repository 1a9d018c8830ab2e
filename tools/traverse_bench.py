"""Traversal-kernel A/B on the GPU box: one-wave vs two-wave kernel on one synthetic index.

    python tools/traverse_bench.py [rows] [dim] [nq] [ef] [threads]

Builds the index once (buildParallel), runs the batch with 1, 2 and 3 waves per query (setTuning),
requires identical output, prints per-launch kernel times from the engine's stage timers.
"""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from minivectordb_amd import HipBridge  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    ef = int(sys.argv[4]) if len(sys.argv) > 4 else 1200
    threads = int(sys.argv[5]) if len(sys.argv) > 5 else 16
    rng = np.random.default_rng(11)
    base = rng.standard_normal((rows, dim), dtype=np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    b8 = np.clip(np.floor(base * 127.0 + 0.5), -127, 127).astype(np.int8)
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q8 = np.clip(np.floor(q * 127.0 + 0.5), -127, 127).astype(np.int8)

    b = HipBridge()
    b.init({"dim": dim, "m": 16, "ef": 300, "efSearch": ef, "capacity": rows, "seed": 7, "resultsCap": ef})
    stop = threading.Event()

    def beat():
        t0 = time.time()
        while not stop.wait(30):
            print(f"[build] {time.time() - t0:.0f}s", file=sys.stderr, flush=True)
    th = threading.Thread(target=beat, daemon=True)
    th.start()
    t0 = time.time()
    dump = os.environ.get("MVDB_BENCH_DUMP")  # reuse one built index across variant libraries
    if dump and os.path.exists(dump):
        b.load(dump)
    else:
        b.buildGpu(b8)
        if dump:
            b.save(dump)
    stop.set()
    build_s = time.time() - t0
    L, h = b._L, b.handle
    L.mvdb_enable_timing(h, 1)

    def stats(stage):
        n, tot = C.c_longlong(0), C.c_double(0.0)
        L.mvdb_stage_stats(h, stage, C.byref(n), C.byref(tot), 1)
        return n.value, tot.value

    out = {"rows": rows, "dim": dim, "nq": nq, "ef": ef, "build_s": round(build_s, 1)}
    results = {}
    for mode in ("1", "2", "3"):
        b.setTuning(traverse_waves=int(mode))
        for _ in range(2):
            r = b.searchManyArrays(q8, ef, with_counters=True)
        stats(1)
        for _ in range(5):
            r = b.searchManyArrays(q8, ef, with_counters=True)
        n1, t1 = stats(1)
        results[mode] = r
        out[f"waves{mode}_ms"] = t1 / max(n1, 1)
    same = all(np.array_equal(x, y) for x, y in zip(results["1"], results["2"])) and all(
        np.array_equal(x, y) for x, y in zip(results["1"], results["3"]))
    out["identical"] = bool(same)
    out["mean_n_dist"] = float(results["2"][3][:, 0].mean())
    out["mean_l0lists"] = float(results["2"][3][:, 1].mean())
    print(json.dumps(out))
    b.close()
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
