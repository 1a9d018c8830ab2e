"""Diagnostic: seconds of a full upload of the device mirror with and without the traversal's relabelled copy.
    python tools/upload_time.py [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minivectordb_amd import HipBridge, resolve_preset
import bench

if __name__ == "__main__":
    n, dim = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000000), 384
    p = resolve_preset("balanced")
    base = bench.make_rows(n, dim, 12345, False)
    b = HipBridge(); b.init(dict(dim=dim, m=p["m"], ef=p["ef_construction"], capacity=n, seed=42, resultsCap=4096))
    qn, q8 = b.prepareQueries(base)
    b.buildGpu(q8)
    b.setEfSearch(64)
    q = q8[:8]
    for on in (2, 0, 1, 0):   # always / never / adaptive / never
        b.setRelabel(on)
        t0 = time.perf_counter(); b.searchManyArrays(q, 10); t1 = time.perf_counter()
        b.searchManyArrays(q, 10); t2 = time.perf_counter()
        print(f"n={n} relabel={on}: first search (full upload) {t1 - t0:.2f} s, second {1e3 * (t2 - t1):.2f} ms, relabelled={b.relabelled()}", flush=True)
