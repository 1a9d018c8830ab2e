"""Bulk insert into a live index: mvdb_append_gpu against one-at-a-time insert (host), 1 M x 384 balanced + 100 k."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivectordb_amd import HipBridge, resolve_preset
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
dim, p = 384, resolve_preset("balanced")
rng = np.random.default_rng(1)
b = HipBridge()
b.init(dict(dim=dim, m=p["m"], ef=p["ef_construction"], capacity=n0 + n1 + 256, seed=42, resultsCap=4096))
b8 = np.empty((n0 + n1 + 256, dim), np.int8)
for s in range(0, b8.shape[0], 131072):
    e = min(b8.shape[0], s + 131072)
    b8[s:e] = b.prepareQueries(rng.standard_normal((e - s, dim), dtype=np.float32))[1]
print("build", b.buildGpu(b8[:n0])["seconds"], flush=True)
q8 = b8[:256]
b.setEfSearch(140)
b.searchManyArrays(q8, 1200)
st = b.appendGpu(b8[n0:n0 + n1])
print(f"appendGpu {n1} into {n0}: {st['seconds']:.3f} s ({n1 / st['seconds']:.0f} inserts/s), passes {st['passes']:.0f}", flush=True)
t0 = time.time(); b.searchManyArrays(q8, 1200); print(f"first search after (full re-upload): {time.time() - t0:.3f} s")
ids = np.arange(n0 + n1, n0 + n1 + 256, dtype=np.int32)
t0 = time.time(); b.insertManyPacked(ids, b8[n0 + n1:], dim); dt = time.time() - t0
print(f"host insert of 256: {dt:.3f} s ({256 / dt:.0f} inserts/s)")
t0 = time.time(); b.searchManyArrays(q8, 1200); print(f"first search after (patch): {time.time() - t0:.3f} s", b.uploadCounters())
