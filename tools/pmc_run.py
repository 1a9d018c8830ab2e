"""Workload for the rocprofv3 --pmc passes (HBM traffic of the traversal kernel).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_run.py
  (tools/pmc_passes.sh runs every pass; tools/pmc.py traffic turns them into profiles/rN/pmc_traffic_*.json)

Launches, on one 1M x 384 balanced index: (1) a calibration launch of dist_pairs_kernel whose byte
count is known exactly (it gathers rows with the same 8-lanes-x-16-B pattern as the traversal), then
(2) three searchMany batches.  FETCH_SIZE on gfx950 is access-pattern dependent
(MI355X_MICROARCH.md, HBM section), so traffic is reported after scaling by the calibration ratio."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minivectordb_amd import HipBridge, resolve_preset, derive_search_params

n = int(os.environ.get("PMC_N", "1000000")); dim, nq = 384, 1000
p = resolve_preset("balanced")
b = HipBridge(); b.init(dict(dim=dim, m=p["m"], ef=p["ef_construction"], capacity=n, seed=42, resultsCap=4096))
rng = np.random.default_rng(12345)
base8 = np.empty((n, dim), np.int8)
clustered = os.environ.get("PMC_CLUSTERED", "0") == "1"  # bench.py's second dataset instead of i.i.d. rows
if clustered:
    import bench
    rows = bench.make_rows(n, dim, 12345 + 777, True)
    base_n = np.empty((n, dim), np.float32)
for s in range(0, n, 131072):
    qn, q8 = b.prepareQueries(rows[s:s + 131072] if clustered else rng.standard_normal((min(131072, n - s), dim), dtype=np.float32))
    base8[s:s + 131072] = q8
    if clustered:
        base_n[s:s + 131072] = qn
t0 = time.time(); b.buildGpu(base8); print("built in", time.time() - t0, flush=True)
q = (bench.make_queries(1, nq, dim, 54321, base_n, True)[0] if clustered
     else np.random.default_rng(54321).standard_normal((nq, dim), dtype=np.float32))
_, qi8 = b.prepareQueries(q)
per_q = 18000
ids = np.random.default_rng(7).integers(0, n, (nq, per_q)).astype(np.int32)
b.distMany(qi8, ids)                       # calibration: nq*per_q rows of `dim` bytes each
tk, ef, ann_k = derive_search_params(10, p)
b.setEfSearch(ef)
if os.environ.get("MVDB_NO_RELABEL"):
    b.setRelabel(0)
for _ in range(3):
    out = b.searchManyArrays(qi8, ann_k, with_counters=True)
ctr = out[3].astype(np.int64)
m0, m = 2 * p["m"], p["m"]
alg = int((ctr[:, 0] * dim + ctr[:, 1] * (4 + 4 * m0) + ctr[:, 2] * (4 + 4 * m) + dim).sum())
print(json.dumps(dict(calib_rows=nq * per_q, calib_bytes=nq * per_q * dim + nq * per_q * 4 + nq * dim,
                      traverse_algorithmic_bytes=alg, n=n, clustered=clustered,
                      mean_n_dist=float(ctr[:, 0].mean()), mean_n_l0lists=float(ctr[:, 1].mean()))), flush=True)
