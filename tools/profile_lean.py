"""Diagnostic: per-phase cycle sums of traverse_lean_kernel (s_memtime stamps, MVDB_TRAVERSE_PROFILE build of the
library: tools/profile_traverse.py --build).  usage: profile_lean.py [n] [nq] [--clustered]
(nq = queries in the launch: 1000 leaves every second place of the chip empty, 2048 fills it)"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minivectordb_amd import _native

if __name__ == "__main__":
    _native.LIB_PATH = os.path.join(ROOT, "minivectordb_amd", "lib", "libmvdb_hip_prof.so")
    from minivectordb_amd import HipBridge, resolve_preset, derive_search_params
    import bench
    clustered = "--clustered" in sys.argv
    if clustered:
        sys.argv.remove("--clustered")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    nq, dim = (int(sys.argv[2]) if len(sys.argv) > 2 else 1000), 384
    p = resolve_preset("balanced")
    base = bench.make_rows(n, dim, 12345 + (777 if clustered else 0), clustered)
    b = HipBridge(); b.init(dict(dim=dim, m=p["m"], ef=p["ef_construction"], capacity=n, seed=42, resultsCap=4096))
    qn, q8 = b.prepareQueries(base)
    b.buildGpu(q8)
    b.setTuning(traverse_waves=0)
    if os.environ.get("MVDB_NO_RELABEL"):
        b.setRelabel(0)
    q = bench.make_queries(1, nq, dim, 54321, qn, clustered)[0]
    _, qi8 = b.prepareQueries(q)
    tk, ef, ann_k = derive_search_params(10, p)
    b.setEfSearch(ef)
    for _ in range(3):
        t0 = time.perf_counter(); ids, dist, cnt, ctr = b.searchManyArrays(qi8, ann_k, with_counters=True); t1 = time.perf_counter()
    prof = np.zeros((nq, 32), np.uint64)
    L = b._L
    L.mvdb_debug_traverse_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert L.mvdb_debug_traverse_profile(b.handle, prof.ctypes.data_as(C.c_void_p), nq)
    print(f"host wall of searchMany (incl. copies): {(t1 - t0) * 1e3:.2f} ms; fallbacks {b.leanFallbacks()}")
    print("counters mean:", ctr.mean(0))
    names = ["query+descent", "list+issue", "heap work (shadow)", "rows+dist", "visited+compact", "admission (incl refill)",
             "  of which refill", "log+predict"]
    heap_parts = prof[:, [16, 17, 18, 2]].astype(np.float64)   # pushes | root + last, check | sift-down | new root, list request
    prof[:, 2] += prof[:, 16] + prof[:, 17] + prof[:, 18]      # slot 2 = the whole heap work in the table below
    tot = prof[:, [0, 1, 2, 3, 4, 5, 7, 13, 14, 15, 19, 20]].sum(1).astype(np.float64)
    ex = ctr[:, 1].mean()
    print(f"mean cycles/query {tot.mean():.0f} max {tot.max():.0f}; expansions {ex:.0f} -> {tot.mean() / ex:.0f} cycles per expansion")
    for i, nm in enumerate(names):
        if i != 6:
            print(f"  {nm:26s} {prof[:, i].mean():12.0f} cyc  {100 * prof[:, i].sum() / tot.sum():5.1f} %   {prof[:, i].mean() / ex:8.0f} /expansion")
    for i, nm in enumerate(["heap: pushes", "heap: root + last, check", "heap: sift-down", "heap: new root, list request"]):
        print(f"  {nm:28s} {heap_parts[:, i].mean() / ex:10.0f} cycles per expansion")
    for i, nm in zip([13, 14, 15, 19, 20, 5], ["adm: preamble", "adm: parallel decision", "adm: inserts + append", "adm: one-at-a-time path",
                                             "adm: refills", "adm: remove largest, log"]):
        print(f"  {nm:28s} {prof[:, i].mean() / ex:10.0f} cycles per expansion   {100 * prof[:, i].sum() / tot.sum():5.1f} %")
    for i, nm in zip([8, 9, 10, 11, 12, 21, 22], ["lists not prefetched", "admitted", "refills", "dense refills", "next candidate from the list",
                                                 "admitted one at a time", "lists decided in parallel"]):
        print(f"  {nm:28s} {prof[:, i].mean():10.1f} per query")
