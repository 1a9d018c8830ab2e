"""Diagnostic: per-phase cycle shares of the traversal kernel (s_memtime stamps).
Builds a separate libmvdb_hip_prof.so with -DMVDB_TRAVERSE_PROFILE; never shipped or benchmarked."""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minivectordb_amd import _native, build as B

def build_prof():
    out = os.path.join(ROOT, "minivectordb_amd", "lib", "libmvdb_hip_prof.so")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
           "-fno-fast-math", "-DMVDB_TRAVERSE_PROFILE", "-I", os.path.join(ROOT, "include"), "-I", B.CSRC, "-x", "hip",
           os.path.join(B.CSRC, "engine.hip"), os.path.join(B.CSRC, "host_graph.cpp"), "-o", out]
    subprocess.run(cmd, check=True)
    return out

if __name__ == "__main__":
    if "--build" in sys.argv:
        print(build_prof()); sys.exit(0)
    _native.LIB_PATH = os.path.join(ROOT, "minivectordb_amd", "lib", "libmvdb_hip_prof.so")
    from minivectordb_amd import HipBridge, resolve_preset, derive_search_params
    clustered = "--clustered" in sys.argv  # bench.py's second dataset (4096 Gaussian clusters) instead of i.i.d. rows
    if clustered:
        sys.argv.remove("--clustered")
        import bench
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    nq, dim = 1000, 384
    p = resolve_preset("balanced")
    rng = np.random.default_rng(12345)
    base = bench.make_rows(n, dim, 12345 + 777, True) if clustered else rng.standard_normal((n, dim), dtype=np.float32)
    b = HipBridge(); b.init(dict(dim=dim, m=p["m"], ef=p["ef_construction"], capacity=n, seed=42, resultsCap=4096))
    qn, q8 = b.prepareQueries(base)
    b.buildGpu(q8)
    b.setTuning(traverse_waves=1)  # the one-wave kernel (tools/profile_lean.py reports the lean kernel's stamps)
    q = (bench.make_queries(1, nq, dim, 54321, qn, True)[0] if clustered
         else np.random.default_rng(54321).standard_normal((nq, dim), dtype=np.float32))
    _, qi8 = b.prepareQueries(q)
    tk, ef, ann_k = derive_search_params(10, p)
    b.setEfSearch(ef)
    for _ in range(3):
        t0 = time.perf_counter(); ids, dist, cnt, ctr = b.searchManyArrays(qi8, ann_k, with_counters=True); t1 = time.perf_counter()
    prof = np.zeros((nq, 32), np.uint64)
    L = b._L
    L.mvdb_debug_traverse_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert L.mvdb_debug_traverse_profile(b.handle, prof.ctypes.data_as(C.c_void_p), nq)
    print(f"host wall of searchMany (incl. copies): {(t1 - t0) * 1e3:.2f} ms")
    print("counters mean:", ctr.mean(0))
    names = ["descent", "cand_pop", "list+visited", "rows+dist", "heap_updates", "final_sort", "output"]
    print("admitted neighbours per query (mean):", prof[:, 7].mean())
    tot = prof[:, :7].sum(1).astype(np.float64)
    print(f"mean cycles/query {tot.mean():.0f}, max {tot.max():.0f}")
    for i, nm in enumerate(names):
        print(f"  {nm:14s} {prof[:, i].mean():12.0f} cyc  {100 * prof[:, i].sum() / tot.sum():5.1f} %")
