"""Generates the golden fixtures under tests/golden/ with the CPU oracle (oracle/mvdb_oracle.c).

    python tests/golden/make_golden.py

The reference (AssemblyScript/WASM + TypeScript) cannot be built or run in this image
 and holds no binary fixtures of its own, so these vectors pin REGRESSIONS of the
oracle and of the HIP path against each other -- not the oracle against the WASM binary.
Each .npz holds: a V3 dump.bin image built by the reference's insert order (seed_rng(42)), the
float32 store, raw (un-normalised) float32 queries, and the expected outputs of every stage.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def make(name, n, dim, preset, nq, topk, seed_base, seed_q, tie_heavy=False):
    p = O.PRESETS[preset]
    rng = np.random.default_rng(seed_base)
    if tie_heavy:  # tiny value range: distance ties everywhere, exercises the heap tie rules
        base = rng.integers(-3, 4, (n, dim)).astype(np.float32)
        base[np.all(base == 0, axis=1), 0] = 1
    else:
        base = rng.standard_normal((n, dim), dtype=np.float32)
    base = O.normalize_rows_f32(base)
    b8 = O.quantize_rows_i8(base)
    ix = O.OracleIndex(dim, p["m"], p["ef_construction"], n + 7, ef_search=p["baseEfSearch"],
                       results_cap=p["resultsCap"], seed=42)
    ix.insert_many(np.arange(n, dtype=np.int32), b8)
    dump = np.frombuffer(ix.save_bytes(), np.uint8)
    q = np.random.default_rng(seed_q).standard_normal((nq, dim), dtype=np.float32) * 3.0
    qn = O.normalize_rows_f32(q)
    q8 = O.quantize_rows_i8(qn)
    tk, ef, ann_k = O.derive_search_params(topk, preset)
    allow = np.random.default_rng(seed_q + 1).integers(0, 2**32, (n + 7 + 31) // 32, dtype=np.uint64).astype(np.uint32)
    out_ids, out_l2, out_n, ann_ids, ann_cnt, ctr = O.search_many_db(ix, q, base, topk, preset)
    f_ids, f_l2, f_n, _, _, _ = O.search_many_db(ix, q, base, topk, preset, allow_bitmap=allow)
    ix.set_ef_search(ef)
    _, ann_dist, _, _ = ix.search_many(q8, ann_k)
    scan_ids, scan_val, scan_cnt = ix.scan_many(q8, 10, "l2")
    np.savez_compressed(os.path.join(HERE, name + ".npz"), dim=dim, m=p["m"], ef_construction=p["ef_construction"],
                        preset=preset, topk=tk, ef_search=ef, ann_k=ann_k, dump=dump, base_f32=base, queries=q,
                        q_norm=qn, q_i8=q8, ann_ids=ann_ids, ann_dist=ann_dist, ann_cnt=ann_cnt,
                        n_dist=ctr["n_dist"], n_l0lists=ctr["n_l0lists"], n_upperlists=ctr["n_upperlists"],
                        top_ids=out_ids, top_l2=out_l2, top_n=out_n, allow=allow, f_ids=f_ids, f_l2=f_l2, f_n=f_n,
                        scan_ids=scan_ids, scan_val=scan_val, scan_cnt=scan_cnt)
    print(name, os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")


if __name__ == "__main__":
    make("g384_balanced", 800, 384, "balanced", 24, 10, 101, 202)
    make("g64_fast_ties", 900, 64, "fast", 24, 10, 303, 404, tie_heavy=True)
