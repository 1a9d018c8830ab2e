"""Stage micro-bench on the GPU box: prep and rerank kernels at the bench shape, without an index build.

    python tools/stage_bench.py [rows] [dim] [nq] [ann_k] [topk]

Synthetic candidates (random distinct ids per query).  Prints per-launch kernel times from the
engine's HIP-event stage timers, checks a sample of queries against the oracle bit-for-bit.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from minivectordb_amd import HipBridge  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker only)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    ann_k = int(sys.argv[4]) if len(sys.argv) > 4 else 1200
    topk = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    rng = np.random.default_rng(5)
    base = rng.standard_normal((rows, dim), dtype=np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    cand = rng.integers(0, rows, size=(nq, ann_k), dtype=np.int64).astype(np.int32)
    cc = np.full(nq, ann_k, np.int32)

    b = HipBridge()
    b.init({"dim": dim, "m": 16, "ef": 100, "capacity": 1024, "seed": 1})
    b.attachF32(base, resident=True)
    L, h = b._L, b.handle
    L.mvdb_enable_timing(h, 1)

    def stats(stage):
        n, tot = C.c_longlong(0), C.c_double(0.0)
        L.mvdb_stage_stats(h, stage, C.byref(n), C.byref(tot), 1)
        return n.value, tot.value

    for _ in range(3):
        qn, qi = b.prepareQueries(q)
        ids, l2, n = b.rerankMany(qn, cand, cc, ann_k, topk)
    stats(0), stats(2)
    reps = 10
    for _ in range(reps):
        qn, qi = b.prepareQueries(q)
        ids, l2, n = b.rerankMany(qn, cand, cc, ann_k, topk)
    n0, t0 = stats(0)
    n2, t2 = stats(2)
    bytes_rr = nq * ann_k * dim * 4
    out = {"rows": rows, "dim": dim, "nq": nq, "ann_k": ann_k, "topk": topk,
           "prep_ms": t0 / max(n0, 1), "rerank_ms": t2 / max(n2, 1),
           "rerank_GBps": bytes_rr / (t2 / max(n2, 1) * 1e-3) / 1e9}

    # parity sample against the oracle
    ns = min(nq, 48)
    qn_o = O.normalize_rows_f32(q[:ns])
    qi_o = O.quantize_rows_i8(qn_o)
    ok = bool(np.array_equal(qn_o.view(np.uint32), qn[:ns].view(np.uint32)) and np.array_equal(qi_o, qi[:ns]))
    o_ids, o_l2, o_n = O.rerank_many(qn_o, base, cand[:ns], cc[:ns], ann_k, topk)
    ok = ok and bool(np.array_equal(o_ids, ids[:ns]) and np.array_equal(o_l2.view(np.uint64), l2[:ns].view(np.uint64))
                     and np.array_equal(o_n, n[:ns]))
    out["parity_sample_ok"] = ok
    print(json.dumps(out))
    b.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
