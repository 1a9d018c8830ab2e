"""GPU bulk builder vs the multi-threaded host builder on one synthetic set (run on the GPU box).

    python tools/build_bench.py [rows] [dim] [preset] [nq] [compare_host(0/1)] [clusters]

Prints one JSON line: build seconds and counters, structural checks of the graph, recall@10 and mean n_dist of
the balanced searchMany pipeline on the GPU-built graph and (optionally) on the host-built one.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from minivectordb_amd import HipBridge, derive_search_params, resolve_preset  # noqa: E402


def make_data(rows, dim, clusters, seed):
    rng = np.random.default_rng(seed)
    if clusters <= 0:
        x = rng.standard_normal((rows, dim), dtype=np.float32)
    else:  # gaussian mixture: neighbourhood structure
        centers = rng.standard_normal((clusters, dim), dtype=np.float32)
        x = centers[rng.integers(0, clusters, rows)] + 0.35 * rng.standard_normal((rows, dim), dtype=np.float32)
    return x


def structure_report(b, n, m, m0):
    import struct
    dump = b.saveBytes()
    hdr = struct.unpack_from("<12i", dump, 0)
    dim = hdr[2]
    off, bad, edges0, levels = 48, 0, 0, np.zeros(4, np.int64)
    full0 = 0
    for _ in range(hdr[8]):
        nid, lvl = struct.unpack_from("<2i", dump, off)
        off += 8 + dim
        levels[lvl] += 1
        for l in range(lvl + 1):
            cap = m0 if l == 0 else m
            cnt = struct.unpack_from("<i", dump, off)[0]
            ids = np.frombuffer(dump, np.int32, cap, off + 4)
            off += 4 + 4 * cap
            v = ids[:cnt]
            if cnt < 0 or cnt > cap or (v < 0).any() or (v >= n).any() or (v == nid).any() or len(set(v.tolist())) != cnt \
                    or (ids[cnt:] != -1).any():
                bad += 1
            if l == 0:
                edges0 += cnt
                full0 += cnt == cap
    return dict(bad_lists=bad, mean_deg0=edges0 / max(hdr[8], 1), full0_frac=full0 / max(hdr[8], 1),
                levels=levels.tolist(), entry=hdr[9], max_level=hdr[10]), dump


def quality(b, base, q, preset, topk=10):
    import torch
    tk, ef, ann_k = derive_search_params(topk, preset)
    b.attachF32(base, resident=True)
    t0 = time.time()
    ids, l2, n, ctr = b.queryMany(q, tk, ef, ann_k, with_counters=True)
    dt = time.time() - t0
    dev = torch.device("cuda", 0)
    qn, _ = b.prepareQueries(q)
    qd = torch.from_numpy(qn).to(dev)
    best_d = torch.full((q.shape[0], tk), float("inf"), device=dev)
    best_i = torch.full((q.shape[0], tk), -1, dtype=torch.int64, device=dev)
    for s in range(0, base.shape[0], 262144):
        bt = torch.from_numpy(base[s:s + 262144]).to(dev)
        d = 2.0 - 2.0 * (qd @ bt.T)
        dd, ii = torch.topk(d, tk, dim=1, largest=False)
        cat_d, cat_i = torch.cat([best_d, dd], 1), torch.cat([best_i, ii + s], 1)
        sel = torch.topk(cat_d, tk, dim=1, largest=False).indices
        best_d, best_i = torch.gather(cat_d, 1, sel), torch.gather(cat_i, 1, sel)
    gt = best_i.cpu().numpy()
    rec = float(np.mean([len(set(ids[i, :tk].tolist()) & set(gt[i].tolist())) / tk for i in range(q.shape[0])]))
    return dict(recall_at_10=round(rec, 4), mean_n_dist=float(ctr[:, 0].mean()), mean_l0lists=float(ctr[:, 1].mean()),
                first_query_call_s=round(dt, 3))


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
    preset_name = sys.argv[3] if len(sys.argv) > 3 else "balanced"
    nq = int(sys.argv[4]) if len(sys.argv) > 4 else 500
    compare = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    clusters = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    max_block = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    import torch
    torch.cuda.init()
    preset = resolve_preset(preset_name)
    m = preset["m"]
    cfg = dict(dim=dim, m=m, ef=preset["ef_construction"], efSearch=preset["baseEfSearch"], capacity=rows, seed=42,
               resultsCap=preset["resultsCap"])
    b = HipBridge()
    b.init(cfg)
    x = make_data(rows, dim, clusters, 12345)
    base = np.empty_like(x)
    b8 = np.empty((rows, dim), np.int8)
    for s in range(0, rows, 131072):
        base[s:s + 131072], b8[s:s + 131072] = b.prepareQueries(x[s:s + 131072])
    del x
    q = make_data(nq, dim, clusters, 54321) if clusters <= 0 else None
    if q is None:  # queries from the same mixture
        rng = np.random.default_rng(999)
        q = base[rng.integers(0, rows, nq)] + 0.05 * rng.standard_normal((nq, dim), dtype=np.float32)
    out = dict(rows=rows, dim=dim, preset=preset_name, clusters=clusters)
    t0 = time.time()
    st = b.buildGpu(b8, max_block)
    out["gpu_build_s"] = round(time.time() - t0, 2)
    out["gpu_build"] = {k: (round(v, 3) if k.endswith("seconds") else int(v)) for k, v in st.items()}
    print(f"[build_bench] gpu build {out['gpu_build_s']}s", file=sys.stderr, flush=True)
    rep, dump = structure_report(b, rows, m, 2 * m) if rows <= 300_000 else (None, None)
    out["gpu_graph"] = rep
    out["gpu_quality"] = quality(b, base, q, preset)
    if compare:
        b2 = HipBridge()
        b2.init(cfg)
        t0 = time.time()
        b2.buildParallel(b8, 16)
        out["host_build_s"] = round(time.time() - t0, 2)
        rep2, _ = structure_report(b2, rows, m, 2 * m) if rows <= 300_000 else (None, None)
        out["host_graph"] = rep2
        out["host_quality"] = quality(b2, base, q, preset)
        b2.close()
    print(json.dumps(out))
    b.close()


if __name__ == "__main__":
    main()
