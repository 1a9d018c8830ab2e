#!/usr/bin/env python3
"""bench.py -- searchMany QPS @ recall@10 on MI355X (BASELINE.json metric).

A "step" = one searchMany batch through the fused device pipeline
(normalise+quantise -> HNSW traversal ef=annK -> float32 rerank -> top-10), queries
and results resident in HBM.  Workload at N=1: BASELINE configs[1] -- 1M x 384-d
synthetic normalised vectors, balanced preset (M=16, efc=300, ef=annK=1200),
1k-query batches.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU modes: "replicas" (default: the index fits one GPU, each rank
answers its own query batch, results all-gathered; weak scaling in queries) and
"sharded" (each rank holds an id-range shard, every rank sees every query, per-shard
top-k all-gathered over RCCL and merged by (l2, id)).

Prints ONE JSON line (rank 0).  The graph is produced by the engine's own multi-threaded
builder (reference insert algorithm, not its sequential order) because no reference
dump of this size exists; parity on that dump is checked against the CPU oracle on a
sample inside the cpu_baseline leg.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def log(msg):
    sys.stderr.write(f"[bench {time.strftime('%H:%M:%S')}] {msg}\n")
    sys.stderr.flush()


def run_with_heartbeat(fn, what):
    """Run a long C call in a thread (ctypes drops the GIL) and print progress lines."""
    box = {}

    def work():
        try:
            box["r"] = fn()
        except BaseException as e:  # noqa: BLE001
            box["e"] = e

    th = threading.Thread(target=work)
    t0 = time.time()
    th.start()
    while th.is_alive():
        th.join(20.0)
        if th.is_alive():
            log(f"{what}: {time.time() - t0:.0f}s elapsed")
    if "e" in box:
        raise box["e"]
    return box.get("r")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=1_000_000, help="vectors per index (per shard when sharded)")
    ap.add_argument("--nq", type=int, default=1000, help="queries per batch")
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--preset", default="balanced")
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--mode", default="replicas", choices=["replicas", "sharded"])
    ap.add_argument("--build-threads", type=int, default=int(os.environ.get("OMP_NUM_THREADS", "16")))
    ap.add_argument("--cpu-sample", type=int, default=256, help="queries timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--in-flight", type=int, default=3, choices=[1, 2, 3, 4],
                    help="batches in flight: batch i runs on stream i %% IN_FLIGHT (the next batches' workgroups "
                         "fill the places of queries that finished early); 1 = one batch at a time")
    ap.add_argument("--pipeline", action="store_true",
                    help="batch i's rerank on a second stream under batch i+1's traversal (measured slower on "
                         "MI355X: 4.52 vs 4.34 ms per batch, DESIGN.md 4.1; default: one batch at a time)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from minivectordb_amd import HipBridge, resolve_preset, derive_search_params
    from minivectordb_amd.distributed import gather_topk
    from minivectordb_amd.pipeline import ResidentSearcher

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    preset = resolve_preset(args.preset)
    N, nq, dim, topk = args.n, args.nq, args.dim, args.topk
    tk, ef_search, ann_k = derive_search_params(topk, preset)

    # ---------------- data: synthetic normalised vectors --------------
    sharded = world > 1 and args.mode == "sharded"
    base_seed = 12345 + (rank if sharded else 0)
    t0 = time.time()
    rng = np.random.default_rng(base_seed)
    base = np.empty((N, dim), np.float32)
    chunk = 131072
    for s in range(0, N, chunk):
        base[s:s + chunk] = rng.standard_normal((min(chunk, N - s), dim), dtype=np.float32)
    log(f"rank {rank}: generated {N}x{dim} gaussian base in {time.time() - t0:.1f}s")

    bridge = HipBridge(device=local_rank)
    bridge.init(dict(dim=dim, m=preset["m"], ef=preset["ef_construction"], efSearch=preset["baseEfSearch"],
                     capacity=N, seed=42, resultsCap=preset["resultsCap"]))
    # normalise + quantise exactly as insertMany would (device prep kernel, bit-exact to the reference)
    t0 = time.time()
    base8 = np.empty((N, dim), np.int8)
    for s in range(0, N, chunk):
        qn, q8 = bridge.prepareQueries(base[s:s + chunk])
        base[s:s + chunk] = qn
        base8[s:s + chunk] = q8
    log(f"rank {rank}: normalised+quantised on device in {time.time() - t0:.1f}s")

    # ---------------- graph: built once (rank 0 in replicas mode), shared as a V3 dump --------
    t0 = time.time()
    if sharded or world == 1 or rank == 0:
        run_with_heartbeat(lambda: bridge.buildParallel(base8, args.build_threads),
                           f"rank {rank}: building HNSW graph ({args.build_threads} threads)")
        log(f"rank {rank}: built {bridge.elementCount()} nodes in {time.time() - t0:.1f}s")
    if world > 1 and not sharded:
        if rank == 0:
            dump = np.frombuffer(bridge.saveBytes(), np.uint8)
            size = torch.tensor([dump.size], dtype=torch.int64, device=dev)
        else:
            size = torch.zeros(1, dtype=torch.int64, device=dev)
        dist.broadcast(size, 0)
        buf = torch.empty(int(size.item()), dtype=torch.uint8, device=dev)
        if rank == 0:
            buf.copy_(torch.from_numpy(dump.copy()))
        dist.broadcast(buf, 0)
        if rank != 0:
            bridge.loadBytes(buf.cpu().numpy().tobytes())
        del buf
    bridge.attachF32(base, resident=True)

    # ---------------- queries ------------------------------------------------------------------
    q_seed = 54321 + (rank if (world > 1 and not sharded) else 0)
    q_host = np.random.default_rng(q_seed).standard_normal((nq, dim), dtype=np.float32)
    q_dev = torch.from_numpy(q_host).to(dev)
    searcher = ResidentSearcher(bridge, nq, tk, dev)
    id_offset = rank * N if sharded else 0

    pipelined = args.pipeline
    interleaved = args.in_flight >= 2 and not pipelined
    if pipelined or interleaved:
        # 2-wave rerank workgroups (34 KB of LDS) find room beside running traversals: +2 % with 3 in flight
        os.environ.setdefault("MVDB_RERANK_WAVES", "2")

    def step():
        if pipelined or interleaved:
            if pipelined:
                slot, ids, l2, n = searcher.submit(q_dev, ef_search, ann_k)
            else:  # q_dev is resident and never rewritten: no need to order the batch behind this stream
                slot, ids, l2, n = searcher.submit_interleaved(q_dev, ef_search, ann_k, sync_with_current=False,
                                                                   depth=args.in_flight)
            if world == 1:
                return ids, l2, n
            searcher.wait(slot)
        else:
            ids, l2, n = searcher.query_many(q_dev, ef_search, ann_k)
            if world == 1:
                return ids, l2, n
        if sharded:
            gid = ids + id_offset if id_offset else ids
            all_ids, all_l2, all_n = gather_topk(gid, l2, n)          # the one exchange step (RCCL all-gather)
            out = searcher.merge_topk(all_ids, all_l2, all_n, world, nq)
        else:  # replicas: every rank ends up with every rank's answers
            out = gather_topk(ids, l2, n)
        if pipelined or interleaved:
            searcher.release(slot)  # the collectives above are the last readers of this slot's outputs
        return out

    # ---------------- warmup + timed region -----------------------------------------------------
    log(f"rank {rank}: warmup {args.warmup} steps (first step uploads the index)")
    for _ in range(args.warmup):
        step()
    if pipelined or interleaved:
        searcher.drain()
    torch.cuda.synchronize(dev)
    bridge._L.mvdb_enable_timing(bridge.handle, 1)
    searcher.stage_stats(0), searcher.stage_stats(1), searcher.stage_stats(2)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    if pipelined or interleaved:
        searcher.drain()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    bridge._L.mvdb_enable_timing(bridge.handle, 0)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ms_per_step = elapsed / args.steps * 1e3
    queries_per_step = nq * (world if (world > 1 and not sharded) else 1)
    qps = queries_per_step / (elapsed / args.steps)

    # ---------------- roofline of the dominant kernel (traversal), live HIP-event timing ---------
    n_launch, trav_ms_total = searcher.stage_stats(1)
    _, rerank_ms_total = searcher.stage_stats(2)
    _, prep_ms_total = searcher.stage_stats(0)
    ctr = searcher.ctr[:nq].cpu().numpy().astype(np.int64)
    m, m0 = preset["m"], 2 * preset["m"]
    # B_ann = n_dist*DIM + n_L0lists*(4+4*M_MAX0) + n_upperlists*(4+4*M) + DIM per query
    b_ann = ctr[:, 0] * dim + ctr[:, 1] * (4 + 4 * m0) + ctr[:, 2] * (4 + 4 * m) + dim
    b_rerank = ctr[:, 3] * dim * 4 + dim * 4 + tk * 8
    trav_ms = trav_ms_total / max(n_launch, 1)
    achieved = float(b_ann.sum()) / (trav_ms * 1e-3) / 1e9 if trav_ms > 0 else 0.0
    # HBM traffic per launch: FETCH_SIZE from a separate rocprofv3 --pmc pass of this workload, corrected
    # by the gfx950 calibration recorded next to it (profiles/r1/pmc_traffic_1m_balanced.json)
    traffic = None
    # two waves per query whenever a neighbour list fits one 64-lane chunk (engine.hip: traverse_use_duo)
    kernel = "traverse_kernel" if (m0 > 64 or os.environ.get("MVDB_TRAVERSE_WAVES") == "1") else "traverse_duo_kernel"
    pmc_file = os.path.join(ROOT, "profiles", "r1", "pmc_traffic_1m_balanced.json")
    if N == 1_000_000 and dim == 384 and nq == 1000 and args.preset == "balanced" and os.path.exists(pmc_file):
        with open(pmc_file) as f:
            pmc = json.load(f)["traverse"]
        if pmc.get("kernel", "traverse_kernel") == kernel:
            traffic = int(pmc["corrected_bytes_per_launch"])
    roofline = dict(bound="hbm", kernel=kernel, achieved=round(achieved, 2), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic,
                    algorithmic_bytes_per_launch=int(b_ann.sum()), avg_launch_ms=round(trav_ms, 4),
                    rerank_bytes_per_launch=int(b_rerank.sum()),
                    rerank_avg_launch_ms=round(rerank_ms_total / max(n_launch, 1), 4),
                    prep_avg_launch_ms=round(prep_ms_total / max(n_launch, 1), 4),
                    # launches overlap when several batches are in flight: `achieved` (per launch, by the
                    # definition) then understates what the chip delivers; bytes of all timed launches / wall:
                    effective_gbs=round(float(b_ann.sum()) * args.steps / elapsed / 1e9, 2),
                    effective_frac=round(float(b_ann.sum()) * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 5),
                    pipeline_gbs=round(float(b_ann.sum() + b_rerank.sum()) / (ms_per_step * 1e-3) / 1e9, 2))

    # ---------------- recall@10 vs exact float32 brute force (rank 0, own queries) --------------
    recall = None
    ids_dev, l2_dev, n_dev = searcher.query_many(q_dev, ef_search, ann_k)
    torch.cuda.synchronize(dev)
    got = ids_dev.cpu().numpy()
    if (pipelined or interleaved) and world == 1:  # the overlapped batches answered the same queries: same bits expected
        p_ids, p_l2, p_n = last
        assert torch.equal(p_ids, ids_dev) and torch.equal(p_l2, l2_dev) and torch.equal(p_n, n_dev), \
            "pipelined and single-stream results differ"
    if not args.no_recall and rank == 0:
        qn_host, _ = bridge.prepareQueries(q_host)
        qn = torch.from_numpy(qn_host).to(dev)
        best_d = torch.full((nq, tk), float("inf"), device=dev)
        best_i = torch.full((nq, tk), -1, dtype=torch.int64, device=dev)
        for s in range(0, N, 262144):  # unit vectors: smallest L2 == largest dot
            bt = torch.from_numpy(base[s:s + 262144]).to(dev)
            d = 2.0 - 2.0 * (qn @ bt.T)
            dd, ii = torch.topk(d, tk, dim=1, largest=False)
            cat_d = torch.cat([best_d, dd], 1)
            cat_i = torch.cat([best_i, ii + s], 1)
            sel = torch.topk(cat_d, tk, dim=1, largest=False).indices
            best_d, best_i = torch.gather(cat_d, 1, sel), torch.gather(cat_i, 1, sel)
            del bt, d
        gt = best_i.cpu().numpy()
        recall = float(np.mean([len(set(got[i, :tk]) & set(gt[i])) / tk for i in range(nq)]))

    # ---------------- CPU baseline: the oracle (C restatement) on a bounded sample ---------------
    cpu_baseline = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import oracle as O
        ns = min(args.cpu_sample, nq)
        ix = O.OracleIndex(dim, preset["m"], preset["ef_construction"], 8, seed=1)
        t0 = time.time()
        assert ix.load_bytes(bridge.saveBytes()), "oracle rejected the engine's dump"
        log(f"oracle loaded the dump in {time.time() - t0:.1f}s; timing {ns} queries on 1 host thread")
        t0 = time.perf_counter()
        o_ids, o_l2, o_n, ann_ids, ann_cnt, octr = O.search_many_db(ix, q_host[:ns], base, topk, args.preset, threads=1)
        cpu_s = time.perf_counter() - t0
        same = bool(np.array_equal(o_ids, got[:ns]) and
                    np.array_equal(o_l2.view(np.uint64), l2_dev[:ns].cpu().numpy().view(np.uint64)))
        ncores = os.cpu_count() or 1
        nthr = min(args.build_threads, ncores)
        t0 = time.perf_counter()
        O.search_many_db(ix, q_host[:ns], base, topk, args.preset, threads=nthr)
        cpu_mt_s = time.perf_counter() - t0
        cpu_baseline = dict(value=round(ns / cpu_s, 2), unit="queries/s", cores=1, kind="port",
                            sample=f"first {ns} of the {nq} bench queries, full searchMany pipeline on the same "
                                   f"{N}-node dump (C restatement of the WASM path, 1 thread = the reference's "
                                   "sequential loop)",
                            all_cores=dict(value=round(ns / cpu_mt_s, 2), cores=nthr),
                            gpu_matches_oracle_on_sample=same)
        ix.close()

    if rank == 0:
        out = dict(metric="searchMany QPS @ recall@10 (topK=10, 384-d int8+f32 rerank)",
                   value=round(qps, 2), unit="queries/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(ms_per_step, 4), higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="i8", dtype_detail="int8 rows, exact i32 distances (traversal); f32 rows, f64 accumulate (rerank)",
                   data="synthetic",
                   config=dict(workload=f"{N}x{dim} int8 HNSW (preset {args.preset}: M={m}, efc="
                                        f"{preset['ef_construction']}, ef=annK={ann_k}) + f32 rerank, "
                                        f"batch {nq} queries, topK={tk}",
                               n=N, dim=dim, nq=nq, topk=tk, ann_k=ann_k, preset=args.preset,
                               mode=("single" if world == 1 else args.mode),
                               batches=("pipelined: rerank of batch i on a second stream under the traversal "
                                        "of batch i+1" if pipelined else
                                        f"{args.in_flight} in flight: batch i on stream i % {args.in_flight}" if interleaved
                                        else "one at a time"),
                               graph="engine parallel builder (reference insert algorithm, unordered)"),
                   recall_at_10=None if recall is None else round(recall, 4),
                   roofline=roofline, cpu_baseline=cpu_baseline,
                   counters=dict(mean_n_dist=float(ctr[:, 0].mean()), mean_n_l0lists=float(ctr[:, 1].mean()),
                                 mean_n_upperlists=float(ctr[:, 2].mean())))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    bridge.close()


if __name__ == "__main__":
    main()
