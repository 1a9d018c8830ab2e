#!/usr/bin/env python3
"""bench.py -- searchMany QPS @ recall@10 on MI355X (BASELINE.json metric).

A "step" = one searchMany batch through the fused device pipeline (normalise + quantise -> HNSW traversal with
ef = annK -> float32 rerank -> top-10), queries and results resident in HBM.  Workload at N=1: BASELINE
configs[1] -- 1M x 384-d synthetic normalised vectors, balanced preset (M=16, efc=300, ef=annK=1200), 1k-query
batches; `--n 10000000` is the north-star size (10M x 384 balanced on one GPU).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

What one run reports (ONE JSON line, rank 0):
  value                    K timed steps; every step copies its query batch from pinned host memory (H2D) and its
                           result block back (D2H) -- SURVEY.md 8(d) counts both; `--in-flight` batches overlap on as
                           many streams, `--batches` DISTINCT query batches rotate (no batch is answered twice in a row)
  resident                 the same steps on batches already in HBM, results left in HBM (no copies)
  sustained                >= 12 windows of >= 20 steps back to back (>= `--sustain-s` seconds); `median_ms_per_step`
                           is the median window duration / steps per window (stream events, no drain between windows)
  host_slots               the same work through mvdb_query_many_submit/_wait on pageable host buffers
  roofline                 traversal kernel: algorithmic bytes per launch (from the kernel's own
                           counters) / mean launch duration (HIP events on the launching stream, timed region only);
                           `frac_structured` = the same figure on the clustered set
  cpu_baseline             the C oracle (a restatement of the WASM path) on a bounded sample, 1 thread and all cores;
                           the same leg asserts that the GPU results equal the oracle's on that sample
  second_dataset           the same pipeline on a clustered set (neighbourhood structure, recall >= 0.9)
  scan_1m_512              BASELINE configs[2]: exact int8 scan on the matrix cores, 1M x 512, 4096 queries
  north_star_10m           the north-star size, 10M x 384 balanced on this GPU: value, recall, roofline, cpu_baseline
(`--legs none` drops the last three; they run by default only for the default 1M x 384 single-GPU line.)
N>1: headline = SHARDED (each rank an id-range shard of `--n` vectors with its own sub-index, every
rank sees every query, ONE packed all-gather of the per-shard top-k per batch + (l2, id) merge); `replicas` = the
ranks answering their own batches with no exchange, as a second field.

`--workload scan`: BASELINE configs[2] -- exact int8 brute-force scan on the matrix cores (1M x 512, 4k queries),
with the HNSW path's recall against that exact result.

The graph comes from the engine's GPU bulk builder (reference insert algorithm applied block-parallel; no
reference dump of this size can exist: the 32-bit WASM holds ~3.5 M nodes); parity on that dump is checked
against the CPU oracle on a sample inside the cpu_baseline leg.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
RANDOM_ROW_GBS = 6490.0  # random 384-byte rows over 4 GiB, 8 lanes x 16 B x 3, nothing else: tools/micro/probe_rate.hip, profiles/r3/probe_rate_random_4B_loads.txt
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA = 2x the ~2.5 PF dense bf16 rate (same guide, Matrix cores)
CLUSTERS, CLUSTER_SIGMA, CLUSTER_QUERY_SIGMA = 4096, 0.35, 0.05


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


def make_rows(n, dim, seed, clustered):
    """Synthetic rows before normalisation: i.i.d. N(0,1) or a Gaussian mixture.
    Above 2 M rows the chunks are drawn by a small thread pool, one generator per chunk (numpy drops the GIL)."""
    out = np.empty((n, dim), np.float32)
    chunk = 131072
    if n > 2_000_000 and not clustered:
        from concurrent.futures import ThreadPoolExecutor

        def fill(ci):
            s0 = ci * chunk
            c = min(chunk, n - s0)
            np.random.default_rng([seed, ci]).standard_normal((c, dim), dtype=np.float32, out=out[s0:s0 + c])
        with ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1))) as ex:
            list(ex.map(fill, range(-(-n // chunk))))
        return out
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((CLUSTERS, dim), dtype=np.float32) if clustered else None
    for s in range(0, n, chunk):
        c = min(chunk, n - s)
        x = rng.standard_normal((c, dim), dtype=np.float32)
        if clustered:
            x = centers[rng.integers(0, CLUSTERS, c)] + CLUSTER_SIGMA * x
        out[s:s + c] = x
    return out


def make_queries(nb, nq, dim, seed, base_norm, clustered):
    rng = np.random.default_rng(seed)
    if not clustered:  # distinct from the base, same distribution
        return rng.standard_normal((nb, nq, dim), dtype=np.float32)
    pick = rng.integers(0, base_norm.shape[0], nb * nq)  # a stored point plus a little noise
    return (base_norm[pick] + CLUSTER_QUERY_SIGMA * rng.standard_normal((nb * nq, dim), dtype=np.float32)
            / np.float32(math.sqrt(dim))).reshape(nb, nq, dim).astype(np.float32)


def normalise_quantise(bridge, rows):
    """normalizeF32InPlace + quantizeToI8 exactly as insertMany would (device prep kernel, bit-exact)."""
    b8 = np.empty(rows.shape, np.int8)
    for s in range(0, rows.shape[0], 131072):
        rows[s:s + 131072], b8[s:s + 131072] = bridge.prepareQueries(rows[s:s + 131072])
    return rows, b8


def exact_topk(torch, dev, base, qn, tk, id_offset=0):
    """Exact float32 top-k by brute force on the device (unit vectors: smallest L2 == largest dot)."""
    nq = qn.shape[0]
    q = torch.from_numpy(qn).to(dev)
    best_d = torch.full((nq, tk), float("inf"), device=dev)
    best_i = torch.full((nq, tk), -1, dtype=torch.int64, device=dev)
    for s in range(0, base.shape[0], 262144):
        bt = torch.from_numpy(base[s:s + 262144]).to(dev)
        d = 2.0 - 2.0 * (q @ bt.T)
        dd, ii = torch.topk(d, min(tk, bt.shape[0]), dim=1, largest=False)
        cat_d, cat_i = torch.cat([best_d, dd], 1), torch.cat([best_i, ii + s + id_offset], 1)
        sel = torch.topk(cat_d, tk, dim=1, largest=False).indices
        best_d, best_i = torch.gather(cat_d, 1, sel), torch.gather(cat_i, 1, sel)
        del bt, d
    return best_d, best_i


def recall_of(got, gt, tk):
    return float(np.mean([len(set(got[i, :tk].tolist()) & set(gt[i].tolist())) / tk for i in range(got.shape[0])]))


class Pipeline:
    """One index + query batches (pinned host copies and resident device copies) + the step of the timed region."""

    def __init__(self, args, torch, dev, rank, world, sharded, clustered, n, label):
        from minivectordb_amd import HipBridge, derive_search_params, resolve_preset
        from minivectordb_amd.distributed import block_bytes
        from minivectordb_amd.pipeline import ResidentSearcher
        self.args, self.torch, self.dev, self.rank, self.world, self.sharded = args, torch, dev, rank, world, sharded
        self.preset = resolve_preset(args.preset)
        self.tk, self.ef_search, self.ann_k = derive_search_params(args.topk, self.preset)
        self.n, self.dim, self.nq, self.nb = n, args.dim, args.nq, max(1, args.batches)
        self.clustered, self.label = clustered, label
        self.id_offset = rank * n if sharded else 0
        p = self.preset
        t0 = time.time()
        rows = make_rows(n, self.dim, 12345 + (rank if world > 1 else 0) + (777 if clustered else 0), clustered)
        self.bridge = HipBridge(device=dev.index)
        self.bridge.init(dict(dim=self.dim, m=p["m"], ef=p["ef_construction"], efSearch=p["baseEfSearch"], capacity=n,
                              seed=42, resultsCap=p["resultsCap"]))
        self.base, self.base8 = normalise_quantise(self.bridge, rows)
        log(f"rank {rank} [{label}]: {n}x{self.dim} rows generated, normalised + quantised in {time.time() - t0:.1f}s")
        t0 = time.time()
        if args.builder == "gpu":
            self.build_stats = run_with_heartbeat(lambda: self.bridge.buildGpu(self.base8), f"rank {rank}: GPU build")
            self.graph = ("engine GPU bulk builder (reference insert algorithm, block-parallel; not the "
                          "reference's sequential edge order)")
        else:
            run_with_heartbeat(lambda: self.bridge.buildParallel(self.base8, args.build_threads),
                               f"rank {rank}: host build ({args.build_threads} threads)")
            self.build_stats = None
            self.graph = "engine multi-threaded host builder (reference insert algorithm, unordered)"
        self.build_s = time.time() - t0
        log(f"rank {rank} [{label}]: built {self.bridge.elementCount()} nodes in {self.build_s:.1f}s")
        self.bridge.attachF32(self.base, resident=not args.host_f32)
        self.bridge.setTuning(traverse_waves=args.traverse_waves, rerank_waves=2 if args.in_flight >= 2 else 0)
        if args.no_relabel:
            self.bridge.setRelabel(0)
        # queries: replicated in sharded mode, per-rank otherwise
        q_seed = 54321 + (rank if (world > 1 and not sharded) else 0)
        self.q_host = make_queries(self.nb, self.nq, self.dim, q_seed, self.base, clustered)
        self.q_dev = torch.from_numpy(self.q_host).to(dev)
        self.searcher = s = ResidentSearcher(self.bridge, self.nq, self.tk, dev)
        # the host side of a step ("H2D/D2H of queries/results included"): pinned query batches,
        # one device staging buffer and one pinned result block per batch in flight
        self.D = D = max(args.in_flight, 1)
        if D >= 2:
            s._pipe_init(self.nq, self.dim, int(self.ann_k), D)
        self.q_pin = torch.from_numpy(self.q_host).pin_memory()
        self.stage = [torch.empty((self.nq, self.dim), dtype=torch.float32, device=dev) for _ in range(D)]
        self.out_pin = [torch.empty(block_bytes(self.nq, self.tk), dtype=torch.uint8).pin_memory() for _ in range(D)]
        self.out_i = 0
        self.step_i = 0

    # -- reference pass: per-batch results (single stream) and work counters
    def reference_pass(self):
        torch = self.torch
        self.ref, self.ctr = [], []
        for i in range(self.nb):
            ids, l2, n = self.searcher.query_many(self.q_dev[i], self.ef_search, self.ann_k)
            torch.cuda.synchronize(self.dev)
            self.ref.append((ids.clone(), l2.clone(), n.clone()))
            self.ctr.append(self.searcher.ctr[:self.nq].cpu().numpy().astype(np.int64))
        self.searcher.check()
        p = self.preset
        m, m0 = p["m"], 2 * p["m"]
        # B_ann = n_dist*DIM + n_L0lists*(4+4*M_MAX0) + n_upperlists*(4+4*M) + DIM per query
        self.b_ann = [int((c[:, 0] * self.dim + c[:, 1] * (4 + 4 * m0) + c[:, 2] * (4 + 4 * m) + self.dim).sum())
                      for c in self.ctr]
        self.b_rerank = [int((c[:, 3] * self.dim * 4 + self.dim * 4 + self.tk * 8).sum()) for c in self.ctr]

    def step(self, exchange=True, host_io=True):
        """One batch.  host_io: the queries start in pinned host memory and the results end there (both copies on
        the batch's stream); otherwise the resident device copy of the batch is used and results stay in HBM."""
        a, s, torch = self.args, self.searcher, self.torch
        i = self.step_i % self.nb
        self.step_i += 1
        shard_io = self.sharded and exchange
        if self.D >= 2:
            slot = s._pipe_i % self.D
            S = s._b_streams[slot]
            q = self.q_dev[i]
            if host_io:
                q = self.stage[slot]
                with torch.cuda.stream(S):   # stream order: the batch that used this staging buffer is D steps back on S
                    q.copy_(self.q_pin[i], non_blocking=True)
            sl, *_ = s.submit_interleaved(q, self.ef_search, self.ann_k, sync_with_current=False, depth=self.D)
            assert sl == slot
            if not shard_io:
                if host_io:
                    with torch.cuda.stream(S):
                        self.out_pin[slot].copy_(s.slot_block(slot).buf, non_blocking=True)
                return slot, i
            s.wait(slot)
            block = s.slot_block(slot)
        else:
            q = self.q_dev[i]
            if host_io:
                q = self.stage[0]
                q.copy_(self.q_pin[i], non_blocking=True)
            s.query_many(q, self.ef_search, self.ann_k)
            if not shard_io:
                if host_io:
                    self.out_pin[0].copy_(s.out_block.buf, non_blocking=True)
                return -1, i
            slot, block = -1, s.out_block
        from minivectordb_amd.distributed import sharded_exchange
        self.merged = sharded_exchange(block, self.id_offset, lambda full: s.merge_blocks(full, self.nq))
        if host_io:
            self.out_pin[self.out_i % self.D].copy_(self.merged.buf, non_blocking=True)
            self.out_i += 1
        if slot >= 0:
            s.release(slot)  # the all-gather above is the last reader of this slot's block
        return slot, i

    def _mark(self):
        """An event on every stream a step may use: the latest of them = 'everything submitted so far is done'."""
        torch = self.torch
        evs = []
        streams = [torch.cuda.current_stream(self.dev)] + (list(self.searcher._b_streams) if self.D >= 2 else [])
        for st in streams:
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            evs.append(e)
        return evs

    def timed(self, steps, exchange=True, host_io=True, window=0):
        """barrier + sync | steps | drain + sync + barrier -> seconds (max over ranks).  window > 0: also the
        durations of consecutive windows of `window` steps, from stream events (no drain between windows)."""
        import torch.distributed as dist
        torch, dev = self.torch, self.dev
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        marks = [self._mark()] if window > 0 else []
        t0 = time.perf_counter()
        last = []
        for j in range(steps):
            last.append(self.step(exchange, host_io))
            if window > 0 and (j + 1) % window == 0:
                marks.append(self._mark())
        if self.D >= 2:
            self.searcher.drain()
        torch.cuda.synchronize(dev)
        if self.world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if self.world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        self.last_steps = last
        self.window_ms = []
        if window > 0:
            base = marks[0][0]
            t = [max(base.elapsed_time(e) for e in m) for m in marks]
            self.window_ms = [t[k + 1] - t[k] for k in range(len(t) - 1)]
        return el

    def check_last_steps(self, host_io=True):
        """Overlapped batches (and their copies in pinned host memory) must be the bits of the one-at-a-time
        reference pass."""
        torch, s = self.torch, self.searcher
        if self.sharded:
            return
        from minivectordb_amd.distributed import TopkBlock
        seen = set()
        for slot, i in reversed(self.last_steps):
            if slot in seen:
                break
            seen.add(slot)
            blk = s.slot_block(slot) if slot >= 0 else s.out_block
            r_ids, r_l2, r_n = self.ref[i]
            assert torch.equal(blk.ids[:self.nq], r_ids) and torch.equal(blk.l2[:self.nq], r_l2) and \
                torch.equal(blk.n[:self.nq], r_n), "overlapped and one-at-a-time results differ"
            if host_io:
                hb = TopkBlock(self.nq, self.tk, buf=self.out_pin[max(slot, 0)])
                assert torch.equal(hb.ids, r_ids.cpu()) and torch.equal(hb.l2, r_l2.cpu()) and \
                    torch.equal(hb.n, r_n.cpu()), "results copied to the host differ from the reference pass"
            if self.D < 2:
                break

    def host_slots(self, steps):
        """The path a host without device pointers uses (Node N-API, plain C): mvdb_query_many_submit / _wait on
        pageable host buffers, D slots in flight.  Results of the first round are checked against the device path."""
        b, D = self.bridge, max(min(self.args.in_flight, 4), 1)
        pending = []

        def run(n, keep=None):
            for j in range(n):
                if len(pending) == D:
                    jj, slot = pending.pop(0)
                    r = b.waitMany(slot)
                    if keep is not None:
                        keep[jj] = r
                b.submitMany(j % D, self.q_host[j % self.nb], self.tk, self.ef_search, self.ann_k)
                pending.append((j, j % D))
            while pending:
                jj, slot = pending.pop(0)
                r = b.waitMany(slot)
                if keep is not None:
                    keep[jj] = r
        first = {}
        run(min(3, self.nb), first)
        ids, l2, n = first[0]
        r_ids, r_l2, r_n = (x.cpu().numpy() for x in self.ref[0])
        same = bool(np.array_equal(n, r_n) and all(
            np.array_equal(ids[q, :n[q]], r_ids[q, :n[q]]) and
            np.array_equal(l2[q, :n[q]].view(np.uint64), np.ascontiguousarray(r_l2[q, :n[q]]).view(np.uint64)) for q in range(self.nq)))
        t0 = time.perf_counter()
        run(steps)
        return time.perf_counter() - t0, same

    def recall(self):
        """recall@10 of batch 0 against the exact float32 top-10 (over ALL shards when sharded)."""
        from minivectordb_amd.distributed import TopkBlock, gather_blocks, merge_blocks_host
        torch, dev = self.torch, self.dev
        qn, _ = self.bridge.prepareQueries(self.q_host[0])
        gt_d, gt_i = exact_topk(torch, dev, self.base, qn, self.tk, self.id_offset)
        if self.sharded:
            blk = TopkBlock(self.nq, self.tk, device=dev).fill(gt_i.to(torch.int32), gt_d.to(torch.float64),
                                                               torch.full((self.nq,), self.tk, dtype=torch.int32))
            gt_i = torch.from_numpy(merge_blocks_host(gather_blocks(blk.buf), self.nq, self.tk)[0])
            self.step_i = 0
            self.step(exchange=True, host_io=False)
            torch.cuda.synchronize(dev)
            got = self.merged.ids.cpu().numpy()
        else:
            got = self.ref[0][0].cpu().numpy()
        return recall_of(got, gt_i.cpu().numpy(), self.tk)

    def close(self):
        self.bridge.close()


def cpu_baseline_leg(pl, args, sample):
    """The C oracle (restatement of the WASM path) on the first queries of batch 0, same dump."""
    from oracle import oracle as O
    p = pl.preset
    ns = min(sample, pl.nq)
    ix = O.OracleIndex(pl.dim, p["m"], p["ef_construction"], 8, seed=1)
    t0 = time.time()
    assert ix.load_bytes(pl.bridge.saveBytes()), "oracle rejected the engine's dump"
    log(f"oracle loaded the dump in {time.time() - t0:.1f}s; timing {ns} queries on 1 host thread")
    q = pl.q_host[0][:ns]
    t0 = time.perf_counter()
    o_ids, o_l2, o_n, ann_ids, ann_cnt, octr = O.search_many_db(ix, q, pl.base, args.topk, args.preset, threads=1)
    cpu_s = time.perf_counter() - t0
    g_ids, g_l2, g_n = (t[:ns].cpu().numpy() for t in pl.ref[0])
    same = bool(np.array_equal(o_ids, g_ids) and np.array_equal(o_n, g_n) and
                np.array_equal(o_l2.view(np.uint64), g_l2.view(np.uint64)) and
                np.array_equal(octr["n_dist"], pl.ctr[0][:ns, 0]))
    # all-cores figure: the oracle's threads each take a contiguous slice of the queries (independent searches on the
    # shared read-only dump); timed on enough queries to give every thread a few (not the reference's mode: its
    # searchMany is one sequential loop)
    nthr = max(1, min(args.build_threads, os.cpu_count() or 1))
    nmt = min(pl.nq, max(ns, 8 * nthr))
    t0 = time.perf_counter()
    O.search_many_db(ix, pl.q_host[0][:nmt], pl.base, args.topk, args.preset, threads=nthr)
    cpu_mt_s = time.perf_counter() - t0
    ix.close()
    return dict(value=round(ns / cpu_s, 2), unit="queries/s", cores=1, kind="port",
                sample=f"first {ns} of the {pl.nq} queries of batch 0, full searchMany pipeline on the same "
                       f"{pl.n}-node dump (C restatement of the WASM path; 1 thread = the reference's sequential loop)",
                all_cores=dict(value=round(nmt / cpu_mt_s, 2), cores=nthr, queries=nmt,
                               what="the same oracle, queries split over host threads (not a mode the reference has)"),
                gpu_matches_oracle_on_sample=same)


def traffic_record(pl, kernel):
    """HBM bytes per traversal launch from the committed PMC passes of THIS workload (profiles/), if any."""
    name = f"pmc_traffic_{pl.n // 1_000_000}m_{pl.args.preset}{'_clustered' if pl.clustered else ''}.json"
    for rnd in ("r3", "r2", "r1"):
        f = os.path.join(ROOT, "profiles", rnd, name)
        if pl.n % 1_000_000 == 0 and pl.dim == 384 and pl.nq == 1000 and os.path.exists(f):
            with open(f) as fh:
                doc = json.load(fh)
            rec = doc.get("traverse", {})
            if rec.get("kernel") == kernel and "corrected_bytes_per_launch" in rec:
                return int(rec["corrected_bytes_per_launch"]), f"profiles/{rnd}/{name} (separate rocprofv3 --pmc passes of this workload)"
    return None, None


def measure(pl, args, steps, warmup, sustain_s, solo):
    """Warm up, time `steps` host-to-host steps, and price the traversal launches of that region."""
    torch, dev, s, bridge = pl.torch, pl.dev, pl.searcher, pl.bridge
    world, sharded = pl.world, pl.sharded
    for _ in range(warmup):
        pl.step()
    if pl.D >= 2:
        s.drain()
    torch.cuda.synchronize(dev)
    bridge._L.mvdb_enable_timing(bridge.handle, 1)
    for st in (0, 1, 2, 4):
        s.stage_stats(st)
    pl.step_i = 0
    elapsed = pl.timed(steps)
    bridge._L.mvdb_enable_timing(bridge.handle, 0)
    s.check()
    pl.check_last_steps()
    per_step_queries = pl.nq * (world if (world > 1 and not sharded) else 1)
    res = dict(value=round(per_step_queries / (elapsed / steps), 2), ms_per_step=round(elapsed / steps * 1e3, 4))

    # ---- roofline of the dominant kernel (traversal): live HIP-event timing of the timed launches only
    n_launch, trav_ms_total = s.stage_stats(1)
    _, rerank_ms_total = s.stage_stats(2)
    _, prep_ms_total = s.stage_stats(0)
    n_replay, replay_ms_total = s.stage_stats(4)
    bytes_timed = sum(pl.b_ann[j % pl.nb] for j in range(steps))
    rerank_bytes_timed = sum(pl.b_rerank[j % pl.nb] for j in range(steps))
    trav_ms = trav_ms_total / max(n_launch, 1)
    bytes_per_launch = bytes_timed / max(steps, 1)
    achieved = bytes_per_launch / (trav_ms * 1e-3) / 1e9 if trav_ms > 0 else 0.0
    m0 = 2 * pl.preset["m"]
    kernel = "traverse_kernel" if (m0 > 64 or args.traverse_waves == 1) else "traverse_lean_kernel"
    traffic, traffic_source = traffic_record(pl, kernel)
    roofline = dict(bound="hbm", kernel=kernel, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_source,
                    algorithmic_bytes_per_launch=int(bytes_per_launch), avg_launch_ms=round(trav_ms, 4),
                    launches_timed=int(n_launch),
                    replay_avg_launch_ms=round(replay_ms_total / max(n_replay, 1), 4),
                    rerank_bytes_per_launch=int(rerank_bytes_timed / max(steps, 1)),
                    rerank_avg_launch_ms=round(rerank_ms_total / max(n_launch, 1), 4),
                    prep_avg_launch_ms=round(prep_ms_total / max(n_launch, 1), 4),
                    # launches overlap when several batches are in flight: `achieved` (per launch, by the
                    # definition) then understates what the chip delivers; bytes of all timed launches / wall:
                    effective_gbs=round(bytes_timed / elapsed / 1e9, 2),
                    effective_frac=round(bytes_timed / elapsed / 1e9 / HBM_PEAK_GBS, 5),
                    pipeline_gbs=round((bytes_timed + rerank_bytes_timed) / elapsed / 1e9, 2))
    if solo and not sharded:
        # the same kernel with one batch at a time: what one launch achieves when it has the chip to itself
        bridge._L.mvdb_enable_timing(bridge.handle, 1)
        s.stage_stats(1)
        k_solo = min(max(steps, 8), 2 * pl.nb)
        for j in range(k_solo):
            s.query_many(pl.q_dev[j % pl.nb], pl.ef_search, pl.ann_k)
        torch.cuda.synchronize(dev)
        n_solo, ms_solo = s.stage_stats(1)
        bridge._L.mvdb_enable_timing(bridge.handle, 0)
        if n_solo > 0 and ms_solo > 0:
            solo_ms = ms_solo / n_solo
            solo_bytes = sum(pl.b_ann[j % pl.nb] for j in range(k_solo)) / k_solo
            roofline["solo_avg_launch_ms"] = round(solo_ms, 4)
            roofline["solo_frac"] = round(solo_bytes / (solo_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    res["roofline"] = roofline
    # the same steps with queries and results resident in HBM (no copies in the step)
    el = pl.timed(steps, host_io=False)
    pl.check_last_steps(host_io=False)
    res["resident"] = dict(value=round(per_step_queries / (el / steps), 2), unit="queries/s",
                           ms_per_step=round(el / steps * 1e3, 4),
                           what="the same steps on query batches already in HBM, results left in HBM")
    if sustain_s > 0:
        # >= 10 windows of >= 20 steps; window durations from stream events, no drain between windows
        nw = 12
        w = max(20, int(math.ceil(sustain_s / max(elapsed / steps, 1e-6) / nw)))
        bridge._L.mvdb_enable_timing(bridge.handle, 1)
        s.stage_stats(1)
        pl.step_i = 0
        el = pl.timed(nw * w, window=w)
        bridge._L.mvdb_enable_timing(bridge.handle, 0)
        n_sus, ms_sus = s.stage_stats(1)
        if n_sus > 0 and ms_sus > 0:   # the same per-launch figure over many more launches than the K timed steps
            sus_bytes = sum(pl.b_ann[j % pl.nb] for j in range(nw * w)) / (nw * w)
            roofline["sustained_avg_launch_ms"] = round(ms_sus / n_sus, 4)
            roofline["sustained_frac"] = round(sus_bytes / (ms_sus / n_sus * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            roofline["sustained_launches"] = int(n_sus)
        if traffic:
            # what the memory side moves per second with the batches overlapping (PMC bytes of a traversal launch + the
            # rerank's rows, x batches per second), beside what it delivers on random 384-byte rows and nothing else
            roofline["chip_traffic_gbs_sustained"] = round((traffic + roofline["rerank_bytes_per_launch"]) / (el / (nw * w)) / 1e9, 1)
            roofline["random_row_rate_gbs"] = RANDOM_ROW_GBS
            roofline["chip_traffic_over_random_row_rate"] = round(roofline["chip_traffic_gbs_sustained"] / RANDOM_ROW_GBS, 4)
        wm = sorted(x / w for x in pl.window_ms)
        res["sustained"] = dict(value=round(per_step_queries / (el / (nw * w)), 2), unit="queries/s", steps=nw * w,
                                seconds=round(el, 3), windows=len(wm), steps_per_window=w,
                                median_window_ms_per_step=round(wm[len(wm) // 2], 4),
                                min_window_ms_per_step=round(wm[0], 4), max_window_ms_per_step=round(wm[-1], 4))
        s.check()
    return res


def run_hnsw(args, torch, dist, dev, rank, world):
    t_start = time.time()
    sharded = world > 1 and args.mode != "replicas"
    legs = set() if args.legs == "none" else set(args.legs.split(","))
    if "auto" in legs:
        legs = {"clustered", "scan", "10m"} if (world == 1 and args.n == 1_000_000 and args.dim == 384) else \
            ({"clustered"} if (world == 1 and args.n <= 2_000_000) else set())
    pl = Pipeline(args, torch, dev, rank, world, sharded, False, args.n, "iid")
    bridge, s = pl.bridge, pl.searcher
    build_per_rank = [round(pl.build_s, 2)]
    if world > 1:
        bt = torch.tensor([pl.build_s], dtype=torch.float64, device=dev)
        allb = [torch.zeros_like(bt) for _ in range(world)]
        dist.all_gather(allb, bt)
        build_per_rank = [round(float(x.item()), 2) for x in allb]
        log(f"rank {rank}: index build seconds per rank {build_per_rank}")
    log(f"rank {rank}: reference pass over {pl.nb} distinct batches (first call uploads the index)")
    pl.reference_pass()
    res = measure(pl, args, args.steps, args.warmup, args.sustain_s, solo=True)
    roofline = res["roofline"]

    extras = {"resident": res["resident"]}
    if "sustained" in res:
        extras["sustained"] = res["sustained"]
    if world > 1 and sharded:  # the same ranks with no exchange: what "replicas" scaling would show
        el = pl.timed(args.steps, exchange=False)
        extras["replicas"] = dict(value=round(world * pl.nq / (el / args.steps), 2), unit="queries/s",
                                  mode="every rank answers its own copy of the batch on its own index, no exchange "
                                       "(embarrassingly parallel; says nothing about the sharded path)")
    if world == 1 and not args.no_pcie:
        el, same = pl.host_slots(args.steps)
        extras["host_slots"] = dict(value=round(pl.nq / (el / args.steps), 2), unit="queries/s",
                                    ms_per_step=round(el / args.steps * 1e3, 4), same_as_device_path=same,
                                    what="mvdb_query_many_submit/_wait on pageable host buffers (the Node / plain C "
                                         "host's way to keep batches in flight), copies inside")
    recall = None if args.no_recall else pl.recall()
    if world > 1 and rank != 0:
        recall = None
    cpu_baseline = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        cpu_baseline = cpu_baseline_leg(pl, args, args.cpu_sample if pl.n <= 2_000_000 else min(args.cpu_sample, 64))

    ctr_all = np.concatenate(pl.ctr, 0)
    out = None
    if rank == 0:
        p = pl.preset
        sus = res.get("sustained")
        out = dict(metric="searchMany QPS @ recall@10 (topK=10, 384-d int8+f32 rerank)",
                   value=res["value"], unit="queries/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=res["ms_per_step"], higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="i8", dtype_detail="int8 rows, exact i32 distances (traversal); f32 rows, f64 accumulate (rerank)",
                   data="synthetic",
                   value_includes="H2D of each query batch from pinned host memory and D2H of its result block "
                                  "(SURVEY.md 8d); `resident` = the same steps without the copies",
                   config=dict(workload=f"{pl.n}x{pl.dim} int8 HNSW (preset {args.preset}: M={p['m']}, efc="
                                        f"{p['ef_construction']}, ef=annK={pl.ann_k}) + f32 rerank, batch {pl.nq} "
                                        f"queries, topK={pl.tk}" + (f", {world} id-range shards of {pl.n}" if sharded else ""),
                               n=pl.n, dim=pl.dim, nq=pl.nq, topk=pl.tk, ann_k=pl.ann_k, preset=args.preset,
                               mode=("single" if world == 1 else ("sharded" if sharded else "replicas")),
                               total_vectors=pl.n * (world if sharded else 1),
                               batches=(f"{args.in_flight} in flight: batch i on stream i % {args.in_flight}"
                                        if args.in_flight >= 2 else "one at a time"),
                               distinct_query_batches=pl.nb, f32_store=("pinned host" if args.host_f32 else "HBM"),
                               graph=pl.graph, build_seconds=round(pl.build_s, 2), build_seconds_per_rank=build_per_rank,
                               build=pl.build_stats,
                               exchange=("one packed all-gather of per-shard top-k blocks per batch + (l2,id) merge kernel"
                                         if sharded else None)),
                   median_ms_per_step=None if sus is None else sus["median_window_ms_per_step"],
                   median_of=None if sus is None else f"{sus['windows']} consecutive windows of {sus['steps_per_window']} steps of the sustained loop",
                   recall_at_10=None if recall is None else round(recall, 4),
                   roofline=roofline, cpu_baseline=cpu_baseline,
                   counters=dict(mean_n_dist=float(ctr_all[:, 0].mean()), mean_n_l0lists=float(ctr_all[:, 1].mean()),
                                 mean_n_upperlists=float(ctr_all[:, 2].mean())),
                   lean_fallbacks=bridge.leanFallbacks(), lean_replays=bridge.leanReplays(),
                   relabelled=bridge.relabelled())
        out.update(extras)
    pl.close()
    del pl, s, bridge
    torch.cuda.empty_cache()

    # ---- second synthetic set with neighbourhood structure (the i.i.d. line above stays the headline)
    if "clustered" in legs and world == 1:
        p2 = Pipeline(args, torch, dev, rank, world, False, True, args.n, "clustered")
        p2.reference_pass()
        r2 = measure(p2, args, args.steps, args.warmup, min(args.sustain_s, 1.0), solo=False)
        c2 = np.concatenate(p2.ctr, 0)
        rf2 = r2["roofline"]
        rec2 = None if args.no_recall else round(p2.recall(), 4)
        out["second_dataset"] = dict(
            data=f"{CLUSTERS} Gaussian clusters (centres N(0,1), sigma {CLUSTER_SIGMA}), queries = stored points + noise",
            value=r2["value"], unit="queries/s", ms_per_step=r2["ms_per_step"], resident=r2["resident"]["value"],
            sustained=r2.get("sustained", {}).get("value"), recall_at_10=rec2, build_seconds=round(p2.build_s, 2),
            mean_n_dist=float(c2[:, 0].mean()), mean_n_l0lists=float(c2[:, 1].mean()),
            frac=rf2["frac"], avg_launch_ms=rf2["avg_launch_ms"], effective_gbs=rf2["effective_gbs"],
            effective_frac=rf2["effective_frac"], algorithmic_bytes_per_launch=rf2["algorithmic_bytes_per_launch"],
            traffic=rf2.get("traffic"), traffic_source=rf2.get("traffic_source"), relabelled=p2.bridge.relabelled(),
            chip_traffic_gbs_sustained=rf2.get("chip_traffic_gbs_sustained"))
        # the same per-launch definition as `frac`, on the set where the index has neighbourhoods to find
        out["roofline"]["frac_structured"] = rf2["frac"]
        out["roofline"]["structured"] = dict(value=r2["value"], unit="queries/s", recall_at_10=rec2,
                                             effective_frac=rf2["effective_frac"], avg_launch_ms=rf2["avg_launch_ms"])
        p2.close()
        del p2
        torch.cuda.empty_cache()

    # ---- BASELINE configs[2] on the same line: exact int8 scan, 1 M x 512, 4096 queries
    if "scan" in legs and world == 1:
        import copy
        a2 = copy.copy(args)
        a2.n, a2.dim, a2.nq, a2.second_dataset, a2.cpu_sample = 1_000_000, 512, 4096, "none", min(args.cpu_sample, 16)
        sc = run_scan(a2, torch, dist, dev, rank, world)
        out["scan_1m_512"] = dict(metric=sc["metric"], value=sc["value"], unit=sc["unit"], ms_per_step=sc["ms_per_step"],
                                  steps=sc["steps"], workload=sc["config"]["workload"], roofline=sc["roofline"],
                                  hnsw_recall_vs_exact_int8=sc["hnsw_recall_vs_exact_int8"], cpu_baseline=sc["cpu_baseline"])
        torch.cuda.empty_cache()

    # ---- the north-star size: 10 M x 384 balanced on one GPU
    if "10m" in legs and world == 1:
        spent = time.time() - t_start
        if spent > args.leg_deadline_s:
            out["north_star_10m"] = dict(skipped=f"{spent:.0f}s already spent (limit {args.leg_deadline_s}s before this leg)")
        else:
            import copy
            a3 = copy.copy(args)
            a3.n = 10_000_000
            p3 = Pipeline(a3, torch, dev, rank, world, False, False, a3.n, "10m")
            p3.reference_pass()
            r3 = measure(p3, a3, args.steps, args.warmup, min(args.sustain_s, 1.0), solo=True)
            rec3 = None if args.no_recall else round(p3.recall(), 4)
            cpu3 = cpu_baseline_leg(p3, a3, min(args.cpu_sample, 64)) if args.cpu_sample > 0 else None
            c3 = np.concatenate(p3.ctr, 0)
            out["north_star_10m"] = dict(
                workload=f"{p3.n}x{p3.dim} int8 HNSW (preset {args.preset}, ef=annK={p3.ann_k}) + f32 rerank, batch "
                         f"{p3.nq} queries, topK={p3.tk}",
                value=r3["value"], unit="queries/s", ms_per_step=r3["ms_per_step"], steps=args.steps,
                resident=r3["resident"]["value"], sustained=r3.get("sustained"), recall_at_10=rec3,
                roofline=r3["roofline"], cpu_baseline=cpu3, build_seconds=round(p3.build_s, 2), build=p3.build_stats,
                mean_n_dist=float(c3[:, 0].mean()),
                lean_fallbacks=p3.bridge.leanFallbacks(), lean_replays=p3.bridge.leanReplays())
            p3.close()
            del p3
            torch.cuda.empty_cache()
    if out is not None:
        out["bench_seconds"] = round(time.time() - t_start, 1)
    return out


def run_scan(args, torch, dist, dev, rank, world):
    """BASELINE configs[2]: exact int8 brute-force scan on the matrix cores + the HNSW path's recall against it."""
    from minivectordb_amd import HipBridge, resolve_preset
    n, dim, nq, k = args.n, args.dim, args.nq, args.topk
    p = resolve_preset(args.preset)
    b = HipBridge(device=dev.index)
    b.init(dict(dim=dim, m=p["m"], ef=p["ef_construction"], efSearch=p["baseEfSearch"], capacity=n, seed=42,
                resultsCap=4096))
    base, b8 = normalise_quantise(b, make_rows(n, dim, 12345 + rank, False))
    st = run_with_heartbeat(lambda: b.buildGpu(b8), "GPU build")
    q8 = normalise_quantise(b, np.random.default_rng(54321 + rank).standard_normal((nq, dim), dtype=np.float32))[1]
    qd = torch.from_numpy(q8).to(dev)
    # `--scan-in-flight` sweeps overlap, each on a stream of its own with its own outputs (the engine keeps the scan's
    # partial lists per stream): the small, latency-bound levels of one sweep run under the large levels of the other
    D = max(1, min(args.scan_in_flight, 4))
    outs = [(torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.int32, device=dev),
             torch.empty((nq,), dtype=torch.int32, device=dev)) for _ in range(D)]
    ids, val, cnt = outs[0]
    L, h, vp = b._L, b.handle, C.c_void_p
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(D - 1)]
    state = {"i": 0}

    def step():
        j = state["i"] % D
        state["i"] += 1
        o = outs[j]
        b._check(L.mvdb_scan_many_dev(h, vp(qd.data_ptr()), nq, k, 0, vp(o[0].data_ptr()), vp(o[1].data_ptr()),
                                      vp(o[2].data_ptr()), vp(streams[j].cuda_stream)))

    def drain():
        for st in streams[1:]:
            streams[0].wait_stream(st)
        torch.cuda.synchronize(dev)
    for _ in range(max(args.warmup, D)):
        step()
    drain()
    state["i"] = 0
    L.mvdb_enable_timing(h, 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    L.mvdb_enable_timing(h, 0)
    nl, tot = C.c_longlong(0), C.c_double(0.0)
    L.mvdb_stage_stats(h, 3, C.byref(nl), C.byref(tot), 1)
    scan_ms = tot.value / max(nl.value, 1)
    ops = 2.0 * nq * n * dim
    # the same sweeps one at a time: what one sweep achieves with the chip to itself
    solo_ms = None
    if D > 1:
        L.mvdb_enable_timing(h, 1)
        for _ in range(min(args.steps, 10)):
            b._check(L.mvdb_scan_many_dev(h, vp(qd.data_ptr()), nq, k, 0, vp(ids.data_ptr()), vp(val.data_ptr()),
                                          vp(cnt.data_ptr()), vp(streams[0].cuda_stream)))
        torch.cuda.synchronize(dev)
        L.mvdb_enable_timing(h, 0)
        L.mvdb_stage_stats(h, 3, C.byref(nl), C.byref(tot), 1)
        solo_ms = tot.value / max(nl.value, 1)
        for o in outs[1:]:   # overlapped sweeps return the bits of the sweep that ran alone
            assert torch.equal(o[0], ids) and torch.equal(o[1], val) and torch.equal(o[2], cnt), "overlapped scans differ"
    # HNSW (balanced traversal, int8 distances) recall@k against this exact int8 result
    b.setEfSearch(p["baseEfSearch"])
    hi, hd, hc = b.searchManyArrays(q8[:1000], k * p["rerankMultiplier"])
    exact = ids[:1000].cpu().numpy()
    hn_recall = recall_of(hi, exact, k)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import oracle as O
        ns = min(16, nq)
        ix = O.OracleIndex(dim, p["m"], p["ef_construction"], 8, seed=1)
        assert ix.load_bytes(b.saveBytes())
        nthr = max(1, min(args.build_threads, os.cpu_count() or 1))
        t1 = time.perf_counter()
        oi, ov, oc = ix.scan_many(q8[:ns], k, "l2", threads=nthr)
        cs = time.perf_counter() - t1
        same = bool(np.array_equal(oi, ids[:ns].cpu().numpy()) and np.array_equal(ov, val[:ns].cpu().numpy()))
        cpu = dict(value=round(ns / cs, 2), unit="queries/s", cores=nthr, kind="port",
                   sample=f"first {ns} queries, exhaustive int8 scan of the same {n} rows (C oracle)",
                   gpu_matches_oracle_on_sample=same)
        ix.close()
    second = None
    if rank == 0 and world == 1 and args.second_dataset != "none":
        # the same scan + the HNSW comparison on a set with neighbourhood structure (the i.i.d. set has none at 1 M)
        b.close()
        b = HipBridge(device=dev.index)
        b.init(dict(dim=dim, m=p["m"], ef=p["ef_construction"], efSearch=p["baseEfSearch"], capacity=n, seed=42,
                    resultsCap=4096))
        base2, c8 = normalise_quantise(b, make_rows(n, dim, 12345 + 777, True))
        run_with_heartbeat(lambda: b.buildGpu(c8), "GPU build (clustered)")
        cq = make_queries(1, nq, dim, 999, base2, True)[0]
        cq8 = normalise_quantise(b, cq)[1]
        qd.copy_(torch.from_numpy(cq8))
        L, h = b._L, b.handle
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        el2 = time.perf_counter() - t1
        b.setEfSearch(p["baseEfSearch"])
        hi2, _, _ = b.searchManyArrays(cq8[:1000], k * p["rerankMultiplier"])
        second = dict(data=f"{CLUSTERS} Gaussian clusters, sigma {CLUSTER_SIGMA}", ms_per_step=round(el2 / args.steps * 1e3, 4),
                      value=round(nq / (el2 / args.steps), 2), unit="queries/s",
                      hnsw_recall_vs_exact_int8=round(recall_of(hi2, ids[:1000].cpu().numpy(), k), 4))
    out = None
    if rank == 0:
        out = dict(metric="exact int8 scan QPS (top-10, int8 MFMA)", value=round(world * nq / (el / args.steps), 2),
                   unit="queries/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(el / args.steps * 1e3, 4), higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="i8", data="synthetic",
                   config=dict(workload=f"{n}x{dim} exact int8 brute-force scan, batch {nq} queries, top-{k}", n=n, dim=dim,
                               nq=nq, topk=k, mode="single" if world == 1 else "replicas", build=st),
                   roofline=dict(bound="mfma", kernel="whole sweep: scan_kernel (small levels) + scan_wide_kernel (levels >= 100k nodes) + merges", achieved=round(ops / (scan_ms * 1e-3) / 1e12, 2),
                                 peak=I8_MFMA_PEAK_TOPS, unit="TOP/s (int8 multiply-add = 2 ops)",
                                 frac=round(ops / (scan_ms * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS, 5), traffic=None,
                                 ops_per_launch=ops, avg_launch_ms=round(scan_ms, 4), sweeps_in_flight=D,
                                 # sweeps overlap when several are in flight: ops of all timed sweeps / wall time
                                 effective_tops=round(ops * args.steps / el / 1e12, 2),
                                 effective_frac=round(ops * args.steps / el / 1e12 / I8_MFMA_PEAK_TOPS, 5),
                                 solo_avg_launch_ms=None if solo_ms is None else round(solo_ms, 4),
                                 solo_frac=None if solo_ms is None else round(ops / (solo_ms * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS, 5)),
                   hnsw_recall_vs_exact_int8=dict(value=round(hn_recall, 4), queries=1000,
                                                  what=f"traversal candidates (ef={k * p['rerankMultiplier']}) top-{k} by "
                                                       f"int8 distance vs the exact scan's top-{k}"),
                   second_dataset=second, cpu_baseline=cpu)
    b.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="hnsw", choices=["hnsw", "scan"])
    # (--rows: torchrun's own parser trips over "--n" in the script's arguments)
    ap.add_argument("--n", "--rows", dest="n", type=int, default=None, help="vectors per index (per shard when sharded); default 1M")
    ap.add_argument("--nq", type=int, default=None, help="queries per batch (default 1000; scan: 4096)")
    ap.add_argument("--dim", type=int, default=None, help="default 384 (scan: 512)")
    ap.add_argument("--preset", default="balanced")
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--mode", default="auto", choices=["auto", "replicas", "sharded"],
                    help="N>1: sharded (default, the north_star path) or replicas")
    ap.add_argument("--builder", default="gpu", choices=["gpu", "host"])
    ap.add_argument("--build-threads", type=int, default=int(os.environ.get("OMP_NUM_THREADS", "16")))
    ap.add_argument("--cpu-sample", type=int, default=256, help="queries timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--no-pcie", action="store_true")
    ap.add_argument("--batches", type=int, default=8, help="distinct query batches rotated through the steps")
    ap.add_argument("--sustain-s", type=float, default=2.0, help="seconds of the sustained leg (0 = skip)")
    ap.add_argument("--second-dataset", default="auto", choices=["auto", "none", "clustered"],
                    help="scan workload only: also run the scan on the clustered set")
    ap.add_argument("--legs", default="auto",
                    help="extra legs of the hnsw line: comma list of clustered,scan,10m | none | auto (= all three "
                         "for the default 1M x 384 single-GPU run, clustered for other sizes up to 2M)")
    ap.add_argument("--leg-deadline-s", type=float, default=330.0,
                    help="the 10m leg is skipped when the run has already taken this long")
    ap.add_argument("--host-f32", action="store_true", help="float32 store in pinned host memory (preloadVectors=false)")
    ap.add_argument("--in-flight", type=int, default=3, choices=[1, 2, 3, 4],  # (the engine keeps four per-stream scratch sets)
                    help="batches in flight: batch i runs on stream i %% IN_FLIGHT; 1 = one batch at a time")
    ap.add_argument("--scan-in-flight", type=int, default=1, choices=[1, 2, 3, 4],
                    help="scan workload: sweeps in flight, each on its own stream")
    ap.add_argument("--traverse-waves", type=int, default=0, choices=[0, 1],
                    help="0 = the engine's default (lean traversal + result-heap replay); 1 = the one-wave kernel with the result heap inside")
    ap.add_argument("--no-relabel", action="store_true",
                    help="walk the index under its own node ids (default: the traversal's copy in breadth-first label order, mvdb_set_relabel)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="wiring rehearsal of the N>1 path on a one-GPU box: every rank uses cuda:0 and the exchange "
                         "goes over gloo (RCCL refuses two ranks on one device); numbers from such a run mean nothing")
    args = ap.parse_args()
    scan = args.workload == "scan"
    args.n = args.n or 1_000_000
    args.nq = args.nq or (4096 if scan else 1000)
    args.dim = args.dim or (512 if scan else 384)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.rehearse_on_one_gpu else "nccl", rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if args.rehearse_on_one_gpu:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(90, exit=True)  # a rehearsal that hangs says where
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    out = (run_scan if scan else run_hnsw)(args, torch, dist, dev, rank, world)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
