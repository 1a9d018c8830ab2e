"""Multi-GPU layout of the retrieval path: one process per GPU.

* ``shard_range``: internal-id range partition, shard g of G holds ids [g*ceil(N/G), (g+1)*ceil(N/G)).
  Every shard is an independent HNSW sub-index with local ids; global id = local id + range begin.
* ``gather_topk``: the ONE exchange step of the sharded path -- an all-gather of each rank's
  per-query topK (global id i32, l2 f64, count i32).  nq*topK*12 bytes per rank (117 KB at 1k x 10):
  latency-bound on xGMI, so one collective per tensor per batch, never a ring of small sends.
  backend "nccl" (= RCCL) for device tensors, "gloo" for the CPU tests.
* merge: ``ResidentSearcher.merge_topk`` (HIP kernel) for device tensors; ``merge_topk_host`` is
  the same (l2 asc, id asc) merge for host tensors (used by the gloo tests and small tools).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_total: int, world: int, rank: int):
    per = -(-int(n_total) // int(world))
    b = min(rank * per, n_total)
    return b, min(b + per, n_total)


def gather_topk(ids: torch.Tensor, l2: torch.Tensor, n: torch.Tensor, group=None):
    """all-gather [nq, topk] results of every rank -> [world, nq, topk] (ids, l2) and [world, nq] (n)."""
    world = dist.get_world_size(group)
    outs = []
    for t in (ids, l2, n):
        t = t.contiguous()
        if t.is_cuda:
            full = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(full, t, group=group)
        else:
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t, group=group)
            full = torch.stack(parts, 0)
        outs.append(full)
    return tuple(outs)


def merge_topk_host(all_ids, all_l2, all_n, topk: int):
    """(l2 asc, id asc) merge of [parts, nq, topk] host arrays; same order as mini-vector-db.ts:637."""
    all_ids, all_l2, all_n = (np.asarray(x) for x in (all_ids, all_l2, all_n))
    parts, nq, k = all_ids.shape
    out_ids = np.full((nq, topk), -1, np.int32)
    out_l2 = np.zeros((nq, topk), np.float64)
    out_n = np.zeros(nq, np.int32)
    for q in range(nq):
        items = [(float(all_l2[p, q, j]), int(all_ids[p, q, j])) for p in range(parts) for j in range(int(all_n[p, q]))]
        items.sort()
        items = items[:topk]
        out_n[q] = len(items)
        for j, (l, i) in enumerate(items):
            out_ids[q, j], out_l2[q, j] = i, l
    return out_ids, out_l2, out_n
