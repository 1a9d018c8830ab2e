"""Multi-GPU layout of the retrieval path: one process per GPU.

* ``shard_range``: internal-id range partition, shard g of G holds ids [g*ceil(N/G), (g+1)*ceil(N/G)).
  Every shard is an independent HNSW sub-index with local ids; global id = local id + range begin.
* ``TopkBlock``: one rank's per-query topK as ONE contiguous block
  ``{l2 f64[nq][topk] | ids i32[nq][topk] | n i32[nq]}``, the whole padded to 8 B, (mvdb_topk_block_bytes).  The engine
  writes its outputs straight into views of the block, so there is nothing to pack.
* ``gather_blocks``: the ONE exchange step of the sharded path -- one all-gather of the blocks per batch
  (nq*topK*12 + nq*4 bytes per rank, 124 KB at 1k x 10: latency-bound on xGMI, so a single collective,
  never one per field and never a ring of small sends).  l2 travels as f64 on purpose: the merge order
  (l2 asc, id asc) must be the reference's (mini-vector-db.ts:637) bit for bit.
  backend "nccl" (= RCCL) for device tensors, "gloo" for the CPU tests.
* merge: ``ResidentSearcher.merge_blocks`` (HIP kernel over the gathered blocks) on the device;
  ``merge_topk_host`` / ``merge_blocks_host`` are the same (l2 asc, id asc) merge on the host (gloo tests).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_total: int, world: int, rank: int):
    per = -(-int(n_total) // int(world))
    b = min(rank * per, n_total)
    return b, min(b + per, n_total)


def block_bytes(nq: int, topk: int) -> int:
    """== mvdb_topk_block_bytes (include/mvdb_hip.h)."""
    return (nq * topk * 12 + nq * 4 + 7) & ~7


class TopkBlock:
    """A rank's per-query topK in the packed layout, with typed views (torch; host or device)."""

    def __init__(self, nq: int, topk: int, device=None, buf: torch.Tensor | None = None):
        self.nq, self.topk = int(nq), int(topk)
        nb = block_bytes(self.nq, self.topk)
        self.buf = torch.zeros(nb, dtype=torch.uint8, device=device) if buf is None else buf
        assert self.buf.numel() == nb and self.buf.dtype == torch.uint8
        e = self.nq * self.topk
        self.l2 = self.buf[:e * 8].view(torch.float64).view(self.nq, self.topk)
        self.ids = self.buf[e * 8:e * 12].view(torch.int32).view(self.nq, self.topk)
        self.n = self.buf[e * 12:e * 12 + self.nq * 4].view(torch.int32)

    def fill(self, ids, l2, n):
        self.ids.copy_(torch.as_tensor(ids))
        self.l2.copy_(torch.as_tensor(l2))
        self.n.copy_(torch.as_tensor(n))
        return self


def gather_blocks(block: torch.Tensor, group=None) -> torch.Tensor:
    """ONE all-gather: every rank's packed block -> [world, block_bytes] (uint8)."""
    world = dist.get_world_size(group)
    block = block.contiguous()
    full = torch.empty((world, block.numel()), dtype=torch.uint8, device=block.device)
    if block.is_cuda and dist.get_backend(group) == "gloo":
        # gloo is the CPU backend (tests, one-GPU rehearsals of the N>1 wiring): stage through the host
        host = torch.empty((world, block.numel()), dtype=torch.uint8)
        dist.all_gather(list(host.unbind(0)), block.cpu(), group=group)
        full.copy_(host)
    elif block.is_cuda:
        # RCCL; flat output = the ranks' blocks back to back
        dist.all_gather_into_tensor(full.view(-1), block.view(-1), group=group)
    else:
        dist.all_gather(list(full.unbind(0)), block, group=group)
    return full


def split_blocks(full: torch.Tensor, nq: int, topk: int):
    """[world, block_bytes] -> ([world, nq, topk] ids, [world, nq, topk] l2, [world, nq] n) host arrays."""
    blocks = [TopkBlock(nq, topk, buf=row) for row in full.cpu().unbind(0)]
    return (np.stack([b.ids.numpy() for b in blocks]), np.stack([b.l2.numpy() for b in blocks]),
            np.stack([b.n.numpy() for b in blocks]))


def gather_topk(ids: torch.Tensor, l2: torch.Tensor, n: torch.Tensor, group=None):
    """Convenience over gather_blocks for separate tensors: -> [world, nq, topk] ids, l2 and [world, nq] n,
    still ONE collective (the tensors are packed into a block first)."""
    nq, topk = ids.shape
    blk = TopkBlock(nq, topk, device=ids.device).fill(ids, l2, n)
    full = gather_blocks(blk.buf, group)
    world = full.shape[0]
    views = [TopkBlock(nq, topk, buf=full[w]) for w in range(world)]
    return (torch.stack([v.ids for v in views]), torch.stack([v.l2 for v in views]), torch.stack([v.n for v in views]))


def merge_topk_host(all_ids, all_l2, all_n, topk: int):
    """(l2 asc, id asc) merge of [parts, nq, topk] host arrays; same order as mini-vector-db.ts:637.
    A part with n = -1 for a query (aborted on that shard) makes the merged n -1, as the device merge does."""
    all_ids, all_l2, all_n = (np.asarray(x) for x in (all_ids, all_l2, all_n))
    parts, nq, k = all_ids.shape
    out_ids = np.full((nq, topk), -1, np.int32)
    out_l2 = np.zeros((nq, topk), np.float64)
    out_n = np.zeros(nq, np.int32)
    for q in range(nq):
        if (all_n[:, q] < 0).any():
            out_n[q] = -1
            continue
        items = [(float(all_l2[p, q, j]), int(all_ids[p, q, j])) for p in range(parts)
                 for j in range(min(int(all_n[p, q]), k))]
        items.sort()
        items = items[:topk]
        out_n[q] = len(items)
        for j, (l, i) in enumerate(items):
            out_ids[q, j], out_l2[q, j] = i, l
    return out_ids, out_l2, out_n


def merge_blocks_host(full: torch.Tensor, nq: int, topk: int):
    return merge_topk_host(*split_blocks(full, nq, topk), topk)


def sharded_exchange(block: TopkBlock, id_offset: int, merge, group=None):
    """What every rank of the sharded path does with its local result block (bench.py's sharded step and the
    gloo tests run exactly this): local ids -> global ids (the rank's id-range begin; rows beyond n stay
    untouched garbage and are never read), ONE all-gather of the blocks, then `merge(gathered [world, bytes])`
    -- ResidentSearcher.merge_blocks on the device, merge_blocks_host on the host."""
    if id_offset:
        block.ids.add_(int(id_offset))
    return merge(gather_blocks(block.buf, group))
