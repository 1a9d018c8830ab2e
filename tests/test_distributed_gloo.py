"""N>1 path on CPU: gloo ranks, each owning an id-range shard.  The per-shard search is done by
the oracle here (no GPU in this suite); what is under test is the product's sharding plan, the ONE packed
all-gather per batch, the global-id offsets and the (l2, id) merge -- `sharded_exchange`, the function
bench.py's sharded step runs -- against a single-process merge of the per-shard oracle outputs.
World sizes 2 and 4, uneven shards, ragged per-shard counts (a shard whose filter leaves fewer than topK)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from minivectordb_amd.distributed import (TopkBlock, block_bytes, gather_topk, merge_blocks_host, merge_topk_host,
                                          shard_range, sharded_exchange, split_blocks)
from oracle import oracle as O

N, DIM, NQ, TOPK, PRESET = 1603, 32, 16, 10, "fast"


def _data():
    rng = np.random.default_rng(12345)
    base = O.normalize_rows_f32(rng.standard_normal((N, DIM), dtype=np.float32))
    q = np.random.default_rng(54321).standard_normal((NQ, DIM), dtype=np.float32)
    return base, q


def _shard_result(rank, world):
    """Local (shard-relative) ids, as a rank's engine returns them."""
    base, q = _data()
    b, e = shard_range(N, world, rank)
    p = O.PRESETS[PRESET]
    ix = O.OracleIndex(DIM, p["m"], p["ef_construction"], e - b, seed=42)
    ix.insert_many(np.arange(e - b, dtype=np.int32), O.quantize_rows_i8(base[b:e]))
    allow = None
    if world == 4 and rank == 1:  # ragged: this shard's filter keeps 4 ids -> fewer than topK results per query
        allow = np.zeros((e - b + 31) // 32, np.uint32)
        for i in (3, 77, 130, 200):
            allow[i >> 5] |= np.uint32(1 << (i & 31))
    ids, l2, n, *_ = O.search_many_db(ix, q, base[b:e], TOPK, PRESET, allow_bitmap=allow)
    return ids.astype(np.int32), l2, n.astype(np.int32), b


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids, l2, n, begin = _shard_result(rank, world)
    blk = TopkBlock(NQ, TOPK).fill(ids, l2, n)
    gathered = {}

    def merge(full):
        gathered["full"] = full.clone()
        return merge_blocks_host(full, NQ, TOPK)

    mi, ml2, mn = sharded_exchange(blk, begin, merge)
    # the convenience wrapper is one collective too and agrees
    a_ids, a_l2, a_n = gather_topk(blk.ids, blk.l2, blk.n)
    assert np.array_equal(a_ids.numpy(), split_blocks(gathered["full"], NQ, TOPK)[0])
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), ids=mi, l2=ml2, n=mn, full=gathered["full"].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n, w in ((10, 3), (1_000_000, 8), (7, 8), (1600, 2), (1603, 4)):
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_block_layout_matches_the_c_abi():
    from minivectordb_amd import _native
    L = _native.lib()
    for nq, topk in ((1, 1), (5, 3), (16, 10), (1000, 10), (7, 1)):
        assert block_bytes(nq, topk) == L.mvdb_topk_block_bytes(nq, topk) and block_bytes(nq, topk) % 8 == 0
    blk = TopkBlock(5, 3).fill(np.arange(15).reshape(5, 3), np.arange(15).reshape(5, 3) / 4, np.arange(5))
    raw = blk.buf.numpy().tobytes()
    assert np.array_equal(np.frombuffer(raw, np.float64, 15), np.arange(15) / 4)
    assert np.array_equal(np.frombuffer(raw, np.int32, 15, 120), np.arange(15))
    assert np.array_equal(np.frombuffer(raw, np.int32, 5, 180), np.arange(5))


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_exchange_and_merge(tmp_path, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [_shard_result(r, world) for r in range(world)]
    sizes = [shard_range(N, world, r) for r in range(world)]
    assert len({e - b for b, e in sizes}) > 1, "shards are meant to be uneven here"
    gids = [np.where(np.arange(TOPK)[None, :] < p[2][:, None], p[0] + p[3], 0).astype(np.int32) for p in parts]
    exp = merge_topk_host(np.stack(gids), np.stack([p[1] for p in parts]), np.stack([p[2] for p in parts]), TOPK)
    if world == 4:
        assert (parts[1][2] <= 4).all() and (parts[1][2] < TOPK).all()      # the ragged shard
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert np.array_equal(z["n"], exp[2])
        for qi in range(NQ):
            k = exp[2][qi]
            assert np.array_equal(z["ids"][qi, :k], exp[0][qi, :k])
            assert np.array_equal(z["l2"][qi, :k].view(np.uint64), exp[1][qi, :k].view(np.uint64))
        assert z["full"].shape == (world, block_bytes(NQ, TOPK))            # ONE gathered buffer of packed blocks
    # the merged result is the exact (l2, id)-ordered top-k of the union of the shards' lists
    base, q = _data()
    qn = O.normalize_rows_f32(q)
    for qi in range(NQ):
        ids = exp[0][qi, :exp[2][qi]]
        l2 = [O.l2sq_f32(qn[qi], base[i]) for i in ids]
        assert l2 == exp[1][qi, :exp[2][qi]].tolist()
        assert all((l2[j], ids[j]) < (l2[j + 1], ids[j + 1]) for j in range(len(ids) - 1))


def test_aborted_shard_makes_the_merged_query_fail():
    ids = np.zeros((3, 4, 2), np.int32)
    l2 = np.zeros((3, 4, 2))
    n = np.full((3, 4), 2, np.int32)
    n[1, 2] = -1
    assert merge_topk_host(ids, l2, n, 2)[2].tolist() == [2, 2, -1, 2]
