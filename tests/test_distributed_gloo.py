"""N>1 path on CPU: two gloo ranks, each owning an id-range shard.  The per-shard
search is done by the oracle here (no GPU in this suite); what is under test is the product's
sharding plan, the all-gather exchange and the (l2, id) merge, against a single-process merge of
the per-shard oracle outputs."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from minivectordb_amd.distributed import gather_topk, merge_topk_host, shard_range
from oracle import oracle as O

N, DIM, NQ, TOPK, PRESET = 1600, 32, 16, 10, "fast"


def _data():
    rng = np.random.default_rng(12345)
    base = O.normalize_rows_f32(rng.standard_normal((N, DIM), dtype=np.float32))
    q = np.random.default_rng(54321).standard_normal((NQ, DIM), dtype=np.float32)
    return base, q


def _shard_result(rank, world):
    base, q = _data()
    b, e = shard_range(N, world, rank)
    p = O.PRESETS[PRESET]
    ix = O.OracleIndex(DIM, p["m"], p["ef_construction"], e - b, seed=42)
    ix.insert_many(np.arange(e - b, dtype=np.int32), O.quantize_rows_i8(base[b:e]))
    ids, l2, n, *_ = O.search_many_db(ix, q, base[b:e], TOPK, PRESET)
    gids = np.where(ids >= 0, ids + b, ids).astype(np.int32)
    return gids, l2, n


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gids, l2, n = _shard_result(rank, world)
    all_ids, all_l2, all_n = gather_topk(torch.from_numpy(gids), torch.from_numpy(l2), torch.from_numpy(n))
    ids, ml2, mn = merge_topk_host(all_ids.numpy(), all_l2.numpy(), all_n.numpy(), TOPK)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), ids=ids, l2=ml2, n=mn)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n, w in ((10, 3), (1_000_000, 8), (7, 8), (1600, 2)):
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_two_rank_sharded_merge(tmp_path):
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [_shard_result(r, world) for r in range(world)]
    exp = merge_topk_host(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]),
                          np.stack([p[2] for p in parts]), TOPK)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert np.array_equal(z["ids"], exp[0]) and np.array_equal(z["n"], exp[2])
        assert np.array_equal(z["l2"].view(np.uint64), exp[1].view(np.uint64))
    # the merged result is the exact (l2, id)-ordered top-k of the union of the shards' lists
    base, q = _data()
    qn = O.normalize_rows_f32(q)
    for qi in range(NQ):
        ids = exp[0][qi, :exp[2][qi]]
        l2 = [O.l2sq_f32(qn[qi], base[i]) for i in ids]
        assert l2 == exp[1][qi, :exp[2][qi]].tolist()
        assert all((l2[j], ids[j]) < (l2[j + 1], ids[j + 1]) for j in range(len(ids) - 1))
