"""The RCCL branch of the sharded path's one exchange step (minivectordb_amd/distributed.py: gather_blocks ->
all_gather_into_tensor on device tensors), executed for real: a world-size-1 "nccl" (= RCCL) process group on the
one GPU of the box.  sharded_exchange on a device TopkBlock -> ResidentSearcher.merge_blocks must give the local
result with the shard's id offset applied, and a query the shard aborted (n = -1) must stay -1 after the merge.
(The multi-rank wiring is covered by the gloo tests; an 8-GPU run is the driver's.)"""
import os
import socket

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_world1_sharded_exchange(gpu):
    import torch
    import torch.distributed as dist
    from minivectordb_amd.distributed import TopkBlock, merge_blocks_host, sharded_exchange
    from minivectordb_amd.pipeline import ResidentSearcher
    from tests._data import bridge_from_dump, built_index, oracle_from_dump
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        dump, base, b8 = built_index(4000, 384, "balanced")
        b = bridge_from_dump(dump, 384, "balanced")
        ix = oracle_from_dump(dump, 384, "balanced")
        b.attachF32(base)
        nq, tk, offset = 48, 10, 1_000_000
        q = np.random.default_rng(3).standard_normal((nq, 384)).astype(np.float32)
        oi, ol, on, _, _, _ = O.search_many_db(ix, q, base, tk, "balanced", threads=8)
        s = ResidentSearcher(b, nq, tk, dev)
        q_dev = torch.from_numpy(q).to(dev)
        for inject in (-1, 5):
            b._L.mvdb_debug_inject_abort(b.handle, inject)
            s.query_many(q_dev, 140, 1200)
            blk = s.out_block
            host_ref = TopkBlock(nq, tk, buf=blk.buf.cpu().clone())
            merged = sharded_exchange(blk, offset, lambda full: s.merge_blocks(full, nq))   # RCCL all-gather inside
            torch.cuda.synchronize(dev)
            m_ids, m_l2, m_n = merged.ids.cpu().numpy(), merged.l2.cpu().numpy(), merged.n.cpu().numpy()
            exp_n = on.copy()
            if inject >= 0:
                exp_n[inject] = -1
            assert np.array_equal(m_n, exp_n)
            for qi in range(nq):
                if exp_n[qi] < 0:
                    continue
                assert np.array_equal(m_ids[qi, :exp_n[qi]], oi[qi, :exp_n[qi]] + offset)
                assert np.array_equal(m_l2[qi, :exp_n[qi]].view(np.uint64), ol[qi, :exp_n[qi]].view(np.uint64))
            # the same block through the host merge (what the gloo tests compare against)
            host_ref.ids.add_(offset)
            h_ids, h_l2, h_n = merge_blocks_host(host_ref.buf.unsqueeze(0), nq, tk)
            assert np.array_equal(h_n, m_n)
            ok = m_n >= 0
            assert np.array_equal(h_ids[ok], m_ids[ok]) and np.array_equal(h_l2[ok], m_l2[ok])
            if inject >= 0:
                with pytest.raises(Exception, match="aborted"):
                    s.check()
        b._L.mvdb_debug_inject_abort(b.handle, -1)
        b.close()
    finally:
        dist.destroy_process_group()
