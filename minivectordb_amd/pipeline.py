"""Resident (device-pointer) use of the engine: queries and results stay in HBM.

torch is plumbing only here: it owns device memory and the stream; every kernel is
launched by libmvdb_hip.so through the `_dev` entry points of include/mvdb_hip.h.

Process rule: when torch shares the process, initialise torch's device first
(``torch.cuda.init()`` / ``torch.cuda.set_device``) and create HipBridge objects afterwards.
"""
from __future__ import annotations

import ctypes as C

import torch

from .bridge import HipBridge
from .distributed import TopkBlock


class ResidentSearcher:
    """searchMany over device tensors: prep -> traversal -> rerank -> top-k in one call."""

    def __init__(self, bridge: HipBridge, nq_max: int, topk: int, device: torch.device):
        self.b, self.L, self.h = bridge, bridge._L, bridge.handle
        self.device = device
        self.topk = int(topk)
        # outputs live in ONE packed block {l2 | ids | n}: the unit of the sharded path's single all-gather
        self.out_block = TopkBlock(nq_max, self.topk, device=device)
        self.out_ids, self.out_l2, self.out_n = self.out_block.ids, self.out_block.l2, self.out_block.n
        self.ctr = torch.zeros((nq_max, 4), dtype=torch.int32, device=device)

    def query_many(self, q_f32: torch.Tensor, ef_search: int, ann_k: int, allow: torch.Tensor | None = None):
        assert q_f32.is_cuda and q_f32.dtype == torch.float32 and q_f32.is_contiguous()
        nq = q_f32.shape[0]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ok = self.L.mvdb_query_many_dev(self.h, C.c_void_p(q_f32.data_ptr()), nq, self.topk, int(ef_search),
                                        int(ann_k), C.c_void_p(allow.data_ptr()) if allow is not None else None,
                                        C.c_void_p(self.out_ids.data_ptr()), C.c_void_p(self.out_l2.data_ptr()),
                                        C.c_void_p(self.out_n.data_ptr()), C.c_void_p(self.ctr.data_ptr()),
                                        C.c_void_p(stream))
        self.b._check(ok)
        return self.out_ids[:nq], self.out_l2[:nq], self.out_n[:nq]

    # ---- consecutive batches, software-pipelined ---------------------------------------------
    # The traversal is latency-bound and leaves HBM idle; the rerank is bandwidth-bound.  With
    # batch i's rerank on a second stream, it runs under batch i+1's traversal (call
    # bridge.setTuning(rerank_waves=2) so that a rerank workgroup fits beside four traversal workgroups on a
    # CU).  Results of a submitted batch are valid after `wait(slot)`/`drain()`.
    def _pipe_init(self, nq: int, dim: int, ann_k: int, depth: int = 2):
        key = (nq, dim, ann_k, depth)
        if getattr(self, "_pipe_key", None) == key:
            return
        dev = self.device
        if getattr(self, "_pipe_key", None) is not None:
            torch.cuda.synchronize(dev)  # batches of the previous shape may still use the old buffers
        self._pipe_key = key
        self._pipe_i = 0
        # the traversal needs all its workgroups resident at once (it is latency-bound): its stream gets
        # the high priority so that a ready rerank never takes their place at dispatch
        self._t_stream = torch.cuda.Stream(device=dev, priority=-1)
        self._r_stream = torch.cuda.Stream(device=dev, priority=0)
        # whole-batch streams of submit_interleaved (the engine keeps up to 4 per-stream scratch sets)
        self._depth = depth
        self._b_streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self._slots = []
        for _ in range(max(2, depth)):
            blk = TopkBlock(nq, self.topk, device=dev)
            self._slots.append(dict(
                block=blk,
                qf=torch.empty((nq, dim), dtype=torch.float32, device=dev),
                q8=torch.empty((nq, dim), dtype=torch.int8, device=dev),
                cand=torch.empty((nq, max(ann_k, 1)), dtype=torch.int32, device=dev),
                dist=torch.empty((nq, max(ann_k, 1)), dtype=torch.float32, device=dev),
                cnt=torch.empty((nq,), dtype=torch.int32, device=dev),
                ctr=torch.zeros((nq, 4), dtype=torch.int32, device=dev),
                ids=blk.ids, l2=blk.l2, n=blk.n,
                ready=torch.cuda.Event(), done=torch.cuda.Event(), consumed=torch.cuda.Event(), used=False,
                has_consumer=False))

    def submit(self, q_f32: torch.Tensor, ef_search: int, ann_k: int, allow: torch.Tensor | None = None):
        """Enqueue one batch; returns (slot, ids, l2, n) -- tensors are valid after wait(slot)."""
        assert q_f32.is_cuda and q_f32.dtype == torch.float32 and q_f32.is_contiguous()
        nq, dim = q_f32.shape
        self._pipe_init(nq, dim, int(ann_k))
        slot = self._pipe_i & 1
        self._pipe_i += 1
        s = self._slots[slot]
        L, h = self.L, self.h
        L.mvdb_set_ef_search(h, int(ef_search))                    # mini-vector-db.ts:572
        L.mvdb_ensure_results_cap_at_least(h, int(ann_k))          # :577
        cur = torch.cuda.current_stream(self.device)
        T, R = self._t_stream, self._r_stream
        T.wait_stream(cur)                       # the queries were produced on the caller's stream
        if s["used"]:
            T.wait_event(s["done"])              # the rerank two batches ago still reads this slot
        vp = C.c_void_p
        self.b._check(L.mvdb_prepare_queries_dev(h, vp(q_f32.data_ptr()), nq, vp(s["qf"].data_ptr()),
                                                 vp(s["q8"].data_ptr()), vp(T.cuda_stream)))
        self.b._check(L.mvdb_search_many_dev(h, vp(s["q8"].data_ptr()), nq, int(ann_k), vp(s["cand"].data_ptr()),
                                             vp(s["dist"].data_ptr()), vp(s["cnt"].data_ptr()),
                                             vp(s["ctr"].data_ptr()), vp(T.cuda_stream)))
        s["ready"].record(T)
        R.wait_event(s["ready"])
        self.b._check(L.mvdb_rerank_many_dev(h, vp(s["qf"].data_ptr()), nq, vp(s["cand"].data_ptr()),
                                             vp(s["cnt"].data_ptr()), max(int(ann_k), 1), int(ann_k),
                                             vp(allow.data_ptr()) if allow is not None else None, self.topk,
                                             vp(s["ids"].data_ptr()), vp(s["l2"].data_ptr()), vp(s["n"].data_ptr()),
                                             vp(R.cuda_stream)))
        s["done"].record(R)
        s["used"] = True
        self.ctr = s["ctr"]
        return slot, s["ids"], s["l2"], s["n"]

    def submit_interleaved(self, q_f32: torch.Tensor, ef_search: int, ann_k: int, allow: torch.Tensor | None = None,
                           sync_with_current: bool = True, depth: int = 2):
        """`depth` (2..4) whole batches in flight: batch i runs prep -> traversal -> rerank on stream i % depth.
        A traversal's queries finish at different times (the slowest takes ~20 % longer than the
        mean) and a 1 k-query batch does not fill the LDS of the chip; the next batch's workgroups take
        the places as they come free.  Same return convention as submit().  `sync_with_current=False`
        when the queries were not produced on the caller's current stream (resident query sets):
        otherwise the batch also waits for whatever else that stream has queued, e.g. the
        all-gather of the previous batch."""
        assert q_f32.is_cuda and q_f32.dtype == torch.float32 and q_f32.is_contiguous()
        nq, dim = q_f32.shape
        assert 2 <= depth <= 4
        self._pipe_init(nq, dim, int(ann_k), depth)
        slot = self._pipe_i % depth
        self._pipe_i += 1
        s = self._slots[slot]
        L, h = self.L, self.h
        L.mvdb_set_ef_search(h, int(ef_search))
        L.mvdb_ensure_results_cap_at_least(h, int(ann_k))
        cur = torch.cuda.current_stream(self.device)
        S = self._b_streams[slot]   # stream order protects the slot's buffers
        if sync_with_current:
            S.wait_stream(cur)
        elif s["has_consumer"]:
            S.wait_event(s["consumed"])  # whoever read this slot's outputs two batches ago (release())
        vp = C.c_void_p
        # the engine keeps one set of intermediates (and visited bitmaps) per stream, four streams at most
        self.b._check(L.mvdb_query_many_dev(h, vp(q_f32.data_ptr()), nq, self.topk, int(ef_search), int(ann_k),
                                            vp(allow.data_ptr()) if allow is not None else None,
                                            vp(s["ids"].data_ptr()), vp(s["l2"].data_ptr()), vp(s["n"].data_ptr()),
                                            vp(s["ctr"].data_ptr()), vp(S.cuda_stream)))
        s["done"].record(S)
        s["used"] = True
        self.ctr = s["ctr"]
        return slot, s["ids"], s["l2"], s["n"]

    def wait(self, slot: int):
        """Make the caller's current stream wait for the batch in `slot`."""
        torch.cuda.current_stream(self.device).wait_event(self._slots[slot]["done"])

    def release(self, slot: int):
        """Tell the pipeline that everything queued so far on the caller's current stream has finished
        with the outputs of `slot` (needed with sync_with_current=False when another stream reads them)."""
        s = self._slots[slot]
        s["consumed"].record(torch.cuda.current_stream(self.device))
        s["has_consumer"] = True

    def check(self):
        """Synchronise the device and raise if a traversal of this handle aborted a query (its results
        carry n = -1; mvdb_check_errors).  Call outside timed regions."""
        torch.cuda.synchronize(self.device)
        self.b.checkErrors()

    def drain(self):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self._t_stream)
        cur.wait_stream(self._r_stream)
        for st in self._b_streams:
            cur.wait_stream(st)

    def merge_topk(self, ids: torch.Tensor, l2: torch.Tensor, n: torch.Tensor, parts: int, nq: int):
        """Merge [parts][nq][topk] per-shard results by (l2, id)."""
        out_ids = torch.empty((nq, self.topk), dtype=torch.int32, device=self.device)
        out_l2 = torch.empty((nq, self.topk), dtype=torch.float64, device=self.device)
        out_n = torch.empty((nq,), dtype=torch.int32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ok = self.L.mvdb_merge_topk_dev(self.h, C.c_void_p(ids.data_ptr()), C.c_void_p(l2.data_ptr()),
                                        C.c_void_p(n.data_ptr()), int(parts), int(nq), self.topk,
                                        C.c_void_p(out_ids.data_ptr()), C.c_void_p(out_l2.data_ptr()),
                                        C.c_void_p(out_n.data_ptr()), C.c_void_p(stream))
        self.b._check(ok)
        return out_ids, out_l2, out_n

    def slot_block(self, slot: int) -> TopkBlock:
        """The packed result block of a submitted batch (valid after wait(slot))."""
        return self._slots[slot]["block"]

    def merge_blocks(self, full: torch.Tensor, nq: int):
        """Merge gathered per-shard blocks [parts, block_bytes] by (l2, id) on the device -> a TopkBlock."""
        parts = int(full.shape[0])
        # merged blocks come from a small ring allocated once per shape (no allocation per step; four merged
        # results may be alive at a time, as many as batches can be in flight)
        ring = getattr(self, "_merged_ring", None)
        if ring is None or ring[0].nq != int(nq):
            ring = self._merged_ring = [TopkBlock(nq, self.topk, device=self.device) for _ in range(4)]
            self._merged_i = 0
        out = ring[self._merged_i % len(ring)]
        self._merged_i += 1
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ok = self.L.mvdb_merge_topk_packed_dev(self.h, C.c_void_p(full.data_ptr()), parts, int(nq), self.topk,
                                               C.c_void_p(out.ids.data_ptr()), C.c_void_p(out.l2.data_ptr()),
                                               C.c_void_p(out.n.data_ptr()), C.c_void_p(stream))
        self.b._check(ok)
        return out

    def stage_stats(self, stage: int, reset: bool = True):
        launches, total = C.c_longlong(0), C.c_double(0.0)
        self.L.mvdb_stage_stats(self.h, int(stage), C.byref(launches), C.byref(total), 1 if reset else 0)
        return launches.value, total.value
