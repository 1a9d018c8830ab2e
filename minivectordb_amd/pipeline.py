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


class ResidentSearcher:
    """searchMany over device tensors: prep -> traversal -> rerank -> top-k in one call."""

    def __init__(self, bridge: HipBridge, nq_max: int, topk: int, device: torch.device):
        self.b, self.L, self.h = bridge, bridge._L, bridge.handle
        self.device = device
        self.topk = int(topk)
        self.out_ids = torch.empty((nq_max, self.topk), dtype=torch.int32, device=device)
        self.out_l2 = torch.empty((nq_max, self.topk), dtype=torch.float64, device=device)
        self.out_n = torch.empty((nq_max,), dtype=torch.int32, device=device)
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

    def stage_stats(self, stage: int, reset: bool = True):
        launches, total = C.c_longlong(0), C.c_double(0.0)
        self.L.mvdb_stage_stats(self.h, int(stage), C.byref(launches), C.byref(total), 1 if reset else 0)
        return launches.value, total.value
