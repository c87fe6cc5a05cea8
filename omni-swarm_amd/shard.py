"""Row-sharded global-descriptor index across the GPUs of one node (new; the reference is single-GPU).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).  Global
row g lives on rank g % world at local slot g // world, which keeps insertion order per shard -- the recency rule
`label <= ntotal - max_index` of loop_detector.cpp:232 is evaluated on GLOBAL ids after the merge.  A search is

    every rank: local exact top-k over its shard (HIP scan + top-k)         -> [nq, k] scores + global ids
    ONE exchange step: all_gather of the (score f32, id i64) lists            (k*12 B per query per rank: latency-bound)
    every rank: world*k-way merge, ties -> lower global id                    (omni_topk_merge, host)

so every rank ends up with the identical faiss::IndexFlatIP::search result.  Inserts touch exactly one shard and need
no collective (every rank calls add() with the same rows and keeps its own).
"""
from __future__ import annotations

import numpy as np

from . import capi


class ShardedIndex:
    def __init__(self, local_index, rank: int, world: int, dist=None, device=None):
        """local_index: object with add(x), search(q, k), ntotal, set_shard(rank, world) -- a capi.IndexFlatIP on the
        GPU box.  dist: the torch.distributed module (initialised) or None for world == 1."""
        self.local = local_index
        self.rank, self.world, self.dist, self.device = rank, world, dist, device
        self.local.set_shard(rank, world)
        self._ntotal = 0

    @property
    def ntotal(self) -> int:
        return self._ntotal

    def add(self, x: np.ndarray):
        """Collective in spirit (all ranks see the same rows), communication-free in practice."""
        x = np.atleast_2d(np.asarray(x, np.float32))
        g = np.arange(self._ntotal, self._ntotal + x.shape[0])
        mine = x[(g % self.world) == self.rank]
        if len(mine):
            self.local.add(mine)
        self._ntotal += x.shape[0]

    def search(self, q: np.ndarray, k: int):
        q = np.atleast_2d(np.asarray(q, np.float32))
        D, I = self.local.search(q, k)                     # global ids already (set_shard)
        if self.world == 1 or self.dist is None:
            return D, I
        import torch
        dev = self.device or "cpu"
        # one all_gather: pack score and id into one int64 tensor so a single collective moves both
        packed = np.empty((q.shape[0], k, 2), np.int64)
        packed[..., 0] = D.view(np.int32).astype(np.int64)
        packed[..., 1] = I
        t = torch.from_numpy(packed).to(dev)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        allp = torch.stack(out).cpu().numpy()              # [world, nq, k, 2]
        Dl = np.ascontiguousarray(allp[..., 0].astype(np.int32)).view(np.float32)
        Il = np.ascontiguousarray(allp[..., 1])
        return capi.topk_merge(Dl, Il, k)
