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


class SwarmIndex:
    """Data-parallel key frames + one row-sharded index: the per-step exchange of bench.py at N > 1.

    Every rank produces m new rows per step (the <=4 direction descriptors of its own key frame).  One step is
        all_gather(rows)                       -> every rank sees the world*m new rows in rank order = global id order
        local add of the rows it owns          (g % world == rank, slot g // world; no further traffic)
        local batched search of the `world` queries (one per rank) over its shard
        all_gather(per-shard top-k)            -> each rank merges the lists for ITS query
    i.e. two small collectives per step (world*m*16 KB and world*k*16 B per rank), both latency-bound on xGMI.
    "add before query" is kept (loop_detector.cpp:89-104): all of the step's rows are visible to all of its queries.
    """

    def __init__(self, local_index, rank: int, world: int, dist, device=None):
        self.local, self.rank, self.world, self.dist, self.device = local_index, rank, world, dist, device or "cpu"
        self.local.set_shard(rank, world)
        self._ntotal = 0

    @property
    def ntotal(self) -> int:
        return self._ntotal

    def preload_local(self, rows_local: np.ndarray, ntotal_global: int):
        """Bulk load: this rank's rows of an index holding ntotal_global rows (ntotal_global % world == 0)."""
        assert ntotal_global % self.world == 0 and rows_local.shape[0] * self.world == ntotal_global
        self.local.add(rows_local)
        self._ntotal = ntotal_global

    def _all_gather(self, arr: np.ndarray) -> np.ndarray:
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return torch.stack(out).cpu().numpy()

    def step(self, rows: np.ndarray, query_row: int, k: int):
        rows = np.ascontiguousarray(rows, np.float32)
        m = rows.shape[0]
        assert self._ntotal % self.world == 0
        allrows = self._all_gather(rows).reshape(self.world * m, -1)       # global ids base .. base + world*m - 1
        g = self._ntotal + np.arange(self.world * m)
        mine = allrows[(g % self.world) == self.rank]
        self.local.add(mine)
        self._ntotal += self.world * m
        queries = allrows.reshape(self.world, m, -1)[:, query_row]         # one query per rank
        D, I = self.local.search(queries, k)                               # [world, k], global ids
        packed = np.empty((self.world, k, 2), np.int64)
        packed[..., 0] = D.view(np.int32).astype(np.int64)
        packed[..., 1] = I
        allp = self._all_gather(packed)                                    # [shard, query, k, 2]
        Dl = np.ascontiguousarray(allp[:, self.rank, :, 0].astype(np.int32)).view(np.float32)[:, None, :]
        Il = np.ascontiguousarray(allp[:, self.rank, :, 1])[:, None, :]
        return capi.topk_merge(Dl, Il, k)

    def step_batch(self, rows: np.ndarray, query_row: int, k: int):
        """F consecutive steps in two collectives and one index synchronisation.  rows [F][m][d]: this rank's m new rows of each of
        its next F key frames.  Step f's rows of all ranks get the global ids base + f*world*m ..., exactly as F calls of step()
        would number them, and step f's queries only see rows up to and including step f's (the shard's prefix search): the
        returned [F] lists of (D [1][k], I [1][k]) equal those of F step() calls."""
        rows = np.ascontiguousarray(rows, np.float32)
        F, m, d = rows.shape
        assert self._ntotal % self.world == 0
        allrows = self._all_gather(rows)                                   # [world][F][m][d]
        ordered = np.ascontiguousarray(allrows.transpose(1, 0, 2, 3)).reshape(F * self.world * m, d)   # global id order
        g = self._ntotal + np.arange(F * self.world * m)
        self.local.add(ordered[(g % self.world) == self.rank])
        queries = np.ascontiguousarray(allrows[:, :, query_row].transpose(1, 0, 2))      # [F][world][d]
        limits = [(self._ntotal + (f + 1) * self.world * m - self.rank + self.world - 1) // self.world for f in range(F)]
        D, I = self.local.search_prefix_many(queries, k, limits)          # [F][world][k], global ids
        self._ntotal += F * self.world * m
        packed = np.empty((F, self.world, k, 2), np.int64)
        packed[..., 0] = D.view(np.int32).astype(np.int64)
        packed[..., 1] = I
        allp = self._all_gather(packed)                                    # [shard][F][query][k][2]
        out = []
        for f in range(F):
            Dl = np.ascontiguousarray(allp[:, f, self.rank, :, 0].astype(np.int32)).view(np.float32)[:, None, :]
            Il = np.ascontiguousarray(allp[:, f, self.rank, :, 1])[:, None, :]
            out.append(capi.topk_merge(Dl, Il, k))
        return out



class NativeSwarmIndex:
    """SwarmIndex with the exchange inside libomni_hip.so: ncclAllGather on device buffers on the index stream, no host bounce, no torch in
    the data path (csrc/shard.hip, omni_shard_*).  Same numbering and results as SwarmIndex.step_batch (checked against it and against
    the unsharded oracle by tests/test_gpu_shard_rccl.py).  unique_id: capi.shard_unique_id() of rank 0, carried to every rank by the
    launcher (bench.py: torch.distributed.broadcast_object_list -- plumbing only)."""

    def __init__(self, ctx, local_index, rank: int, world: int, unique_id: bytes):
        self.local, self.rank, self.world = local_index, rank, world
        self.shard = capi.Shard(ctx, local_index, rank, world, unique_id)
        self._stage = None

    @property
    def ntotal(self) -> int:
        return self.shard.ntotal

    def preload_local(self, rows_local: np.ndarray, ntotal_global: int):
        self.shard.preload_local(rows_local, ntotal_global)

    def step_batch_dev(self, F: int, m: int, rows_dev: int, query_row: int, k: int):
        """rows_dev: [F][m][d] fp32 in HBM (e.g. MobileNetVLAD's output buffer, complete w.r.t. the index stream).  -> [F] x (D [1][k], I [1][k])"""
        D, I = self.shard.step_batch_dev(F, m, rows_dev, query_row, k)
        return [(D[f:f + 1], I[f:f + 1]) for f in range(F)]

    def step_batch(self, rows: np.ndarray, query_row: int, k: int):
        rows = np.ascontiguousarray(rows, np.float32)
        F, m, _ = rows.shape
        ctx = self.shard.ctx
        dev = ctx.to_device(rows)
        try:
            return self.step_batch_dev(F, m, dev, query_row, k)
        finally:
            ctx.free(dev)

    def search(self, q: np.ndarray, k: int):
        return self.shard.search(q, k)

    def close(self):
        self.shard.close()
