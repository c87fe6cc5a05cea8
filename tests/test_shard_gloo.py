"""world_size-2 test of the sharded-index exchange step on CPU (gloo).  The per-shard search is a CPU stand-in with the
capi.IndexFlatIP interface (the oracle's exact search) so that the routing, the single all_gather and the merge --
the multi-GPU logic of omni-swarm_amd/shard.py -- run exactly as they do over RCCL on the GPU box."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _CpuShard:
    """CPU stand-in for capi.IndexFlatIP (test only)."""

    def __init__(self, d):
        from oracle import match_ref as M
        self._M, self.d, self.rows, self.rank, self.world = M, d, [], 0, 1

    @property
    def ntotal(self):
        return len(self.rows)

    def set_shard(self, rank, world):
        self.rank, self.world = rank, world

    def add(self, x):
        self.rows += [r.copy() for r in np.atleast_2d(x)]

    def search(self, q, k, n_limit=None):
        rows = self.rows if n_limit is None else self.rows[:n_limit]
        db = np.stack(rows) if rows else np.zeros((0, self.d), np.float32)
        D, I = self._M.ip_search(db, q, k)
        return D, np.where(I >= 0, I * self.world + self.rank, -1)

    def search_prefix_many(self, q, k, limits):
        res = [self.search(q[f], k, int(limits[f])) for f in range(len(limits))]
        return np.stack([r[0] for r in res]), np.stack([r[1] for r in res])


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import omni_loader
    omni = omni_loader.load()
    from omni_swarm_amd import shard
    from oracle import match_ref as M
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    db = rng.standard_normal((257, 128)).astype(np.float32)
    db[200] = db[3]
    q = np.stack([db[3], db[50], -db[9]]).astype(np.float32)
    idx = shard.ShardedIndex(_CpuShard(128), rank, world, dist)
    idx.add(db[:100])
    idx.add(db[100:101])
    idx.add(db[101:])
    assert idx.ntotal == 257
    assert idx.local.ntotal == len(range(rank, 257, world))
    D, I = idx.search(q, 9)
    Dr, Ir = M.ip_search(db, q, 9)
    ok = bool(np.array_equal(I, Ir) and np.array_equal(D, Dr) and I[0, 0] == 3 and I[0, 1] == 200)
    # SwarmIndex: data-parallel key frames, two collectives per step
    sw = shard.SwarmIndex(_CpuShard(128), rank, world, dist)
    pre = rng.standard_normal((40, 128)).astype(np.float32)             # same stream on both ranks -> same global DB
    sw.preload_local(pre[rank::world], 40)
    ref_db = [r for r in pre]
    for step in range(3):
        new = np.random.default_rng(100 + step).standard_normal((world, 4, 128)).astype(np.float32)
        new[1, 1] = pre[6] * 1.01 if step == 1 else new[1, 1]           # rank 1's query hits a preloaded row
        D, I = sw.step(new[rank], query_row=1, k=7)
        ref_db += [r for r in new.reshape(world * 4, 128)]
        Dr, Ir = M.ip_search(np.stack(ref_db), new[rank, 1], 7)
        ok = ok and bool(np.array_equal(I, Ir) and np.array_equal(D, Dr)) and sw.ntotal == len(ref_db)
        ok = ok and sw.local.ntotal == len(ref_db) // world
    # step_batch: three more steps in two collectives; must equal three step() calls (ids numbered per step, queries see only the
    # rows up to their own step)
    sw2 = shard.SwarmIndex(_CpuShard(128), rank, world, dist)
    sw2.preload_local(pre[rank::world], 40)
    news = [np.random.default_rng(100 + step).standard_normal((world, 4, 128)).astype(np.float32) for step in range(3)]
    news[1][1, 1] = pre[6] * 1.01
    news[2][0, 1] = news[0][1, 2] * 0.99                               # step 2's query of rank 0 hits a row added in step 0 by rank 1
    seq = []
    sw3 = shard.SwarmIndex(_CpuShard(128), rank, world, dist)
    sw3.preload_local(pre[rank::world], 40)
    for step in range(3):
        seq.append(sw3.step(news[step][rank], query_row=1, k=7))
    got = sw2.step_batch(np.stack([n[rank] for n in news]), query_row=1, k=7)
    for (Db, Ib), (Ds, Is) in zip(got, seq):
        ok = ok and bool(np.array_equal(Ib, Is) and np.array_equal(Db, Ds))
    ok = ok and sw2.ntotal == sw3.ntotal and sw2.local.ntotal == sw3.local.ntotal
    if rank == 0:
        ok = ok and int(got[2][1][0, 0]) == 40 + 0 * world * 4 + 1 * 4 + 2   # the planted row's global id (step 0, rank 1, row 2)
    out[rank] = ok
    dist.destroy_process_group()


def test_sharded_search_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] and out[1]
