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

    def search(self, q, k):
        db = np.stack(self.rows) if self.rows else np.zeros((0, self.d), np.float32)
        D, I = self._M.ip_search(db, q, k)
        return D, np.where(I >= 0, I * self.world + self.rank, -1)


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
