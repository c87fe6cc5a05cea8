"""Host logic of LoopDetector.on_images_recv_batch without a GPU: the plan (which rows, which searches, over how many rows) and the
replay must reproduce on_image_recv frame by frame.  The index and the context are host stand-ins with the product's method names
(exact IP over numpy rows, "device" pointers into a byte arena); the real index is covered by the -m gpu twin of this test."""
import importlib

import numpy as np
import pytest

from oracle import match_ref as M
from tests import detector_stream as DS

DIM = 4096


class ArenaCtx:
    """alloc / to_device / from_device / free / sync over host memory; addresses are plain integers so pointer arithmetic works."""

    def __init__(self):
        self.blocks, self.next, self.syncs = {}, 4096, 0

    def alloc(self, nbytes):
        base = self.next
        self.blocks[base] = np.zeros(nbytes, np.uint8)
        self.next += (nbytes + 4095) // 4096 * 4096 + 4096
        return base

    def to_device(self, arr):
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        p = self.alloc(raw.size)
        self.blocks[p][:] = raw
        return p

    def view(self, addr, nbytes):
        for base, blk in self.blocks.items():
            if base <= addr and addr + nbytes <= base + blk.size:
                return blk[addr - base:addr - base + nbytes]
        raise AssertionError(f"wild pointer {addr}+{nbytes}")

    def from_device(self, p, shape, dtype):
        self.syncs += 1
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.view(p, n).view(dtype).reshape(shape).copy()

    def free(self, p):
        del self.blocks[p]

    def sync(self):
        self.syncs += 1


class HostIndex:
    def __init__(self, ctx):
        self.ctx, self.rows, self.host_syncs = ctx, np.zeros((0, DIM), np.float32), 0

    @property
    def ntotal(self):
        return len(self.rows)

    def add(self, x):
        self.host_syncs += 1
        self.rows = np.concatenate([self.rows, np.atleast_2d(x).astype(np.float32)])

    def search(self, q, k):
        self.host_syncs += 1
        return M.ip_search(self.rows, np.atleast_2d(q), k)

    def add_dev(self, n, x_dev):
        self.rows = np.concatenate([self.rows, self.ctx.view(x_dev, n * DIM * 4).view(np.float32).reshape(n, DIM)])

    def search_prefix_dev(self, nq, q_dev, k, n_limit, D_dev, I_dev):
        q = self.ctx.view(q_dev, nq * DIM * 4).view(np.float32).reshape(nq, DIM)
        D, I = M.ip_search(self.rows[:min(n_limit, len(self.rows))], q, k)
        self.ctx.view(D_dev, nq * k * 4)[:] = D.astype(np.float32).view(np.uint8).reshape(-1)
        self.ctx.view(I_dev, nq * k * 8)[:] = I.astype(np.int64).view(np.uint8).reshape(-1)


@pytest.fixture()
def detector_mod():
    return importlib.import_module("omni-swarm_amd.detector")


@pytest.mark.parametrize("batch,on_device", [(1, False), (3, False), (4, True), (16, True)])
def test_batch_equals_sequential_and_oracle(detector_mod, batch, on_device):
    frames = DS.make_stream(seed=41, n_frames=150, n_places=16)
    ctx = ArenaCtx()
    seq = DS.trace(DS.run_product(frames, ctx, detector_mod, index_factory=lambda: HostIndex(ctx)))
    ctx2 = ArenaCtx()
    made = []

    def factory():
        made.append(HostIndex(ctx2))
        return made[-1]

    got = DS.trace(DS.run_product_batched(frames, ctx2, detector_mod, batch=batch, rows_on_device=on_device, index_factory=factory))
    assert np.array_equal(got, seq)
    assert np.array_equal(got, DS.trace(DS.run_oracle(frames)))
    assert all(ix.host_syncs == 0 for ix in made)                      # no per-row / per-query host round trips left
    assert ctx2.syncs <= (len(frames) + batch - 1) // batch            # at most one synchronisation per batch


def test_batch_with_nothing_to_do(detector_mod):
    ctx = ArenaCtx()
    det = detector_mod.LoopDetector(ctx, 1, index_factory=lambda: HostIndex(ctx), **DS.PARAMS)
    assert det.on_images_recv_batch([]) == []
    empty = detector_mod.FisheyeFrameDescriptor(msg_id=1, drone_id=2, landmark_num=0, images=[])
    remote_first = detector_mod.FisheyeFrameDescriptor(
        msg_id=2, drone_id=2, landmark_num=400,
        images=[detector_mod.ImageDescriptor(drone_id=2, landmark_num=100, image_desc=np.ones(DIM, np.float32)) for _ in range(4)])
    recs = det.on_images_recv_batch([empty, remote_first])              # remote frame while the database is empty: dropped (:36-38)
    assert [r["added"] for r in recs] == [False, False] and det.database_size() == 0 and ctx.syncs == 0
