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

    def save(self, path):
        np.save(path + ".npy", self.rows)

    def load(self, path):
        self.rows = np.load(path + ".npy")

    def search_prefix_dev(self, nq, q_dev, k, n_limit, D_dev, I_dev):
        q = self.ctx.view(q_dev, nq * DIM * 4).view(np.float32).reshape(nq, DIM)
        D, I = M.ip_search(self.rows[:min(n_limit, len(self.rows))], q, k)
        self.ctx.view(D_dev, nq * k * 4)[:] = D.astype(np.float32).view(np.uint8).reshape(-1)
        self.ctx.view(I_dev, nq * k * 8)[:] = I.astype(np.int64).view(np.uint8).reshape(-1)


    def search_batch_prefix_dev(self, rows_dev, row_idx, k, limits, D_dev, I_dev):
        nq = len(limits)
        for j in range(nq):
            r = j if row_idx is None else int(row_idx[j])
            self.search_prefix_dev(1, rows_dev + r * DIM * 4, k, int(limits[j]), D_dev + j * k * 4, I_dev + j * k * 8)

    def truncate(self, n):
        self.rows = self.rows[:n]


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


def test_checkpoint_resume_continues_the_same_trace(detector_mod, tmp_path):
    """save() half-way through a stream, load() into a fresh detector, feed the rest to both: identical decisions (id maps, init-mode
    counters, known nodes, stored frames and both index shards all survive)."""
    frames = DS.make_stream(seed=51, n_frames=120, n_places=12)

    def mk(fr):
        return detector_mod.FisheyeFrameDescriptor(
            msg_id=fr["msg_id"], drone_id=fr["drone_id"], landmark_num=fr["landmark_num"], prevent_adding_db=fr["prevent_adding_db"],
            images=[detector_mod.ImageDescriptor(drone_id=i["drone_id"], landmark_num=i["landmark_num"], image_desc=i["image_desc"],
                                                 feature_descriptor=np.full((3, 64), float(fr["msg_id"] % 7), np.float32),
                                                 landmarks_2d=np.array([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]], np.float32)) for i in fr["images"]])

    loop = lambda n, o, dn, do, im: DS.loop_ok(n.msg_id, o.msg_id)
    ctx = ArenaCtx()
    a = detector_mod.LoopDetector(ctx, DS.SELF_ID, compute_loop=loop, index_factory=lambda: HostIndex(ctx), **DS.PARAMS)
    for fr in frames[:70]:
        a.on_image_recv(mk(fr))
    prefix = str(tmp_path / "det")
    a.save(prefix)
    b = detector_mod.LoopDetector(ctx, DS.SELF_ID, compute_loop=loop, index_factory=lambda: HostIndex(ctx), **DS.PARAMS)
    b.load(prefix)
    assert b.database_size() == a.database_size() and b.imgid2fisheye == a.imgid2fisheye and b.all_nodes == a.all_nodes
    some = next(iter(a.fisheyeframe_database))
    assert np.array_equal(b.fisheyeframe_database[some].images[1].feature_descriptor, a.fisheyeframe_database[some].images[1].feature_descriptor)
    ra = [a.on_image_recv(mk(fr)) for fr in frames[70:]]
    rb = b.on_images_recv_batch([mk(fr) for fr in frames[70:]])       # and the resumed one through the batched entry point
    assert np.array_equal(DS.trace(ra), DS.trace(rb))
    assert (DS.trace(ra)[:, 4] != -1).any()
    with pytest.raises(ValueError):
        detector_mod.LoopDetector(ctx, 2, index_factory=lambda: HostIndex(ctx), **DS.PARAMS).load(prefix)



def _frame(detector_mod, msg_id, drone, descs, lms):
    return detector_mod.FisheyeFrameDescriptor(
        msg_id=msg_id, drone_id=drone, landmark_num=int(sum(lms)),
        images=[detector_mod.ImageDescriptor(drone_id=drone, landmark_num=lm, image_desc=d) for d, lm in zip(descs, lms)])


def test_direction_never_received_has_an_empty_descriptor(detector_mod, tmp_path):
    """A direction that was not received carries landmark_num = 0 and an EMPTY image_desc (the ImageDescriptor default; loop_net's remote
    frames, loop_net.cpp:206-218): sequential, batched and checkpointed paths must all accept it."""
    rng = np.random.default_rng(5)
    def unit():
        v = rng.standard_normal(DIM).astype(np.float32)
        return v / np.linalg.norm(v)
    frames = []
    for i in range(12):
        descs = [unit(), unit(), unit(), np.zeros(0, np.float32)]
        frames.append((1000 + i, 1 if i % 3 else 2, descs, [50, 60, 70, 0]))
    ctx = ArenaCtx()
    a = detector_mod.LoopDetector(ctx, DS.SELF_ID, index_factory=lambda: HostIndex(ctx), **DS.PARAMS)
    ra = [a.on_image_recv(_frame(detector_mod, *f)) for f in frames]
    b = detector_mod.LoopDetector(ctx, DS.SELF_ID, index_factory=lambda: HostIndex(ctx), **DS.PARAMS)
    rb = b.on_images_recv_batch([_frame(detector_mod, *f) for f in frames[:5]]) + b.on_images_recv_batch([_frame(detector_mod, *f) for f in frames[5:]])
    assert np.array_equal(DS.trace(ra), DS.trace(rb)) and a.database_size() == b.database_size() > 0
    prefix = str(tmp_path / "empty")
    b.save(prefix)
    c = detector_mod.LoopDetector(ctx, DS.SELF_ID, index_factory=lambda: HostIndex(ctx), **DS.PARAMS)
    c.load(prefix)
    some = next(iter(c.fisheyeframe_database.values()))
    assert some.images[3].image_desc.size == 0 and some.images[0].image_desc.size == DIM
    short = _frame(detector_mod, 5000, 1, [unit()], [500])            # fewer images than the queried direction: no crash, no query hit
    assert b.on_images_recv_batch([short])[0]["old_msg_id"] == -1


def test_compute_loop_exception_leaves_batch_state_consistent(detector_mod):
    """compute_loop raising in the middle of a batch: every frame's bookkeeping is still applied (ntotal == len(id map)), the exception
    surfaces afterwards, and the detector keeps working -- the sequential path would have stopped at that frame instead, but never with
    orphan rows in the index."""
    frames = DS.make_stream(seed=61, n_frames=90, n_places=6)
    ctx = ArenaCtx()
    calls = [0]

    def loop(n, o, dn, do, im):
        calls[0] += 1
        if calls[0] == 3:
            raise RuntimeError("geometry failed")
        return True

    det = detector_mod.LoopDetector(ctx, DS.SELF_ID, compute_loop=loop, index_factory=lambda: HostIndex(ctx), **DS.PARAMS)
    mk = lambda fr: detector_mod.FisheyeFrameDescriptor(
        msg_id=fr["msg_id"], drone_id=fr["drone_id"], landmark_num=fr["landmark_num"], prevent_adding_db=fr["prevent_adding_db"],
        images=[detector_mod.ImageDescriptor(drone_id=i["drone_id"], landmark_num=i["landmark_num"], image_desc=i["image_desc"]) for i in fr["images"]])
    raised = False
    for s in range(0, len(frames), 10):
        try:
            det.on_images_recv_batch([mk(fr) for fr in frames[s:s + 10]])
        except RuntimeError:
            raised = True
        n_ids = len(det.imgid2fisheye)
        assert det.local_index.ntotal + det.remote_index.ntotal == n_ids
        assert det._deferred is None
    assert raised and calls[0] > 3
