"""LoopDetector: keyframe database + loop-candidate decision rules on top of the HIP index (host logic mirror).

Mirrors /root/reference/swarm_loop/src/loop_detector.cpp (class LoopDetector, include/swarm_loop/loop_detector.h:24-111):
    on_image_recv                      :11-137    gating (drop / add / query / hand to compute_loop)
    add_to_database                    :150-173   <=4 rows per keyframe into local_index / remote_index
    query_from_database (4- and 6-arg) :176-242   top-(5+max_index) IP search + recency/threshold rule, incl. the
                                                   fall-through return of :241 and the shared `distance` of :184-186
    query_fisheyeframe_from_database   :245-287
The two faiss::IndexFlatIP members are omni_index handles (exact IP, HBM-resident); geometry (compute_loop, :627-836:
BFMatcher + findHomography + solvePnPRansac + odometry gate) stays on the host and is passed in as a callback, as the
kernels' scope ends at the match candidate.

Unlike the reference (no mutex, entered from the ROS and LCM threads) calls are serialised by a lock.
"""
from __future__ import annotations

import threading
from dataclasses import dataclass, field

import numpy as np

from . import capi

REMOTE_MAGIN_NUMBER = 1000000   # loop_detector.h:22
SEARCH_NEAREST_NUM = 5          # loop_defines.h:32
DEEP_DESC_SIZE = 4096           # loop_defines.h:30
STEREO_PINHOLE, STEREO_FISHEYE, PINHOLE_DEPTH = 0, 1, 2


def _pose():
    """Pose_t as fromROSPose fills it (loop_cam.cpp:366-368): position xyz + orientation quaternion wxyz."""
    return np.array([0, 0, 0, 1, 0, 0, 0], np.float64)


@dataclass
class ImageDescriptor:
    """swarm_msgs::ImageDescriptor_t (un-vendored): every field the reference reads or writes on this path -- loop_cam.cpp:529-585 (extractor),
    :362-374,:434-440 (stereo part), loop_net.cpp:51-79,206-218 (wire split / reassembly), loop_detector.cpp:539-603 (matching).  Same fields
    as omni::ImageDescriptor (host/omni_swarm.hpp)."""
    drone_id: int = 0
    landmark_num: int = 0
    image_desc: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))                    # 4096 (empty: direction not received)
    feature_descriptor: np.ndarray = field(default_factory=lambda: np.zeros((0, 64), np.float32))      # n x 64
    landmarks_2d: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))             # pixel key points
    landmarks_2d_norm: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))        # lifted (x/z, y/z), loop_cam.cpp:558-569
    landmarks_3d: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))             # triangulated, loop_cam.cpp:434-440
    landmarks_flag: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint8))                  # 1 = has a 3-D point
    direction: int = 0
    msg_id: int = 0
    frame_id: int = 0
    timestamp: float = 0.0
    prevent_adding_db: bool = False
    pose_drone: np.ndarray = field(default_factory=_pose)
    camera_extrinsic: np.ndarray = field(default_factory=_pose)
    image_width: int = 0
    image_height: int = 0
    image: bytes = b""                                                                                 # optional JPEG (loop_cam.cpp:49-71), opaque


@dataclass
class FisheyeFrameDescriptor:    # swarm_msgs::FisheyeFrameDescriptor_t
    msg_id: int = 0
    drone_id: int = 0
    landmark_num: int = 0
    prevent_adding_db: bool = False
    images: list = field(default_factory=list)
    image_num: int = 0
    timestamp: float = 0.0
    pose_drone: np.ndarray = field(default_factory=_pose)


class LoopDetector:
    def __init__(self, ctx: capi.Context, self_id: int, *, inner_product_thres=0.6, init_mode_product_thres=0.3,
                 match_index_dist=10, min_loop_num=15, min_direction_loop=3, inter_drone_init_frames=50,
                 camera_configuration=STEREO_FISHEYE, compute_loop=None, storage=capi.STORE_F32,
                 index_factory=None):
        self.self_id = self_id
        self.INNER_PRODUCT_THRES = inner_product_thres
        self.INIT_MODE_PRODUCT_THRES = init_mode_product_thres
        self.MATCH_INDEX_DIST = match_index_dist
        self.MIN_LOOP_NUM = min_loop_num
        self.MIN_DIRECTION_LOOP = min_direction_loop
        self.inter_drone_init_frames = inter_drone_init_frames
        self.camera_configuration = camera_configuration
        make = index_factory or (lambda: capi.IndexFlatIP(ctx, DEEP_DESC_SIZE, storage))
        self.local_index = make()      # LoopDetector::LoopDetector, loop_detector.cpp:842-844
        self.remote_index = make()
        self.imgid2fisheye: dict[int, int] = {}
        self.imgid2dir: dict[int, int] = {}
        self.fisheyeframe_database: dict[int, FisheyeFrameDescriptor] = {}
        self.inter_drone_loop_count: dict[tuple[int, int], int] = {}
        self.all_nodes: set[int] = set()
        self.compute_loop = compute_loop or (lambda *a: False)
        self.log: list[dict] = []
        self._mu = threading.Lock()
        self._deferred = None          # results of searches enqueued ahead by on_images_recv_batch
        self._batch_bufs = None

    def database_size(self) -> int:
        if self._deferred is not None:                 # replaying a batch: the sizes as of this frame's turn, not the final ones
            return self._sim_local + self._sim_remote
        return self.local_index.ntotal + self.remote_index.ntotal

    def _add_image(self, img: ImageDescriptor) -> int:
        if self._deferred is not None:                 # on_images_recv_batch: the row is already in the index, appended ahead in this order
            row = self._row_ids.pop(0)
            if row >= REMOTE_MAGIN_NUMBER:
                self._sim_remote += 1
            else:
                self._sim_local += 1
            return row
        if img.drone_id == self.self_id:
            self.local_index.add(img.image_desc)
            return self.local_index.ntotal - 1
        self.remote_index.add(img.image_desc)
        return self.remote_index.ntotal - 1 + REMOTE_MAGIN_NUMBER

    def add_to_database(self, frame: FisheyeFrameDescriptor) -> int:
        for i, img in enumerate(frame.images):
            if img.landmark_num > 0:
                index = self._add_image(img)
                self.imgid2fisheye[index] = frame.msg_id
                self.imgid2dir[index] = i
        self.fisheyeframe_database[frame.msg_id] = frame
        return frame.msg_id

    def _query_index(self, img, index, remote_db: bool, thres: float, max_index: int, distance: list) -> int:
        search_num = SEARCH_NEAREST_NUM + max_index
        if self._deferred is not None:                 # on_images_recv_batch: the search ran ahead on the stream, over the rows of its turn
            D, I, ntotal = self._deferred.pop(0)
        else:
            D, I = index.search(img.image_desc, search_num)
            ntotal = index.ntotal
        return self._decide(D, I, ntotal, remote_db, thres, max_index, distance)

    def _decide(self, D, I, ntotal: int, remote_db: bool, thres: float, max_index: int, distance: list) -> int:
        """The scan over the top-(5 + max_index) list, loop_detector.cpp:215-241 (incl. its fall-through return)."""
        index_offset = REMOTE_MAGIN_NUMBER if remote_db else 0
        return_msg_id = -1
        for i in range(SEARCH_NEAREST_NUM + max_index):
            label = int(I[0, i])
            if label < 0:
                continue
            if label + index_offset not in self.imgid2fisheye:
                continue
            return_msg_id = label + index_offset
            if label <= ntotal - max_index and float(D[0, i]) > thres:
                distance[0] = float(D[0, i])
                return return_msg_id
        return return_msg_id

    def query_from_database(self, img, init_mode: bool, nonkeyframe: bool, distance: list) -> int:
        thres = self.INIT_MODE_PRODUCT_THRES if init_mode else self.INNER_PRODUCT_THRES
        if img.drone_id == self.self_id:
            _id = self._query_index(img, self.remote_index, True, thres, 1, distance)
            if not nonkeyframe:
                return self._query_index(img, self.local_index, False, thres, self.MATCH_INDEX_DIST, distance)
            elif _id != -1:
                return _id
        else:
            return self._query_index(img, self.local_index, False, thres, 1, distance)
        return -1

    def query_fisheyeframe_from_database(self, frame, init_mode: bool, nonkeyframe: bool):
        direction_new = 1 if self.camera_configuration == STEREO_FISHEYE else 0
        if len(frame.images) > direction_new and frame.images[direction_new].landmark_num > 0:
            distance = [-1.0]
            _id = self.query_from_database(frame.images[direction_new], init_mode, nonkeyframe, distance)
            if _id != -1 and distance[0] > -1:
                return (self.fisheyeframe_database[self.imgid2fisheye[_id]], direction_new, self.imgid2dir[_id], _id,
                        distance[0])
        return None, direction_new, -1, -1, -1.0

    def on_image_recv(self, frame: FisheyeFrameDescriptor) -> dict:
        with self._mu:
            return self._on_image_recv(frame)

    def _on_image_recv(self, frame):
        rec = {"msg_id": frame.msg_id, "drone_id": frame.drone_id, "added": False, "queried": False, "image_id": -1,
               "old_msg_id": -1, "dir_new": -1, "dir_old": -1, "loop": False}
        self.log.append(rec)
        if len(frame.images) == 0:
            return rec
        drone_id = frame.drone_id
        if drone_id != self.self_id and self.database_size() == 0:
            return rec
        new_node = frame.drone_id not in self.all_nodes
        self.all_nodes.add(frame.drone_id)
        if sum(1 for img in frame.images if img.landmark_num > 0) < self.MIN_DIRECTION_LOOP:
            return rec
        if frame.landmark_num < self.MIN_LOOP_NUM:
            return rec
        init_mode = False
        if drone_id != self.self_id:
            init_mode = self.inter_drone_loop_count.get((drone_id, self.self_id), 0) < self.inter_drone_init_frames
        if (not frame.prevent_adding_db) or new_node:
            self.add_to_database(frame)
            rec["added"] = True
        if self.database_size() > self.MATCH_INDEX_DIST or init_mode or drone_id != self.self_id:
            rec["queried"] = True
            old, d_new, d_old, image_id, dist = self.query_fisheyeframe_from_database(frame, init_mode, frame.prevent_adding_db)
            if d_old >= 0:
                rec.update(image_id=image_id, old_msg_id=old.msg_id, dir_new=d_new, dir_old=d_old, distance=dist)
                success, pair = False, None
                if old.drone_id == self.self_id:
                    success = self.compute_loop(frame, old, d_new, d_old, init_mode)
                    pair = (frame.drone_id, old.drone_id)
                elif frame.drone_id == self.self_id:
                    success = self.compute_loop(old, frame, d_old, d_new, init_mode)
                    pair = (old.drone_id, frame.drone_id)
                if success:
                    a, b = pair
                    self.inter_drone_loop_count[(a, b)] = self.inter_drone_loop_count.get((a, b), 0) + 1
                    self.inter_drone_loop_count[(b, a)] = self.inter_drone_loop_count.get((b, a), 0) + 1
                    rec["loop"] = True
        return rec

    def on_images_recv_batch(self, frames: list, rows_dev: int | None = None) -> list:
        """on_image_recv for several frames in arrival order with ONE host synchronisation instead of ~6 per frame (4 row appends,
        1-2 searches): which rows a frame appends and which searches it runs (loop_detector.cpp:36-98) does not depend on any search
        RESULT, so the whole batch is planned first, enqueued on the index stream -- all appends, then every search restricted to the
        rows its frame would have seen (omni_index_search_prefix_dev) -- fetched with one copy, and the decision rules are then replayed
        frame by frame through the unchanged _on_image_recv.  Records are identical to calling on_image_recv per frame.
        rows_dev: optional device pointer to the frames' global descriptors, [sum of len(f.images)][4096] fp32 in frame order (e.g.
        MobileNetVLAD's output buffer, still in HBM): rows are then appended and queried without touching the host copies.  ORDERING: the
        index works on its own context's stream -- rows_dev must be complete before this call (LoopCam.fetch() has waited for the
        MobileNetVLAD stream), there is no cross-stream wait inside."""
        with self._mu:
            if not all(hasattr(ix, "search_batch_prefix_dev") and hasattr(ix, "add_dev") for ix in (self.local_index, self.remote_index)):
                return [self._on_image_recv(f) for f in frames]        # e.g. a sharded index factory: frame by frame
            return self._recv_batch(frames, rows_dev)

    def _recv_batch(self, frames, rows_dev):
        ctx = self.local_index.ctx
        # ---- plan: the gating of _on_image_recv with simulated index sizes (no GPU work)
        sim_local, sim_remote = self.local_index.ntotal, self.remote_index.ntotal
        nodes = set(self.all_nodes)
        adds, searches, row_ids = [], [], []      # adds: (index, flat image row); searches: (index, flat row, max_index, n_limit)
        base = 0
        for f in frames:
            first = base
            base += len(f.images)
            if len(f.images) == 0 or (f.drone_id != self.self_id and sim_local + sim_remote == 0):
                continue
            new_node = f.drone_id not in nodes
            nodes.add(f.drone_id)
            if sum(1 for img in f.images if img.landmark_num > 0) < self.MIN_DIRECTION_LOOP or f.landmark_num < self.MIN_LOOP_NUM:
                continue
            if (not f.prevent_adding_db) or new_node:
                for i, img in enumerate(f.images):
                    if img.landmark_num > 0:
                        if img.drone_id == self.self_id:
                            adds.append((self.local_index, first + i)); row_ids.append(sim_local); sim_local += 1
                        else:
                            adds.append((self.remote_index, first + i)); row_ids.append(sim_remote + REMOTE_MAGIN_NUMBER); sim_remote += 1
            # `database_size() > MATCH_INDEX_DIST || init_mode || drone_id != self_id`: init_mode implies a remote drone
            if sim_local + sim_remote > self.MATCH_INDEX_DIST or f.drone_id != self.self_id:
                d = 1 if self.camera_configuration == STEREO_FISHEYE else 0
                img = f.images[d] if len(f.images) > d else None
                if img is not None and img.landmark_num > 0:
                    if img.drone_id == self.self_id:
                        searches.append((self.remote_index, first + d, 1, sim_remote))
                        if not f.prevent_adding_db:
                            searches.append((self.local_index, first + d, self.MATCH_INDEX_DIST, sim_local))
                    else:
                        searches.append((self.local_index, first + d, 1, sim_local))
        # ---- enqueue: rows to HBM (unless they are there already), appends, prefix searches, one result copy
        row_bytes = DEEP_DESC_SIZE * 4
        own_rows = None
        if rows_dev is None and (adds or searches):
            # a direction that was never received carries an empty image_desc (the ImageDescriptor default; loop_net's remote frames):
            # its row is never appended or queried (landmark_num == 0), it only keeps the frame-major row numbering -- zero-filled
            stage = np.zeros((base, DEEP_DESC_SIZE), np.float32)
            r = 0
            for f in frames:
                for img in f.images:
                    d = np.asarray(img.image_desc, np.float32).reshape(-1)
                    if d.size == DEEP_DESC_SIZE:
                        stage[r] = d
                    r += 1
            own_rows = rows_dev = ctx.to_device(stage)
        start_local, start_remote = self.local_index.ntotal, self.remote_index.ntotal
        try:
            a = 0
            while a < len(adds):                                # consecutive rows of one index go in as one append
                b = a + 1
                while b < len(adds) and adds[b][0] is adds[a][0] and adds[b][1] == adds[b - 1][1] + 1:
                    b += 1
                adds[a][0].add_dev(b - a, rows_dev + adds[a][1] * row_bytes)
                a = b
            # searches of one index with one k share ONE pass over that index (omni_index_search_batch_prefix_dev, <= 64 queries per
            # pass): every query still only sees the rows of its own turn, but the database is read once per micro-batch, not per frame
            live = [j for j, s in enumerate(searches) if s[3] > 0]
            groups = {}
            for j in live:
                index, row, max_index, n_limit = searches[j]
                groups.setdefault((id(index), max_index), (index, max_index, []))[2].append(j)
            chunks, need = [], 0                                 # (index, k, [search j], I offset, D offset)
            for index, max_index, js in groups.values():
                k = SEARCH_NEAREST_NUM + max_index
                for c0 in range(0, len(js), 64):
                    part = js[c0:c0 + 64]
                    chunks.append((index, k, part, need, need + len(part) * k * 8))
                    need += len(part) * k * 12
            where = {}
            if chunks:
                if self._batch_bufs is None or self._batch_bufs[1] < need:
                    if self._batch_bufs is not None:
                        ctx.free(self._batch_bufs[0])
                    self._batch_bufs = (ctx.alloc(need), need)
                buf = self._batch_bufs[0]
                for index, k, part, off_i, off_d in chunks:
                    index.search_batch_prefix_dev(rows_dev, [searches[j][1] for j in part], k, [searches[j][3] for j in part],
                                                  buf + off_d, buf + off_i)
                    for pos, j in enumerate(part):
                        where[j] = (k, off_i + pos * k * 8, off_d + pos * k * 4)
                raw = ctx.from_device(buf, (need,), np.uint8)                               # the batch's only synchronisation
            elif adds:
                ctx.sync()
        except Exception:
            # nothing has been recorded in the id maps yet: drop the rows appended ahead so that ntotal (and with it the recency rule and
            # every later row id) stays what the frame-by-frame path would have
            self._rollback(start_local, start_remote)
            raise
        finally:
            if own_rows is not None:
                ctx.free(own_rows)
        results = []
        for j, (index, row, max_index, n_limit) in enumerate(searches):
            k = SEARCH_NEAREST_NUM + max_index
            if j in where:
                _, oi, od = where[j]
                results.append((raw[od:od + k * 4].view(np.float32).reshape(1, k), raw[oi:oi + k * 8].view(np.int64).reshape(1, k), n_limit))
            else:                                            # faiss pads an empty index's result with -1 labels
                results.append((np.full((1, k), -3.4028235e38, np.float32), np.full((1, k), -1, np.int64), 0))
        # ---- replay the decision rules frame by frame on the fetched results.  A compute_loop that raises must not leave rows in the
        # index without their id-map entries: the exception is deferred until every frame's bookkeeping has been applied.
        self._deferred, self._row_ids, self._sim_local, self._sim_remote = results, row_ids, start_local, start_remote
        pending_exc = []
        user_cb = self.compute_loop

        def guarded(*a):
            try:
                return user_cb(*a)
            except Exception as e:                           # noqa: BLE001 -- re-raised below
                pending_exc.append(e)
                return False

        self.compute_loop = guarded
        try:
            out = [self._on_image_recv(f) for f in frames]
            assert not self._deferred and not self._row_ids, "plan and replay diverged"
        finally:
            self._deferred = None
            self.compute_loop = user_cb
        if pending_exc:
            raise pending_exc[0]
        return out

    def _rollback(self, n_local: int, n_remote: int):
        for index, n in ((self.local_index, n_local), (self.remote_index, n_remote)):
            if index.ntotal > n and hasattr(index, "truncate"):
                index.truncate(n)

    # ---- checkpoint / resume (new: the reference's key-frame database lives in RAM only) --------------------
    def save(self, prefix: str):
        """Everything on_image_recv depends on: the two index shards (OMNX1 snapshots, omni_index_save) and `prefix`.state.npz with the
        id maps, inter-drone loop counters, known nodes and the stored frames (global + local descriptors, key points).  Plain arrays
        only (no pickle)."""
        with self._mu:
            self.local_index.save(prefix + ".local.omnx")
            self.remote_index.save(prefix + ".remote.omnx")
            ids = sorted(self.imgid2fisheye)
            frames = [self.fisheyeframe_database[k] for k in sorted(self.fisheyeframe_database)]
            imgs = [im for f in frames for im in f.images]
            feat = [np.asarray(im.feature_descriptor, np.float32).reshape(-1) for im in imgs]
            kps = [np.asarray(im.landmarks_2d, np.float32).reshape(-1) for im in imgs]
            pairs = sorted(self.inter_drone_loop_count)
            np.savez(prefix + ".state.npz",
                     params=np.array([self.self_id, self.camera_configuration], np.int64),
                     imgid=np.array(ids, np.int64), imgid_fisheye=np.array([self.imgid2fisheye[i] for i in ids], np.int64),
                     imgid_dir=np.array([self.imgid2dir[i] for i in ids], np.int64),
                     loop_pairs=np.array(pairs, np.int64).reshape(-1, 2),
                     loop_counts=np.array([self.inter_drone_loop_count[p] for p in pairs], np.int64),
                     nodes=np.array(sorted(self.all_nodes), np.int64),
                     frame_meta=np.array([[f.msg_id, f.drone_id, f.landmark_num, int(f.prevent_adding_db), len(f.images)] for f in frames],
                                         np.int64).reshape(-1, 5),
                     img_meta=np.array([[im.drone_id, im.landmark_num] for im in imgs], np.int64).reshape(-1, 2),
                     img_desc=self._stack_desc(imgs),
                     img_desc_len=np.array([np.asarray(im.image_desc).size for im in imgs], np.int64),
                     feat_dim=np.array([(np.asarray(im.feature_descriptor).shape[1] if np.asarray(im.feature_descriptor).ndim == 2
                                         and np.asarray(im.feature_descriptor).size else 64) for im in imgs], np.int64),
                     det_params=np.array([self.INNER_PRODUCT_THRES, self.INIT_MODE_PRODUCT_THRES, self.MATCH_INDEX_DIST, self.MIN_LOOP_NUM,
                                          self.MIN_DIRECTION_LOOP, self.inter_drone_init_frames], np.float64),
                     feat=np.concatenate(feat) if feat else np.zeros(0, np.float32),
                     feat_len=np.array([len(x) for x in feat], np.int64),
                     kps=np.concatenate(kps) if kps else np.zeros(0, np.float32),
                     kps_len=np.array([len(x) for x in kps], np.int64))

    @staticmethod
    def _stack_desc(imgs):
        out = np.zeros((len(imgs), DEEP_DESC_SIZE), np.float32)
        for i, im in enumerate(imgs):
            d = np.asarray(im.image_desc, np.float32).reshape(-1)
            if d.size == DEEP_DESC_SIZE:
                out[i] = d
        return out

    def load(self, prefix: str):
        """Restores a save() into this (freshly constructed, same parameters) detector."""
        with self._mu:
            z = np.load(prefix + ".state.npz", allow_pickle=False)
            if int(z["params"][0]) != self.self_id or int(z["params"][1]) != self.camera_configuration:
                raise ValueError("snapshot was written by a detector with another self_id / camera configuration")
            mine = np.array([self.INNER_PRODUCT_THRES, self.INIT_MODE_PRODUCT_THRES, self.MATCH_INDEX_DIST, self.MIN_LOOP_NUM,
                             self.MIN_DIRECTION_LOOP, self.inter_drone_init_frames], np.float64)
            if "det_params" in z and not np.array_equal(z["det_params"], mine):
                raise ValueError(f"snapshot was written with detector parameters {z['det_params'].tolist()}, this detector has {mine.tolist()}")
            desc_len = z["img_desc_len"] if "img_desc_len" in z else np.full(len(z["img_meta"]), DEEP_DESC_SIZE, np.int64)
            feat_dim = z["feat_dim"] if "feat_dim" in z else np.full(len(z["img_meta"]), 64, np.int64)
            self.local_index.load(prefix + ".local.omnx")
            self.remote_index.load(prefix + ".remote.omnx")
            self.imgid2fisheye = {int(i): int(f) for i, f in zip(z["imgid"], z["imgid_fisheye"])}
            self.imgid2dir = {int(i): int(d) for i, d in zip(z["imgid"], z["imgid_dir"])}
            self.inter_drone_loop_count = {(int(a), int(b)): int(c) for (a, b), c in zip(z["loop_pairs"], z["loop_counts"])}
            self.all_nodes = {int(n) for n in z["nodes"]}
            self.fisheyeframe_database = {}
            fo = np.concatenate([[0], np.cumsum(z["feat_len"])])
            ko = np.concatenate([[0], np.cumsum(z["kps_len"])])
            j = 0
            for msg_id, drone_id, lm, prevent, n_img in z["frame_meta"]:
                images = []
                for _ in range(int(n_img)):
                    images.append(ImageDescriptor(drone_id=int(z["img_meta"][j, 0]), landmark_num=int(z["img_meta"][j, 1]),
                                                  image_desc=(z["img_desc"][j].copy() if int(desc_len[j]) == DEEP_DESC_SIZE
                                                              else np.zeros(0, np.float32)),
                                                  feature_descriptor=z["feat"][fo[j]:fo[j + 1]].reshape(-1, int(feat_dim[j])).copy(),
                                                  landmarks_2d=z["kps"][ko[j]:ko[j + 1]].reshape(-1, 2).copy()))
                    j += 1
                self.fisheyeframe_database[int(msg_id)] = FisheyeFrameDescriptor(msg_id=int(msg_id), drone_id=int(drone_id), landmark_num=int(lm),
                                                                                prevent_adding_db=bool(prevent), images=images)

