"""Loop-closure matcher oracle (CPU) -- test infrastructure only.

Restates, from /root/reference/swarm_loop/src/loop_detector.cpp:
  IndexFlatIPRef                      faiss::IndexFlatIP add/search/ntotal as used at :166,169,213
                                      (faiss is un-vendored: PARITY UNPINNED; exact IP, ties -> lower row)
  LoopDetectorRef.add_to_database     :150-173
  LoopDetectorRef.query_from_database :176-242  (incl. the fall-through quirk at :241 and the shared
                                                 ``distance`` variable at :184-186)
  LoopDetectorRef.query_fisheyeframe_from_database :245-287
  LoopDetectorRef.on_image_recv       :11-137   (gating only; compute_loop is geometry, a callback here)
  bf_match                            cv::BFMatcher(NORM_L2, crossCheck=true).match (:564-567,
                                      loop_cam.cpp:147-150) via oracle/csrc/oracle.c
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field

import numpy as np

from . import postproc_ref as _P

REMOTE_MAGIN_NUMBER = 1000000   # loop_detector.h:22 (sic)
SEARCH_NEAREST_NUM = 5          # loop_defines.h:32
DEEP_DESC_SIZE = 4096           # loop_defines.h:30
FEATURE_DESC_SIZE = 64          # loop_defines.h:67


def ip_search(db: np.ndarray, q: np.ndarray, k: int):
    """Exact inner-product top-k, descending, ties -> lower row, missing -> (-FLT_MAX, -1)."""
    db = np.ascontiguousarray(db, np.float32)
    q = np.ascontiguousarray(np.atleast_2d(q), np.float32)
    nq, d = q.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    n = db.shape[0] if db.size else 0
    dbp = db if n else np.zeros((1, d), np.float32)
    _P.lib().oracle_ip_search(_P._fp(dbp), n, d, _P._fp(q), nq, k, _P._fp(D),
                              I.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return D, I


def ip_search_numpy(db: np.ndarray, q: np.ndarray, k: int):
    """Same contract with a BLAS matmul (the cpu_baseline stand-in for faiss's sgemv)."""
    q = np.atleast_2d(np.asarray(q, np.float32))
    n = db.shape[0]
    s = q @ db.T if n else np.zeros((q.shape[0], 0), np.float32)
    D = np.full((q.shape[0], k), -3.402823466e+38, np.float32)
    I = np.full((q.shape[0], k), -1, np.int64)
    kk = min(k, n)
    for qi in range(q.shape[0]):
        order = np.lexsort((np.arange(n), -s[qi].astype(np.float64)))[:kk]
        D[qi, :kk] = s[qi, order]
        I[qi, :kk] = order
    return D, I


def ip_search_blocked(blocks, q: np.ndarray, k: int, extra: int = 32):
    """The same contract (exact inner-product top-k, descending, ties -> lower row) for databases too big for the scalar C loop: `blocks`
    yields (first_row, rows[float32]) in any order, each block scored with one BLAS sgemm, its k + extra best per query re-scored in
    float64 (the sgemm's own summation noise, ~1e-6, is far below the score gap over `extra` ranks of a random database), then merged by
    (score desc, row asc).  Returns (D float32, I int64, S float64): S = the float64 scores, for the near-tie rule of the caller."""
    q = np.ascontiguousarray(np.atleast_2d(q), np.float32)
    nq = q.shape[0]
    q64 = q.astype(np.float64)
    cand_s = [[] for _ in range(nq)]
    cand_i = [[] for _ in range(nq)]
    for row0, blk in blocks:
        blk = np.ascontiguousarray(blk, np.float32)
        n = blk.shape[0]
        if n == 0:
            continue
        s = q @ blk.T                                      # [nq][n] fp32
        kk = min(n, k + extra)
        part = np.argpartition(-s, kk - 1, axis=1)[:, :kk] if kk < n else np.tile(np.arange(n), (nq, 1))
        for qi in range(nq):
            rows = np.sort(part[qi])
            cand_s[qi].append(blk[rows].astype(np.float64) @ q64[qi])
            cand_i[qi].append(rows.astype(np.int64) + int(row0))
    D = np.full((nq, k), -3.402823466e+38, np.float32)
    I = np.full((nq, k), -1, np.int64)
    S = np.full((nq, k), -np.inf, np.float64)
    for qi in range(nq):
        if not cand_s[qi]:
            continue
        sc, ids = np.concatenate(cand_s[qi]), np.concatenate(cand_i[qi])
        order = np.lexsort((ids, -sc))[:k]
        D[qi, :len(order)] = sc[order].astype(np.float32)
        I[qi, :len(order)] = ids[order]
        S[qi, :len(order)] = sc[order]
    return D, I, S


def bf_match(q: np.ndarray, t: np.ndarray, mode: int = 0):
    """-> (query_idx, train_idx, distance); mode 0 = OpenCV batchDistance crosscheck, 1 = strict mutual NN."""
    q = np.ascontiguousarray(q, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    nq, nt = q.shape[0], t.shape[0]
    qi = np.zeros(max(nq, 1), np.int32)
    ti = np.zeros(max(nq, 1), np.int32)
    dd = np.zeros(max(nq, 1), np.float32)
    if nq == 0 or nt == 0:
        return qi[:0], ti[:0], dd[:0]
    n = _P.lib().oracle_bf_match(_P._fp(q), nq, _P._fp(t), nt, q.shape[1], mode, _P._ip(qi), _P._ip(ti), _P._fp(dd))
    return qi[:n].copy(), ti[:n].copy(), dd[:n].copy()


class IndexFlatIPRef:
    def __init__(self, d: int):
        self.d = d
        self.rows: list[np.ndarray] = []

    @property
    def ntotal(self) -> int:
        return len(self.rows)

    def add(self, x: np.ndarray):
        for r in np.atleast_2d(np.asarray(x, np.float32)):
            self.rows.append(r.copy())

    def matrix(self) -> np.ndarray:
        return np.stack(self.rows) if self.rows else np.zeros((0, self.d), np.float32)

    def search(self, q: np.ndarray, k: int, use_numpy=False):
        f = ip_search_numpy if use_numpy else ip_search
        return f(self.matrix(), q, k)


# --- message PODs (swarm_msgs is un-vendored; fields as used in loop_cam.cpp:529-551, loop_detector.cpp) ----
@dataclass
class ImageDesc:
    drone_id: int = 0
    landmark_num: int = 0
    image_desc: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))   # 4096
    feature_descriptor: np.ndarray = field(default_factory=lambda: np.zeros((0, FEATURE_DESC_SIZE), np.float32))
    landmarks_2d: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))


@dataclass
class FisheyeFrameDesc:
    msg_id: int = 0
    drone_id: int = 0
    landmark_num: int = 0
    prevent_adding_db: bool = False
    images: list = field(default_factory=list)


STEREO_PINHOLE, STEREO_FISHEYE, PINHOLE_DEPTH = 0, 1, 2   # loop_defines.h:110-115


class LoopDetectorRef:
    """DB + decision rules of LoopDetector; ``compute_loop`` (geometry, :627-836) is a callback
    ``(new_frame, old_frame, dir_new, dir_old, init_mode) -> bool``."""

    def __init__(self, self_id: int, *, inner_product_thres=0.6, init_mode_product_thres=0.3,
                 match_index_dist=10, min_loop_num=15, min_direction_loop=3, inter_drone_init_frames=50,
                 camera_configuration=STEREO_FISHEYE, compute_loop=None, use_numpy_search=False):
        self.self_id = self_id
        self.INNER_PRODUCT_THRES = inner_product_thres        # swarm_loop.cpp:236
        self.INIT_MODE_PRODUCT_THRES = init_mode_product_thres  # :237
        self.MATCH_INDEX_DIST = match_index_dist              # :222
        self.MIN_LOOP_NUM = min_loop_num                      # :223
        self.MIN_DIRECTION_LOOP = min_direction_loop
        self.inter_drone_init_frames = inter_drone_init_frames
        self.camera_configuration = camera_configuration
        self.local_index = IndexFlatIPRef(DEEP_DESC_SIZE)
        self.remote_index = IndexFlatIPRef(DEEP_DESC_SIZE)
        self.imgid2fisheye: dict[int, int] = {}
        self.imgid2dir: dict[int, int] = {}
        self.fisheyeframe_database: dict[int, FisheyeFrameDesc] = {}
        self.inter_drone_loop_count: dict[tuple[int, int], int] = {}
        self.all_nodes: set[int] = set()
        self.compute_loop = compute_loop or (lambda *a: False)
        self.use_numpy_search = use_numpy_search
        self.log: list[dict] = []     # one record per on_image_recv: the match-ID parity trace

    def database_size(self):                                   # :290-292
        return self.local_index.ntotal + self.remote_index.ntotal

    def _add_image(self, img: ImageDesc) -> int:               # :164-173
        if img.drone_id == self.self_id:
            self.local_index.add(img.image_desc)
            return self.local_index.ntotal - 1
        self.remote_index.add(img.image_desc)
        return self.remote_index.ntotal - 1 + REMOTE_MAGIN_NUMBER

    def add_to_database(self, frame: FisheyeFrameDesc) -> int:  # :150-162
        for i, img in enumerate(frame.images):
            if img.landmark_num > 0:
                index = self._add_image(img)
                self.imgid2fisheye[index] = frame.msg_id
                self.imgid2dir[index] = i
        self.fisheyeframe_database[frame.msg_id] = frame
        return frame.msg_id

    def _query_index(self, img: ImageDesc, index: IndexFlatIPRef, remote_db: bool, thres: float,
                     max_index: int, distance: list) -> int:    # :199-242 ; distance = [value] (by reference)
        index_offset = REMOTE_MAGIN_NUMBER if remote_db else 0
        search_num = SEARCH_NEAREST_NUM + max_index
        D, I = index.search(img.image_desc, search_num, use_numpy=self.use_numpy_search)
        distances, labels = D[0], I[0]
        return_msg_id = -1
        for i in range(search_num):
            if labels[i] < 0:
                continue
            if int(labels[i]) + index_offset not in self.imgid2fisheye:
                continue
            return_msg_id = int(labels[i]) + index_offset
            if labels[i] <= index.ntotal - max_index and float(distances[i]) > thres:
                distance[0] = float(distances[i])
                return return_msg_id
        return return_msg_id                                    # :241 fall-through: last examined label

    def query_from_database(self, img: ImageDesc, init_mode: bool, nonkeyframe: bool, distance: list) -> int:
        thres = self.INIT_MODE_PRODUCT_THRES if init_mode else self.INNER_PRODUCT_THRES   # :177-180
        if img.drone_id == self.self_id:                        # :182-190
            _id = self._query_index(img, self.remote_index, True, thres, 1, distance)
            if not nonkeyframe:
                return self._query_index(img, self.local_index, False, thres, self.MATCH_INDEX_DIST, distance)
            elif _id != -1:
                return _id
        else:                                                   # :191-195
            return self._query_index(img, self.local_index, False, thres, 1, distance)
        return -1

    def query_fisheyeframe_from_database(self, frame: FisheyeFrameDesc, init_mode: bool, nonkeyframe: bool):
        """-> (old_frame or None, direction_new, direction_old, best_image_id, distance)   (:245-287)"""
        direction_new = 1 if self.camera_configuration == STEREO_FISHEYE else 0
        if frame.images[direction_new].landmark_num > 0:
            distance = [-1.0]
            _id = self.query_from_database(frame.images[direction_new], init_mode, nonkeyframe, distance)
            if _id != -1 and distance[0] > -1:
                return (self.fisheyeframe_database[self.imgid2fisheye[_id]], direction_new,
                        self.imgid2dir[_id], _id, distance[0])
        return None, direction_new, -1, -1, -1.0

    def on_image_recv(self, frame: FisheyeFrameDesc):           # :11-137
        rec = {"msg_id": frame.msg_id, "drone_id": frame.drone_id, "added": False, "queried": False,
               "image_id": -1, "old_msg_id": -1, "dir_new": -1, "dir_old": -1, "loop": False}
        self.log.append(rec)
        if len(frame.images) == 0:
            return rec
        drone_id = frame.drone_id
        if drone_id != self.self_id and self.database_size() == 0:   # :36-38
            return rec
        new_node = frame.drone_id not in self.all_nodes
        self.all_nodes.add(frame.drone_id)
        dir_count = sum(1 for img in frame.images if img.landmark_num > 0)
        if dir_count < self.MIN_DIRECTION_LOOP:                 # :60-63
            return rec
        if frame.landmark_num < self.MIN_LOOP_NUM:              # :65,130-132
            return rec
        init_mode = False
        if drone_id != self.self_id:                            # :67-72
            init_mode = self.inter_drone_loop_count.get((drone_id, self.self_id), 0) < self.inter_drone_init_frames
        if (not frame.prevent_adding_db) or new_node:           # :89-94
            self.add_to_database(frame)
            rec["added"] = True
        if self.database_size() > self.MATCH_INDEX_DIST or init_mode or drone_id != self.self_id:   # :98
            rec["queried"] = True
            old, d_new, d_old, image_id, dist = self.query_fisheyeframe_from_database(
                frame, init_mode, frame.prevent_adding_db)
            if d_old >= 0:
                rec.update(image_id=image_id, old_msg_id=old.msg_id, dir_new=d_new, dir_old=d_old, distance=dist)
                success = False
                if old.drone_id == self.self_id:                # :110-111
                    success = self.compute_loop(frame, old, d_new, d_old, init_mode)
                    pair = (frame.drone_id, old.drone_id)
                elif frame.drone_id == self.self_id:            # :114-115
                    success = self.compute_loop(old, frame, d_old, d_new, init_mode)
                    pair = (old.drone_id, frame.drone_id)
                if success:                                     # :826-827
                    a, b = pair
                    self.inter_drone_loop_count[(a, b)] = self.inter_drone_loop_count.get((a, b), 0) + 1
                    self.inter_drone_loop_count[(b, a)] = self.inter_drone_loop_count.get((b, a), 0) + 1
                    rec["loop"] = True
        return rec
