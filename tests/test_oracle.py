"""CPU tests of the oracle itself: restatements agree with each other and with the committed golden vectors
(tools/gen_golden.py pinned those against the reference's own PyTorch module in the build container)."""
import os

import numpy as np
import pytest

from oracle import match_ref as M
from oracle import mobilenetvlad_ref as V
from oracle import postproc_ref as P
from oracle import superpoint_ref as S
from omni_swarm_amd import synth          # seeded synthetic inputs (data generators; shared by bench.py)
from tests import detector_stream as DS


def test_nms2_literal_c_equals_python_equals_characterised():
    rng = np.random.default_rng(0)
    for t in range(40):
        h, w = int(rng.integers(9, 48)), int(rng.integers(9, 64))
        prob = rng.random((h, w)).astype(np.float32) ** 5
        if t % 3 == 0:
            prob = (np.round(prob * 16) / 16).astype(np.float32)     # many exact ties
        thr = [0.05, 0.2, 0.5][t % 3]
        a = P.get_keypoints(prob, thr, 64)
        b = P.get_keypoints_py_literal(prob, thr, 64)
        c = P.get_keypoints_characterised(prob, thr, 64)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[0], c[0])
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[1], c[1])


def test_nms2_is_order_dependent():
    # confs .3, .5, .9 spaced 3 px apart along the scan: only .9 survives one way, {.9,.3} the other
    prob = np.zeros((9, 24), np.float32)
    prob[4, 6], prob[4, 9], prob[4, 12] = 0.3, 0.5, 0.9
    xy, conf, _, _ = P.get_keypoints(prob, 0.1, 10)
    assert {tuple(p) for p in xy.tolist()} == {(12, 4)}
    prob2 = prob[:, ::-1].copy()
    xy2, conf2, _, _ = P.get_keypoints(prob2, 0.1, 10)
    assert {tuple(p) for p in xy2.tolist()} == {(11, 4), (17, 4)}


def test_nms2_equal_confidences_do_not_suppress_and_order_is_fixed():
    prob = np.zeros((16, 16), np.float32)
    prob[5, 5] = prob[5, 6] = prob[6, 5] = 0.7
    xy, conf, nc, ns = P.get_keypoints(prob, 0.5, 10)
    assert ns == 3 and xy.tolist() == [[5, 5], [6, 5], [5, 6]]      # conf ties -> row-major index ascending
    xy1, _, _, _ = P.get_keypoints(prob, 0.5, 2)
    assert xy1.tolist() == [[5, 5], [6, 5]]


def test_nms2_column_wrap_quirk_only_matters_at_the_border():
    rng = np.random.default_rng(3)
    prob = rng.random((32, 40)).astype(np.float32) ** 4
    interior = prob.copy()
    interior[:, :4] = 0
    interior[:, -4:] = 0
    a = P.get_keypoints(interior, 0.3, 100, wrap_columns=False)
    b = P.get_keypoints(interior, 0.3, 100, wrap_columns=True)
    assert np.array_equal(a[0], b[0])


def test_superpoint_oracle_matches_golden_small(golden):
    g = golden("sp_small.npz")
    w = S.synth_weights(0)
    semi, desc = S.forward(w, S.preprocess_u8(g["image"]))
    # golden came from the reference notebook module on the build container; allow cross-CPU kernel differences
    assert np.abs(semi[0] - g["semi"]).max() < 2e-6
    assert np.abs(desc[0] - g["desc"]).max() < 2e-5
    xy, conf, nc, ns = P.get_keypoints(g["semi"], float(g["thres"]), 200)
    assert np.array_equal(xy, g["kps"]) and np.array_equal(conf, g["conf"])
    comp, mean = synth.pca()
    d64, d256 = P.compute_descriptors(g["desc"], xy, 96, 64, comp, mean)
    assert np.allclose(d64, g["desc64"], atol=1e-5) and np.allclose(d256, g["desc256"], atol=1e-5)


def test_descriptor_normalisation_is_per_channel_across_keypoints(golden):
    # superpoint_tensorrt.cpp:211-215: torch::norm(desc[256,n], 2, dim=1) -> every CHANNEL has unit norm over the key points
    g = golden("sp_small.npz")
    d256 = g["desc256"]
    assert np.allclose(np.linalg.norm(d256, axis=0), 1.0, atol=1e-4)
    assert not np.allclose(np.linalg.norm(d256, axis=1), 1.0, atol=1e-2)


def test_preprocess_is_opencv_convert_to():
    u = np.arange(256, dtype=np.uint8)[None]
    x = S.preprocess_u8(u)
    assert x.dtype == np.float32 and x[0, 255] == np.float32(1.0) and x[0, 0] == 0
    assert np.array_equal(x[0], u[0].astype(np.float32) * np.float32(1.0 / 255.0))          # OpenCV 3.4 cvt_32f: float scale
    img = np.full((8, 4), 9, np.uint8)
    assert (S.preprocess_u8(img, fisheye_mask=True)[6:] == 0).all() and (S.preprocess_u8(img, True)[:6] > 0).all()


def test_ip_search_c_equals_numpy_and_tie_rule():
    db = synth.global_db(500, dim=512, seed=9)
    db[77] = db[13]
    db[400] = db[13]
    q = db[[13, 250]] * np.float32(1.0)
    D1, I1 = M.ip_search(db, q, 7)
    D2, I2 = M.ip_search_numpy(db, q, 7)
    assert np.array_equal(I1, I2) and np.allclose(D1, D2, atol=1e-6)
    assert I1[0, :3].tolist() == [13, 77, 400]                       # ties -> lower row first
    D3, I3 = M.ip_search(db[:3], q, 5)
    assert (I3[:, 3:] == -1).all() and (D3[:, 3:] < -1e38).all()     # k > n padding (faiss semantics)


def test_blocked_ip_search_equals_the_scalar_oracle():
    """ip_search_blocked (BLAS per block + float64 re-scoring: the oracle of the full-size GPU index tests) == the scalar C loop, incl. ties across
    blocks -> lower row, k > rows of a block, blocks handed over out of order, and the golden database."""
    db = synth.global_db(3000, seed=3)
    db[2900] = db[41]
    db[1200] = db[41]
    q, rows = synth.queries_from_db(db, 6, seed=4)
    q[5] = db[41]
    Dr, Ir = M.ip_search(db, q, 12)
    for edges in ([0, 3000], [0, 700, 1400, 3000], [0, 5, 2999, 3000]):
        blocks = [(a, db[a:b]) for a, b in zip(edges[:-1], edges[1:])]
        D, I, S = M.ip_search_blocked(iter(blocks[::-1]), q, 12)
        assert np.array_equal(I, Ir) and np.allclose(D, Dr, rtol=1e-6, atol=1e-7) and np.allclose(S, Dr, rtol=1e-6, atol=1e-7)
    assert I[5, :3].tolist() == [41, 1200, 2900]
    D, I, S = M.ip_search_blocked(iter([(0, db[:4])]), q[:2], 9)
    assert (I[:, 4:] == -1).all() and (D[:, 4:] < -1e38).all()


def test_ip_search_matches_golden(golden):
    g = golden("match.npz")
    db = synth.global_db(3000, seed=3)
    q, rows = synth.queries_from_db(db, 8, seed=4)
    D, I = M.ip_search(db, q, 10)
    assert np.array_equal(I, g["ip_I"]) and np.allclose(D, g["ip_D"], atol=1e-6) and np.array_equal(rows, g["ip_rows"])


def test_bf_match_opencv_semantics_differs_from_strict_mutual():
    # d(a1,b1)=1, d(a2,b1)=.5, d(a1,b2)=2, d(a2,b2)=3: OpenCV's crosscheck still pairs a1 with b2
    a = np.array([[0.0, 0], [1.5, 0]], np.float32)
    b = np.array([[1.0, 0], [-2.0, 0]], np.float32)
    q0, t0, d0 = M.bf_match(a, b, 0)
    q1, t1, d1 = M.bf_match(a, b, 1)
    assert list(zip(q0, t0)) == [(0, 1), (1, 0)] and np.allclose(d0, [2.0, 0.5])
    assert list(zip(q1, t1)) == [(1, 0)]


def test_bf_match_golden_and_first_minimum_wins(golden):
    g = golden("match.npz")
    q0, t0, d0 = M.bf_match(g["bf_a"], g["bf_b"], 0)
    assert np.array_equal(q0, g["bf0_q"]) and np.array_equal(t0, g["bf0_t"]) and np.array_equal(d0, g["bf0_d"])
    a = np.zeros((3, 8), np.float32)
    b = np.zeros((2, 8), np.float32)            # all distances equal -> query 0 gets train 0 only
    q, t, d = M.bf_match(a, b, 0)
    assert q.tolist() == [0] and t.tolist() == [0]
    assert M.bf_match(a[:0], b, 0)[0].size == 0


def test_detector_trace_matches_golden(golden):
    log = DS.run_oracle(DS.make_stream(seed=11))
    tr = DS.trace(log)
    assert np.array_equal(tr, golden("detector.npz")["trace"])
    assert (tr[:, 4] >= 0).sum() > 20 and tr[:, 6].sum() > 5      # the stream does exercise candidates and loops


def test_detector_fall_through_quirk_is_restated():
    # remote hit + local miss: the shared `distance` leaks and the LAST examined local label is returned (:184-186,241)
    det = M.LoopDetectorRef(1, match_index_dist=1, min_loop_num=1, min_direction_loop=1)
    rng = np.random.default_rng(0)
    def unit():
        v = rng.standard_normal(4096).astype(np.float32)
        return v / np.linalg.norm(v)
    target = unit()
    def frame(mid, did, vec):
        img = M.ImageDesc(drone_id=did, landmark_num=50, image_desc=vec)
        return M.FisheyeFrameDesc(msg_id=mid, drone_id=did, landmark_num=200, images=[img, img, img, img])
    for i in range(4):
        det.on_image_recv(frame(10 + i, 1, unit()))
    det.on_image_recv(frame(50, 2, target))
    rec = det.on_image_recv(frame(60, 1, target * np.float32(0.999)))
    assert rec["queried"] and rec["image_id"] >= 0 and rec["image_id"] < M.REMOTE_MAGIN_NUMBER
    assert rec["distance"] > 0.9                 # the remote hit's score, attached to a local label


class _RefDetector:
    """oracle/_ref/libref_detector.so: the REFERENCE'S OWN text of LoopDetector::on_image_recv (loop_detector.cpp:11-137), add_to_database (x2),
    query_from_database (x2), query_fisheyeframe_from_database and database_size (:150-292), compiled verbatim against the stand-ins of
    oracle/ref_build/detector_shim.h (faiss::IndexFlatIP = oracle_ip_search, the geometry stage = a callback).  Built where /root/reference
    exists (oracle/Makefile), travels as a binary otherwise."""

    def __init__(self, self_id, verdict, *, inner_product_thres=0.6, init_mode_product_thres=0.3, match_index_dist=10, min_loop_num=15,
                 min_direction_loop=3, inter_drone_init_frames=50, camera_configuration=M.STEREO_FISHEYE):
        import ctypes
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"])
        path = os.path.join(root, "oracle", "_ref", "libref_detector.so")
        if not os.path.exists(path):
            pytest.skip("oracle/_ref/libref_detector.so not built (needs /root/reference once)")
        self.C = ctypes
        L = self.L = ctypes.CDLL(path)
        self._cb_t = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int64, ctypes.c_int64)
        self._cb = self._cb_t(lambda a, b: int(bool(verdict(a, b))))
        L.ref_det_create.restype = ctypes.c_void_p
        L.ref_det_create.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, self._cb_t]
        L.ref_det_destroy.argtypes = [ctypes.c_void_p]
        ip, fp, lp, dp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)
        L.ref_det_on_image_recv.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, ip, fp, lp]
        L.ref_det_query_index.argtypes = [ctypes.c_void_p, ctypes.c_int, fp, ctypes.c_double, ctypes.c_int, dp]
        L.ref_det_query_image.argtypes = [ctypes.c_void_p, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, dp]
        L.ref_det_database_size.argtypes = [ctypes.c_void_p]
        L.ref_det_inter_drone_loop_count.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        self.h = L.ref_det_create(self_id, inner_product_thres, init_mode_product_thres, match_index_dist, min_loop_num, min_direction_loop,
                                  inter_drone_init_frames, camera_configuration, self._cb)

    def close(self):
        if self.h:
            self.L.ref_det_destroy(self.h)
            self.h = None

    def on_image_recv(self, fr):
        """fr: a tests/detector_stream frame dict -> [added, queried, image_id, old_msg_id, dir_old, loop, database_size, compute_loop calls]"""
        C = self.C
        n = len(fr["images"])
        lm = np.array([i["landmark_num"] for i in fr["images"]], np.int32)
        did = np.array([i["drone_id"] for i in fr["images"]], np.int32)
        desc = np.ascontiguousarray(np.stack([i["image_desc"] for i in fr["images"]]), np.float32) if n else np.zeros((1, 4096), np.float32)
        out = np.zeros(8, np.int64)
        self.L.ref_det_on_image_recv(self.h, int(fr["msg_id"]), int(fr["drone_id"]), int(fr["landmark_num"]), int(fr["prevent_adding_db"]), n,
                                     lm.ctypes.data_as(C.POINTER(C.c_int)), did.ctypes.data_as(C.POINTER(C.c_int)),
                                     desc.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_int64)))
        return out

    def query_index(self, remote_db, vec, thres, max_index, distance):
        C = self.C
        d = C.c_double(distance)
        v = np.ascontiguousarray(vec, np.float32)
        r = self.L.ref_det_query_index(self.h, int(remote_db), v.ctypes.data_as(C.POINTER(C.c_float)), float(thres), int(max_index), C.byref(d))
        return r, d.value

    def query_image(self, drone_id, vec, init_mode, nonkeyframe, distance=-1.0):
        C = self.C
        d = C.c_double(distance)
        v = np.ascontiguousarray(vec, np.float32)
        r = self.L.ref_det_query_image(self.h, int(drone_id), v.ctypes.data_as(C.POINTER(C.c_float)), int(init_mode), int(nonkeyframe), C.byref(d))
        return r, d.value


@pytest.mark.parametrize("seed,n_drones,n_frames", [(11, 3, 90), (5, 5, 240), (23, 5, 240), (7, 3, 150)])
def test_detector_restatement_is_pinned_to_the_reference_text(seed, n_drones, n_frames):
    """oracle/match_ref.LoopDetectorRef -- what every detector parity test (Python product, C++ product, batched, GPU, the 10k-frame replay)
    is compared with -- against the reference's own on_image_recv / add_to_database / query_* text on multi-drone streams that exercise
    every branch (remote frames on an empty database, too few directions / landmarks, non-keyframes of known and new nodes, init mode ending
    after inter_drone_init_frames loops, remote and local hits, candidates whose old frame is remote): the same frames are added, the same
    frames query, the same candidate (row id, frame, direction) comes back, the same loops are accepted, the database has the same size."""
    frames = DS.make_stream(seed=seed, n_frames=n_frames, n_drones=n_drones)
    ref = _RefDetector(DS.SELF_ID, DS.loop_ok, **DS.PARAMS)
    tr = DS.trace(DS.run_oracle(frames))
    got = np.array([ref.on_image_recv(fr) for fr in frames])
    assert np.array_equal(got[:, :6], tr[:, 1:7]), np.nonzero((got[:, :6] != tr[:, 1:7]).any(1))[0][:10]
    det = M.LoopDetectorRef(DS.SELF_ID, compute_loop=lambda n, o, dn, do, im: DS.loop_ok(n.msg_id, o.msg_id), **DS.PARAMS)
    for fr, g in zip(DS._frames_ref(frames), got):
        det.on_image_recv(fr)
        assert det.database_size() == g[6]
    for (a, b), c in det.inter_drone_loop_count.items():
        assert ref.L.ref_det_inter_drone_loop_count(ref.h, a, b) == c
    assert (tr[:, 4] >= 0).sum() > 15 and tr[:, 6].sum() > 3
    ref.close()


def test_query_rule_is_pinned_to_the_reference_text_on_random_index_states():
    """query_from_database, both overloads (loop_detector.cpp:176-242), on 1000 random states of the two indexes: return value AND the in/out
    `distance` -- including the :241 fall-through (returns the last examined label, `distance` untouched) and the `distance` shared by the
    remote and the local query of a self frame (:184-186)."""
    rng = np.random.default_rng(99)
    n_fall, n_leak = 0, 0
    for trial in range(40):
        mid = int(rng.integers(1, 8))
        ref = _RefDetector(1, lambda a, b: 0, match_index_dist=mid, min_loop_num=1, min_direction_loop=1, inner_product_thres=0.5, init_mode_product_thres=0.25)
        det = M.LoopDetectorRef(1, match_index_dist=mid, min_loop_num=1, min_direction_loop=1, inner_product_thres=0.5, init_mode_product_thres=0.25)
        places = rng.standard_normal((6, 4096)).astype(np.float32)
        places /= np.linalg.norm(places, axis=1, keepdims=True)

        def vec(p):
            c = rng.uniform(0.3, 0.95)
            nz = rng.standard_normal(4096).astype(np.float32)
            v = c * places[p] + np.sqrt(1 - c * c) * nz / np.linalg.norm(nz)
            return (v / np.linalg.norm(v)).astype(np.float32)
        for f in range(int(rng.integers(0, 14))):               # fill both indexes through the front door
            drone = 1 if rng.random() < 0.6 else 2
            imgs = [{"drone_id": drone, "landmark_num": int(rng.integers(0, 3) > 0) * 50, "image_desc": vec(int(rng.integers(0, 6)))} for _ in range(4)]
            fr = {"msg_id": 100 + f, "drone_id": drone, "landmark_num": 200, "prevent_adding_db": False, "images": imgs}
            ref.on_image_recv(fr)
            det.on_image_recv(DS._frames_ref([fr])[0])
        for _ in range(25):
            q = vec(int(rng.integers(0, 6)))
            img = M.ImageDesc(drone_id=1, landmark_num=50, image_desc=q)
            thres, max_index, d0 = float(rng.uniform(0.1, 0.9)), int(rng.integers(1, 9)), float(rng.choice([-1.0, 0.77]))
            for remote in (0, 1):
                dist = [d0]
                r = det._query_index(img, det.remote_index if remote else det.local_index, bool(remote), thres, max_index, dist)
                rr, dd = ref.query_index(remote, q, thres, max_index, d0)
                assert (r, dist[0]) == (rr, dd), (trial, remote, r, rr, dist[0], dd)
                n_fall += int(r != -1 and dist[0] == d0)
            for drone, init_mode, nonkey in ((1, 0, 0), (1, 0, 1), (1, 1, 0), (2, 1, 0), (2, 0, 1)):
                dist = [-1.0]
                r = det.query_from_database(M.ImageDesc(drone_id=drone, landmark_num=50, image_desc=q), bool(init_mode), bool(nonkey), dist)
                rr, dd = ref.query_image(drone, q, init_mode, nonkey)
                assert (r, dist[0]) == (rr, dd), (trial, drone, init_mode, nonkey, r, rr, dist[0], dd)
                n_leak += int(drone == 1 and not nonkey and r != -1 and r < M.REMOTE_MAGIN_NUMBER and dist[0] > -1)
        ref.close()
    assert n_fall > 20 and n_leak > 20          # the quirks were exercised


def test_product_architecture_tables_equal_the_oracles_own():
    """The oracle states the two architectures itself (superpoint.ipynb:143-160; MobileNetV2 table at width 0.35 + NetVLAD head, assumed); the
    product's tables (omni-swarm_amd/weights.py, what the HIP side is built from) must say the same."""
    from omni_swarm_amd import weights as W
    assert S.LAYERS == W.SUPERPOINT_LAYERS
    assert V.BLOCKS == list(W.VLAD_BLOCKS) and V.layer_specs() == W.mobilenetvlad_layer_specs()
    assert (V.N_CLUSTERS, V.OUT_DIM, V.STEM_OUT, V.FEAT_DIM, V.VLAD_DIM) == (W.VLAD_N_CLUSTERS, W.VLAD_OUT_DIM, W.VLAD_STEM_OUT, W.VLAD_FEAT_DIM, W.VLAD_DIM)


def test_mobilenetvlad_oracle_golden(golden):
    g = golden("vlad_small.npz")
    y = V.forward(V.synth_weights(), g["images"])
    assert np.allclose(np.linalg.norm(y, axis=1), 1, atol=1e-5)
    assert np.abs(y - g["out"]).max() < 1e-5


def _ref_nms2():
    """oracle/_ref/libref_nms2.so: the REFERENCE'S OWN NMS2() text (superpoint_tensorrt.cpp:232-310) compiled verbatim against stand-in
    cv::Mat / cv::Point2f types (oracle/Makefile, oracle/ref_build/).  Built where /root/reference exists, travels as a binary otherwise."""
    import ctypes
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"])
    path = os.path.join(root, "oracle", "_ref", "libref_nms2.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_nms2.so not built (needs /root/reference once)")
    L = ctypes.CDLL(path)
    fp = ctypes.POINTER(ctypes.c_float)
    L.ref_nms2.restype = ctypes.c_int
    L.ref_nms2.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]

    def run(prob, thres, max_num, dist=4):
        """getKeyPoints (:164-189: mask = prob > thres, findNonZero row-major) feeding the compiled NMS2"""
        ys, xs = np.nonzero(prob > thres)                      # row-major, like cv::findNonZero
        xy = np.stack([xs, ys], 1).astype(np.float32)
        conf = np.ascontiguousarray(prob[ys, xs], np.float32)
        out = np.zeros((max(max_num, 1), 2), np.float32)
        n = L.ref_nms2(np.ascontiguousarray(xy).ctypes.data_as(fp), conf.ctypes.data_as(fp), len(xy), prob.shape[1], prob.shape[0], dist, max_num,
                       out.ctypes.data_as(fp))
        return out[:n].astype(np.int32)
    return run


def test_nms2_restatement_is_pinned_to_the_reference_text():
    """The literal restatement (oracle_nms2_literal, the checker of every key-point parity test) against the reference's own function compiled
    from its own source.  Interior-only maps (no candidate within 4 px of the frame: no out-of-bounds access in the reference) must agree
    exactly with the fixed-spec mode; with candidates at the left / right edge the reference's column wrap-around (contiguous cv::Mat rows)
    must agree with the restatement's wrap_columns mode -- the documented deviation of the fixed spec.  Confidences are distinct, so the
    reference's unstable std::sort has one valid result."""
    ref = _ref_nms2()
    rng = np.random.default_rng(42)
    for trial in range(30):
        H, W = int(rng.integers(24, 70)), int(rng.integers(24, 90))
        prob = np.zeros((H, W), np.float32)
        n = int(rng.integers(5, H * W // 6))
        ys, xs = rng.integers(4, H - 4, n), rng.integers(4, W - 4, n)
        prob[ys, xs] = rng.permutation(n).astype(np.float32)[: len(ys)] / n * 0.9 + 0.05        # distinct values in (0.05, 0.95]
        for thr, max_num in ((0.0, 10_000), (0.3, 10_000), (0.1, 12)):
            xy, conf, _, _ = P.get_keypoints(prob, thr, max_num)
            got = ref(prob, thr, max_num)
            assert np.array_equal(got, xy), (trial, thr, max_num)
    # left / right edges: reference = wrap-around, restated by wrap_columns=True (rows kept 4 px from the top and bottom)
    diff = 0
    for trial in range(30):
        H, W = int(rng.integers(24, 60)), int(rng.integers(12, 40))
        prob = np.zeros((H, W), np.float32)
        n = int(rng.integers(20, H * W // 4))
        ys, xs = rng.integers(4, H - 4, n), rng.integers(0, W, n)
        prob[ys, xs] = rng.permutation(n).astype(np.float32) / n * 0.9 + 0.05
        xy_w, _, _, _ = P.get_keypoints(prob, 0.0, 10_000, wrap_columns=True)
        assert np.array_equal(ref(prob, 0.0, 10_000), xy_w), trial
        diff += not np.array_equal(xy_w, P.get_keypoints(prob, 0.0, 10_000)[0])
    assert diff > 0            # the quirk is real (and only reachable at the frame's left / right edge)


def _ref_sp_post():
    """oracle/_ref/libref_sp_post.so (+ _nopca): the REFERENCE'S OWN SuperPointTensorRT::getKeyPoints and ::computeDescriptors text
    (superpoint_tensorrt.cpp:164-230, with pt_conf_comp / NMS2 behind them) compiled verbatim against the real libtorch C++ API of this image,
    the cv:: stand-in and a three-expression Eigen stand-in (oracle/ref_build/sp_post_wrap.cpp)."""
    import ctypes
    import subprocess
    import torch                                                    # libtorch must be in the process before the .so is opened
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"])
    libs = []
    for name in ("libref_sp_post.so", "libref_sp_post_nopca.so"):
        path = os.path.join(root, "oracle", "_ref", name)
        if not os.path.exists(path):
            pytest.skip("oracle/_ref/%s not built (needs /root/reference once)" % name)
        libs.append(ctypes.CDLL(path))
    fp = ctypes.POINTER(ctypes.c_float)
    ci = ctypes.c_int

    def fn(L, name, argtypes):
        f = getattr(L, name)
        f.restype = ci
        f.argtypes = argtypes
        return f
    gk = fn(libs[0], "ref_sp_get_keypoints", [fp, ci, ci, ctypes.c_float, ci, fp])
    cd = fn(libs[0], "ref_sp_compute_descriptors", [fp, ci, ci, fp, ci, ci, ci, fp, fp, ci, fp])
    cd_raw = fn(libs[1], "ref_sp_compute_descriptors_nopca", [fp, ci, ci, fp, ci, ci, ci, fp, fp, ci, fp])

    def keypoints(prob, thres, max_num):
        prob = np.ascontiguousarray(prob, np.float32)
        out = np.zeros((prob.size, 2), np.float32)
        n = gk(prob.ctypes.data_as(fp), prob.shape[1], prob.shape[0], thres, max_num, out.ctypes.data_as(fp))
        return out[:n].copy()

    def descriptors(desc, xy, W, H, comp=None, mean=None):
        desc = np.ascontiguousarray(desc, np.float32)
        xy = np.ascontiguousarray(xy, np.float32)
        d_out = 256 if comp is None else comp.shape[0]
        out = np.zeros((max(len(xy), 1), d_out), np.float32)
        if comp is None:
            n = cd_raw(desc.ctypes.data_as(fp), desc.shape[1], desc.shape[2], xy.ctypes.data_as(fp), len(xy), W, H, None, None, 256, out.ctypes.data_as(fp))
        else:
            comp = np.ascontiguousarray(comp, np.float32)
            mean = np.ascontiguousarray(mean, np.float32)
            n = cd(desc.ctypes.data_as(fp), desc.shape[1], desc.shape[2], xy.ctypes.data_as(fp), len(xy), W, H, comp.ctypes.data_as(fp), mean.ctypes.data_as(fp),
                   d_out, out.ctypes.data_as(fp))
        assert n == len(xy) * d_out
        return out[:len(xy)]
    return keypoints, descriptors


def test_postprocessing_restatement_is_pinned_to_the_reference_text():
    """oracle.postproc_ref.get_keypoints / compute_descriptors -- the checkers of every key-point and descriptor parity test -- against the
    reference's own getKeyPoints and computeDescriptors compiled from their own text with the real libtorch:
      * key points: `prob > thres`, findNonZero order, the confidence column, NMS2(border 0, dist 4, width, height, max_num): identical sets in
        identical order, on heat maps of the network itself (seeded weights, synthetic frames) and on random maps, interior candidates
        (the left / right edge wrap-around of NMS2 is covered by test_nms2_restatement_is_pinned_to_the_reference_text);
      * descriptors in front of the PCA (grid = 2 x / width - 1, grid_sampler(…, 0, 0, 0), the norm over dim 1 = ACROSS the key points, div,
        transpose): bit-identical to the restatement;
      * with the PCA: (desc - mean) * comp^T to 2e-6 (the Eigen product is a stand-in: summation order unpinned)."""
    import torch
    from oracle import superpoint_ref as S
    from omni_swarm_amd import synth
    ref_kp, ref_desc = _ref_sp_post()
    comp, mean = synth.pca()
    w = S.synth_weights(0)
    checked = 0
    for (H, W, seed) in ((96, 128, 3), (120, 160, 5), (64, 96, 8)):
        img = synth.image_u8(900 + seed, H, W, n_shapes=40)[None]
        semi, desc = S.forward(w, S.preprocess_u8(img, False))
        prob = semi[0].copy()
        prob[:4] = 0; prob[-4:] = 0; prob[:, :4] = 0; prob[:, -4:] = 0                 # interior candidates only (NMS2's edge quirk is pinned elsewhere)
        for thr, maxn in ((0.015, 200), (0.05, 30), (0.3, 200)):
            xy, _, _, _ = P.get_keypoints(prob, thr, maxn)
            got = ref_kp(prob, thr, maxn)
            assert np.array_equal(got, xy.astype(np.float32)), (H, W, thr, maxn)
            if len(xy) == 0:
                continue
            d64, d256 = P.compute_descriptors(desc[0], xy, W, H, comp, mean)
            r256 = ref_desc(desc[0], xy, W, H)
            assert np.array_equal(r256, d256), np.abs(r256 - d256).max()               # the torch part: same library, same calls, same bits
            r64 = ref_desc(desc[0], xy, W, H, comp, mean)
            assert np.abs(r64 - d64).max() < 2e-6 * max(1.0, np.abs(d64).max())
            checked += len(xy)
    rng = np.random.default_rng(7)
    for trial in range(10):                                                             # random maps and descriptor planes, odd sizes
        H, W = 8 * int(rng.integers(4, 10)), 8 * int(rng.integers(4, 12))
        prob = np.zeros((H, W), np.float32)
        n = int(rng.integers(10, 200))
        ys, xs = rng.integers(4, H - 4, n), rng.integers(4, W - 4, n)
        prob[ys, xs] = rng.permutation(n).astype(np.float32) / n * 0.9 + 0.05
        desc = rng.standard_normal((256, H // 8, W // 8)).astype(np.float32)
        xy, _, _, _ = P.get_keypoints(prob, 0.1, 50)
        assert np.array_equal(ref_kp(prob, 0.1, 50), xy.astype(np.float32))
        d64, d256 = P.compute_descriptors(desc, xy, W, H, comp, mean)
        assert np.array_equal(ref_desc(desc, xy, W, H), d256)
        assert np.abs(ref_desc(desc, xy, W, H, comp, mean) - d64).max() < 2e-6 * max(1.0, np.abs(d64).max())
        checked += len(xy)
    assert checked > 500


def test_mobilenetvlad_backbone_matches_an_independent_mobilenetv2():
    """The reference ships no MobileNetVLAD graph (parity unpinned, assumed architecture).  What CAN be checked offline: the assumed backbone --
    MobileNetV2 at width 0.35 up to the 112-channel block -- against an INDEPENDENT implementation of that architecture (Hugging Face
    `transformers.MobileNetV2Model`, written after TF-slim's definition): same layer table (channels, strides, groups of all 51 convolutions)
    and, with the oracle's seeded weights loaded into it (batch norm set to identity + the conv bias), the same feature map.  A wrong row in
    omni-swarm_amd/weights.py:VLAD_BLOCKS, a wrong padding or a misplaced residual would show up here."""
    import torch
    transformers = pytest.importorskip("transformers")
    from transformers import MobileNetV2Config, MobileNetV2Model
    cfg = MobileNetV2Config(num_channels=3, depth_multiplier=0.35, min_depth=8, depth_divisible_by=8, expand_ratio=6, output_stride=32,
                            first_layer_is_expansion=True, finegrained_output=True, tf_padding=False, hidden_act="relu6")
    hf = MobileNetV2Model(cfg, add_pooling_layer=False).eval()
    convs = [(n, m) for n, m in hf.named_modules() if isinstance(m, torch.nn.Conv2d) and not n.startswith("conv_1x1")]
    specs = V.layer_specs()
    assert len(convs) == len(specs) == 51
    w = V.synth_weights()
    with torch.no_grad():
        for (hname, conv), (name, kind, cin, cout, stride) in zip(convs, specs):
            groups = cin if kind == "dw3x3_relu6" else 1
            assert (conv.in_channels, conv.out_channels, conv.stride[0], conv.groups) == (cin, cout, stride, groups), (hname, name)
            assert conv.kernel_size[0] == (1 if kind.startswith("pw") else 3)
            conv.weight.copy_(torch.from_numpy(w[name + ".weight"]))
            bn = dict(hf.named_modules())[hname.rsplit(".", 1)[0] + ".normalization"]
            bn.weight.fill_(1.0); bn.running_mean.zero_(); bn.eps = 1e-12; bn.running_var.fill_(1.0 - 1e-12)      # identity up to 1 ulp
            bn.bias.copy_(torch.from_numpy(w[name + ".bias"]))
        img = synth.image_u8(321, 96, 128, n_shapes=60)
        _, feat, _ = V.forward(w, img, return_features=True)
        x = ((torch.from_numpy(img).float() - 128.0) / 128.0)[None, None].repeat(1, 3, 1, 1)
        h = hf.conv_stem(x)
        for layer in hf.layer:
            h = layer(h)
    assert h.shape == feat.shape == (1, 112, 3, 4)
    assert np.abs(h.numpy() - feat).max() < 1e-4 * max(1.0, np.abs(feat).max())
