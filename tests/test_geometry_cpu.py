"""Host geometry (omni-swarm_amd/host/geometry.hpp, loop_geometry.hpp: triangulation + 3-D flags, homography-RANSAC mask, PnP-RANSAC,
compute_loop -> LoopEdge; loop_cam.cpp:73-106,397-444, loop_detector.cpp:295-836) against the numpy oracle (oracle/geometry_ref.py: LAPACK
instead of the product's Jacobi sweeps) and against the ground truth of seeded synthetic stereo-fisheye scenes.  CPU only: g++ builds
tests/cpp/geometry_check.cpp, which talks a text protocol; the descriptor matcher there is the oracle's cv::BFMatcher restatement."""
import math
import os
import subprocess

import numpy as np
import pytest

from oracle import geometry_ref as G
from oracle import match_ref as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, CX, CY = 300.0, 300.0, 240.0          # 600 x 480 virtual pinhole views with a 90 degree horizontal field of view


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("geom") / "geometry_check")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-o", out, os.path.join(ROOT, "tests", "cpp", "geometry_check.cpp"),
                           "-L", os.path.join(ROOT, "oracle"), "-loracle", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}"])
    return out


def run(exe, text):
    r = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return [ln.split() for ln in r.stdout.strip().split("\n")]


def fmt(*arrs):
    return " ".join(repr(float(x)) for a in arrs for x in np.asarray(a, np.float64).reshape(-1))


def fpose(p):
    return fmt(p[0], p[1])


def rpy_quat(roll, pitch, yaw):
    return G.qmul(G.qmul(G.q_from_yaw(yaw), np.array([math.cos(pitch / 2), 0, math.sin(pitch / 2), 0])), np.array([math.cos(roll / 2), math.sin(roll / 2), 0, 0]))


R_BC = np.array([[0.0, 0, 1], [-1, 0, 0], [0, -1, 0]])       # camera (x right, y down, z forward) in the body (x forward, y left, z up)


def extrinsics(direction, up):
    yaw = math.pi / 2 * direction
    Rz = np.array([[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1]])
    return G.pose([0.0, 0.0, 0.05 if up else -0.05], G.q_from_R(Rz @ R_BC))


def observe(world_pts, desc, pose_drone, ext, rng, max_num=200, desc_noise=0.05):
    """Project landmarks into one virtual camera: integer pixel key points (as SuperPoint gives them), normalised points lifted from the
    pixels, noisy unit descriptors.  Returns dict + the landmark id of every key point."""
    cam = G.pmul(pose_drone, ext)
    pc = (world_pts - cam[0]) @ G.qR(cam[1])                    # R^T (X - t)
    vis = np.nonzero((pc[:, 2] > 0.3) & (np.abs(pc[:, 0] / pc[:, 2]) < 0.98) & (np.abs(pc[:, 1] / pc[:, 2]) < 0.78))[0]
    vis = vis[:max_num]
    px = np.rint(np.stack([F * pc[vis, 0] / pc[vis, 2] + CX, F * pc[vis, 1] / pc[vis, 2] + CY], 1))
    norm = np.stack([(px[:, 0] - CX) / F, (px[:, 1] - CY) / F], 1)
    d = desc[vis] + desc_noise * rng.standard_normal((len(vis), 64))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return {"landmark_num": len(vis), "camera_extrinsic": ext, "landmarks_2d": px.astype(np.float32).astype(np.float64), "landmarks_2d_norm": norm.astype(np.float32).astype(np.float64),
            "feature_descriptor": d, "landmarks_3d": np.zeros((len(vis), 3)), "landmarks_flag": np.zeros(len(vis), np.uint8)}, vis


def make_frame(world_pts, desc, pose_drone, msg_id, drone_id, rng, triangle_thres=0.006):
    images = []
    for d in range(4):
        up, _ = observe(world_pts, desc, pose_drone, extrinsics(d, True), rng)
        down, _ = observe(world_pts, desc, pose_drone, extrinsics(d, False), rng)
        qi, ti, _ = M.bf_match(up["feature_descriptor"], down["feature_descriptor"], 0)
        cnt, l3u, fu, _, _ = G.stereo_landmarks(pose_drone, up["camera_extrinsic"], down["camera_extrinsic"], up["landmarks_2d_norm"], down["landmarks_2d_norm"],
                                                qi, ti, triangle_thres)
        up["landmarks_3d"], up["landmarks_flag"] = l3u.astype(np.float32).astype(np.float64), fu
        up["stereo"] = (down, qi, ti, cnt)
        images.append(up)
    return {"msg_id": msg_id, "drone_id": drone_id, "timestamp": 100.0 + msg_id, "pose_drone": pose_drone, "images": images,
            "landmark_num": int(sum(i["landmark_num"] for i in images))}


def frame_text(f):
    out = [f"{f['msg_id']} {f['drone_id']} {f['timestamp']!r} {f['landmark_num']} {fpose(f['pose_drone'])} {len(f['images'])}"]
    for im in f["images"]:
        out.append(f"{im['landmark_num']} {fpose(im['camera_extrinsic'])}")
        for i in range(im["landmark_num"]):
            out.append(f"{fmt(im['landmarks_2d'][i], im['landmarks_2d_norm'][i], im['landmarks_3d'][i])} {int(im['landmarks_flag'][i])} {fmt(im['feature_descriptor'][i])}")
    return "\n".join(out)


@pytest.fixture(scope="module")
def scene():
    rng = np.random.default_rng(2026)
    n = 1500
    dirs = rng.standard_normal((n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    pts = dirs * rng.uniform(2.0, 7.0, (n, 1)) + np.array([0, 0, 1.0])
    desc = rng.standard_normal((n, 64))
    desc /= np.linalg.norm(desc, axis=1, keepdims=True)
    pose_old = G.pose([0.0, 0.0, 1.0], rpy_quat(0.01, -0.02, 0.10))
    pose_new = G.pose([0.35, -0.25, 1.08], rpy_quat(-0.015, 0.01, 0.42))
    return {"pts": pts, "desc": desc, "pose_old": pose_old, "pose_new": pose_new,
            "old": make_frame(pts, desc, pose_old, 7, 1, rng), "new": make_frame(pts, desc, pose_new, 9, 1, rng), "rng": rng}


def test_cv_rng_sequence(exe):
    r = G.CvRng()
    assert [int(x) for x in run(exe, "rng 40")[0][1:]] == [r.uniform(0, 1000) for _ in range(40)]


def test_triangulate_point(exe):
    rng = np.random.default_rng(1)
    cases, text = [], ["tri 60"]
    for _ in range(60):
        X = rng.uniform(-3, 3, 3) + np.array([0, 0, 6.0])
        a = G.pose(rng.uniform(-0.2, 0.2, 3), rpy_quat(*rng.uniform(-0.1, 0.1, 3)))
        b = G.pose(rng.uniform(-0.2, 0.2, 3) + np.array([0, 0.1, 0]), rpy_quat(*rng.uniform(-0.1, 0.1, 3)))
        pa, pb = (X - a[0]) @ G.qR(a[1]), (X - b[0]) @ G.qR(b[1])
        p0, p1 = pa[:2] / pa[2] + rng.normal(0, 1e-4, 2), pb[:2] / pb[2] + rng.normal(0, 1e-4, 2)
        cases.append((a, b, p0, p1, X))
        text.append(f"{fpose(a)} {fpose(b)} {fmt(p0, p1)}")
    out = run(exe, "\n".join(text))
    for (a, b, p0, p1, X), ln in zip(cases, out):
        err, Xr = G.triangulate_point(a[1], a[0], b[1], b[0], p0, p1)
        got = np.array(ln[1:], np.float64)
        assert abs(got[0] - err) < 1e-9 and np.abs(got[1:] - Xr).max() < 1e-7 * max(1.0, np.abs(Xr).max())
        assert np.linalg.norm(got[1:] - X) < 0.2                      # 1e-4 noise at a 0.1-0.3 m baseline, 6 m away


def test_stereo_landmarks_and_flags(exe, scene):
    f = scene["new"]
    n3d = 0
    for d in range(4):
        up = f["images"][d]
        down, qi, ti, cnt = up["stereo"]
        text = (f"stereo {fpose(f['pose_drone'])} {fpose(up['camera_extrinsic'])} {fpose(down['camera_extrinsic'])} {up['landmark_num']} {down['landmark_num']} "
                f"{len(qi)} 0.006\n{fmt(up['landmarks_2d_norm'])}\n{fmt(down['landmarks_2d_norm'])}\n" + " ".join(f"{a} {b}" for a, b in zip(qi, ti)))
        out = run(exe, text)
        c, l3u, fu, l3d, fd = G.stereo_landmarks(f["pose_drone"], up["camera_extrinsic"], down["camera_extrinsic"], up["landmarks_2d_norm"], down["landmarks_2d_norm"], qi, ti, 0.006)
        assert int(out[0][1]) == c == cnt
        got = np.array(out[0][2:], np.float64).reshape(-1, 4)
        assert np.array_equal(got[:, 0].astype(np.uint8), fu) and np.abs(got[:, 1:] - l3u).max() < 1e-6
        assert np.array_equal(np.array(out[1][1:], np.uint8), fd)
        n3d += c
    assert n3d > 150                                                     # the scene does triangulate


def test_homography_ransac_mask(exe):
    rng = np.random.default_rng(3)
    H = np.array([[1.02, 0.03, 12.0], [-0.02, 0.98, -7.0], [1e-5, -2e-5, 1.0]])
    for n, n_out in ((60, 18), (25, 5), (9, 0), (4, 0)):
        src = np.rint(rng.uniform([20, 20], [580, 460], (n, 2)))
        w = src @ H[:2, :2].T + H[:2, 2]
        dst = np.rint(w / (src @ H[2, :2] + 1.0)[:, None] + rng.normal(0, 0.3, (n, 2)))
        bad = rng.choice(n, n_out, replace=False)
        dst[bad] += rng.uniform(25, 80, (n_out, 2)) * rng.choice([-1, 1], (n_out, 2))
        out = run(exe, f"homo {n} 3.0\n" + "\n".join(fmt(s, d) for s, d in zip(src, dst)))
        Hr, mask = G.find_homography_ransac(src, dst, 3.0)
        got = np.array(out[0][2:], np.uint8)
        assert int(out[0][1]) == 1 and np.array_equal(got, mask), (n, got, mask)
        assert not got[bad].any() and got.sum() >= n - n_out - 2
        assert np.abs(np.array(out[1][1:], np.float64).reshape(3, 3) - Hr).max() < 1e-6 * np.abs(Hr).max()
    # degenerate: all points on one line -> no model, mask all zero (OpenCV returns an empty H and a zero mask)
    src = np.stack([np.arange(10.0) * 10, np.arange(10.0) * 5], 1)
    out = run(exe, "homo 10 3.0\n" + "\n".join(fmt(s, s + 1) for s in src))
    assert int(out[0][1]) == 0 and not any(int(x) for x in out[0][2:])
    assert G.find_homography_ransac(src, src + 1, 3.0)[0] is None


def test_pnp_ransac_pose(exe):
    rng = np.random.default_rng(4)
    for n, n_out, thr in ((80, 0, 3.0), (60, 12, 0.02), (12, 0, 3.0)):
        X = rng.uniform(-3, 3, (n, 3)) + np.array([0.5, 0.2, 6.0])
        R, t = G.rodrigues(rng.uniform(-0.3, 0.3, 3)), rng.uniform(-0.5, 0.5, 3)
        c = X @ R.T + t
        u = c[:, :2] / c[:, 2:3] + rng.normal(0, 2e-3, (n, 2))
        bad = rng.choice(n, n_out, replace=False)
        u[bad] += rng.uniform(0.1, 0.4, (n_out, 2))
        out = run(exe, f"pnp {n} 100 {thr} 0.99\n" + "\n".join(fmt(a, b) for a, b in zip(X, u)))
        (Rr, tr), inl = G.solve_pnp_ransac(X, u, 100, thr, 0.99)
        got = np.array(out[0][3:], np.float64)
        assert int(out[0][1]) == 1 and int(out[0][2]) == len(inl)
        assert np.abs(got[:9].reshape(3, 3) - Rr).max() < 1e-7 and np.abs(got[9:] - tr).max() < 1e-7
        assert np.abs(got[:9].reshape(3, 3) - R).max() < 5e-3 and np.abs(got[9:] - t).max() < 3e-2
        if n_out:
            assert not set(bad.tolist()) & set(inl)
    assert run(exe, "pnp 5 100 3.0 0.99\n" + "\n".join(fmt(rng.uniform(0, 1, 5)) for _ in range(5)))[0][1] == "0"     # fewer than 6 points


def test_epnp_minimal_solver(exe):
    """EPnP on minimal sets (cv::solvePnPRansac's kernel: 5 points, OpenCV 3.4 epnp.cpp restated): exact data -> the exact pose, in the C++ host code
    and in the numpy oracle, which agree with each other to 1e-7 -- on random point clouds and on the walls-of-a-room configuration of the
    rendered-scene test (points on four planes, image points rotated into the main camera: |u| up to ~50).  Larger sets too (the solver is
    general).  EPnP fixes the solution's sign by the depth of the FIRST point (epnp.cpp solve_for_sign), i.e. it assumes the points in front of the
    camera: sets with points behind it (the reference feeds such: image points of the side / rear directions rotated into the main camera) are
    only compared between the two implementations, as are exactly coplanar sets (outside EPnP's general case, in OpenCV too)."""
    rng = np.random.default_rng(11)
    text, cases = [], []
    for trial in range(120):
        n = 5 if trial % 3 else int(rng.integers(6, 40))
        kind = trial % 4
        if kind == 3:                                            # a square room seen from its centre: points on the walls x = +-2, y = +-2
            wall = rng.integers(0, 4, n)
            a, b = rng.uniform(-1.9, 1.9, n), rng.uniform(-1.2, 1.2, n)
            X = np.stack([np.where(wall == 0, 2.0, np.where(wall == 1, -2.0, a)), np.where(wall == 2, 2.0, np.where(wall == 3, -2.0, a)), b], 1)
            X = X[:, [1, 2, 0]] + np.array([0.1, 0.05, 0.0])     # camera looks along +z of this frame: some points beside / behind it
            R, t = G.rodrigues(rng.uniform(-0.05, 0.05, 3)), rng.uniform(-0.1, 0.1, 3)
            c = X @ R.T + t
            if np.abs(c[:, 2]).min() < 0.05 or len(set(wall.tolist())) < 2:
                continue
        else:
            X = rng.uniform(-3, 3, (n, 3)) + np.array([0.5, 0.2, 6.0])
            R, t = G.rodrigues(rng.uniform(-0.4, 0.4, 3)), rng.uniform(-0.5, 0.5, 3)
            c = X @ R.T + t
        u = c[:, :2] / c[:, 2:3]
        cases.append((X, u, R, t, bool((c[:, 2] > 0).all())))
        text.append(f"epnp {n}\n" + "\n".join(fmt(a, b) for a, b in zip(X, u)))
    out = run(exe, "\n".join(text))
    worst = 0.0
    n_front = 0
    for (X, u, R, t, front), o in zip(cases, out):
        got = np.array(o[2:], np.float64)
        ref = G.epnp(X, u)
        assert o[1] == "1" and ref is not None
        assert np.abs(got[:9].reshape(3, 3) - ref[0]).max() < 1e-7 and np.abs(got[9:] - ref[1]).max() < 1e-7 * max(1.0, np.abs(ref[1]).max())
        if front:
            n_front += 1
            worst = max(worst, np.abs(got[:9].reshape(3, 3) - R).max(), np.abs(got[9:] - t).max())
    assert worst < 1e-6 and n_front > 80 and len(cases) - n_front > 5, (worst, n_front, len(cases))
    # coplanar minimal set: defined output, nothing more
    Xp = np.array([[0, 0, 5.0], [1, 0, 5], [0, 1, 5], [1, 1, 5], [0.3, 0.7, 5]])
    o = run(exe, "epnp 5\n" + "\n".join(fmt(a, a[:2] / a[2]) for a in Xp))[0]
    assert all(np.isfinite(float(x)) for x in o[2:])


def test_compute_loop_end_to_end(exe, scene):
    new, old = scene["new"], scene["old"]
    for (dn, dold, init_mode, is4) in ((1, 1, 0, 1), (1, 1, 1, 0), (0, 0, 0, 1)):
        out = run(exe, f"loop {dn} {dold} {init_mode} {is4}\n{frame_text(new)}\n{frame_text(old)}")[0]
        ref = G.compute_loop(new, old, dn, dold, bool(init_mode), lambda a, b: M.bf_match(a, b, 0), is_4dof=bool(is4))
        assert ref is not None and out[1] == "1"
        assert int(out[2]) == ref["n_corr"] and int(out[3]) == ref["inliers"]
        assert (int(out[4]), int(out[5]), int(out[6]), int(out[7])) == (old["msg_id"], new["msg_id"], old["drone_id"], new["drone_id"])
        dp = np.array(out[8:], np.float64)
        assert np.abs(dp[:3] - ref["relative_pose"][0]).max() < 1e-6 and min(np.abs(dp[3:] - ref["relative_pose"][1]).max(), np.abs(dp[3:] + ref["relative_pose"][1]).max()) < 1e-6
        truth = G.delta_pose(scene["pose_old"], scene["pose_new"], bool(is4))
        assert np.linalg.norm(dp[:3] - truth[0]) < 0.10, (dp[:3], truth[0])      # 10 cm stereo baseline, integer-pixel key points, landmarks 2-7 m away
        assert abs(G.wrap_angle(G.quat2eulers(dp[3:])[2] - G.quat2eulers(truth[1])[2])) < math.radians(1.0)
        assert int(out[3]) > 100
    # a frame from another place (other landmarks, other descriptors): no loop, in both implementations
    rng = np.random.default_rng(9)
    pts2 = rng.standard_normal((800, 3)) * 3 + np.array([0, 0, 1.0])
    d2 = rng.standard_normal((800, 64))
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    other = make_frame(pts2, d2, scene["pose_new"], 11, 1, rng)
    out = run(exe, f"loop 1 1 0 1\n{frame_text(other)}\n{frame_text(old)}")[0]
    assert out[1] == "0" and G.compute_loop(other, old, 1, 1, False, lambda a, b: M.bf_match(a, b, 0)) is None
    # too few landmarks in the new frame (:633)
    few = dict(new, landmark_num=10)
    assert run(exe, f"loop 1 1 0 1\n{frame_text(few)}\n{frame_text(old)}")[0][1] == "0"


# ---- the control flow of LoopGeometry against the reference's own text ----------------------------------------------------------------------------
@pytest.fixture(scope="module")
def pin_exe():
    """tests/cpp/loopgeo_pin.cpp: loop_detector.cpp:317-836 (+ PnPRestoCamPose, reduceVector, loop_params.cpp), extracted at build time and compiled
    verbatim next to LoopGeometry (oracle/Makefile, _ref/loopgeo_pin)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "loopgeo_pin")
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/loopgeo_pin"])
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/loopgeo_pin is built from /root/reference, which is absent here")
    return exe


def _thin(frame, keep_dirs=None, every=1, drone_id=None, msg_id=None):
    """a copy of a frame with some directions emptied / every n-th landmark kept / another sender"""
    f = dict(frame)
    imgs = []
    for d, im in enumerate(frame["images"]):
        im = dict(im)
        if keep_dirs is not None and d not in keep_dirs:
            sel = np.zeros(0, np.int64)
        else:
            sel = np.arange(0, im["landmark_num"], every)
        for k in ("landmarks_2d", "landmarks_2d_norm", "landmarks_3d", "landmarks_flag", "feature_descriptor"):
            im[k] = im[k][sel]
        im["landmark_num"] = len(sel)
        imgs.append(im)
    f["images"] = imgs
    f["landmark_num"] = int(sum(i["landmark_num"] for i in imgs))
    if drone_id is not None:
        f["drone_id"] = drone_id
    if msg_id is not None:
        f["msg_id"] = msg_id
    return f


def test_loop_geometry_control_flow_is_pinned_to_the_reference_text(pin_exe, scene):
    """One session of key-frame pairs through LoopGeometry::compute_loop and through the reference's own LoopDetector::compute_loop (same numerical
    kernels behind both): the same verdicts, inlier counts, LoopEdge fields, edge ids (self_id * MAX_LOOP_ID + loop_count, numbered only when the
    odometry gate lets the edge through) and relative poses; the frame-pair correspondence function alone returns the same points, 3-D
    landmarks, per-direction index lists and direction pairs (and its index maps agree with them); the inter-drone counters grow by two per
    accepted loop.  Cases: main directions equal and rotated, init mode, 6-dof, a direction without landmarks, thinned frames (direction and
    feature-count gates), another place, an intra-drone edge refused by the odometry gate, an inter-drone edge that the gate does not touch."""
    new, old = scene["new"], scene["old"]
    rng = np.random.default_rng(5)
    pts2 = rng.standard_normal((800, 3)) * 3 + np.array([0, 0, 1.0])
    d2 = rng.standard_normal((800, 64))
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    other = make_frame(pts2, d2, scene["pose_new"], 11, 1, rng)
    cases = [
        (1, 1, 0, 1, 0, new, old), (1, 1, 1, 0, 0, new, old), (0, 0, 0, 1, 0, new, old), (2, 2, 0, 1, 0, new, old), (3, 3, 1, 1, 0, new, old),
        (2, 1, 0, 1, 0, new, old),                                         # main directions that do not face each other: pairs (2,1) (3,2) (0,3) (1,0)
        (1, 1, 0, 1, 0, _thin(new, keep_dirs=(0, 1, 2)), old),             # a direction without landmarks on one side
        (1, 1, 0, 1, 0, new, _thin(old, keep_dirs=(1, 2))),                # only two common directions: MIN_DIRECTION_LOOP fails
        (1, 1, 0, 1, 0, _thin(new, every=6), old),                         # thin frames: per-direction and total feature gates
        (1, 1, 1, 1, 0, _thin(new, every=6), _thin(old, every=2)),
        (1, 1, 0, 1, 0, _thin(new, every=25), old),                        # fewer than 4 flagged matches per image pair: the ignored return value
        (1, 1, 0, 1, 0, other, old),                                       # another place
        (1, 1, 0, 1, 0, dict(new, landmark_num=10), old),                  # :633
        (1, 1, 0, 1, 1, new, old),                                         # the odometry gate refuses an intra-drone edge: no id consumed
        (1, 1, 0, 1, 1, _thin(new, drone_id=2, msg_id=21), old),           # ... and does not look at an inter-drone one
        (1, 1, 0, 1, 0, _thin(new, drone_id=3, msg_id=31), old),
        (0, 0, 1, 1, 0, new, old),
    ]
    ver = run(pin_exe, "verify\n")[-1]                                     # the gates' constants and comparison operators on a straddling grid
    assert ver[0] == "VERIFY" and int(ver[1]) > 5000 and ver[2] == "0", ver
    text = "\n".join(f"loop {dn} {dold} {im} {is4} {rej}\n{frame_text(a)}\n{frame_text(b)}" for dn, dold, im, is4, rej, a, b in cases)
    out = [ln for ln in run(pin_exe, text) if ln and ln[0] in ("PROD", "REF", "CORR", "COUNTS")]     # (the reference text prints its own progress lines)
    assert len(out) == 4 * len(cases)
    accepted = 0
    for i, case in enumerate(cases):
        prod, ref, corr, counts = out[4 * i: 4 * i + 4]
        assert prod[0] == "PROD" and ref[0] == "REF" and corr[0] == "CORR" and counts[0] == "COUNTS"
        assert prod[1:8] == ref[1:8], (i, prod[:11], ref[:11])             # verdict, inliers, id, key-frame ids, drone ids
        assert prod[10] == ref[10]                                           # loops numbered so far
        if prod[1] == "1":
            accepted += 1
            assert abs(float(prod[8]) - float(ref[8])) < 1e-6 and abs(float(prod[9]) - float(ref[9])) < 1e-6          # time stamps (sec + nsec)
            a, b = np.array(prod[11:18], np.float64), np.array(ref[11:18], np.float64)
            assert np.abs(a - b).max() < 1e-9, (i, a, b)
            assert prod[18:] == ref[18:]                                     # covariances
        assert corr[1] == "1" and corr[2] == corr[3], (i, corr)              # the correspondence sets, element by element
        assert int(counts[1]) == 2 * accepted
    verdicts = [out[4 * i][1] for i in range(len(cases))]
    assert verdicts.count("1") >= 8 and verdicts.count("0") >= 5, verdicts  # both outcomes are exercised
    assert verdicts[13] == "0" and verdicts[14] == "1"                      # the gate: intra-drone refused, inter-drone untouched
    assert out[4 * 14][3] == str(1 * 100000000 + sum(v == "1" for v in verdicts[:14]))     # the refused edge consumed no id


# ---- the stereo half (triangulation, acceptance, flags) against the reference's own text -------------------------------------------------------------
@pytest.fixture(scope="module")
def cam_pin_exe():
    """tests/cpp/loopcam_pin.cpp: triangulatePoint, match_HFNet_local_features and generate_stereo_image_descriptor (loop_cam.cpp:73-106, 141-175,
    341-523), extracted at build time and compiled verbatim next to fill_stereo_landmarks (oracle/Makefile, _ref/loopcam_pin)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "loopcam_pin")
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/loopcam_pin"])
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/loopcam_pin is built from /root/reference, which is absent here")
    return exe


def test_stereo_landmarks_are_pinned_to_the_reference_text(cam_pin_exe, scene):
    """The up / down key points and descriptors of every direction of two key frames through fill_stereo_landmarks (the matcher's list, the
    pipeline's double lifting) and through the reference's own generate_stereo_image_descriptor (its own matching call, lifting, triangulatePoint,
    acceptance test, flag and landmark bookkeeping): the same number of landmarks, the same flags on BOTH images, the same 3-D points (float
    message fields: equal to 1e-6 relative); nothing is triangulated unless the up image has more than ACCEPT_MIN_3D_PTS key points (:385)."""
    total = 0
    for f in (scene["new"], scene["old"]):
        for d in range(4):
            up = f["images"][d]
            down = up["stereo"][0]
            for accept_min in (50, 10 ** 6):
                text = (f"stereo {fpose(f['pose_drone'])} {fpose(up['camera_extrinsic'])} {fpose(down['camera_extrinsic'])} {up['landmark_num']} {down['landmark_num']} "
                        f"0.006 {accept_min} {F} {F} {CX} {CY}\n{fmt(up['landmarks_2d'])}\n{fmt(up['feature_descriptor'])}\n{fmt(down['landmarks_2d'])}\n{fmt(down['feature_descriptor'])}")
                out = [ln for ln in run(cam_pin_exe, text) if ln and ln[0] in ("PROD", "REF", "META")]
                prod, ref, meta = out
                assert prod[1] == ref[1], (d, accept_min, prod[1], ref[1])
                sep_p, sep_r = prod.index("|"), ref.index("|")
                assert sep_p == sep_r
                for a, b in ((prod[2:sep_p], ref[2:sep_r]), (prod[sep_p + 1:], ref[sep_r + 1:])):
                    a, b = np.array(a, np.float64).reshape(-1, 4), np.array(b, np.float64).reshape(-1, 4)
                    assert np.array_equal(a[:, 0], b[:, 0])                                   # flags, up and down
                    assert np.abs(a[:, 1:] - b[:, 1:]).max(initial=0) <= 1e-6 * max(1.0, np.abs(b[:, 1:]).max(initial=0))
                if accept_min > up["landmark_num"]:
                    assert prod[1] == "0"
                else:
                    total += int(prod[1])
                    assert int(prod[1]) == up["stereo"][3]                                     # and the numpy oracle's count
                assert meta[1:4] == ["1", "77", "1"]                                           # drone id, frame id, the up camera's extrinsic
    assert total > 300


# ---- the messages themselves (on_flattened_images -> generate_stereo_image_descriptor -> extractor_img_desc_deepnet) against the reference's own text ------
@pytest.fixture(scope="module")
def front_pin_exe():
    """tests/cpp/loopfront_pin.cpp: the whole message-building chain of LoopCam (loop_cam.cpp:178-229, 341-523, 525-634; the two networks are hooks),
    extracted at build time and compiled verbatim next to fill_image_descriptor / stamp_image_descriptor / finish_frame_descriptor and
    omni_fisheye_mask_rows (oracle/Makefile, _ref/loopfront_pin)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "loopfront_pin")
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/loopfront_pin"])
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/loopfront_pin is built from /root/reference, which is absent here")
    return exe


def test_keyframe_messages_and_mask_rows_are_pinned_to_the_reference_text(front_pin_exe, scene):
    """One key frame's network outputs (key points, descriptors, global descriptors of 4 x (up, down) images) through the functions
    KeyframePipeline::finish() builds its frames with, and through the reference's own on_flattened_images -> generate_stereo_image_descriptor ->
    extractor_img_desc_deepnet: the same frame stamps, the same per-image stamps, the same key points, float-lifted points, flags and landmarks,
    a global descriptor on the main camera's images only; the rows the reference blanks in every image (cv::Rect(0, rows*3/4, cols, rows/4) on
    a cv::Mat that shares its pixels with the caller's) are the rows omni_fisheye_mask_rows names -- also for heights that are not multiples
    of 4 -- and BOTH networks see the blanked image."""
    rng = np.random.default_rng(77)
    cases = 0
    for f, (H, W), accept_min, gdim in ((scene["new"], (48, 60), 50, 8), (scene["old"], (47, 33), 50, 5), (scene["new"], (45, 16), 10 ** 6, 3),
                                        (scene["old"], (42, 21), 50, 4), (scene["new"], (7, 9), 50, 2)):
        kf_id, stamp, self_id = 1000 + cases, 17.25 + cases, 3
        seed = int(rng.integers(1, 2 ** 31))
        lines = [f"frame {H} {W} 4 {gdim} {accept_min} 0.006 {self_id} {kf_id} {stamp!r} {F} {F} {CX} {CY} {seed} {fpose(f['pose_drone'])}"]
        gdescs = rng.standard_normal((4, gdim)).astype(np.float32)
        for d in range(4):
            up = f["images"][d]
            down = up["stereo"][0]
            lines.append(f"{fpose(up['camera_extrinsic'])} {fpose(down['camera_extrinsic'])}")
            for im in (up, down):
                lines.append(f"{im['landmark_num']} {fmt(im['landmarks_2d'])} {fmt(im['feature_descriptor'])}")
            lines.append(fmt(gdescs[d]))
        out = [ln for ln in run(front_pin_exe, "\n".join(lines)) if ln and ln[0] in ("PROD", "REF")]
        prod = [ln[1:] for ln in out if ln[0] == "PROD"]
        ref = [ln[1:] for ln in out if ln[0] == "REF"]
        assert not any(ln[0] == "SIZES-DIFFER" for ln in ref)                                  # the *_size fields equal the vectors' sizes, image_size 0
        # frame stamps: exact
        assert prod[0] == ref[0] and prod[0][0] == "FRAME", (prod[0], ref[0])
        assert prod[0][prod[0].index("image_num") + 1] == "4" and prod[0][prod[0].index("msg_id") + 1] == str(kf_id)
        assert int(prod[0][prod[0].index("landmark_num") + 1]) == sum(im["landmark_num"] for im in f["images"])
        # images: header fields exact (direction, drone, frame id, time, extrinsic, pose, counts, sizes, descriptor sums), arrays to float precision
        for d in range(4):
            p, r = prod[1 + d], ref[1 + d]
            cp, cr = p.index(":"), r.index(":")
            assert p[:cp] == r[:cr], (p[:cp], r[:cr])
            assert p[p.index("dir") + 1] == str(d) and p[p.index("gd") + 1] == str(gdim)
            a, b = np.array(p[cp + 1:], np.float64).reshape(-1, 8), np.array(r[cr + 1:], np.float64).reshape(-1, 8)
            assert np.array_equal(a[:, :5], b[:, :5])                                          # pixels, float-lifted points, flags
            assert np.abs(a[:, 5:] - b[:, 5:]).max(initial=0) <= 1e-6 * max(1.0, np.abs(b[:, 5:]).max(initial=0))
            n_flag = int(a[:, 4].sum())
            assert n_flag == (0 if accept_min > f["images"][d]["landmark_num"] else f["images"][d]["stereo"][3])
        # the mask: the caller's images after the reference ran == the rows the kernels blank; both networks saw exactly that
        pix_p, pix_r = prod[5], ref[5]
        assert pix_p[0] == pix_r[0] == "PIX" and pix_p == pix_r, (H, W)
        seen_sp, seen_vlad, kfc = ref[6], ref[7], ref[8]
        assert seen_sp[0] == "SEEN-SP" and seen_sp[1:] == pix_r[1:]                            # SuperPoint: all 8 images, blanked
        assert seen_vlad[0] == "SEEN-VLAD"
        assert seen_vlad[1::2] == pix_r[1::2] and all(v == "0" for v in seen_vlad[2::2])       # MobileNetVLAD: the up images only, blanked
        assert kfc == ["KF-COUNT", "1"]
        cases += 1
    assert cases == 5


def test_pinhole_depth_keyframe_is_pinned_to_the_reference_text(front_pin_exe):
    """CameraConfig::PINHOLE_DEPTH (launch/realsense.launch, BASELINE.json configs[0]: ONE 640 x 480 gray image + its 16-bit depth image per key frame):
    fill_image_descriptor + stamp_image_descriptor + fill_depth_landmarks + finish_frame_descriptor against the reference's own
    on_flattened_images -> generate_gray_depth_image_descriptor (loop_cam.cpp:231-339) compiled from its text: the same stamps, key points, lifted
    points; a landmark exactly where the depth under the ROUNDED pixel lies strictly inside (DEPTH_NEAR_THRES, DEPTH_FAR_THRES) -- the harness puts
    depths at, one millimetre under and one over both thresholds below every third key point -- and none at all when the image has at most
    ACCEPT_MIN_3D_PTS key points; the image is NOT blanked in this mode and both networks run once on it."""
    rng = np.random.default_rng(404)
    cases = 0
    for (H, W), n, accept_min, gdim, near, far in (((480, 640), 200, 50, 8, 0.3, 7.0), ((480, 640), 50, 50, 4, 0.3, 7.0), ((480, 640), 51, 50, 4, 0.5, 3.0),
                                                   ((120, 160), 90, 10, 3, 1.0, 12.0), ((480, 640), 0, 50, 2, 0.3, 7.0)):
        kf_id, stamp, self_id = 5000 + cases, 91.5 + cases, 2
        # key points as the detector reports them: integer pixels in [0, W) x [0, H); a few half-way values to exercise the rounding of cv::Mat::at(Point2f)
        kps = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.float32)
        if n > 8:
            kps[1] += np.float32(0.25); kps[2] -= np.float32(0.25)
            kps[3] = (0, 0); kps[4] = (W - 1, H - 1)
            kps[1] = np.clip(kps[1], 0, [W - 1, H - 1]); kps[2] = np.clip(kps[2], 0, [W - 1, H - 1])
        desc = rng.standard_normal((n, 64)).astype(np.float32)
        gdesc = rng.standard_normal(gdim).astype(np.float32)
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        pose = dict(pos=rng.standard_normal(3) * 3, quat=q)
        e = np.array([0.5, -0.5, 0.5, -0.5]) + rng.standard_normal(4) * 0.01; e /= np.linalg.norm(e)
        ext = dict(pos=rng.standard_normal(3) * 0.1, quat=e)
        fp = lambda p: " ".join(repr(float(v)) for v in list(p["pos"]) + list(p["quat"]))
        seed = int(rng.integers(1, 2 ** 31))
        text = (f"depthframe {H} {W} {gdim} {accept_min} {near!r} {far!r} {self_id} {kf_id} {stamp!r} {F} {F} {W / 2} {H / 2} {seed} {fp(pose)} {fp(ext)}\n"
                f"{n} {fmt(kps)} {fmt(desc)}\n{fmt(gdesc)}")
        out = [ln for ln in run(front_pin_exe, text) if ln and ln[0] in ("PROD", "REF")]
        prod = [ln[1:] for ln in out if ln[0] == "PROD"]
        ref = [ln[1:] for ln in out if ln[0] == "REF"]
        assert not any(ln[0] == "SIZES-DIFFER" for ln in ref)
        assert prod[0][0] == ref[0][0] == "COUNT3D" and prod[0] == ref[0], (prod[0], ref[0])
        count3d = int(prod[0][1])
        assert (count3d == 0) == (n <= accept_min), (n, accept_min, count3d)
        assert n <= accept_min or 0 < count3d < n                                                # both outcomes occur (a quarter of the depth image is 0)
        assert prod[1] == ref[1] and prod[1][0] == "FRAME"
        assert prod[1][prod[1].index("image_num") + 1] == "1" and prod[1][prod[1].index("landmark_num") + 1] == str(n)
        p, r = prod[2], ref[2]
        cp, cr = p.index(":"), r.index(":")
        assert p[:cp] == r[:cr], (p[:cp], r[:cr])
        assert p[p.index("dir") + 1] == "0" and p[p.index("gd") + 1] == str(gdim)
        a, b = np.array(p[cp + 1:], np.float64).reshape(-1, 8), np.array(r[cr + 1:], np.float64).reshape(-1, 8)
        assert np.array_equal(a[:, :5], b[:, :5])                                                # pixels, float-lifted points, FLAGS: the same landmarks chosen
        assert int(a[:, 4].sum()) == count3d
        assert np.abs(a[:, 5:] - b[:, 5:]).max(initial=0) <= 1e-6 * max(1.0, np.abs(b[:, 5:]).max(initial=0))
        # the oracle's restatement (oracle/geometry_ref.py depth_landmarks) on the harness's depth image (its LCG, replayed here)
        st, dep = seed, np.empty(H * W, np.uint16)
        for _ in range(H * W):
            st = (st * 1664525 + 1013904223) & 0xFFFFFFFF                                        # (the gray image's draws come first)
        for i in range(H * W):
            st = (st * 1664525 + 1013904223) & 0xFFFFFFFF
            dep[i] = 0 if (st >> 30) == 0 else 100 + (st >> 8) % 13900
        dep = dep.reshape(H, W)
        e_mm = [round(near * 1000), round(near * 1000) + 1, round(near * 1000) - 1, round(far * 1000), round(far * 1000) - 1, round(far * 1000) + 1]
        for i in range(0, n, 3):
            dep[int(np.rint(kps[i, 1])), int(np.rint(kps[i, 0]))] = e_mm[(i // 3) % 6]
        lift = lambda xy: np.stack([(xy[:, 0] - W / 2) / F, (xy[:, 1] - H / 2) / F], 1)
        oc, ol3, ofl = G.depth_landmarks((pose["pos"], pose["quat"]), (ext["pos"], ext["quat"]), kps, lift, dep, near, far, accept_min)
        assert oc == count3d and np.array_equal(ofl, a[:, 4].astype(np.uint8))
        assert np.abs(ol3 - a[:, 5:]).max(initial=0) <= 1e-6 * max(1.0, np.abs(ol3).max(initial=0))
        assert prod[3] == ref[3] and prod[3][0] == "PIX"                                         # nothing blanked
        calls = ref[4]
        assert calls[:3] == ["CALLS", "1", "1"] and calls[4] == calls[5] == prod[3][1]           # SuperPoint and MobileNetVLAD: once each, on the untouched image
        cases += 1
    assert cases == 5


@pytest.fixture(scope="module")
def ingest_pin_exe():
    exe = os.path.join(ROOT, "oracle", "_ref", "ingest_pin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ingest_pin is built from /root/reference, which is absent here")
    return exe


def test_keyframe_intake_is_pinned_to_the_reference_text(ingest_pin_exe):
    """omni::KeyframeIntake (host/keyframe_intake.hpp) next to SwarmLoop's own intake compiled from its text (find_images_raw, odometry_callback,
    odometry_keyframe_callback, VIOnonKF_callback, VIOKF_callback, pub_node_frame: swarm_loop.cpp:32-53, 100-187) on jittered streams: camera frames
    at ~20 Hz of which some never arrive (drops), an odometry message per frame stamped within +-0.4 ms of it (a few off by more than the 1 ms window,
    a few duplicated), VIO key-frame poses every ~0.35-2.5 s (faster than max_freq allows: the rate limit bites), silent stretches longer than
    nonkeyframe_waitsec (the non-key-frame path: first frame after 1 s, later ones after the wait, prevent_adding_db when the drone moved less than
    min_movement_keyframe), frames on which the networks report no landmark (they do not count as key frames).  Both sides must extract the same
    frames with the same poses, deliver the same ones with the same prevent_adding_db flag and node_frame, miss the same key-frame poses, and end in the
    same state (received_image, last_invoke, last_kftime, frames left in the queue)."""
    rng = np.random.default_rng(2024)
    cases = 0
    for max_freq, min_move, waitsec, duration, speed in ((1.0, 0.3, 5.0, 60.0, 0.05), (2.0, 0.3, 3.0, 45.0, 0.5), (0.5, 1.0, 5.0, 90.0, 0.02),
                                                         (1.0, 0.3, 5.0, 30.0, 0.0), (10.0, 0.05, 1.5, 20.0, 1.0)):
        events, t, idx = [], 1600000000.0 + 100 * cases, 0
        pos = np.zeros(3)
        next_kf = t + rng.uniform(1.5, 8.0)                       # the first VIO key frame comes late: the non-key-frame path starts the stream
        quiet_from, quiet_to = t + duration * 0.45, t + duration * 0.45 + waitsec * 2.2        # a stretch without VIO key frames
        while t < 1600000000.0 + 100 * cases + duration:
            t += 0.05 + rng.uniform(-0.004, 0.004)
            pos = pos + speed * 0.05 * np.array([1.0, 0.3, 0.0]) + rng.normal(0, 0.002, 3)
            arrived = rng.random() > 0.08                         # 8 % of the camera frames are lost
            if arrived:
                events.append(f"I {float(t)!r} {idx} {0 if rng.random() < 0.07 else int(rng.integers(30, 800))}")
                idx += 1
            r = rng.random()
            ot = t + (rng.uniform(-4e-4, 4e-4) if r > 0.05 else rng.choice([-1, 1]) * rng.uniform(1.6e-3, 4e-3))       # 5 %: outside the 1 ms window
            is_kf = t >= next_kf and not (quiet_from < t < quiet_to)
            pose = " ".join(repr(float(v)) for v in pos)
            if is_kf:
                events.append(f"K {float(ot)!r} {pose}")
                next_kf = t + rng.uniform(0.35, 2.5)
                if rng.random() < 0.1:
                    events.append(f"K {float(ot)!r} {pose}")        # a repeated key-frame pose: its frame is gone
            else:
                events.append(f"O {float(ot)!r} {pose}")
        text = f"{max_freq!r} {min_move!r} {waitsec!r} {len(events)}\n" + "\n".join(events)
        out = [ln for ln in run(ingest_pin_exe, text) if ln and ln[0] in ("PROD", "REF")]
        prod = [ln[1:] for ln in out if ln[0] == "PROD"]
        ref = [ln[1:] for ln in out if ln[0] == "REF"]
        assert not any(ln[0] == "SINKS-DIFFER" for ln in ref)      # the network, the detector and the node_frame topic got the same frames
        assert prod == ref, (cases, [p for p, r in zip(prod, ref) if p != r][:3], len(prod), len(ref))
        kinds = [ln[0] for ln in ref]
        n_ext, n_del, n_miss = kinds.count("EXTRACT"), kinds.count("DELIVER"), kinds.count("MISS")
        assert n_ext >= 5 and n_miss >= 1 and kinds[-1] == "STATE"
        if cases < 3:
            assert n_del < n_ext                                   # frames without landmarks were extracted and not delivered
        prevented = sum(int(ln[2]) for ln in ref if ln[0] == "DELIVER")
        if speed == 0.0:
            assert prevented >= 1                                  # a hovering drone: non-key frames are matched but not added
        assert all(ln[3] == "1" for ln in ref if ln[0] == "DELIVER")
        cases += 1
    assert cases == 5
