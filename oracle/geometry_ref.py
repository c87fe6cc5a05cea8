"""Geometric-verification oracle (CPU, numpy, f64) -- test infrastructure only.  PARITY UNPINNED (OpenCV 3.4, Eigen, swarm_msgs un-vendored).

An independent restatement (LAPACK SVD / eigh instead of the product's hand-written Jacobi sweeps) of
    triangulatePoint                       /root/reference/swarm_loop/src/loop_cam.cpp:73-106
    the up/down triangulation loop         loop_cam.cpp:397-444
    landmarks from the depth image         loop_cam.cpp:260-304 (CameraConfig::PINHOLE_DEPTH; pinned to the reference text through tests/cpp/loopfront_pin.cpp,
                                           tests/test_geometry_cpu.py::test_pinhole_depth_keyframe_is_pinned_to_the_reference_text)
    cv::findHomography(RANSAC, 3) mask     as used at swarm_loop/src/loop_detector.cpp:589-598 (OpenCV 3.4 ptsetreg.cpp / fundam.cpp)
    cv::solvePnPRansac(K = I)              as used at loop_detector.cpp:390-391: cv::RNG driven RANSAC over EPnP models of 5 points (OpenCV 3.4
                                           epnp.cpp restated) + solvePnP(ITERATIVE) on the inliers (DLT start + LM refit)
    PnPRestoCamPose, RPerror, pnp_result_verify, rotate_pt_norm2d, compute_correspond_features, compute_relative_pose, compute_loop
                                           loop_utils.cpp:69-81, loop_detector.cpp:317-353,415-429,431-624,355-413,627-836
Poses are (pos[3], quat wxyz[4]) tuples of float64 arrays.
"""
from __future__ import annotations

import math

import numpy as np

FLT_EPSILON = 1.1920929e-07
DBL_MIN = 2.2250738585072014e-308


# ---- quaternions / poses (Eigen conventions) ----------------------------------------------------------------------------------------------
def qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def qinv(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def qnorm(q):
    return np.asarray(q, np.float64) / np.linalg.norm(q)


def qR(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def q_from_R(R):
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        return np.array([w, (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    v = [0.0, 0.0, 0.0]
    v[i] = 0.5 * s
    s = 0.5 / s
    w = (R[k, j] - R[j, k]) * s
    v[j] = (R[j, i] + R[i, j]) * s
    v[k] = (R[k, i] + R[i, k]) * s
    return np.array([w, v[0], v[1], v[2]])


def q_from_yaw(yaw):
    return np.array([math.cos(yaw / 2), 0, 0, math.sin(yaw / 2)])


def quat2eulers(q):
    w, x, y, z = q
    return np.array([math.atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y)), math.asin(max(-1.0, min(1.0, 2 * (w * y - z * x)))),
                     math.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))])


def wrap_angle(a):
    while a > math.pi:
        a -= 2 * math.pi
    while a < -math.pi:
        a += 2 * math.pi
    return a


def pose(pos, q):
    return (np.asarray(pos, np.float64), qnorm(q))


def pmul(a, b):
    return (a[0] + qR(a[1]) @ b[0], qnorm(qmul(a[1], b[1])))


def pinv(a):
    qi = qinv(a[1])
    return (-(qR(qi) @ a[0]), qi)


def pyaw(a):
    return quat2eulers(a[1])[2]


def delta_pose(a, b, yaw_only):
    if not yaw_only:
        return (qR(qinv(a[1])) @ (b[0] - a[0]), qnorm(qmul(qinv(a[1]), b[1])))
    ya = pyaw(a)
    dyaw = wrap_angle(pyaw(b) - ya)
    dp = b[0] - a[0]
    return (np.array([math.cos(-ya) * dp[0] - math.sin(-ya) * dp[1], math.sin(-ya) * dp[0] + math.cos(-ya) * dp[1], dp[2]]), q_from_yaw(dyaw))


# ---- triangulation (loop_cam.cpp:73-106, 397-444) ---------------------------------------------------------------------------------------------
def triangulate_point(q0, t0, q1, t1, p0, p1):
    R0, R1 = qR(q0), qR(q1)
    P0 = np.hstack([R0.T, (-R0.T @ t0)[:, None]])
    P1 = np.hstack([R1.T, (-R1.T @ t1)[:, None]])
    D = np.stack([p0[0] * P0[2] - P0[0], p0[1] * P0[2] - P0[1], p1[0] * P1[2] - P1[0], p1[1] * P1[2] - P1[1]])
    v = np.linalg.svd(D)[2][-1]
    X = v[:3] / v[3]
    return float(np.linalg.norm(D @ np.append(X, 1.0)) / 4), X


def stereo_landmarks(pose_drone, ext_up, ext_down, norm_up, norm_down, ids_up, ids_down, triangle_thres):
    pu, pd = pmul(pose_drone, ext_up), pmul(pose_drone, ext_down)
    l3u, fu = np.zeros((len(norm_up), 3)), np.zeros(len(norm_up), np.uint8)
    l3d, fd = np.zeros((len(norm_down), 3)), np.zeros(len(norm_down), np.uint8)
    count = 0
    for iu, idn in zip(ids_up, ids_down):
        err, X = triangulate_point(pu[1], pu[0], pd[1], pd[0], norm_up[iu], norm_down[idn])
        pt_cam = qR(qinv(pu[1])) @ (X - pu[0])
        if err > triangle_thres or pt_cam[2] < 0:
            continue
        l3u[iu], fu[iu], l3d[idn], fd[idn] = X, 1, X, 1
        count += 1
    return count, l3u, fu, l3d, fd


def depth_landmarks(pose_drone, ext, kps_xy, lift, depth_mm, near, far, accept_min_3d_pts):
    """generate_gray_depth_image_descriptor's loop (loop_cam.cpp:266-304): kps_xy float32 pixels, lift(xy) -> normalised point, depth_mm u16 [H][W].
    -> (count, landmarks_3d f64 [n][3], flags u8 [n]).  cv::Mat::at<ushort>(Point2f) reads pixel (lrint(y), lrint(x)): round half to even."""
    n = len(kps_xy)
    l3, fl = np.zeros((n, 3)), np.zeros(n, np.uint8)
    if n <= accept_min_3d_pts:
        return 0, l3, fl
    pc = pmul(pose_drone, ext)
    R = qR(pc[1])
    count = 0
    for i in range(n):
        x, y = float(kps_xy[i][0]), float(kps_xy[i][1])
        if x < 0 or x > 640 or y < 0 or y > 480:                     # the literal gate of :276 (whatever the image size)
            continue
        px, py = int(np.rint(x)), int(np.rint(y))
        if not (0 <= px < depth_mm.shape[1] and 0 <= py < depth_mm.shape[0]):
            continue                                                 # (the reference reads whatever lies there; the product skips the key point)
        dep = float(depth_mm[py, px]) / 1000.0
        if near < dep < far:
            nx, ny = lift(np.array([[x, y]]))[0]
            l3[i] = R @ (np.array([nx, ny, 1.0]) * dep) + pc[0]
            fl[i] = 1
            count += 1
    return count, l3, fl


# ---- cv::RNG + RANSAC driver (OpenCV 3.4 modules/calib3d/src/ptsetreg.cpp) ----------------------------------------------------------------------
class CvRng:
    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state or 0xFFFFFFFF

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else self.next() % (b - a) + a


def ransac_update_num_iters(p, ep, model_points, max_iters):
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, DBL_MIN)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < DBL_MIN:
        return 0
    num, denom = math.log(num), math.log(denom)
    if denom >= 0 or -num >= max_iters * (-denom):
        return max_iters
    return int(np.rint(num / denom))


def ransac_run(count, model_points, threshold, confidence, max_iters, run_kernel, errors, check_subset=lambda idx: True):
    """-> (best model or None, mask)."""
    if count < model_points:
        return None, np.zeros(count, np.uint8)
    if count == model_points:
        m = run_kernel(list(range(count)))
        return (m, np.ones(count, np.uint8)) if m is not None else (None, np.zeros(count, np.uint8))
    rng = CvRng()
    niters, max_good = max(max_iters, 1), 0
    best, best_mask = None, np.zeros(count, np.uint8)
    t = np.float32(threshold * threshold)
    it = 0
    while it < niters:
        it += 1
        attempts, idx = 0, []
        while attempts < 10000:
            idx = []
            while len(idx) < model_points:
                v = rng.uniform(0, count)
                if v in idx:
                    continue
                idx.append(v)
            if check_subset(idx):
                break
            attempts += 1
        if attempts >= 10000:
            if it == 1:
                return None, np.zeros(count, np.uint8)
            break
        m = run_kernel(idx)
        if m is None:
            continue
        mask = (errors(m).astype(np.float32) <= t).astype(np.uint8)
        good = int(mask.sum())
        if good > max(max_good, model_points - 1):
            best, best_mask, max_good = m, mask, good
            niters = ransac_update_num_iters(confidence, (count - good) / count, model_points, niters)
    return (best, best_mask) if max_good > 0 else (None, np.zeros(count, np.uint8))


# ---- findHomography(RANSAC): only the mask matters (loop_detector.cpp:589-598) ----------------------------------------------------------------------
def _collinear(m, idx):
    i = len(idx) - 1
    for j in range(i):
        d1 = m[idx[j]] - m[idx[i]]
        for k in range(j):
            d2 = m[idx[k]] - m[idx[i]]
            if abs(d2[0] * d1[1] - d2[1] * d1[0]) <= FLT_EPSILON * (abs(d1[0]) + abs(d1[1]) + abs(d2[0]) + abs(d2[1])):
                return True
    return False


def _homography_check_subset(src, dst, idx):
    if _collinear(src, idx) or _collinear(dst, idx):
        return False
    if len(idx) == 4:
        neg = 0
        for t in ((0, 1, 2), (1, 2, 3), (0, 2, 3), (0, 1, 3)):
            A = np.array([[src[idx[k]][0], src[idx[k]][1], 1.0] for k in t])
            B = np.array([[dst[idx[k]][0], dst[idx[k]][1], 1.0] for k in t])
            neg += int(np.linalg.det(A) * np.linalg.det(B) < 0)
        if neg not in (0, 4):
            return False
    return True


def _homography_kernel(src, dst, idx):
    M, m = src[idx], dst[idx]
    n = len(idx)
    cM, cm = M.mean(0), m.mean(0)
    sM, sm = np.abs(M - cM).sum(0), np.abs(m - cm).sum(0)
    if min(abs(sM[0]), abs(sM[1]), abs(sm[0]), abs(sm[1])) < np.finfo(np.float64).eps:
        return None
    sM, sm = n / sM, n / sm
    inv_hnorm = np.array([[1 / sm[0], 0, cm[0]], [0, 1 / sm[1], cm[1]], [0, 0, 1]])
    hnorm2 = np.array([[sM[0], 0, -cM[0] * sM[0]], [0, sM[1], -cM[1] * sM[1]], [0, 0, 1]])
    x, y = (m[:, 0] - cm[0]) * sm[0], (m[:, 1] - cm[1]) * sm[1]
    X, Y = (M[:, 0] - cM[0]) * sM[0], (M[:, 1] - cM[1]) * sM[1]
    o, z = np.ones(n), np.zeros(n)
    L = np.concatenate([np.stack([X, Y, o, z, z, z, -x * X, -x * Y, -x], 1), np.stack([z, z, z, X, Y, o, -y * X, -y * Y, -y], 1)])
    w, v = np.linalg.eigh(L.T @ L)
    h0 = v[:, 0].reshape(3, 3)
    H = inv_hnorm @ h0 @ hnorm2
    if abs(H[2, 2]) < 1e-300:
        return None
    return H / H[2, 2]


def _homography_errors(src, dst, H):
    ww = 1.0 / (H[2, 0] * src[:, 0] + H[2, 1] * src[:, 1] + 1.0)
    dx = (H[0, 0] * src[:, 0] + H[0, 1] * src[:, 1] + H[0, 2]) * ww - dst[:, 0]
    dy = (H[1, 0] * src[:, 0] + H[1, 1] * src[:, 1] + H[1, 2]) * ww - dst[:, 1]
    return dx * dx + dy * dy


def find_homography_ransac(src, dst, thr=3.0, max_iters=2000, confidence=0.995):
    src, dst = np.asarray(src, np.float64).reshape(-1, 2), np.asarray(dst, np.float64).reshape(-1, 2)
    H, mask = ransac_run(len(src), 4, thr if thr > 0 else 3.0, confidence, max_iters, lambda idx: _homography_kernel(src, dst, idx),
                         lambda H: _homography_errors(src, dst, H), lambda idx: _homography_check_subset(src, dst, idx))
    return H, mask


# ---- PnP (K = I) -------------------------------------------------------------------------------------------------------------------------------------
def rodrigues(r):
    th = np.linalg.norm(r)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    k = K / th
    return np.eye(3) + math.sin(th) * k + (1 - math.cos(th)) * (k @ k)


def pnp_dlt(X, u, idx):
    if len(idx) < 6:
        return None
    P, uu = X[idx], u[idx]
    c = P.mean(0)
    sc = np.linalg.norm(P - c, axis=1).sum()
    sc = len(idx) / sc if sc > 0 else 1.0
    p = sc * (P - c)
    n = len(idx)
    o, z = np.ones(n), np.zeros(n)
    x, y = uu[:, 0], uu[:, 1]
    A = np.concatenate([np.stack([p[:, 0], p[:, 1], p[:, 2], o, z, z, z, z, -x * p[:, 0], -x * p[:, 1], -x * p[:, 2], -x], 1),
                        np.stack([z, z, z, z, p[:, 0], p[:, 1], p[:, 2], o, -y * p[:, 0], -y * p[:, 1], -y * p[:, 2], -y], 1)])
    w, v = np.linalg.eigh(A.T @ A)
    Pm = v[:, 0].reshape(3, 4)
    M, t = Pm[:, :3], Pm[:, 3]
    d = np.linalg.det(M)
    s = abs(d) ** (1.0 / 3.0)
    if s < 1e-300:
        return None
    if d < 0:
        s = -s
    M, t = M / s, t / s
    U, S, Vt = np.linalg.svd(M)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        U[:, 2] = -U[:, 2]
        R = U @ Vt
    tt = t / sc - R @ c
    if not np.all(np.isfinite(tt)):
        return None
    return R, tt


def pnp_errors(X, u, Rt, idx=None):
    R, t = Rt
    P = X if idx is None else X[idx]
    uu = u if idx is None else u[idx]
    c = P @ R.T + t
    d = c[:, :2] / c[:, 2:3] - uu
    return (d * d).sum(1)


def pnp_refine(X, u, idx, Rt, max_iters=30):
    R, t = Rt
    P, uu = X[idx], u[idx]

    def cost(R, t):
        return float(pnp_errors(P, uu, (R, t)).sum())

    lam, e0 = 1e-3, cost(R, t)
    for _ in range(max_iters):
        c = P @ R.T + t
        iz = 1.0 / c[:, 2]
        rx, ry = c[:, 0] * iz - uu[:, 0], c[:, 1] * iz - uu[:, 1]
        z = np.zeros(len(P))
        o = np.ones(len(P))
        dcx = np.stack([z, c[:, 2], -c[:, 1], o, z, z], 1)
        dcy = np.stack([-c[:, 2], z, c[:, 0], z, o, z], 1)
        dcz = np.stack([c[:, 1], -c[:, 0], z, z, z, o], 1)
        jx = iz[:, None] * dcx - (c[:, 0] * iz * iz)[:, None] * dcz
        jy = iz[:, None] * dcy - (c[:, 1] * iz * iz)[:, None] * dcz
        JtJ = jx.T @ jx + jy.T @ jy
        Jtr = jx.T @ rx + jy.T @ ry
        improved = False
        for _try in range(8):
            A = JtJ + lam * np.diag(np.diag(JtJ) + 1e-12)
            try:
                d = np.linalg.solve(A, -Jtr)
            except np.linalg.LinAlgError:
                lam *= 10
                continue
            dR = rodrigues(d[:3])
            R1, t1 = dR @ R, dR @ t + d[3:]
            e1 = cost(R1, t1)
            if e1 < e0:
                step = float(np.linalg.norm(d))
                R, t, improved, lam = R1, t1, True, max(lam * 0.1, 1e-12)
                done = (e0 - e1) <= 1e-16 * (e0 + 1e-300) or step < 1e-14
                e0 = e1
                if done:
                    return R, t
                break
            lam *= 10
        if not improved:
            break
    return R, t


# ---- EPnP (Lepetit, Moreno-Noguer, Fua 2009) as OpenCV 3.4 implements it (modules/calib3d/src/epnp.cpp), K = I: the minimal solver of
# cv::solvePnPRansac (5 points, SOLVEPNP_EPNP) ---------------------------------------------------------------------------------------
def _epnp_null_basis(MtM, n_points):
    """the four eigenvectors of M^T M with the smallest eigenvalues, v[0] = smallest.  With exactly 5 points M has rank <= 10: the two smallest
    eigenvalues are zero and ANY basis of that plane is a valid eigen-decomposition (it depends on the SVD routine); a canonical one is taken --
    the projections of e0, e1, ... onto the plane, Gram-Schmidt -- so that two implementations walk the same numbers."""
    w, V = np.linalg.eigh(MtM)                  # ascending
    v = [V[:, i].copy() for i in range(4)]
    if n_points == 5:
        P = np.outer(v[0], v[0]) + np.outer(v[1], v[1])        # projector onto the null plane: basis independent
        basis = []
        for k in range(12):
            c = P[:, k].copy()
            for b in basis:
                c -= (c @ b) * b
            nrm = np.linalg.norm(c)
            if nrm > 1e-3:
                basis.append(c / nrm)
            if len(basis) == 2:
                break
        if len(basis) == 2:
            v[0], v[1] = basis
    for i in range(4):                          # an eigenvector's sign is arbitrary: largest component positive
        if v[i][np.argmax(np.abs(v[i]))] < 0:
            v[i] = -v[i]
    return v


def _svd3(M):
    """M = U diag(s) V^T for a 3x3 matrix through the eigen-decomposition of M^T M (s descending; a zero singular value's u is the cross product
    of the other two) -- the same arithmetic path as the product's host code, so that the two agree beyond the solver's tolerance"""
    w, V = np.linalg.eigh(M.T @ M)
    w, V = w[::-1], V[:, ::-1]
    s = np.sqrt(np.maximum(w, 0.0))
    U = np.zeros((3, 3))
    for i in range(2):
        U[:, i] = M @ V[:, i] / s[i] if s[i] > 1e-300 else np.eye(3)[:, i]
    u2 = np.cross(U[:, 0], U[:, 1])
    if s[2] > 1e-12 * max(s[0], 1e-300):
        m2 = M @ V[:, 2] / s[2]
        u2 = u2 if u2 @ m2 >= 0 else -u2
    U[:, 2] = u2
    return U, s, V


def _pinv3(A, rel=1e-10):
    """pseudo-inverse of a 3x3 matrix through the eigen-decomposition of A^T A; singular values below rel * the largest count as zero"""
    w, V = np.linalg.eigh(A.T @ A)
    inv = np.where(w > (rel * rel) * max(w.max(), 1e-300), 1.0 / np.maximum(w, 1e-300), 0.0)
    return (V * inv) @ V.T @ A.T


def epnp(X, u):
    """-> (R, t) or None.  X [n, 3], u [n, 2] normalised image points, n >= 4."""
    X, u = np.asarray(X, np.float64), np.asarray(u, np.float64)
    n = len(X)
    if n < 4:
        return None
    # choose_control_points
    c0 = X.mean(0)
    PW0 = X - c0
    dc, uct = np.linalg.eigh(PW0.T @ PW0)
    dc, uct = dc[::-1], uct[:, ::-1].T                          # descending, eigenvectors as rows (cvSVD with CV_SVD_U_T)
    for i in range(3):                                          # a fixed sign per axis (the SVD's is arbitrary): largest component positive
        if uct[i, np.argmax(np.abs(uct[i]))] < 0:
            uct[i] = -uct[i]
    cws = np.stack([c0] + [c0 + math.sqrt(max(dc[i], 0.0) / n) * uct[i] for i in range(3)])
    # compute_barycentric_coordinates
    CC = (cws[1:] - cws[0]).T
    a123 = (_pinv3(CC) @ (X - cws[0]).T).T                                          # cvInvert(CC, CC_inv, CV_SVD): pseudo-inverse (coplanar points: rank 2)
    alphas = np.concatenate([1.0 - a123.sum(1, keepdims=True), a123], 1)            # [n, 4]
    # fill_M (fu = fv = 1, uc = vc = 0)
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = alphas[:, j]
        M[0::2, 3 * j + 2] = -alphas[:, j] * u[:, 0]
        M[1::2, 3 * j + 1] = alphas[:, j]
        M[1::2, 3 * j + 2] = -alphas[:, j] * u[:, 1]
    v = _epnp_null_basis(M.T @ M, n)
    # compute_L_6x10 / compute_rho
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    dv = np.array([[v[i][3 * a:3 * a + 3] - v[i][3 * b:3 * b + 3] for (a, b) in pairs] for i in range(4)])          # [4][6][3]
    L = np.zeros((6, 10))
    for r in range(6):
        d = dv[:, r]
        L[r] = [d[0] @ d[0], 2 * d[0] @ d[1], d[1] @ d[1], 2 * d[0] @ d[2], 2 * d[1] @ d[2], d[2] @ d[2], 2 * d[0] @ d[3], 2 * d[1] @ d[3], 2 * d[2] @ d[3],
                d[3] @ d[3]]
    rho = np.array([((cws[a] - cws[b]) ** 2).sum() for (a, b) in pairs])

    def lsq(A, b):
        """least squares through the eigen-decomposition of A^T A (pseudo-inverse, eigenvalues below 1e-26 x the largest = 0): the product's path"""
        w, V = np.linalg.eigh(A.T @ A)
        keep = w > 1e-26 * max(w.max(), 1e-300)
        c = (V.T @ (A.T @ b))
        return V[:, keep] @ (c[keep] / w[keep])

    def approx_1():
        b4 = lsq(L[:, [0, 1, 3, 6]], rho)
        if b4[0] < 0:
            b0 = math.sqrt(-b4[0])
            return np.array([b0, -b4[1] / b0, -b4[2] / b0, -b4[3] / b0]) if b0 > 0 else None
        b0 = math.sqrt(b4[0])
        return np.array([b0, b4[1] / b0, b4[2] / b0, b4[3] / b0]) if b0 > 0 else None

    def approx_2():
        b3 = lsq(L[:, [0, 1, 2]], rho)
        if b3[0] < 0:
            be = [math.sqrt(-b3[0]), math.sqrt(-b3[2]) if b3[2] < 0 else 0.0]
        else:
            be = [math.sqrt(b3[0]), math.sqrt(b3[2]) if b3[2] > 0 else 0.0]
        if b3[1] < 0:
            be[0] = -be[0]
        return np.array([be[0], be[1], 0.0, 0.0])

    def approx_3():
        b5 = lsq(L[:, [0, 1, 2, 3, 4]], rho)
        if b5[0] < 0:
            be = [math.sqrt(-b5[0]), math.sqrt(-b5[2]) if b5[2] < 0 else 0.0]
        else:
            be = [math.sqrt(b5[0]), math.sqrt(b5[2]) if b5[2] > 0 else 0.0]
        if b5[1] < 0:
            be[0] = -be[0]
        if be[0] == 0:
            return None
        return np.array([be[0], be[1], b5[3] / be[0], 0.0])

    def gauss_newton(be):
        be = be.copy()
        for _ in range(5):
            A = np.stack([2 * L[:, 0] * be[0] + L[:, 1] * be[1] + L[:, 3] * be[2] + L[:, 6] * be[3],
                          L[:, 1] * be[0] + 2 * L[:, 2] * be[1] + L[:, 4] * be[2] + L[:, 7] * be[3],
                          L[:, 3] * be[0] + L[:, 4] * be[1] + 2 * L[:, 5] * be[2] + L[:, 8] * be[3],
                          L[:, 6] * be[0] + L[:, 7] * be[1] + L[:, 8] * be[2] + 2 * L[:, 9] * be[3]], 1)
            bb = rho - (L[:, 0] * be[0] ** 2 + L[:, 1] * be[0] * be[1] + L[:, 2] * be[1] ** 2 + L[:, 3] * be[0] * be[2] + L[:, 4] * be[1] * be[2] +
                        L[:, 5] * be[2] ** 2 + L[:, 6] * be[0] * be[3] + L[:, 7] * be[1] * be[3] + L[:, 8] * be[2] * be[3] + L[:, 9] * be[3] ** 2)
            be = be + lsq(A, bb)
        return be

    def r_and_t(be):
        ccs = sum(be[i] * v[i].reshape(4, 3) for i in range(4))                      # control points in the camera frame
        pcs = alphas @ ccs
        if pcs[0, 2] < 0:                                                           # solve_for_sign
            ccs, pcs = -ccs, -pcs
        pc0, pw0 = pcs.mean(0), X.mean(0)
        ABt = (pcs - pc0).T @ (X - pw0)
        U, _, V = _svd3(ABt)
        R = U @ V.T
        if np.linalg.det(R) < 0:
            R[2] = -R[2]
        t = pc0 - R @ pw0
        c = X @ R.T + t
        err = np.sqrt(((c[:, :2] / c[:, 2:3] - u) ** 2).sum(1)).sum() / n
        return R, t, err

    best = None
    for ap in (approx_1, approx_2, approx_3):
        be = ap()
        if be is None or not np.all(np.isfinite(be)):
            continue
        R, t, err = r_and_t(gauss_newton(be))
        if np.isfinite(err) and (best is None or err < best[2]):
            best = (R, t, err)
    return None if best is None else (best[0], best[1])


def solve_pnp_ransac(X, u, iterations, reproj_error, confidence):
    """cv::solvePnPRansac (OpenCV 3.4, flags = SOLVEPNP_ITERATIVE): RANSAC over EPnP models of 5 points, then solvePnP(ITERATIVE) on the inliers
    = a DLT start + the Levenberg-Marquardt least-squares refit."""
    X, u = np.asarray(X, np.float64).reshape(-1, 3), np.asarray(u, np.float64).reshape(-1, 2)
    n = len(X)
    if n < 6:
        return None, []

    def kernel(idx):
        return epnp(X[idx], u[idx])

    best, mask = ransac_run(n, 5, reproj_error, confidence, iterations, kernel, lambda m: pnp_errors(X, u, m))
    if best is None:
        return None, []
    inl = [int(i) for i in np.nonzero(mask)[0]]
    if len(inl) < 6:
        return None, []
    fit = pnp_dlt(X, u, inl) or best
    return pnp_refine(X, u, inl, fit, 30), inl


def pnp_res_to_cam_pose(Rt):
    R, t = Rt
    return (R.T @ (-t), q_from_R(R.T))


def rotate_pt_norm2d(pt, q):
    p = qR(q) @ np.array([pt[0], pt[1], 1.0])
    if 0 < p[2] < 1e-3:
        p[2] = 1e-3
    if -1e-3 < p[2] < 0:
        p[2] = -1e-3
    return np.array([np.float32(p[0] / p[2]), np.float32(p[1] / p[2])], np.float64)


def rp_error(p_drone_old_in_new, drone_pose_old, drone_pose_now):
    dp6 = delta_pose(p_drone_old_in_new, drone_pose_now, False)
    pred = pmul(drone_pose_old, dp6)
    a_old, a_new = qnorm(pred[1]), qnorm(drone_pose_now[1])
    dyaw = quat2eulers(a_new)[2] - quat2eulers(a_old)[2]
    a_old = qmul(q_from_yaw(dyaw), a_old)
    return float(np.linalg.norm(quat2eulers(a_old) - quat2eulers(a_new)))


DEG2RAD = 0.01745277777777778          # loop_defines.h:24 (not pi / 180: the reference's own literal)


def pnp_result_verify(ok, init_mode, inliers, rperr, dp, min_loop_num=15, init_min=10, yaw_rad=30 * DEG2RAD, max_dis=5.0, rperr_thres=10 * DEG2RAD):
    if not ok or rperr > rperr_thres:
        return False
    need = init_min if init_mode else min_loop_num
    return inliers >= need and abs(pyaw(dp)) < yaw_rad and np.linalg.norm(dp[0]) < max_dis


# ---- compute_correspond_features / compute_relative_pose / compute_loop on dict messages ------------------------------------------------------------------
def correspond_image(new, old, bf_match):
    qi, ti, _ = bf_match(new["feature_descriptor"], old["feature_descriptor"])
    keep = [(a, b) for a, b in zip(qi.tolist(), ti.tolist()) if new["landmarks_flag"][a]]
    new_idx, old_idx = [a for a, _ in keep], [b for _, b in keep]
    if len(keep) >= 4:
        _, mask = find_homography_ransac(old["landmarks_2d"][old_idx], new["landmarks_2d"][new_idx], 3.0)
        new_idx = [a for a, m in zip(new_idx, mask) if m]
        old_idx = [b for b, m in zip(old_idx, mask) if m]
    return new_idx, old_idx


def correspond_frames(new, old, main_dir_new, main_dir_old, bf_match, max_dirs=4, min_match_per_dir=15, min_direction_loop=3):
    dirs = []
    for d in range(main_dir_new, main_dir_new + max_dirs):
        dn, do = d % max_dirs, ((main_dir_old - main_dir_new + max_dirs) % max_dirs + d) % max_dirs
        if dn < len(new["images"]) and do < len(old["images"]) and old["images"][do]["landmark_num"] > 0 and new["images"][dn]["landmark_num"] > 0:
            dirs.append((dn, do))
    mq_new, mq_old = new["images"][main_dir_new]["camera_extrinsic"][1], old["images"][main_dir_old]["camera_extrinsic"][1]
    new_3d, new_norm, old_norm, count = [], [], [], 0
    for dn, do in dirs:
        a, b = new["images"][dn], old["images"][do]
        ni, oi = correspond_image(a, b, bf_match)
        if len(ni) >= min_match_per_dir:
            count += 1
        dq_new, dq_old = qmul(qinv(mq_new), a["camera_extrinsic"][1]), qmul(qinv(mq_old), b["camera_extrinsic"][1])
        new_3d += [a["landmarks_3d"][i] for i in ni]
        old_norm += [rotate_pt_norm2d(b["landmarks_2d_norm"][i], dq_old) for i in oi]
        new_norm += [rotate_pt_norm2d(a["landmarks_2d_norm"][i], dq_new) for i in ni]
    ok = len(new_norm) > 0 and count >= min_direction_loop
    return ok, np.array(new_3d, np.float64).reshape(-1, 3), np.array(new_norm, np.float64).reshape(-1, 2), np.array(old_norm, np.float64).reshape(-1, 2)


def compute_loop(new, old, main_dir_new, main_dir_old, init_mode, bf_match, is_4dof=True, min_loop_num=15, init_min=10, **kw):
    """-> None, or dict(relative_pose, inliers, n_corr)."""
    if new["landmark_num"] < min_loop_num:
        return None
    ok, new_3d, new_norm, old_norm = correspond_frames(new, old, main_dir_new, main_dir_old, bf_match, **kw)
    if not ok or not (len(new_norm) > min_loop_num or (init_mode and len(new_norm) > init_min)):
        return None
    Rt, inl = solve_pnp_ransac(new_3d, old_norm, 1000 if init_mode else 100, 3.0, 0.99)
    if Rt is None:
        return None
    p_cam = pnp_res_to_cam_pose(Rt)
    p_drone_old_in_new = pmul(p_cam, pinv(old["images"][main_dir_old]["camera_extrinsic"]))
    dp = delta_pose(p_drone_old_in_new, new["pose_drone"], is_4dof)
    rperr = rp_error(p_drone_old_in_new, old["pose_drone"], new["pose_drone"])
    if not pnp_result_verify(True, init_mode, len(inl), rperr, dp, min_loop_num, init_min):
        return None
    return {"relative_pose": dp, "inliers": len(inl), "n_corr": len(new_norm)}
