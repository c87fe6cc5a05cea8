// geometry.hpp -- the host-side geometry of the loop-closure path (f64, no Eigen / OpenCV: neither is available to this build).
//
//   triangulate_point            triangulatePoint                          swarm_loop/src/loop_cam.cpp:73-106
//   stereo_landmarks             the up/down triangulation loop            swarm_loop/src/loop_cam.cpp:397-444 (landmarks_3d / landmarks_flag)
//   find_homography_ransac       cv::findHomography(old, new, RANSAC, 3, mask)   used at swarm_loop/src/loop_detector.cpp:589-598
//   solve_pnp_ransac             cv::solvePnPRansac(3d, 2d, K=I, D, r, t, false, iters, 3, 0.99, inliers)   loop_detector.cpp:390-391
//   rp_error, pnp_result_verify  RPerror, pnp_result_verify                loop_detector.cpp:317-353
//   rotate_pt_norm2d             rotate_pt_norm2d                          loop_detector.cpp:415-429
//
// OpenCV 3.4 is an un-vendored dependency of the reference: PARITY UNPINNED.  The RANSAC driver (cv::RNG((uint64)-1)
// multiply-with-carry generator, getSubset re-draw rule, RANSACUpdateNumIters, inlier test err <= thresh^2 in float) and the homography
// kernel (Hartley-normalised DLT through the 9x9 LtL eigen-decomposition, collinearity + orientation checkSubset) are restated from the
// published OpenCV 3.4 sources (modules/calib3d/src/ptsetreg.cpp, fundam.cpp); only the inlier MASK of findHomography is consumed by the
// reference, so the LM refinement of H is not needed.  solvePnPRansac (round 3): the minimal kernel is EPnP on 5 points restated from OpenCV's
// epnp.cpp, the final pose is, as in OpenCV (SOLVEPNP_ITERATIVE), the Levenberg-Marquardt least-squares refit on the inlier set started from
// a DLT (cvFindExtrinsicCameraParams2's own start for >= 6 non-coplanar points).
// swarm_msgs (Swarm::Pose, DeltaPose, quat2eulers) is un-vendored too; the definitions below are the conventional ones the call sites
// imply (pose composition p*q, DeltaPose(a,b) = a^-1 b or its yaw-only form, ZYX Euler angles).
#pragma once
#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

namespace omni {
namespace geom {

struct Vec2 { double x = 0, y = 0; };
struct Vec3 { double x = 0, y = 0, z = 0; };
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(Vec3 a) { return std::sqrt(dot(a, a)); }

struct Mat3 {
    double m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    Vec3 operator*(Vec3 v) const { return {m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z, m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z}; }
    Mat3 operator*(const Mat3& o) const { Mat3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { r.m[i][j] = 0; for (int k = 0; k < 3; ++k) r.m[i][j] += m[i][k] * o.m[k][j]; } return r; }
    Mat3 T() const { Mat3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[j][i]; return r; }
};
inline double det(const Mat3& a) {
    return a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1]) - a.m[0][1] * (a.m[1][0] * a.m[2][2] - a.m[1][2] * a.m[2][0]) +
           a.m[0][2] * (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]);
}

struct Quat {                      // unit quaternion, Eigen convention (w, x, y, z)
    double w = 1, x = 0, y = 0, z = 0;
    Quat normalized() const { const double n = std::sqrt(w * w + x * x + y * y + z * z); return {w / n, x / n, y / n, z / n}; }
    Quat inverse() const { return {w, -x, -y, -z}; }
    Quat operator*(const Quat& o) const {
        return {w * o.w - x * o.x - y * o.y - z * o.z, w * o.x + x * o.w + y * o.z - z * o.y, w * o.y - x * o.z + y * o.w + z * o.x, w * o.z + x * o.y - y * o.x + z * o.w};
    }
    Mat3 R() const {               // Eigen::Quaterniond::toRotationMatrix
        Mat3 r;
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
        r.m[0][0] = 1 - (tyy + tzz); r.m[0][1] = txy - twz; r.m[0][2] = txz + twy;
        r.m[1][0] = txy + twz; r.m[1][1] = 1 - (txx + tzz); r.m[1][2] = tyz - twx;
        r.m[2][0] = txz - twy; r.m[2][1] = tyz + twx; r.m[2][2] = 1 - (txx + tyy);
        return r;
    }
    Vec3 operator*(Vec3 v) const { return R() * v; }
};
inline Quat quat_from_R(const Mat3& r) {                     // Eigen's Shepperd branch order
    Quat q;
    const double t = r.m[0][0] + r.m[1][1] + r.m[2][2];
    if (t > 0) {
        double s = std::sqrt(t + 1.0); q.w = 0.5 * s; s = 0.5 / s;
        q.x = (r.m[2][1] - r.m[1][2]) * s; q.y = (r.m[0][2] - r.m[2][0]) * s; q.z = (r.m[1][0] - r.m[0][1]) * s;
    } else {
        int i = 0;
        if (r.m[1][1] > r.m[0][0]) i = 1;
        if (r.m[2][2] > r.m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(r.m[i][i] - r.m[j][j] - r.m[k][k] + 1.0);
        double v[3];
        v[i] = 0.5 * s; s = 0.5 / s;
        q.w = (r.m[k][j] - r.m[j][k]) * s; v[j] = (r.m[j][i] + r.m[i][j]) * s; v[k] = (r.m[k][i] + r.m[i][k]) * s;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}
inline Quat quat_from_yaw(double yaw) { return {std::cos(yaw / 2), 0, 0, std::sin(yaw / 2)}; }
// quat2eulers: (roll, pitch, yaw), ZYX
inline Vec3 quat2eulers(const Quat& q) {
    return {std::atan2(2 * (q.w * q.x + q.y * q.z), 1 - 2 * (q.x * q.x + q.y * q.y)), std::asin(std::max(-1.0, std::min(1.0, 2 * (q.w * q.y - q.z * q.x)))),
            std::atan2(2 * (q.w * q.z + q.x * q.y), 1 - 2 * (q.y * q.y + q.z * q.z))};
}
inline double wrap_angle(double a) { while (a > M_PI) a -= 2 * M_PI; while (a < -M_PI) a += 2 * M_PI; return a; }

struct Pose {                      // Swarm::Pose: position + attitude of a body in its parent frame
    Vec3 pos;
    Quat att;
    Pose operator*(const Pose& o) const { return {pos + att * o.pos, (att * o.att).normalized()}; }
    Pose inverse() const { const Quat qi = att.inverse(); return {-1.0 * (qi * pos), qi}; }
    double yaw() const { return quat2eulers(att).z; }
    // Pose::DeltaPose(a, b, use_yaw_only): b expressed in a (6-dof), or the 4-dof version: translation rotated by -yaw(a), yaw difference
    static Pose DeltaPose(const Pose& a, const Pose& b, bool use_yaw_only) {
        if (!use_yaw_only) return {a.att.inverse() * (b.pos - a.pos), (a.att.inverse() * b.att).normalized()};
        const double ya = a.yaw(), dyaw = wrap_angle(b.yaw() - ya);
        const Vec3 dp = b.pos - a.pos;
        return {{std::cos(-ya) * dp.x - std::sin(-ya) * dp.y, std::sin(-ya) * dp.x + std::cos(-ya) * dp.y, dp.z}, quat_from_yaw(dyaw)};
    }
};

// ---- cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 12): eigenvalues DESCENDING, eigenvectors as ROWS of V
// (the layout of cv::eigen, which findHomography's kernel indexes as V[8]) ------------------------------------------------------------
template <int N>
inline void jacobi_eigen(double A[N][N], double W[N], double V[N][N]) {
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < N; ++i) { diag += A[i][i] * A[i][i]; for (int j = i + 1; j < N; ++j) off += A[i][j] * A[i][j]; }
        if (off <= 1e-30 * (diag + 1e-300)) break;
        for (int p = 0; p < N - 1; ++p)
            for (int q = p + 1; q < N; ++q) {
                if (std::fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < N; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
                for (int k = 0; k < N; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < N; ++k) { const double vpk = V[p][k], vqk = V[q][k]; V[p][k] = c * vpk - s * vqk; V[q][k] = s * vpk + c * vqk; }
            }
    }
    int order[N];
    for (int i = 0; i < N; ++i) { order[i] = i; W[i] = A[i][i]; }
    std::sort(order, order + N, [&](int a, int b) { return W[a] > W[b]; });
    double Wt[N], Vt[N][N];
    for (int i = 0; i < N; ++i) { Wt[i] = W[order[i]]; for (int k = 0; k < N; ++k) Vt[i][k] = V[order[i]][k]; }
    for (int i = 0; i < N; ++i) { W[i] = Wt[i]; for (int k = 0; k < N; ++k) V[i][k] = Vt[i][k]; }
}

// ---- triangulatePoint (loop_cam.cpp:73-106): DLT on normalised image points of two cameras with poses (q, t) in the world; returns
// |design * [X;1]| / 4.  The smallest right singular vector of the 4x4 design matrix = the eigenvector of design^T design with the
// smallest eigenvalue (JacobiSVD in the reference). ---------------------------------------------------------------------------------
inline double triangulate_point(const Quat& q0, Vec3 t0, const Quat& q1, Vec3 t1, Vec2 p0, Vec2 p1, Vec3& point_3d) {
    const Mat3 R0t = q0.R().T(), R1t = q1.R().T();
    const Vec3 c0 = -1.0 * (R0t * t0), c1 = -1.0 * (R1t * t1);
    double P0[3][4], P1[3][4];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) { P0[i][j] = R0t.m[i][j]; P1[i][j] = R1t.m[i][j]; } }
    P0[0][3] = c0.x; P0[1][3] = c0.y; P0[2][3] = c0.z; P1[0][3] = c1.x; P1[1][3] = c1.y; P1[2][3] = c1.z;
    double D[4][4];
    for (int j = 0; j < 4; ++j) {
        D[0][j] = p0.x * P0[2][j] - P0[0][j]; D[1][j] = p0.y * P0[2][j] - P0[1][j];
        D[2][j] = p1.x * P1[2][j] - P1[0][j]; D[3][j] = p1.y * P1[2][j] - P1[1][j];
    }
    double A[4][4], W[4], V[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { A[i][j] = 0; for (int k = 0; k < 4; ++k) A[i][j] += D[k][i] * D[k][j]; }
    jacobi_eigen<4>(A, W, V);
    const double* v = V[3];
    point_3d = {v[0] / v[3], v[1] / v[3], v[2] / v[3]};
    double e2 = 0;
    for (int i = 0; i < 4; ++i) { const double e = D[i][0] * point_3d.x + D[i][1] * point_3d.y + D[i][2] * point_3d.z + D[i][3]; e2 += e * e; }
    return std::sqrt(e2) / 4;
}

// ---- the triangulation loop of generate_stereo_image_descriptor (loop_cam.cpp:397-444): for every up/down match, triangulate from the
// normalised points, keep it when err <= triangle_thres and the point is in front of the up camera; set landmarks_3d / landmarks_flag of
// BOTH images at the matched key-point indices.  Returns count_3d. -------------------------------------------------------------------------
inline int stereo_landmarks(const Pose& pose_drone, const Pose& extrinsic_up, const Pose& extrinsic_down, const std::vector<Vec2>& norm_up,
                            const std::vector<Vec2>& norm_down, const int* ids_up, const int* ids_down, int n_matches, double triangle_thres,
                            std::vector<Vec3>& l3d_up, std::vector<uint8_t>& flag_up, std::vector<Vec3>& l3d_down, std::vector<uint8_t>& flag_down) {
    const Pose pose_up = pose_drone * extrinsic_up, pose_down = pose_drone * extrinsic_down;
    l3d_up.assign(norm_up.size(), Vec3{}); flag_up.assign(norm_up.size(), 0);          // extractor_img_desc_deepnet :571-576
    l3d_down.assign(norm_down.size(), Vec3{}); flag_down.assign(norm_down.size(), 0);
    int count = 0;
    for (int i = 0; i < n_matches; ++i) {
        const int iu = ids_up[i], id = ids_down[i];
        Vec3 p;
        const double err = triangulate_point(pose_up.att, pose_up.pos, pose_down.att, pose_down.pos, norm_up[iu], norm_down[id], p);
        const Vec3 pt_cam = pose_up.att.inverse() * (p - pose_up.pos);
        if (err > triangle_thres || pt_cam.z < 0) continue;
        l3d_up[iu] = p; flag_up[iu] = 1; l3d_down[id] = p; flag_down[id] = 1;
        ++count;
    }
    return count;
}

// ---- cv::RNG (multiply-with-carry) and the RANSAC driver of cv::RANSACPointSetRegistrator (OpenCV 3.4 ptsetreg.cpp) --------------------
struct CvRng {
    uint64_t state;
    explicit CvRng(uint64_t s = 0xffffffffffffffffull) : state(s ? s : 0xffffffffull) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};
inline int ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = std::min(std::max(p, 0.), 1.); ep = std::min(std::max(ep, 0.), 1.);
    double num = std::max(1. - p, DBL_MIN), denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num); denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}
// Model: run_kernel(idx[], n) -> bool (fills the candidate), error(i) -> float squared error under the candidate, check_subset(idx[], n)
template <typename Model>
inline bool ransac_run(Model& model, int count, int model_points, double threshold, double confidence, int max_iters, std::vector<uint8_t>& best_mask) {
    if (count < model_points) return false;
    best_mask.assign(count, 0);
    std::vector<int> idx(model_points);
    if (count == model_points) {
        for (int i = 0; i < count; ++i) idx[i] = i;
        if (!model.run_kernel(idx.data(), count)) return false;
        model.keep_best();
        best_mask.assign(count, 1);
        return true;
    }
    CvRng rng;
    int niters = std::max(max_iters, 1), max_good = 0;
    std::vector<uint8_t> mask(count);
    const float t = (float)(threshold * threshold);
    for (int iter = 0; iter < niters; ++iter) {
        int i = 0, attempts = 0;
        for (; attempts < 10000; ++attempts) {                       // getSubset
            for (i = 0; i < model_points;) {
                int v, j;
                for (;;) { v = idx[i] = rng.uniform(0, count); for (j = 0; j < i; ++j) if (v == idx[j]) break; if (j == i) break; }
                ++i;
            }
            if (!model.check_subset(idx.data(), model_points)) continue;
            break;
        }
        if (attempts >= 10000) { if (iter == 0) return false; break; }
        if (!model.run_kernel(idx.data(), model_points)) continue;
        int good = 0;
        for (int k = 0; k < count; ++k) { mask[k] = model.error(k) <= t; good += mask[k]; }
        if (good > std::max(max_good, model_points - 1)) {
            std::swap(mask, best_mask);
            model.keep_best();
            max_good = good;
            niters = ransac_update_num_iters(confidence, (double)(count - good) / count, model_points, niters);
        }
    }
    return max_good > 0;
}

// ---- cv::findHomography(src, dst, RANSAC, 3, mask): only the mask is used by the reference (loop_detector.cpp:589-598) ----------------
struct HomographyModel {
    const std::vector<Vec2>& src; const std::vector<Vec2>& dst;     // points are float in OpenCV: callers pass float-valued coordinates
    double H[9], best[9];
    static bool collinear(const std::vector<Vec2>& m, const int* idx, int count) {      // haveCollinearPoints
        const int i = count - 1;
        for (int j = 0; j < i; ++j) {
            const double dx1 = m[idx[j]].x - m[idx[i]].x, dy1 = m[idx[j]].y - m[idx[i]].y;
            for (int k = 0; k < j; ++k) {
                const double dx2 = m[idx[k]].x - m[idx[i]].x, dy2 = m[idx[k]].y - m[idx[i]].y;
                if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2))) return true;
            }
        }
        return false;
    }
    bool check_subset(const int* idx, int count) const {
        if (collinear(src, idx, count) || collinear(dst, idx, count)) return false;
        if (count == 4) {
            static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
            int negative = 0;
            for (int i = 0; i < 4; ++i) {
                Mat3 A, B;
                for (int r = 0; r < 3; ++r) {
                    A.m[r][0] = src[idx[tt[i][r]]].x; A.m[r][1] = src[idx[tt[i][r]]].y; A.m[r][2] = 1;
                    B.m[r][0] = dst[idx[tt[i][r]]].x; B.m[r][1] = dst[idx[tt[i][r]]].y; B.m[r][2] = 1;
                }
                negative += det(A) * det(B) < 0;
            }
            if (negative != 0 && negative != 4) return false;
        }
        return true;
    }
    bool run_kernel(const int* idx, int count) {                       // HomographyEstimatorCallback::runKernel (M = src, m = dst)
        Vec2 cM, cm, sM, sm;
        for (int i = 0; i < count; ++i) { cm.x += dst[idx[i]].x; cm.y += dst[idx[i]].y; cM.x += src[idx[i]].x; cM.y += src[idx[i]].y; }
        cm.x /= count; cm.y /= count; cM.x /= count; cM.y /= count;
        for (int i = 0; i < count; ++i) {
            sm.x += std::fabs(dst[idx[i]].x - cm.x); sm.y += std::fabs(dst[idx[i]].y - cm.y);
            sM.x += std::fabs(src[idx[i]].x - cM.x); sM.y += std::fabs(src[idx[i]].y - cM.y);
        }
        if (std::fabs(sm.x) < DBL_EPSILON || std::fabs(sm.y) < DBL_EPSILON || std::fabs(sM.x) < DBL_EPSILON || std::fabs(sM.y) < DBL_EPSILON) return false;
        sm.x = count / sm.x; sm.y = count / sm.y; sM.x = count / sM.x; sM.y = count / sM.y;
        const double invHnorm[9] = {1. / sm.x, 0, cm.x, 0, 1. / sm.y, cm.y, 0, 0, 1};
        const double Hnorm2[9] = {sM.x, 0, -cM.x * sM.x, 0, sM.y, -cM.y * sM.y, 0, 0, 1};
        double LtL[9][9] = {}, W[9], V[9][9];
        for (int i = 0; i < count; ++i) {
            const double x = (dst[idx[i]].x - cm.x) * sm.x, y = (dst[idx[i]].y - cm.y) * sm.y;
            const double X = (src[idx[i]].x - cM.x) * sM.x, Y = (src[idx[i]].y - cM.y) * sM.y;
            const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x}, Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
            for (int j = 0; j < 9; ++j) for (int k = j; k < 9; ++k) LtL[j][k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
        }
        for (int j = 0; j < 9; ++j) for (int k = 0; k < j; ++k) LtL[j][k] = LtL[k][j];
        jacobi_eigen<9>(LtL, W, V);
        const double* h0 = V[8];
        double tmp[9], out[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { tmp[r * 3 + c] = 0; for (int k = 0; k < 3; ++k) tmp[r * 3 + c] += invHnorm[r * 3 + k] * h0[k * 3 + c]; }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { out[r * 3 + c] = 0; for (int k = 0; k < 3; ++k) out[r * 3 + c] += tmp[r * 3 + k] * Hnorm2[k * 3 + c]; }
        if (std::fabs(out[8]) < 1e-300) return false;
        for (int k = 0; k < 9; ++k) H[k] = out[k] / out[8];
        return true;
    }
    float error(int i) const {
        const double ww = 1. / (H[6] * src[i].x + H[7] * src[i].y + 1.);
        const double dx = (H[0] * src[i].x + H[1] * src[i].y + H[2]) * ww - dst[i].x, dy = (H[3] * src[i].x + H[4] * src[i].y + H[5]) * ww - dst[i].y;
        return (float)(dx * dx + dy * dy);
    }
    void keep_best() { for (int k = 0; k < 9; ++k) best[k] = H[k]; }
};
// mask[i] = 1 for the inliers of the best RANSAC model; returns false (mask all 0) when no model was found -- OpenCV's behaviour
inline bool find_homography_ransac(const std::vector<Vec2>& src, const std::vector<Vec2>& dst, double reproj_threshold, std::vector<uint8_t>& mask,
                                   double* H_out = nullptr, int max_iters = 2000, double confidence = 0.995) {
    const int n = (int)src.size();
    HomographyModel m{src, dst, {}, {}};
    bool ok;
    if (n == 4) { int idx[4] = {0, 1, 2, 3}; ok = m.run_kernel(idx, 4); if (ok) m.keep_best(); mask.assign(4, ok ? 1 : 0); }
    else ok = ransac_run(m, n, 4, reproj_threshold <= 0 ? 3 : reproj_threshold, confidence, max_iters, mask);
    if (!ok) mask.assign(n > 0 ? n : 0, 0);
    if (ok && H_out) for (int k = 0; k < 9; ++k) H_out[k] = m.best[k];
    return ok;
}

// ---- PnP: X_cam = R X + t, u = X_cam.xy / X_cam.z (K = I: the reference feeds normalised image points) ----------------------------------
struct Rt { Mat3 R; Vec3 t; };
inline Mat3 rodrigues(Vec3 r) {
    const double th = norm(r);
    Mat3 R;
    if (th < 1e-12) { R.m[0][1] = -r.z; R.m[0][2] = r.y; R.m[1][0] = r.z; R.m[1][2] = -r.x; R.m[2][0] = -r.y; R.m[2][1] = r.x; return R; }
    const double c = std::cos(th), s = std::sin(th), c1 = 1 - c, x = r.x / th, y = r.y / th, z = r.z / th;
    R.m[0][0] = c + c1 * x * x; R.m[0][1] = c1 * x * y - s * z; R.m[0][2] = c1 * x * z + s * y;
    R.m[1][0] = c1 * x * y + s * z; R.m[1][1] = c + c1 * y * y; R.m[1][2] = c1 * y * z - s * x;
    R.m[2][0] = c1 * x * z - s * y; R.m[2][1] = c1 * y * z + s * x; R.m[2][2] = c + c1 * z * z;
    return R;
}
// nearest rotation to M (polar decomposition through the eigen-decomposition of M^T M), det forced to +1
inline Mat3 nearest_rotation(const Mat3& M) {
    double A[3][3], W[3], V[3][3];
    const Mat3 MtM = M.T() * M;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = MtM.m[i][j];
    jacobi_eigen<3>(A, W, V);
    Mat3 S;                                                   // (M^T M)^-1/2
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { S.m[i][j] = 0; for (int k = 0; k < 3; ++k) S.m[i][j] += V[k][i] * V[k][j] / std::sqrt(std::max(W[k], 1e-300)); }
    Mat3 R = M * S;
    if (det(R) < 0) {                                          // reflect along the weakest axis
        Mat3 F;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) F.m[i][j] = (i == j ? 1.0 : 0.0) - 2 * V[2][i] * V[2][j];
        R = R * F;
    }
    return R;
}
// DLT on >= 6 points: 12x12 eigen problem, projection onto SO(3), sign chosen so that the points lie in front of the camera
inline bool pnp_dlt(const std::vector<Vec3>& X, const std::vector<Vec2>& u, const int* idx, int n, Rt& out) {
    if (n < 6) return false;
    Vec3 c;
    for (int i = 0; i < n; ++i) c = c + X[idx[i]];
    c = (1.0 / n) * c;
    double sc = 0;
    for (int i = 0; i < n; ++i) sc += norm(X[idx[i]] - c);
    sc = sc > 0 ? n / sc : 1.0;                                // centred, unit mean distance (conditioning)
    double A[12][12] = {}, W[12], V[12][12];
    for (int i = 0; i < n; ++i) {
        const Vec3 p = sc * (X[idx[i]] - c);
        const double x = u[idx[i]].x, y = u[idx[i]].y;
        const double r1[12] = {p.x, p.y, p.z, 1, 0, 0, 0, 0, -x * p.x, -x * p.y, -x * p.z, -x};
        const double r2[12] = {0, 0, 0, 0, p.x, p.y, p.z, 1, -y * p.x, -y * p.y, -y * p.z, -y};
        for (int j = 0; j < 12; ++j) for (int k = j; k < 12; ++k) A[j][k] += r1[j] * r1[k] + r2[j] * r2[k];
    }
    for (int j = 0; j < 12; ++j) for (int k = 0; k < j; ++k) A[j][k] = A[k][j];
    jacobi_eigen<12>(A, W, V);
    const double* p = V[11];
    Mat3 M;
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) M.m[r][cc] = p[r * 4 + cc];
    Vec3 t{p[3], p[7], p[11]};
    double s = std::cbrt(std::fabs(det(M)));
    if (s < 1e-300) return false;
    if (det(M) < 0) s = -s;
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) M.m[r][cc] /= s;
    t = (1.0 / s) * t;
    out.R = nearest_rotation(M);
    // undo the conditioning: X_cam = R (sc (X - c)) + t  ->  scale so that rotation acts on X directly
    out.t = (1.0 / sc) * t - out.R * c;
    // translation was estimated for the scaled points: R sc (X-c) + t ~ lambda x  =>  divide by sc
    return std::isfinite(out.t.x) && std::isfinite(out.t.y) && std::isfinite(out.t.z);
}
inline float pnp_error(const Rt& p, Vec3 X, Vec2 u) {
    const Vec3 c = p.R * X + p.t;
    const double dx = c.x / c.z - u.x, dy = c.y / c.z - u.y;
    return (float)(dx * dx + dy * dy);
}
// Levenberg-Marquardt on (rotation, translation) minimising the summed squared reprojection error over the listed points
inline void pnp_refine(const std::vector<Vec3>& X, const std::vector<Vec2>& u, const int* idx, int n, Rt& p, int max_iters = 30) {
    auto cost = [&](const Rt& q) { double e = 0; for (int i = 0; i < n; ++i) { const Vec3 c = q.R * X[idx[i]] + q.t; const double dx = c.x / c.z - u[idx[i]].x, dy = c.y / c.z - u[idx[i]].y; e += dx * dx + dy * dy; } return e; };
    double lambda = 1e-3, e0 = cost(p);
    for (int it = 0; it < max_iters; ++it) {
        double JtJ[6][6] = {}, Jtr[6] = {};
        for (int i = 0; i < n; ++i) {
            const Vec3 c = p.R * X[idx[i]] + p.t;
            const double iz = 1 / c.z, rx = c.x * iz - u[idx[i]].x, ry = c.y * iz - u[idx[i]].y;
            // d(proj)/d(c) then d(c)/d(omega, t) with R <- exp(omega^) R: dc = omega x c + dt
            const double a[3] = {iz, 0, -c.x * iz * iz}, b[3] = {0, iz, -c.y * iz * iz};
            const double dcx[6] = {0, c.z, -c.y, 1, 0, 0}, dcy[6] = {-c.z, 0, c.x, 0, 1, 0}, dcz[6] = {c.y, -c.x, 0, 0, 0, 1};
            double jx[6], jy[6];
            for (int k = 0; k < 6; ++k) { jx[k] = a[0] * dcx[k] + a[2] * dcz[k]; jy[k] = b[1] * dcy[k] + b[2] * dcz[k]; }
            for (int j = 0; j < 6; ++j) { Jtr[j] += jx[j] * rx + jy[j] * ry; for (int k = j; k < 6; ++k) JtJ[j][k] += jx[j] * jx[k] + jy[j] * jy[k]; }
        }
        for (int j = 0; j < 6; ++j) for (int k = 0; k < j; ++k) JtJ[j][k] = JtJ[k][j];
        bool improved = false;
        for (int tries = 0; tries < 8 && !improved; ++tries) {
            double M[6][7];
            for (int j = 0; j < 6; ++j) { for (int k = 0; k < 6; ++k) M[j][k] = JtJ[j][k] + (j == k ? lambda * (JtJ[j][j] + 1e-12) : 0); M[j][6] = -Jtr[j]; }
            bool sing = false;
            for (int cidx = 0; cidx < 6 && !sing; ++cidx) {            // Gauss-Jordan with partial pivoting
                int piv = cidx;
                for (int r = cidx + 1; r < 6; ++r) if (std::fabs(M[r][cidx]) > std::fabs(M[piv][cidx])) piv = r;
                if (std::fabs(M[piv][cidx]) < 1e-300) { sing = true; break; }
                if (piv != cidx) for (int k = 0; k < 7; ++k) std::swap(M[piv][k], M[cidx][k]);
                for (int r = 0; r < 6; ++r) if (r != cidx) { const double f = M[r][cidx] / M[cidx][cidx]; for (int k = cidx; k < 7; ++k) M[r][k] -= f * M[cidx][k]; }
            }
            if (sing) { lambda *= 10; continue; }
            double d[6];
            for (int j = 0; j < 6; ++j) d[j] = M[j][6] / M[j][j];
            Rt q;
            q.R = rodrigues({d[0], d[1], d[2]}) * p.R;
            q.t = rodrigues({d[0], d[1], d[2]}) * p.t + Vec3{d[3], d[4], d[5]};      // dc = omega x c + dt with c = R X + t
            const double e1 = cost(q);
            if (e1 < e0) {
                const double step = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
                p = q; improved = true; lambda = std::max(lambda * 0.1, 1e-12);
                const bool done = (e0 - e1) <= 1e-16 * (e0 + 1e-300) || step < 1e-14;
                e0 = e1;
                if (done) return;
            } else lambda *= 10;
        }
        if (!improved) return;
    }
}
// ---- EPnP (Lepetit, Moreno-Noguer, Fua 2009) as OpenCV 3.4 implements it (modules/calib3d/src/epnp.cpp) with K = I: the minimal solver of
// cv::solvePnPRansac (5 points, SOLVEPNP_EPNP).  Steps and names follow epnp.cpp: choose_control_points, compute_barycentric_coordinates,
// fill_M, the four null vectors of M^T M, compute_L_6x10 / compute_rho, find_betas_approx_1..3 + gauss_newton, compute_R_and_t (Arun's
// alignment), best of the three by reprojection error.  Where an SVD's output is not unique the choice is fixed (eigenvector signs: largest
// component positive; the two-dimensional null space of a 5-point M: the Gram-Schmidt of its projections of e0, e1, ...) so that this code and the
// numpy oracle walk the same numbers.
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// M = U diag(s) V^T through the eigen-decomposition of M^T M (s descending; U, V as columns)
inline void svd3(const Mat3& M, Mat3& U, double s[3], Mat3& V) {
    double A[3][3], W[3], E[3][3];
    const Mat3 MtM = M.T() * M;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = MtM.m[i][j];
    jacobi_eigen<3>(A, W, E);
    Vec3 v[3], u[3];
    for (int i = 0; i < 3; ++i) { v[i] = {E[i][0], E[i][1], E[i][2]}; s[i] = std::sqrt(std::max(W[i], 0.0)); }
    for (int i = 0; i < 2; ++i) u[i] = s[i] > 1e-300 ? (1.0 / s[i]) * (M * v[i]) : Vec3{i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, 0};
    u[2] = cross(u[0], u[1]);
    if (s[2] > 1e-12 * std::max(s[0], 1e-300)) { const Vec3 m2 = (1.0 / s[2]) * (M * v[2]); if (dot(u[2], m2) < 0) u[2] = -1.0 * u[2]; }
    for (int i = 0; i < 3; ++i) { U.m[0][i] = u[i].x; U.m[1][i] = u[i].y; U.m[2][i] = u[i].z; V.m[0][i] = v[i].x; V.m[1][i] = v[i].y; V.m[2][i] = v[i].z; }
}
// least squares x = argmin |A x - b| for a ROWS x N system (N <= 5) through the eigen-decomposition of A^T A (pseudo-inverse: eigenvalues below
// 1e-26 x the largest count as zero -- numpy.linalg.lstsq's cut-off on the singular values, squared)
template <int ROWS, int N>
inline void lstsq_small(const double (&A)[ROWS][N], const double (&b)[ROWS], double (&x)[N]) {
    double AtA[N][N], Atb[N], W[N], V[N][N];
    for (int i = 0; i < N; ++i) { Atb[i] = 0; for (int r = 0; r < ROWS; ++r) Atb[i] += A[r][i] * b[r]; for (int j = 0; j < N; ++j) { AtA[i][j] = 0; for (int r = 0; r < ROWS; ++r) AtA[i][j] += A[r][i] * A[r][j]; } }
    jacobi_eigen<N>(AtA, W, V);
    for (int i = 0; i < N; ++i) x[i] = 0;
    for (int k = 0; k < N; ++k) {
        if (!(W[k] > 1e-26 * std::max(W[0], 1e-300))) continue;
        double c = 0;
        for (int i = 0; i < N; ++i) c += V[k][i] * Atb[i];
        c /= W[k];
        for (int i = 0; i < N; ++i) x[i] += c * V[k][i];
    }
}
inline bool epnp(const std::vector<Vec3>& X, const std::vector<Vec2>& u, const int* idx, int n, Rt& out) {
    if (n < 4 || n > 64) return false;
    Vec3 P[64]; Vec2 q[64];
    for (int i = 0; i < n; ++i) { P[i] = X[idx[i]]; q[i] = u[idx[i]]; }
    // choose_control_points: the centroid and the principal axes scaled by sqrt(eigenvalue / n)
    Vec3 c0;
    for (int i = 0; i < n; ++i) c0 = c0 + P[i];
    c0 = (1.0 / n) * c0;
    double C[3][3] = {}, dc[3], uct[3][3];
    for (int i = 0; i < n; ++i) { const double d[3] = {P[i].x - c0.x, P[i].y - c0.y, P[i].z - c0.z}; for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) C[a][b] += d[a] * d[b]; }
    jacobi_eigen<3>(C, dc, uct);
    Vec3 cws[4];
    cws[0] = c0;
    for (int i = 0; i < 3; ++i) {
        int big = 0;
        for (int k = 1; k < 3; ++k) if (std::fabs(uct[i][k]) > std::fabs(uct[i][big])) big = k;
        const double sg = uct[i][big] < 0 ? -1.0 : 1.0, k = std::sqrt(std::max(dc[i], 0.0) / n);
        cws[i + 1] = c0 + (sg * k) * Vec3{uct[i][0], uct[i][1], uct[i][2]};
    }
    // compute_barycentric_coordinates: pseudo-inverse of [c1-c0 | c2-c0 | c3-c0] (cvInvert(..., CV_SVD): coplanar points give rank 2)
    Mat3 CC;
    for (int j = 0; j < 3; ++j) { const Vec3 d = cws[j + 1] - cws[0]; CC.m[0][j] = d.x; CC.m[1][j] = d.y; CC.m[2][j] = d.z; }
    Mat3 CCinv;
    {
        double A[3][3], W[3], V[3][3];
        const Mat3 CtC = CC.T() * CC;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = CtC.m[i][j];
        jacobi_eigen<3>(A, W, V);
        Mat3 S;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            S.m[i][j] = 0;
            for (int k = 0; k < 3; ++k) if (W[k] > 1e-20 * std::max(W[0], 1e-300)) S.m[i][j] += V[k][i] * V[k][j] / W[k];
        }
        CCinv = S * CC.T();
    }
    double al[64][4];
    for (int i = 0; i < n; ++i) {
        const Vec3 a = CCinv * (P[i] - cws[0]);
        al[i][1] = a.x; al[i][2] = a.y; al[i][3] = a.z; al[i][0] = 1.0 - a.x - a.y - a.z;
    }
    // fill_M (fu = fv = 1, uc = vc = 0) and M^T M
    double MtM[12][12] = {}, Wm[12], Vm[12][12];
    for (int i = 0; i < n; ++i) {
        double m1[12], m2[12];
        for (int j = 0; j < 4; ++j) { m1[3 * j] = al[i][j]; m1[3 * j + 1] = 0; m1[3 * j + 2] = -al[i][j] * q[i].x; m2[3 * j] = 0; m2[3 * j + 1] = al[i][j]; m2[3 * j + 2] = -al[i][j] * q[i].y; }
        for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) MtM[a][b] += m1[a] * m1[b] + m2[a] * m2[b];
    }
    jacobi_eigen<12>(MtM, Wm, Vm);
    double v[4][12];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 12; ++k) v[i][k] = Vm[11 - i][k];          // v[0] = the smallest eigenvalue's vector
    if (n == 5) {                                                                         // canonical basis of the two-dimensional null space
        double Pn[12][12];
        for (int a = 0; a < 12; ++a) for (int b = 0; b < 12; ++b) Pn[a][b] = v[0][a] * v[0][b] + v[1][a] * v[1][b];
        double basis[2][12];
        int nb = 0;
        for (int k = 0; k < 12 && nb < 2; ++k) {
            double c[12];
            for (int a = 0; a < 12; ++a) c[a] = Pn[a][k];
            for (int b = 0; b < nb; ++b) { double d = 0; for (int a = 0; a < 12; ++a) d += c[a] * basis[b][a]; for (int a = 0; a < 12; ++a) c[a] -= d * basis[b][a]; }
            double nr = 0;
            for (int a = 0; a < 12; ++a) nr += c[a] * c[a];
            nr = std::sqrt(nr);
            if (nr > 1e-3) { for (int a = 0; a < 12; ++a) basis[nb][a] = c[a] / nr; ++nb; }
        }
        if (nb == 2) for (int a = 0; a < 12; ++a) { v[0][a] = basis[0][a]; v[1][a] = basis[1][a]; }
    }
    for (int i = 0; i < 4; ++i) {
        int big = 0;
        for (int k = 1; k < 12; ++k) if (std::fabs(v[i][k]) > std::fabs(v[i][big])) big = k;
        if (v[i][big] < 0) for (int k = 0; k < 12; ++k) v[i][k] = -v[i][k];
    }
    // compute_L_6x10, compute_rho
    static const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    double L[6][10], rho[6];
    for (int r = 0; r < 6; ++r) {
        double d[4][3];
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 3; ++k) d[i][k] = v[i][3 * pa[r] + k] - v[i][3 * pb[r] + k];
        auto dt = [&](int a, int b) { return d[a][0] * d[b][0] + d[a][1] * d[b][1] + d[a][2] * d[b][2]; };
        const double row[10] = {dt(0, 0), 2 * dt(0, 1), dt(1, 1), 2 * dt(0, 2), 2 * dt(1, 2), dt(2, 2), 2 * dt(0, 3), 2 * dt(1, 3), 2 * dt(2, 3), dt(3, 3)};
        for (int k = 0; k < 10; ++k) L[r][k] = row[k];
        const Vec3 dd = cws[pa[r]] - cws[pb[r]];
        rho[r] = dot(dd, dd);
    }
    auto gauss_newton = [&](double (&be)[4]) {
        for (int it = 0; it < 5; ++it) {
            double A[6][4], bb[6], x[4];
            for (int r = 0; r < 6; ++r) {
                const double* l = L[r];
                A[r][0] = 2 * l[0] * be[0] + l[1] * be[1] + l[3] * be[2] + l[6] * be[3];
                A[r][1] = l[1] * be[0] + 2 * l[2] * be[1] + l[4] * be[2] + l[7] * be[3];
                A[r][2] = l[3] * be[0] + l[4] * be[1] + 2 * l[5] * be[2] + l[8] * be[3];
                A[r][3] = l[6] * be[0] + l[7] * be[1] + l[8] * be[2] + 2 * l[9] * be[3];
                bb[r] = rho[r] - (l[0] * be[0] * be[0] + l[1] * be[0] * be[1] + l[2] * be[1] * be[1] + l[3] * be[0] * be[2] + l[4] * be[1] * be[2] + l[5] * be[2] * be[2] +
                                  l[6] * be[0] * be[3] + l[7] * be[1] * be[3] + l[8] * be[2] * be[3] + l[9] * be[3] * be[3]);
            }
            lstsq_small<6, 4>(A, bb, x);
            for (int k = 0; k < 4; ++k) be[k] += x[k];
        }
    };
    // compute_R_and_t: control points in the camera frame from the betas, sign from the first point's depth, Arun's alignment, reprojection error
    auto r_and_t = [&](const double (&be)[4], Rt& rt) -> double {
        Vec3 ccs[4];
        for (int j = 0; j < 4; ++j) { double c[3] = {0, 0, 0}; for (int i = 0; i < 4; ++i) for (int k = 0; k < 3; ++k) c[k] += be[i] * v[i][3 * j + k]; ccs[j] = {c[0], c[1], c[2]}; }
        Vec3 pcs[64];
        for (int i = 0; i < n; ++i) { Vec3 p; for (int j = 0; j < 4; ++j) p = p + al[i][j] * ccs[j]; pcs[i] = p; }
        if (pcs[0].z < 0) { for (int j = 0; j < 4; ++j) ccs[j] = -1.0 * ccs[j]; for (int i = 0; i < n; ++i) pcs[i] = -1.0 * pcs[i]; }
        Vec3 pc0, pw0;
        for (int i = 0; i < n; ++i) { pc0 = pc0 + pcs[i]; pw0 = pw0 + P[i]; }
        pc0 = (1.0 / n) * pc0; pw0 = (1.0 / n) * pw0;
        Mat3 ABt;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) ABt.m[a][b] = 0;
        for (int i = 0; i < n; ++i) {
            const Vec3 a = pcs[i] - pc0, b = P[i] - pw0;
            const double av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) ABt.m[r][c] += av[r] * bv[c];
        }
        Mat3 U, V;
        double sv[3];
        svd3(ABt, U, sv, V);
        rt.R = U * V.T();
        if (det(rt.R) < 0) for (int c = 0; c < 3; ++c) rt.R.m[2][c] = -rt.R.m[2][c];
        rt.t = pc0 - rt.R * pw0;
        double sum = 0;
        for (int i = 0; i < n; ++i) { const Vec3 c = rt.R * P[i] + rt.t; const double dx = c.x / c.z - q[i].x, dy = c.y / c.z - q[i].y; sum += std::sqrt(dx * dx + dy * dy); }
        return sum / n;
    };
    bool have = false;
    double best_err = 0;
    for (int ap = 1; ap <= 3; ++ap) {
        double be[4] = {0, 0, 0, 0};
        bool ok = true;
        if (ap == 1) {                                         // betas10 columns (B11 B12 B13 B14)
            double A[6][4], x[4];
            for (int r = 0; r < 6; ++r) { A[r][0] = L[r][0]; A[r][1] = L[r][1]; A[r][2] = L[r][3]; A[r][3] = L[r][6]; }
            lstsq_small<6, 4>(A, rho, x);
            const double b0 = std::sqrt(std::fabs(x[0])), sg = x[0] < 0 ? -1.0 : 1.0;
            if (!(b0 > 0)) ok = false;
            else { be[0] = b0; be[1] = sg * x[1] / b0; be[2] = sg * x[2] / b0; be[3] = sg * x[3] / b0; }
        } else if (ap == 2) {                                  // (B11 B12 B22)
            double A[6][3], x[3];
            for (int r = 0; r < 6; ++r) { A[r][0] = L[r][0]; A[r][1] = L[r][1]; A[r][2] = L[r][2]; }
            lstsq_small<6, 3>(A, rho, x);
            if (x[0] < 0) { be[0] = std::sqrt(-x[0]); be[1] = x[2] < 0 ? std::sqrt(-x[2]) : 0.0; }
            else { be[0] = std::sqrt(x[0]); be[1] = x[2] > 0 ? std::sqrt(x[2]) : 0.0; }
            if (x[1] < 0) be[0] = -be[0];
        } else {                                               // (B11 B12 B22 B13 B23)
            double A[6][5], x[5];
            for (int r = 0; r < 6; ++r) for (int k = 0; k < 5; ++k) A[r][k] = L[r][k];
            lstsq_small<6, 5>(A, rho, x);
            if (x[0] < 0) { be[0] = std::sqrt(-x[0]); be[1] = x[2] < 0 ? std::sqrt(-x[2]) : 0.0; }
            else { be[0] = std::sqrt(x[0]); be[1] = x[2] > 0 ? std::sqrt(x[2]) : 0.0; }
            if (x[1] < 0) be[0] = -be[0];
            if (be[0] == 0) ok = false; else be[2] = x[3] / be[0];
        }
        if (!ok || !std::isfinite(be[0]) || !std::isfinite(be[1]) || !std::isfinite(be[2]) || !std::isfinite(be[3])) continue;
        gauss_newton(be);
        Rt rt;
        const double err = r_and_t(be, rt);
        if (std::isfinite(err) && (!have || err < best_err)) { have = true; best_err = err; out = rt; }
    }
    return have;
}

struct PnPModel {
    const std::vector<Vec3>& X; const std::vector<Vec2>& u;
    Rt cur, best;
    bool check_subset(const int*, int) const { return true; }
    bool run_kernel(const int* idx, int n) { return epnp(X, u, idx, n, cur); }
    float error(int i) const { return pnp_error(cur, X[i], u[i]); }
    void keep_best() { best = cur; }
};
// solvePnPRansac(objectPoints, imagePoints, K = I, no distortion, rvec, tvec, false, iterations, reprojectionError, confidence, inliers), OpenCV 3.4
// with flags = SOLVEPNP_ITERATIVE: RANSAC over EPnP models of 5 points, then solvePnP(ITERATIVE) on the inliers = a DLT start + the
// Levenberg-Marquardt least-squares refit.  inliers = indices, ascending.
inline bool solve_pnp_ransac(const std::vector<Vec3>& X, const std::vector<Vec2>& u, int iterations, double reproj_error, double confidence,
                             Rt& pose, std::vector<int>& inliers) {
    inliers.clear();
    const int n = (int)X.size();
    if (n < 6 || (int)u.size() != n) return false;
    PnPModel m{X, u, {}, {}};
    std::vector<uint8_t> mask;
    if (!ransac_run(m, n, 5, reproj_error, confidence, iterations, mask)) return false;
    for (int i = 0; i < n; ++i) if (mask[i]) inliers.push_back(i);
    if ((int)inliers.size() < 6) return false;
    Rt fit;
    if (!pnp_dlt(X, u, inliers.data(), (int)inliers.size(), fit)) fit = m.best;
    pnp_refine(X, u, inliers.data(), (int)inliers.size(), fit, 30);
    pose = fit;
    return true;
}
// PnPRestoCamPose (loop_utils.cpp:69-81): camera pose in the frame of the 3-D points from (R, t) of X_cam = R X + t
inline Pose pnp_res_to_cam_pose(const Rt& p) { const Mat3 Rwc = p.R.T(); return {Rwc * (-1.0 * p.t), quat_from_R(Rwc)}; }

// rotate_pt_norm2d (loop_detector.cpp:415-429)
inline Vec2 rotate_pt_norm2d(Vec2 pt, const Quat& q) {
    Vec3 p = q * Vec3{pt.x, pt.y, 1};
    if (p.z < 1e-3 && p.z > 0) p.z = 1e-3;
    if (p.z > -1e-3 && p.z < 0) p.z = -1e-3;
    return {(double)(float)(p.x / p.z), (double)(float)(p.y / p.z)};            // cv::Point2f
}

// RPerror (loop_detector.cpp:337-351)
inline double rp_error(const Pose& p_drone_old_in_new, const Pose& drone_pose_old, const Pose& drone_pose_now) {
    const Pose dp6 = Pose::DeltaPose(p_drone_old_in_new, drone_pose_now, false);
    const Pose predict_new_in_old = drone_pose_old * dp6;
    Quat att_new_in_old = predict_new_in_old.att.normalized();
    const Quat att_new_in_new = drone_pose_now.att.normalized();
    const double dyaw = quat2eulers(att_new_in_new).z - quat2eulers(att_new_in_old).z;
    att_new_in_old = quat_from_yaw(dyaw) * att_new_in_old;
    const Vec3 a = quat2eulers(att_new_in_old), b = quat2eulers(att_new_in_new);
    return norm(a - b);
}

struct VerifyParams {              // loop_defines.h:16-26,62 and the launch parameters of swarm_loop.cpp:221-226
    int min_loop_num = 15, init_mode_min_loop_num = 10;
    // loop_defines.h:24 spells the degree as DEG2RAD = 0.01745277777777778 (pi / 180 is 0.0174532925...): the gates sit 3e-5 below 30 and 10 degrees
    static constexpr double DEG2RAD = 0.01745277777777778;
    double accept_loop_yaw_rad = 30 * DEG2RAD, max_loop_dis = 5.0, rperr_thres = 10 * DEG2RAD;
};
// pnp_result_verify (loop_detector.cpp:317-334)
inline bool pnp_result_verify(bool pnp_success, bool init_mode, int inliers, double rperr, const Pose& dp_old_to_new, const VerifyParams& vp) {
    if (!pnp_success) return false;
    if (rperr > vp.rperr_thres) return false;
    const int need = init_mode ? vp.init_mode_min_loop_num : vp.min_loop_num;
    return inliers >= need && std::fabs(dp_old_to_new.yaw()) < vp.accept_loop_yaw_rad && norm(dp_old_to_new.pos) < vp.max_loop_dis;
}

}  // namespace geom
}  // namespace omni
