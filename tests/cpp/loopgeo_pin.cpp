// Pins omni::LoopGeometry (omni-swarm_amd/host/loop_geometry.hpp) to the TEXT of the geometry half of the reference's LoopDetector:
// pnp_result_verify, RPerror, LoopDetector::compute_relative_pose, rotate_pt_norm2d, LoopDetector::compute_correspond_features (image pair and
// frame pair) and LoopDetector::compute_loop (swarm_loop/src/loop_detector.cpp:317-836, with the two #defines of :8-9), PnPRestoCamPose
// (loop_utils.cpp:69-80), reduceVector (utils.h:18-26) and the parameter definitions of loop_params.cpp are extracted at build time by
// oracle/Makefile (oracle/_ref/loopgeo_*.inc: git-ignored, nothing of the reference enters the repository) and compiled VERBATIM into this program
// against the stand-ins of oracle/ref_build/loopgeo_shim.h; loop_defines.h is the reference's own header, included from where it lies.
//
// The four numerical kernels (BFMatcher, findHomography, solvePnPRansac, Rodrigues) and the pose algebra (Swarm::Pose, Eigen -- both absent)
// are routed to the SAME functions LoopGeometry uses, so the comparison isolates what this test is about: the control flow -- direction
// pairing, the 3-D-flag filter, what the homography mask reduces, the rotation into the main direction, which points go into the PnP, the
// feature-count and inlier gates, the ignored return value of the image-pair function, the LoopEdge fields, the edge numbering and the
// inter-drone counters.
//
// stdin: the "loop" command of tests/cpp/geometry_check.cpp, repeated;  stdout: one PROD and one REF line per command (+ CORR: the two
// correspondence sets compared element by element)
#include <cstdio>
#include <iostream>

#include "../../omni-swarm_amd/host/loop_geometry.hpp"

extern "C" int oracle_bf_match(const float* q, int nq, const float* t, int nt, int dim, int mode, int* q_idx, int* t_idx, float* dist_out);

// ---------------------------------------------------------------------------------------------------------------- the reference side
#include "../../oracle/ref_build/loopgeo_shim.h"
#include "swarm_loop/loop_defines.h"             // the reference's own header (-I <reference>/swarm_loop/include)

using namespace swarm_msgs;

namespace Eigen {                                // the pieces of Eigen :317-836 touch, on the pose algebra of host/geometry.hpp
namespace og = omni::geom;
struct Vector3d {
    og::Vec3 v;
    Vector3d() {}
    Vector3d(double a, double b, double c) : v{a, b, c} {}
    explicit Vector3d(og::Vec3 a) : v(a) {}
    double& x() { return v.x; } double& y() { return v.y; } double& z() { return v.z; }
    double x() const { return v.x; } double y() const { return v.y; } double z() const { return v.z; }
    double norm() const { return og::norm(v); }
    static Vector3d UnitZ() { return Vector3d(0, 0, 1); }
    Vector3d operator-() const { return Vector3d(-1.0 * v); }
};
inline Vector3d operator-(const Vector3d& a, const Vector3d& b) { return Vector3d(a.v - b.v); }
inline Vector3d operator*(const Vector3d& a, double s) { return Vector3d(s * a.v); }
inline Vector3d operator/(const Vector3d& a, double s) { return Vector3d(a.v.x / s, a.v.y / s, a.v.z / s); }
struct Matrix3d {
    og::Mat3 m;
    Matrix3d transpose() const { Matrix3d o; o.m = m.T(); return o; }
};
inline Vector3d operator*(const Matrix3d& a, const Vector3d& b) { return Vector3d(a.m * b.v); }
struct Quaterniond {
    og::Quat q;
    Quaterniond() {}
    explicit Quaterniond(og::Quat a) : q(a) {}
    Quaterniond inverse() const { return Quaterniond(q.inverse()); }
    Quaterniond normalized() const { return Quaterniond(q.normalized()); }
    Quaterniond operator*(const Quaterniond& o) const { return Quaterniond(q * o.q); }
    Vector3d operator*(const Vector3d& p) const { return Vector3d(q * p.v); }
};
struct AngleAxisd {
    double angle;
    AngleAxisd(double a, const Vector3d& axis) : angle(a) { assert(axis.x() == 0 && axis.y() == 0 && axis.z() == 1); (void)axis; }
};
inline Quaterniond operator*(const AngleAxisd& a, const Quaterniond& q) { return Quaterniond(og::quat_from_yaw(a.angle) * q.q); }
struct Isometry3d {
    og::Pose p;
    Isometry3d inverse() const { return Isometry3d{p.inverse()}; }
};
}  // namespace Eigen

inline Eigen::Vector3d quat2eulers(const Eigen::Quaterniond& q) { return Eigen::Vector3d(omni::geom::quat2eulers(q.q)); }

namespace Swarm {                                // Swarm::Pose (swarm_msgs, absent): the definitions both sides assume (host/geometry.hpp:17-18)
class Pose {
public:
    omni::geom::Pose p;
    Pose() {}
    explicit Pose(const omni::geom::Pose& q) : p(q) {}
    Pose(const Pose_t& m) {
        omni::PoseMsg pm;
        for (int k = 0; k < 3; ++k) pm.position[k] = m.position[k];
        for (int k = 0; k < 4; ++k) pm.quat_wxyz[k] = m.orientation[k];
        p = omni::to_pose(pm);
    }
    Pose(const Eigen::Matrix3d& R, const Eigen::Vector3d& T) { p = {T.v, omni::geom::quat_from_R(R.m)}; }
    Eigen::Quaterniond att() const { return Eigen::Quaterniond(p.att); }
    Eigen::Vector3d pos() const { return Eigen::Vector3d(p.pos); }
    double yaw() const { return p.yaw(); }
    Eigen::Vector3d rpy() const { return Eigen::Vector3d(omni::geom::quat2eulers(p.att)); }
    std::string tostr() const { return ""; }
    Eigen::Isometry3d to_isometry() const { return Eigen::Isometry3d{p}; }
    geometry_msgs::Pose to_ros_pose() const {
        geometry_msgs::Pose o;
        o.position.x = p.pos.x; o.position.y = p.pos.y; o.position.z = p.pos.z;
        o.orientation.w = p.att.w; o.orientation.x = p.att.x; o.orientation.y = p.att.y; o.orientation.z = p.att.z;
        return o;
    }
    static Pose DeltaPose(const Pose& a, const Pose& b, bool use_yaw_only = false) { return Pose(omni::geom::Pose::DeltaPose(a.p, b.p, use_yaw_only)); }
    friend Pose operator*(Pose a, Pose b) { return Pose(a.p * b.p); }
    friend Pose operator*(Pose a, Eigen::Isometry3d b) { return Pose(a.p * b.p); }
};
}  // namespace Swarm

namespace cv {
// the rotation travels from the solvePnPRansac hook to PnPRestoCamPose as the matrix itself (the "rvec" of this program is 3 x 3)
inline void cv2eigen(const Mat& m, Eigen::Matrix3d& o) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o.m.m[i][j] = m.at<double>(i, j); }
inline void cv2eigen(const Mat& m, Eigen::Vector3d& o) { o = Eigen::Vector3d(m.at<double>(0, 0), m.at<double>(1, 0), m.at<double>(2, 0)); }
}  // namespace cv

// the class, reduced to what :317-836 reads and writes (loop_detector.h:24-110); the odometry gate (:295-315, Swarm::DroneTrajectory) is a hook
class LoopDetector {
public:
    std::map<int, std::map<int, int>> inter_drone_loop_count;
    double t0 = -1;
    int loop_count = 0;
    int self_id = -1;
    bool enable_visualize = false;
    std::function<bool(LoopEdge&)> consistency;
    bool check_loop_odometry_consistency(LoopEdge& e) const { return consistency ? consistency(e) : true; }
    bool compute_loop(const FisheyeFrameDescriptor_t& new_fisheye_desc, const FisheyeFrameDescriptor_t& old_fisheye_desc, int main_dir_new, int main_dir_old,
                      std::vector<cv::Mat> img_new, std::vector<cv::Mat> img_old, LoopEdge& ret, bool init_mode = false);
    bool compute_correspond_features(const ImageDescriptor_t& new_img_desc, const ImageDescriptor_t& old_img_desc, std::vector<cv::Point2f>& new_norm_2d,
                                     std::vector<cv::Point3f>& new_3d, std::vector<int>& new_idx, std::vector<cv::Point2f>& old_norm_2d,
                                     std::vector<cv::Point3f>& old_3d, std::vector<int>& old_idx);
    bool compute_correspond_features(const FisheyeFrameDescriptor_t& new_img_desc, const FisheyeFrameDescriptor_t& old_img_desc, int main_dir_new, int main_dir_old,
                                     std::vector<cv::Point2f>& new_norm_2d, std::vector<cv::Point3f>& new_3d, std::vector<std::vector<int>>& new_idx,
                                     std::vector<cv::Point2f>& old_norm_2d, std::vector<cv::Point3f>& old_3d, std::vector<std::vector<int>>& old_idx,
                                     std::vector<int>& dirs_new, std::vector<int>& dirs_old, std::map<int, std::pair<int, int>>& index2dirindex_new,
                                     std::map<int, std::pair<int, int>>& index2dirindex_old);
    int compute_relative_pose(const std::vector<cv::Point2f> now_norm_2d, const std::vector<cv::Point3f> now_3d, const std::vector<cv::Point2f> old_norm_2d,
                              const std::vector<cv::Point3f> old_3d, Swarm::Pose old_extrinsic, Swarm::Pose drone_pose_now, Swarm::Pose drone_pose_old,
                              Swarm::Pose& DP_old_to_new, bool init_mode, int drone_id_new, int drone_id_old, std::vector<cv::DMatch>& matches, int& inlier_num);
};

#include REF_LOOPGEO_PARAMS                      // loop_params.cpp behind its include: the definitions of the launch parameters
#include REF_LOOPGEO_UTILS                       // utils.h: reduceVector
Swarm::Pose PnPRestoCamPose(cv::Mat rvec, cv::Mat tvec);
#include REF_LOOPGEO_DEFS                        // loop_detector.cpp:8-9: USE_FUNDMENTAL, MAX_LOOP_ID
#include REF_LOOPGEO_SRC                         // loop_detector.cpp:317-836
#include REF_LOOPGEO_PNPRES                      // loop_utils.cpp:69-80: PnPRestoCamPose

// ---------------------------------------------------------------------------------------------------------------- the harness
namespace {

using omni::FisheyeFrameDescriptor;
using omni::ImageDescriptor;
namespace og = omni::geom;

og::Pose read_pose() { og::Pose p; std::cin >> p.pos.x >> p.pos.y >> p.pos.z >> p.att.w >> p.att.x >> p.att.y >> p.att.z; return p; }

FisheyeFrameDescriptor read_frame() {            // the frame text of tests/test_geometry_cpu.py (frame_text)
    FisheyeFrameDescriptor f;
    int n_img;
    std::cin >> f.msg_id >> f.drone_id >> f.timestamp >> f.landmark_num;
    f.pose_drone = omni::to_msg(read_pose());
    std::cin >> n_img;
    f.images.resize(n_img);
    for (auto& im : f.images) {
        std::cin >> im.landmark_num;
        im.camera_extrinsic = omni::to_msg(read_pose());
        im.pose_drone = f.pose_drone;
        im.drone_id = f.drone_id;
        const int n = im.landmark_num;
        im.landmarks_2d.resize(n); im.landmarks_2d_norm.resize(n); im.landmarks_3d.resize(n); im.landmarks_flag.resize(n); im.feature_descriptor.resize((size_t)n * 64);
        for (int i = 0; i < n; ++i) {
            int flag;
            std::cin >> im.landmarks_2d[i].x >> im.landmarks_2d[i].y >> im.landmarks_2d_norm[i].x >> im.landmarks_2d_norm[i].y >> im.landmarks_3d[i].x >>
                im.landmarks_3d[i].y >> im.landmarks_3d[i].z >> flag;
            im.landmarks_flag[i] = (uint8_t)flag;
            for (int k = 0; k < 64; ++k) std::cin >> im.feature_descriptor[(size_t)i * 64 + k];
        }
    }
    return f;
}

Pose_t to_lcm(const omni::PoseMsg& m) {
    Pose_t p;
    for (int k = 0; k < 3; ++k) p.position[k] = m.position[k];
    for (int k = 0; k < 4; ++k) p.orientation[k] = m.quat_wxyz[k];
    return p;
}
FisheyeFrameDescriptor_t to_ref(const FisheyeFrameDescriptor& f) {
    FisheyeFrameDescriptor_t r;
    r.image_num = (int)f.images.size(); r.msg_id = f.msg_id; r.drone_id = f.drone_id; r.landmark_num = f.landmark_num;
    r.timestamp.sec = (int32_t)std::floor(f.timestamp); r.timestamp.nsec = (int32_t)std::llround((f.timestamp - std::floor(f.timestamp)) * 1e9);
    r.pose_drone = to_lcm(f.pose_drone);
    for (size_t d = 0; d < f.images.size(); ++d) {
        const ImageDescriptor& im = f.images[d];
        ImageDescriptor_t o;
        o.timestamp = r.timestamp; o.drone_id = im.drone_id; o.landmark_num = im.landmark_num; o.direction = (int)d;
        o.pose_drone = to_lcm(im.pose_drone); o.camera_extrinsic = to_lcm(im.camera_extrinsic);
        for (auto& p : im.landmarks_2d) { Point2d_t q; q.x = p.x; q.y = p.y; o.landmarks_2d.push_back(q); }
        for (auto& p : im.landmarks_2d_norm) { Point2d_t q; q.x = p.x; q.y = p.y; o.landmarks_2d_norm.push_back(q); }
        for (auto& p : im.landmarks_3d) { Point3d_t q; q.x = p.x; q.y = p.y; q.z = p.z; o.landmarks_3d.push_back(q); }
        for (auto v : im.landmarks_flag) o.landmarks_flag.push_back((int8_t)v);
        o.feature_descriptor = im.feature_descriptor;
        r.images.push_back(o);
    }
    return r;
}

void matcher(const float* q, int nq, const float* t, int nt, int dim, std::vector<omni::DMatch>& out) {     // cv::BFMatcher(NORM_L2, true), the oracle's restatement
    out.clear();
    if (nq <= 0 || nt <= 0) return;
    std::vector<int> qi(nq), ti(nq);
    std::vector<float> dd(nq);
    const int n = oracle_bf_match(q, nq, t, nt, dim, 0, qi.data(), ti.data(), dd.data());
    for (int i = 0; i < n; ++i) out.push_back({qi[i], ti[i], dd[i]});
}

void install_hooks() {
    cv::Hooks& h = cv::hooks();
    h.bf_match_l2_crosscheck = [](const cv::Mat& q, const cv::Mat& t, std::vector<cv::DMatch>& out) {
        std::vector<omni::DMatch> m;
        matcher(reinterpret_cast<const float*>(q.data), q.rows, reinterpret_cast<const float*>(t.data), t.rows, q.cols, m);
        out.clear();
        for (auto& x : m) out.push_back(cv::DMatch(x.queryIdx, x.trainIdx, x.distance));
    };
    h.find_homography_ransac = [](const std::vector<cv::Point2f>& src, const std::vector<cv::Point2f>& dst, double thr, std::vector<unsigned char>& mask) {
        std::vector<og::Vec2> a, b;
        for (auto& p : src) a.push_back({p.x, p.y});
        for (auto& p : dst) b.push_back({p.x, p.y});
        std::vector<uint8_t> mk;
        og::find_homography_ransac(a, b, thr, mk);
        mask.assign(mk.begin(), mk.end());
    };
    h.solve_pnp_ransac = [](const std::vector<cv::Point3f>& obj, const std::vector<cv::Point2f>& img, int iters, float reproj, double conf, cv::Mat& rvec, cv::Mat& tvec,
                            cv::Mat& inliers) {
        std::vector<og::Vec3> X;
        std::vector<og::Vec2> u;
        for (auto& p : obj) X.push_back({p.x, p.y, p.z});
        for (auto& p : img) u.push_back({p.x, p.y});
        og::Rt rt;
        std::vector<int> in;
        const bool ok = og::solve_pnp_ransac(X, u, iters, reproj, conf, rt, in);
        rvec.create(3, 3, CV_64F); tvec.create(3, 1, CV_64F);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) rvec.at<double>(i, j) = rt.R.m[i][j];
        tvec.at<double>(0, 0) = rt.t.x; tvec.at<double>(1, 0) = rt.t.y; tvec.at<double>(2, 0) = rt.t.z;
        inliers.create(ok ? (int)in.size() : 0, 1, CV_32S);
        for (size_t i = 0; ok && i < in.size(); ++i) inliers.at<int>((int)i, 0) = in[i];
        return ok;
    };
    h.rodrigues = [](const cv::Mat& rvec, cv::Mat& R) { R = rvec; };
}

void print_pose(const og::Pose& p) { std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g", p.pos.x, p.pos.y, p.pos.z, p.att.w, p.att.x, p.att.y, p.att.z); }

}  // namespace

int main() {
    install_hooks();
    OUTPUT_PATH = "";
    // one detector of each kind for the whole session: the edge numbering and the inter-drone counters run on
    omni::LoopGeometry g;
    g.match = matcher;
    LoopDetector ref;
    bool first = true;
    std::string cmd;
    while (std::cin >> cmd) {
        if (cmd == "verify") {
            // pnp_result_verify (:317-334) on a grid that straddles every gate by one part in 1e9: the constants (RPERR_THRES, ACCEPT_LOOP_YAW_RAD,
            // MAX_LOOP_DIS: the reference's own macros, incl. its DEG2RAD literal) and the comparison operators
            MIN_LOOP_NUM = g.MIN_LOOP_NUM; INIT_MODE_MIN_LOOP_NUM = g.INIT_MODE_MIN_LOOP_NUM;
            og::VerifyParams vp = g.verify;
            vp.min_loop_num = g.MIN_LOOP_NUM; vp.init_mode_min_loop_num = g.INIT_MODE_MIN_LOOP_NUM;
            int n = 0, bad = 0;
            const double rp[] = {0.0, RPERR_THRES * (1 - 1e-9), RPERR_THRES, RPERR_THRES * (1 + 1e-9), 10 * M_PI / 180, 1.0};
            const double yw[] = {0.0, ACCEPT_LOOP_YAW_RAD * (1 - 1e-9), ACCEPT_LOOP_YAW_RAD, ACCEPT_LOOP_YAW_RAD * (1 + 1e-9), -ACCEPT_LOOP_YAW_RAD * (1 - 1e-9),
                                 -ACCEPT_LOOP_YAW_RAD * (1 + 1e-9), 30 * M_PI / 180};
            const double ds[] = {0.0, MAX_LOOP_DIS * (1 - 1e-9), MAX_LOOP_DIS, MAX_LOOP_DIS * (1 + 1e-9)};
            const int inl[] = {0, 9, 10, 11, 14, 15, 16, 500};
            for (int ok = 0; ok < 2; ++ok) for (int im = 0; im < 2; ++im) for (int i : inl) for (double r : rp) for (double y : yw) for (double d : ds) {
                const og::Pose dp{{d, 0, 0}, og::quat_from_yaw(y)};
                const bool a = og::pnp_result_verify(ok != 0, im != 0, i, r, dp, vp), b = pnp_result_verify(ok != 0, im != 0, i, r, Swarm::Pose(dp));
                ++n; bad += a != b;
            }
            std::printf("VERIFY %d %d\n", n, bad);
            continue;
        }
        if (cmd != "loop") { std::fprintf(stderr, "unknown command %s\n", cmd.c_str()); return 2; }
        int dn, dold, init_mode, is4, reject;
        std::cin >> dn >> dold >> init_mode >> is4 >> reject;          // reject = 1: the odometry gate refuses this edge (both sides)
        const FisheyeFrameDescriptor nw = read_frame(), old = read_frame();
        if (first) { g.self_id = old.drone_id; ref.self_id = old.drone_id; first = false; }
        // launch parameters (swarm_loop.cpp:221-250), the same on both sides
        MIN_LOOP_NUM = g.MIN_LOOP_NUM; INIT_MODE_MIN_LOOP_NUM = g.INIT_MODE_MIN_LOOP_NUM; MIN_MATCH_PRE_DIR = g.MIN_MATCH_PRE_DIR;
        MIN_DIRECTION_LOOP = g.MIN_DIRECTION_LOOP; MAX_DIRS = g.MAX_DIRS; loop_cov_pos = g.loop_cov_pos; loop_cov_ang = g.loop_cov_ang;
        is_4dof = is4 != 0; g.is_4dof = is4 != 0;
        g.relative_odometry = nullptr;
        g.debug_no_reject = false;
        if (reject) {                                                   // an ego-motion estimate a mile off with a tight covariance
            g.relative_odometry = [](double, double, og::Pose& rel, double cov6[6]) { rel = og::Pose{{50, 50, 0}, og::Quat{}}; for (int k = 0; k < 6; ++k) cov6[k] = 1e-4; return true; };
            ref.consistency = [](LoopEdge& e) { return e.drone_id_a != e.drone_id_b; };          // :296-299: only intra-drone loops are gated
        } else {
            ref.consistency = nullptr;
        }
        // ---- product
        omni::LoopEdge e;
        omni::LoopGeometry::Correspondence c;
        const bool ok = g.compute_loop(nw, old, dn, dold, e, init_mode != 0, &c);
        std::printf("PROD %d %d %lld %lld %lld %d %d %.9f %.9f %d ", ok ? 1 : 0, ok ? e.pnp_inlier_num : 0, ok ? (long long)e.id : -1LL, (long long)e.keyframe_id_a,
                    (long long)e.keyframe_id_b, e.drone_id_a, e.drone_id_b, e.ts_a, e.ts_b, g.loop_count);
        print_pose(e.relative_pose);
        std::printf(" %g %g\n", e.pos_cov[0], e.ang_cov[2]);
        // ---- reference text
        const FisheyeFrameDescriptor_t rn = to_ref(nw), ro = to_ref(old);
        LoopEdge re;
        const bool rok = ref.compute_loop(rn, ro, dn, dold, std::vector<cv::Mat>(), std::vector<cv::Mat>(), re, init_mode != 0);
        og::Pose rp{{re.relative_pose.position.x, re.relative_pose.position.y, re.relative_pose.position.z},
                    og::Quat{re.relative_pose.orientation.w, re.relative_pose.orientation.x, re.relative_pose.orientation.y, re.relative_pose.orientation.z}};
        std::printf("REF %d %d %lld %lld %lld %d %d %.9f %.9f %d ", rok ? 1 : 0, rok ? re.pnp_inlier_num : 0, rok ? (long long)re.id : -1LL, (long long)re.keyframe_id_a,
                    (long long)re.keyframe_id_b, re.drone_id_a, re.drone_id_b, re.ts_a.toSec(), re.ts_b.toSec(), ref.loop_count);
        print_pose(rp);
        std::printf(" %g %g\n", re.pos_cov.x, re.ang_cov.z);
        // ---- the correspondence sets (the frame-pair function alone, both sides)
        {
            std::vector<cv::Point2f> n2, o2;
            std::vector<cv::Point3f> n3, o3;
            std::vector<std::vector<int>> ni, oi;
            std::vector<int> dnw, dol;
            std::map<int, std::pair<int, int>> m1, m2;
            const bool rs = ref.compute_correspond_features(rn, ro, dn, dold, n2, n3, ni, o2, o3, oi, dnw, dol, m1, m2);
            omni::LoopGeometry::Correspondence pc;
            const bool ps = g.compute_correspond_features(nw, old, dn, dold, pc);
            bool same = rs == ps && n2.size() == pc.new_norm_2d.size() && o2.size() == pc.old_norm_2d.size() && n3.size() == pc.new_3d.size() &&
                        o3.size() == pc.old_3d.size() && dnw == pc.dirs_new && dol == pc.dirs_old && ni == pc.new_idx && oi == pc.old_idx;
            for (size_t i = 0; same && i < n2.size(); ++i) same = n2[i].x == (float)pc.new_norm_2d[i].x && n2[i].y == (float)pc.new_norm_2d[i].y;
            for (size_t i = 0; same && i < o2.size(); ++i) same = o2[i].x == (float)pc.old_norm_2d[i].x && o2[i].y == (float)pc.old_norm_2d[i].y;
            for (size_t i = 0; same && i < n3.size(); ++i) same = n3[i].x == (float)pc.new_3d[i].x && n3[i].y == (float)pc.new_3d[i].y && n3[i].z == (float)pc.new_3d[i].z;
            for (size_t i = 0; same && i < o3.size(); ++i) same = o3[i].x == (float)pc.old_3d[i].x && o3[i].y == (float)pc.old_3d[i].y && o3[i].z == (float)pc.old_3d[i].z;
            // the index maps of the reference (:513-521) against the per-direction index lists
            size_t k = 0;
            for (size_t d = 0; same && d < ni.size(); ++d)
                for (size_t j = 0; same && j < ni[d].size(); ++j, ++k) same = m1[(int)k] == std::make_pair(dnw[d], ni[d][j]) && m2[(int)k] == std::make_pair(dol[d], oi[d][j]);
            std::printf("CORR %d %d %d %zu %zu\n", same ? 1 : 0, rs ? 1 : 0, ps ? 1 : 0, n2.size(), dnw.size());
        }
        int cnt = 0;
        for (auto& a : ref.inter_drone_loop_count) for (auto& b : a.second) cnt += b.second;
        std::printf("COUNTS %d\n", cnt);
    }
    return 0;
}
