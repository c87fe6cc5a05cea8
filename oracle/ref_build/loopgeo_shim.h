// TEST INFRASTRUCTURE ONLY (oracle/): stand-ins for what the geometry half of the reference's LoopDetector needs from ROS, OpenCV, Eigen and
// the un-vendored swarm_msgs package -- pnp_result_verify, RPerror, compute_relative_pose, rotate_pt_norm2d, compute_correspond_features (both),
// compute_loop (swarm_loop/src/loop_detector.cpp:317-836), PnPRestoCamPose (loop_utils.cpp:69-80), reduceVector (utils.h:18-26) and the
// parameter definitions of loop_params.cpp -- so that THEIR TEXT, extracted at build time into oracle/_ref/ (git-ignored), compiles verbatim
// into tests/cpp/loopgeo_pin.cpp and runs next to omni::LoopGeometry on the same key-frame pairs.
//
// What is a stand-in and therefore NOT pinned by that test:
//   * the numerical kernels behind cv::BFMatcher::match, cv::findHomography, cv::solvePnPRansac, cv::Rodrigues: hooks (ref_geo::hooks) that the
//     test program points at the SAME functions LoopGeometry calls -- what is compared is everything around them: which points are handed
//     to which kernel, the flag / mask / threshold logic, the direction pairing and rotation, the gates, the LoopEdge that comes out;
//   * Swarm::Pose / Eigen: declared by the test program on top of the same pose algebra LoopGeometry uses (swarm_msgs and Eigen are absent:
//     DeltaPose, quat2eulers and the pose product are the conventional definitions both sides assume);
//   * the drawing calls of compute_loop's visualisation block (enable_visualize is false in the test): empty functions;
//   * ROS_INFO etc.: no-ops; ros::Time: a double.
// loop_defines.h itself (constants such as RPERR_THRES, ACCEPT_LOOP_YAW_RAD, MAX_LOOP_DIS) is the reference's own header, included from where
// it lies.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#define ROS_INFO(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_ERROR(...) ((void)0)

namespace ros {
struct Duration { double d; double toSec() const { return d; } };
struct Time {
    double t = 0;
    Time() {}
    explicit Time(double s) : t(s) {}
    double toSec() const { return t; }
};
inline Duration operator-(const Time& a, const Time& b) { return Duration{a.t - b.t}; }
}  // namespace ros

namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
}  // namespace geometry_msgs

namespace swarm_msgs {
struct Time_t { int32_t sec = 0, nsec = 0; };
struct Pose_t { double position[3] = {0, 0, 0}; double orientation[4] = {1, 0, 0, 0}; };          // orientation as (w, x, y, z)
struct Point2d_t { float x = 0, y = 0; };
struct Point3d_t { float x = 0, y = 0, z = 0; };
struct ImageDescriptor_t {
    Time_t timestamp;
    int32_t drone_id = 0;
    int64_t msg_id = 0, frame_id = 0;
    Pose_t pose_drone, camera_extrinsic;
    int32_t landmark_num = 0;
    std::vector<Point2d_t> landmarks_2d_norm, landmarks_2d;
    std::vector<Point3d_t> landmarks_3d;
    std::vector<int8_t> landmarks_flag;
    std::vector<float> feature_descriptor;
    int32_t direction = 0;
    int32_t image_desc_size = 0, feature_descriptor_size = 0, image_size = 0;
    std::vector<float> image_desc;
};
struct FisheyeFrameDescriptor_t {
    int32_t image_num = 0;
    Time_t timestamp;
    std::vector<ImageDescriptor_t> images;
    int64_t msg_id = 0;
    Pose_t pose_drone;
    int32_t landmark_num = 0, drone_id = 0;
    bool prevent_adding_db = false;               // (FisheyeFrameDescriptor_t.lcm: set by SwarmLoop::VIOKF_callback, swarm_loop.cpp:156)
};
struct Vector3Cov { double x = 0, y = 0, z = 0; };
struct LoopEdge {                                 // the ROS message, as compute_loop fills it (:789-811)
    int64_t id = 0, keyframe_id_a = 0, keyframe_id_b = 0;
    int32_t drone_id_a = 0, drone_id_b = 0, pnp_inlier_num = 0;
    ros::Time ts_a, ts_b;
    geometry_msgs::Pose relative_pose, self_pose_a, self_pose_b;
    Vector3Cov pos_cov, ang_cov;
};
inline ros::Time toROSTime(const Time_t& t) { return ros::Time(t.sec + 1e-9 * t.nsec); }
inline geometry_msgs::Pose toROSPose(const Pose_t& p) {
    geometry_msgs::Pose o;
    o.position.x = p.position[0]; o.position.y = p.position[1]; o.position.z = p.position[2];
    o.orientation.w = p.orientation[0]; o.orientation.x = p.orientation[1]; o.orientation.y = p.orientation[2]; o.orientation.z = p.orientation[3];
    return o;
}
}  // namespace swarm_msgs

#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_64F 6
#define CV_32S 4
#define CV_RANSAC 8
#define CV_FONT_HERSHEY_SIMPLEX 0

namespace cv {
enum { NORM_L2 = 4, COLOR_GRAY2BGR = 8, FONT_HERSHEY_SIMPLEX = 0 };
struct Point2f { float x, y; Point2f() : x(0), y(0) {} Point2f(float a, float b) : x(a), y(b) {} };
inline Point2f operator+(const Point2f& a, const Point2f& b) { return Point2f(a.x + b.x, a.y + b.y); }
inline Point2f operator-(const Point2f& a, const Point2f& b) { return Point2f(a.x - b.x, a.y - b.y); }
struct Point3f { float x, y, z; Point3f() : x(0), y(0), z(0) {} Point3f(float a, float b, float c) : x(a), y(b), z(c) {} };
struct DMatch { int queryIdx, trainIdx; float distance; DMatch() : queryIdx(-1), trainIdx(-1), distance(0) {} DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), distance(d) {} };
struct Scalar { double v0; Scalar(double a = 0, double = 0, double = 0, double = 0) : v0(a) {} };
struct Rect { int x, y, width, height; Rect(int a, int b, int c, int d) : x(a), y(b), width(c), height(d) {} };
class Mat {                                      // cv::Mat's sharing semantics: copies and ROIs are views of one buffer, copyTo / clone are deep
public:
    int rows = 0, cols = 0, esz = 0;
    size_t step = 0;                             // bytes between rows
    std::shared_ptr<std::vector<unsigned char>> buf;
    unsigned char* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* user) { create(r, c, type); if (r > 0 && c > 0) std::memcpy(data, user, (size_t)r * c * esz); }     // (a copy: the views are read-only here)
    void create(int r, int c, int type) {
        rows = r; cols = c; esz = type == CV_64F ? 8 : (type == CV_8U ? 1 : (type == CV_16U ? 2 : 4));
        step = (size_t)c * esz;
        buf = std::make_shared<std::vector<unsigned char>>((size_t)r * c * esz + 8, 0);
        data = buf->data();
    }
    Mat operator()(const Rect& r) const {
        assert(r.x >= 0 && r.y >= 0 && r.x + r.width <= cols && r.y + r.height <= rows);
        Mat o = *this;
        o.rows = r.height; o.cols = r.width; o.data = data + (size_t)r.y * step + (size_t)r.x * esz;
        return o;
    }
    Mat& setTo(const Scalar& s) {
        assert(esz == 1);
        for (int r = 0; r < rows; ++r) std::memset(data + (size_t)r * step, (int)s.v0, (size_t)cols);
        return *this;
    }
    template <typename T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + (size_t)c * esz); }
    template <typename T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + (size_t)r * step + (size_t)c * esz); }
    // Mat::at(Point pt) with a Point2f argument: cv::Point_<int>(const Point_<float>&) = saturate_cast<int> = cvRound (round half to even), then row pt.y, column pt.x
    template <typename T> const T& at(Point2f p) const { return at<T>((int)std::lrint((double)p.y), (int)std::lrint((double)p.x)); }
    bool empty() const { return rows == 0 || cols == 0; }
    int channels() const { return 1; }
    Mat clone() const {
        Mat o;
        o.rows = rows; o.cols = cols; o.esz = esz; o.step = (size_t)cols * esz;
        o.buf = std::make_shared<std::vector<unsigned char>>((size_t)rows * cols * esz + 8, 0);
        o.data = o.buf->data();
        for (int r = 0; r < rows; ++r) std::memcpy(o.data + (size_t)r * o.step, data + (size_t)r * step, (size_t)cols * esz);
        return o;
    }
    void copyTo(Mat& o) const { o = clone(); }
};
template <typename T>
class Mat_ : public Mat {
public:
    Mat_(int r, int c) : Mat(r, c, sizeof(T) == 8 ? CV_64F : CV_32F) {}
    struct Init {
        Mat m; int i;
        Init& operator,(T v) { reinterpret_cast<T*>(m.data)[i++] = v; return *this; }
        operator Mat() const { return m; }
    };
    Init operator<<(T v) { Init it{*this, 0}; it, v; return it; }
};
// the four numerical kernels: hooks, set by the test program
struct Hooks {
    std::function<void(const Mat& query, const Mat& train, std::vector<DMatch>& out)> bf_match_l2_crosscheck;
    std::function<void(const std::vector<Point2f>& src, const std::vector<Point2f>& dst, double thr, std::vector<unsigned char>& mask)> find_homography_ransac;
    std::function<bool(const std::vector<Point3f>& obj, const std::vector<Point2f>& img, int iters, float reproj, double conf, Mat& rvec, Mat& tvec, Mat& inliers)> solve_pnp_ransac;
    std::function<void(const Mat& rvec, Mat& R)> rodrigues;
};
inline Hooks& hooks() { static Hooks h; return h; }
class BFMatcher {
public:
    BFMatcher(int norm_type, bool cross_check) { assert(norm_type == NORM_L2 && cross_check); (void)norm_type; (void)cross_check; }
    void match(const Mat& q, const Mat& t, std::vector<DMatch>& out) const { hooks().bf_match_l2_crosscheck(q, t, out); }
};
inline Mat findHomography(const std::vector<Point2f>& src, const std::vector<Point2f>& dst, int method, double thr, std::vector<unsigned char>& mask) {
    assert(method == CV_RANSAC); (void)method;
    hooks().find_homography_ransac(src, dst, thr, mask);
    return Mat();
}
inline bool solvePnPRansac(const std::vector<Point3f>& obj, const std::vector<Point2f>& img, const Mat& K, const Mat& D, Mat& rvec, Mat& tvec, bool use_guess, int iters,
                           float reproj, double conf, Mat& inliers) {
    assert(!use_guess && K.at<double>(0, 0) == 1.0 && K.at<double>(1, 1) == 1.0 && K.at<double>(0, 2) == 0.0 && D.empty()); (void)K; (void)D; (void)use_guess;
    return hooks().solve_pnp_ransac(obj, img, iters, reproj, conf, rvec, tvec, inliers);
}
inline void Rodrigues(const Mat& rvec, Mat& R) { hooks().rodrigues(rvec, R); }
// drawing / windows (compute_loop's visualisation block: compiled, never run)
inline void line(Mat&, Point2f, Point2f, Scalar, int = 1) {}
inline void circle(Mat&, Point2f, int, Scalar, int = 1) {}
inline void putText(Mat&, const char*, Point2f, int, double, Scalar, double = 1) {}
inline void vconcat(const Mat&, const Mat&, Mat&) {}
inline void hconcat(const Mat&, const Mat&, Mat&) {}
inline void cvtColor(const Mat&, Mat&, int) {}
inline void imshow(const char*, const Mat&) {}
inline void imwrite(const std::string&, const Mat&) {}
inline void waitKey(int) {}
}  // namespace cv

namespace swarm_msgs {
// swarm_lcm_converter.hpp (absent): element-wise conversions
inline cv::Point2f toCV(const Point2d_t& p) { return cv::Point2f(p.x, p.y); }
inline cv::Point3f toCV(const Point3d_t& p) { return cv::Point3f(p.x, p.y, p.z); }
inline std::vector<cv::Point2f> toCV(const std::vector<Point2d_t>& v) { std::vector<cv::Point2f> o; for (auto& p : v) o.push_back(toCV(p)); return o; }
inline std::vector<cv::Point3f> toCV(const std::vector<Point3d_t>& v) { std::vector<cv::Point3f> o; for (auto& p : v) o.push_back(toCV(p)); return o; }
}  // namespace swarm_msgs
