// Stand-ins for everything the reference's LoopDetector::on_image_recv / add_to_database / query_from_database /
// query_fisheyeframe_from_database / database_size (swarm_loop/src/loop_detector.cpp:11-137,150-292) touch, so that the reference's own
// function TEXT compiles verbatim into oracle/_ref/libref_detector.so (ROS, OpenCV, faiss, LCM, swarm_msgs are not available here).
// TEST INFRASTRUCTURE ONLY: it pins oracle/match_ref.LoopDetectorRef -- the restatement every detector parity test compares against.
//
// What is a stand-in (interfaces only; none of the decision logic):
//   faiss::IndexFlatIP          add / search / ntotal over oracle_ip_search (oracle/csrc/oracle.c: exact IP, ties -> lower row): the SAME
//                               search the Python restatement calls, so what is pinned is the decision rule around it
//   ImageDescriptor_t, FisheyeFrameDescriptor_t   the fields those functions read (swarm_msgs is un-vendored)
//   ROS_INFO / ROS_WARN / ...   no-ops;  TicToc, toROSTime, Swarm::Pose, Swarm::DroneTrajectory, cv::Mat: inert
//   LoopCam::get_camera_configuration             returns the configured CameraConfig
//   LoopDetector::compute_loop                    the geometry stage (loop_detector.cpp:627-836) is OUTSIDE the pinned text: here a callback
//                                                 gives its verdict, and the two counter increments it makes on success (:826-827) are restated
//   LoopDetector::decode_image                    never reached (enable_visualize = false)
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <map>
#include <set>
#include <vector>

extern "C" void oracle_ip_search(const float* db, long n, int d, const float* q, int nq, int k, float* D, int64_t* I);

#define ROS_INFO(...) do { } while (0)
#define ROS_WARN(...) do { } while (0)
#define ROS_DEBUG(...) do { } while (0)
#define ROS_ERROR(...) do { } while (0)
#define REMOTE_MAGIN_NUMBER 1000000      /* loop_detector.h:22 */
#define SEARCH_NEAREST_NUM 5             /* loop_defines.h:32 */
#define CV_8U 0

// launch parameters (loop_defines.h externs, swarm_loop.cpp:221-237): set per detector instance before every call by the wrapper
static int MIN_LOOP_NUM = 15, MIN_DIRECTION_LOOP = 3, MATCH_INDEX_DIST = 10, inter_drone_init_frames = 50, width = 0, height = 0;
static double INNER_PRODUCT_THRES = 0.6, INIT_MODE_PRODUCT_THRES = 0.3;

namespace faiss {
struct Index { typedef int64_t idx_t; };
struct IndexFlatIP {
    int d; int64_t ntotal = 0; std::vector<float> xb;
    explicit IndexFlatIP(int d_) : d(d_) {}
    void add(int64_t n, const float* x) { xb.insert(xb.end(), x, x + n * d); ntotal += n; }
    void search(int64_t n, const float* x, int64_t k, float* D, Index::idx_t* I) const {
        static const float none = 0.f;
        oracle_ip_search(ntotal ? xb.data() : &none, (long)ntotal, d, x, (int)n, (int)k, D, I);
    }
};
}  // namespace faiss

namespace cv {
struct Scalar { Scalar(int) {} };
struct Mat { Mat() {} Mat(int, int, int, Scalar) {} bool empty() const { return true; } };
}  // namespace cv

struct TicToc { double toc() { return 0; } };
struct RosTimeStub { double toSec() const { return 0; } };
template <typename T> static RosTimeStub toROSTime(const T&) { return RosTimeStub(); }
struct PoseMsgStub {};
namespace Swarm {
struct Pose { Pose() {} explicit Pose(const PoseMsgStub&) {} };
struct DroneTrajectory { void push(RosTimeStub, Pose) {} };
}  // namespace Swarm
namespace swarm_msgs { struct LoopEdge {}; }
using swarm_msgs::LoopEdge;

struct Point2fStub { float x, y; };
struct ImageDescriptor_t {
    int drone_id = 0, landmark_num = 0;
    std::vector<float> image_desc, feature_descriptor;
    std::vector<Point2fStub> landmarks_2d;
    std::vector<uint8_t> image;
};
struct FisheyeFrameDescriptor_t {
    int64_t timestamp = 0; int64_t msg_id = 0; int drone_id = 0, landmark_num = 0; bool prevent_adding_db = false;
    PoseMsgStub pose_drone;
    std::vector<ImageDescriptor_t> images;
};

enum CameraConfig { STEREO_PINHOLE = 0, STEREO_FISHEYE = 1, PINHOLE_DEPTH = 2 };     // loop_defines.h:110-115
struct LoopCam { CameraConfig cfg = STEREO_FISHEYE; CameraConfig get_camera_configuration() const { return cfg; } };

// the members the pinned functions use, declared as swarm_loop/include/swarm_loop/loop_detector.h:24-111 declares them
class LoopDetector {
public:
    faiss::IndexFlatIP local_index{4096}, remote_index{4096};
    std::map<int, int64_t> imgid2fisheye;
    std::map<int, int> imgid2dir;
    std::map<int, std::map<int, int>> inter_drone_loop_count;
    std::map<int64_t, FisheyeFrameDescriptor_t> fisheyeframe_database;
    std::map<int64_t, std::vector<cv::Mat>> msgid2cvimgs;
    double t0 = -1;
    std::set<int> all_nodes;
    Swarm::DroneTrajectory ego_motion_traj;
    int self_id = -1;
    LoopCam* loop_cam = nullptr;
    bool enable_visualize = false;

    void on_image_recv(const FisheyeFrameDescriptor_t& img_des, std::vector<cv::Mat> img = std::vector<cv::Mat>(0));
    // the reference's text of these is compiled under the names *_REF (detector_wrap.cpp renames the identifiers with two macros around the
    // snippet, the text itself is untouched); on_image_recv reaches them through the two observers below
    int add_to_database_REF(const FisheyeFrameDescriptor_t& new_fisheye_desc);
    int add_to_database_REF(const ImageDescriptor_t& new_img_desc);
    FisheyeFrameDescriptor_t& query_fisheyeframe_from_database_REF(const FisheyeFrameDescriptor_t& new_img_desc, bool init_mode, bool nonkeyframe, int& direction_new, int& direction_old);
    int query_from_database(const ImageDescriptor_t& new_img_desc, bool init_mode, bool nonkeyframe, double& distance);
    int query_from_database(const ImageDescriptor_t& new_img_desc, faiss::IndexFlatIP& index, bool remote_db, double thres, int max_index, double& distance);
    int database_size() const;
    cv::Mat decode_image(const ImageDescriptor_t&) { return cv::Mat(); }

    // ---- outside the pinned text -----------------------------------------------------------------------------------------------------
    // what one on_image_recv call did, as far as it can be observed without touching the function's text
    struct Trace { int64_t old_msg_id = -1; int added = 0, queried = 0, init_mode = 0, dir_new = -1, dir_old = -1, loop = 0, n_compute_loop = 0; } tr;
    int add_to_database(const FisheyeFrameDescriptor_t& f) { tr.added = 1; return add_to_database_REF(f); }
    FisheyeFrameDescriptor_t& query_fisheyeframe_from_database(const FisheyeFrameDescriptor_t& f, bool init_mode, bool nonkeyframe, int& direction_new, int& direction_old) {
        tr.queried = 1; tr.init_mode = init_mode;
        FisheyeFrameDescriptor_t& r = query_fisheyeframe_from_database_REF(f, init_mode, nonkeyframe, direction_new, direction_old);
        tr.dir_new = direction_new; tr.dir_old = direction_old;
        if (direction_old >= 0) tr.old_msg_id = r.msg_id;          // (the not-found return value is a dangling reference in the reference: not touched)
        return r;
    }
    std::function<int(int64_t new_msg_id, int64_t old_msg_id)> loop_verdict;
    bool compute_loop(const FisheyeFrameDescriptor_t& new_fisheye_desc, const FisheyeFrameDescriptor_t& old_fisheye_desc, int main_dir_new, int main_dir_old,
                      std::vector<cv::Mat>, std::vector<cv::Mat>, LoopEdge&, bool /*init_mode*/ = false) {
        ++tr.n_compute_loop;
        const bool ok = loop_verdict && loop_verdict(new_fisheye_desc.msg_id, old_fisheye_desc.msg_id) != 0;
        if (ok) {                                   // what the reference's compute_loop does to the detector's state on success (:826-827)
            inter_drone_loop_count[new_fisheye_desc.drone_id][old_fisheye_desc.drone_id]++;
            inter_drone_loop_count[old_fisheye_desc.drone_id][new_fisheye_desc.drone_id]++;
        }
        return ok;
    }
    void on_loop_connection(LoopEdge&) { tr.loop = 1; }
};
