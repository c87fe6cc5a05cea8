// ingest_pin.cpp -- TEST INFRASTRUCTURE.  omni::KeyframeIntake (host/keyframe_intake.hpp) next to the reference's own key-frame intake, compiled
// from its text: SwarmLoop::find_images_raw, odometry_callback, odometry_keyframe_callback, VIOnonKF_callback, VIOKF_callback, pub_node_frame
// (/root/reference/swarm_loop/src/swarm_loop.cpp:32-53, 100-187; extracted by oracle/Makefile into oracle/_ref/swarmloop_*.inc) against stand-in
// ROS types (a double for ros::Time, plain structs for the messages, recording stubs for LoopCam / LoopNet / LoopDetector / the publisher).
// stdin:  max_freq min_movement nonkf_waitsec n_events, then per event:  I stamp idx landmark_num   (a camera frame is queued)
//                                                                         O stamp x y z            (an odometry message)
//                                                                         K stamp x y z            (a key-frame pose)
// stdout: for both sides (REF / PROD) every frame that reached the CNN (EXTRACT idx stamp prevent pose landmark_num), every frame that went on to the
// network, the detector and the node_frame topic (DELIVER idx ...), every key-frame pose without a frame (MISS stamp), and the final state.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <mutex>
#include <queue>
#include <vector>

#include "../../omni-swarm_amd/host/keyframe_intake.hpp"
#include "../../oracle/ref_build/loopgeo_shim.h"

using namespace swarm_msgs;
using namespace std::chrono;

namespace Eigen {
struct Vector3d {
    double v[3];
    Vector3d(double x = 0, double y = 0, double z = 0) : v{x, y, z} {}
    Vector3d operator-(const Vector3d& o) const { return Vector3d(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
};
}  // namespace Eigen
namespace std_msgs { struct Header { ros::Time stamp; }; }
namespace geometry_msgs { struct PoseWithCovariance { Pose pose; }; }
namespace nav_msgs { struct Odometry { std_msgs::Header header; geometry_msgs::PoseWithCovariance pose; }; }
namespace swarm_msgs {
struct node_frame { std_msgs::Header header; geometry_msgs::Point position; geometry_msgs::Quaternion quat; bool vo_available = false; int drone_id = 0; int64_t keyframe_id = 0; };
}
#undef ROS_WARN
static std::vector<double> g_miss;
#define ROS_WARN(fmt, ...) ref_warn(fmt, ##__VA_ARGS__)
static void ref_warn(const char*, ...) {}
static void ref_warn(const char* fmt, double stamp) { if (std::string(fmt).find("not found") != std::string::npos) g_miss.push_back(stamp); }
static void ref_warn(const char*, int) {}

struct StereoFrame {                              // loop_cam.h:29-77 without the images (the intake does not look at them)
    ros::Time stamp;
    int keyframe_id = 0;
    geometry_msgs::Pose pose_drone;
    StereoFrame() : stamp(0) {}
};
struct Extracted { int idx; double stamp; bool prevent; double x, y, z; int landmark_num; };
static std::vector<int> g_landmarks;              // per frame index: what the CNN front end will report
struct LoopCam {
    std::vector<Extracted> calls;
    FisheyeFrameDescriptor_t on_flattened_images(const StereoFrame& msg, std::vector<cv::Mat>&) {
        FisheyeFrameDescriptor_t f;
        f.landmark_num = g_landmarks[msg.keyframe_id];
        f.msg_id = msg.keyframe_id;
        f.timestamp.sec = (int32_t)std::floor(msg.stamp.toSec()); f.timestamp.nsec = (int32_t)std::llround((msg.stamp.toSec() - std::floor(msg.stamp.toSec())) * 1e9);
        f.pose_drone.position[0] = msg.pose_drone.position.x; f.pose_drone.position[1] = msg.pose_drone.position.y; f.pose_drone.position[2] = msg.pose_drone.position.z;
        f.drone_id = 7;
        calls.push_back({msg.keyframe_id, msg.stamp.toSec(), false, msg.pose_drone.position.x, msg.pose_drone.position.y, msg.pose_drone.position.z, (int)f.landmark_num});
        return f;
    }
};
struct Delivered { int64_t idx; bool prevent; };
struct LoopNet { std::vector<Delivered> sent; void broadcast_fisheye_desc(FisheyeFrameDescriptor_t& f) { sent.push_back({f.msg_id, f.prevent_adding_db}); } };
struct LoopDetector { std::vector<Delivered> got; void on_image_recv(const FisheyeFrameDescriptor_t& f, std::vector<cv::Mat>) { got.push_back({f.msg_id, f.prevent_adding_db}); } };
struct Publisher { std::vector<swarm_msgs::node_frame> out; void publish(const swarm_msgs::node_frame& n) { out.push_back(n); } };

double ACCEPT_NONKEYFRAME_WAITSEC = 5.0;          // loop_params.cpp:36
#define INIT_ACCEPT_NONKEYFRAME_WAITSEC 1.0       // loop_defines.h:34

class SwarmLoop {                                 // swarm_loop.h: the members these functions touch
public:
    LoopDetector* loop_detector = nullptr;
    LoopCam* loop_cam = nullptr;
    LoopNet* loop_net = nullptr;
    Publisher keyframe_pub;
    double min_movement_keyframe = 0.3;
    bool received_image = false;
    ros::Time last_kftime;
    Eigen::Vector3d last_keyframe_position = Eigen::Vector3d(10000, 10000, 10000);
    std::queue<StereoFrame> raw_stereo_images;
    std::mutex raw_stereo_image_lock;
    double last_invoke = 0;
    double max_freq = 1.0;
    StereoFrame find_images_raw(const nav_msgs::Odometry& odometry);
    void odometry_callback(const nav_msgs::Odometry& odometry);
    void odometry_keyframe_callback(const nav_msgs::Odometry& odometry);
    void VIOnonKF_callback(const StereoFrame& viokf);
    void VIOKF_callback(const StereoFrame& viokf, bool nonkeyframe = false);
    void pub_node_frame(const FisheyeFrameDescriptor_t& viokf);
};

#include REF_SWARMLOOP_FIND                       // swarm_loop.cpp:32-53    find_images_raw
#include REF_SWARMLOOP_KF                         // swarm_loop.cpp:100-187  odometry_callback .. pub_node_frame

int main() {
    double max_freq, min_move, waitsec; int n;
    if (!(std::cin >> max_freq >> min_move >> waitsec >> n)) return 2;
    struct Ev { char k; double stamp, x, y, z; int idx, lm; };
    std::vector<Ev> evs(n);
    for (auto& e : evs) {
        std::cin >> e.k >> e.stamp;
        if (e.k == 'I') { std::cin >> e.idx >> e.lm; if ((int)g_landmarks.size() <= e.idx) g_landmarks.resize(e.idx + 1); g_landmarks[e.idx] = e.lm; }
        else std::cin >> e.x >> e.y >> e.z;
    }
    // ---- the reference's text ------------------------------------------------------------------------------------------------------------------
    {
        LoopCam cam; LoopNet net; LoopDetector det;
        SwarmLoop sl;
        sl.loop_cam = &cam; sl.loop_net = &net; sl.loop_detector = &det;
        sl.max_freq = max_freq; sl.min_movement_keyframe = min_move; ACCEPT_NONKEYFRAME_WAITSEC = waitsec;
        for (auto& e : evs) {
            if (e.k == 'I') {                     // what the image callbacks do (:55-98): a StereoFrame stamped with the image header's stamp is queued
                StereoFrame f; f.stamp = ros::Time(e.stamp); f.keyframe_id = e.idx;
                sl.raw_stereo_image_lock.lock(); sl.raw_stereo_images.push(f); sl.raw_stereo_image_lock.unlock();
            } else {
                nav_msgs::Odometry o; o.header.stamp = ros::Time(e.stamp); o.pose.pose.position.x = e.x; o.pose.pose.position.y = e.y; o.pose.pose.position.z = e.z;
                if (e.k == 'O') sl.odometry_callback(o); else sl.odometry_keyframe_callback(o);
            }
        }
        for (auto& c : cam.calls) std::printf("REF EXTRACT %d %.9f %.17g %.17g %.17g %d\n", c.idx, c.stamp, c.x, c.y, c.z, c.landmark_num);
        for (size_t i = 0; i < net.sent.size(); ++i) {
            const auto& nf = sl.keyframe_pub.out.at(i);
            std::printf("REF DELIVER %lld %d %d %lld %.9f %.17g\n", (long long)net.sent[i].idx, (int)net.sent[i].prevent, (int)(det.got.at(i).idx == net.sent[i].idx && det.got[i].prevent == net.sent[i].prevent),
                        (long long)nf.keyframe_id, nf.header.stamp.toSec(), nf.position.x);
        }
        if (det.got.size() != net.sent.size() || sl.keyframe_pub.out.size() != net.sent.size()) std::printf("REF SINKS-DIFFER\n");
        for (double m : g_miss) std::printf("REF MISS %.9f\n", m);
        std::printf("REF STATE %d %.9f %.9f %zu\n", (int)sl.received_image, sl.last_invoke, sl.last_kftime.toSec(), sl.raw_stereo_images.size());
    }
    // ---- the product ------------------------------------------------------------------------------------------------------------------------------
    {
        omni::KeyframeIntake<int> in;
        in.max_freq = max_freq; in.min_movement_keyframe = min_move; in.accept_nonkeyframe_waitsec = waitsec;
        struct Call { int idx; double stamp; bool prevent; double x, y, z; int lm; };
        std::vector<Call> calls;
        in.extract = [&](const omni::KeyframeIntake<int>::Queued& q, bool prevent) {
            calls.push_back({q.frame, q.stamp, prevent, q.pose_drone.position[0], q.pose_drone.position[1], q.pose_drone.position[2], g_landmarks[q.frame]});
            return g_landmarks[q.frame];
        };
        std::vector<double> miss;
        for (auto& e : evs) {
            omni::PoseMsg p; p.position[0] = e.x; p.position[1] = e.y; p.position[2] = e.z;
            if (e.k == 'I') in.push_images(e.stamp, e.idx);
            else if (e.k == 'O') in.odometry(e.stamp, p);
            else if (!in.odometry_keyframe(e.stamp, p)) miss.push_back(e.stamp);
        }
        for (auto& c : calls) std::printf("PROD EXTRACT %d %.9f %.17g %.17g %.17g %d\n", c.idx, c.stamp, c.x, c.y, c.z, c.lm);
        for (auto& c : calls)
            if (c.lm != 0) std::printf("PROD DELIVER %d %d 1 %d %.9f %.17g\n", c.idx, (int)c.prevent, c.idx, c.stamp, c.x);
        for (double m : miss) std::printf("PROD MISS %.9f\n", m);
        std::printf("PROD STATE %d %.9f %.9f %zu\n", (int)in.received_image(), in.last_invoke(), in.last_kftime(), in.queued());
    }
    return 0;
}
