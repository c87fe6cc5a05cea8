// TEST INFRASTRUCTURE ONLY (oracle/): stand-ins for what swarm_loop/src/loop_net.cpp and the class LoopNet of
// swarm_loop/include/swarm_loop/loop_net.h need from ROS, LCM and the un-vendored swarm_msgs package, so that the reference's OWN
// text -- extracted at build time into oracle/_ref/ (git-ignored; nothing of the reference enters the repository) -- compiles
// verbatim and can be run next to omni::LoopNetWire (tests/cpp/wire_pin.cpp, tests/test_wire_cpu.py).
//
// What is a stand-in here and therefore NOT pinned by that test:
//   * the message structs: swarm_msgs' .lcm files are absent, so only the members loop_net.cpp reads or writes exist, with the types its
//     expressions need; getEncodedSize() (used for log lines and a byte counter) returns 0;
//   * generate_null_img_desc() and toLCMLoopEdge() (swarm_msgs/swarm_lcm_converter.hpp, absent): an image with landmark_num = 0 and empty
//     arrays; a field-for-field copy;
//   * lcm::LCM: publish() hands a copy of the message to ref_net::sink (the transport itself is not the reference's code);
//   * ros::Time::now(): ref_net::now, set by the test;  ROS_INFO / ROS_ERROR: no-ops.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#define ROS_INFO(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_WARN(...) ((void)0)

namespace ref_net {
inline double& now_ref() { static double t = 0; return t; }
}
namespace ros {
struct Time {
    double t;
    static Time now() { return Time{ref_net::now_ref()}; }
    double toSec() const { return t; }
};
}

// loop_defines.h / loop_params.cpp: the parameters loop_net.cpp reads
#define FEATURE_DESC_SIZE 64
bool IS_PC_REPLAY = false;
bool SEND_ALL_FEATURES = false;
int MIN_DIRECTION_LOOP = 3;

namespace swarm_msgs {
struct Time_t { int32_t sec = 0, nsec = 0; };
struct Pose_t { double position[3] = {0, 0, 0}; double orientation[4] = {1, 0, 0, 0}; };
struct Point2d_t { float x = 0, y = 0; };
struct Point3d_t { float x = 0, y = 0, z = 0; };
struct ImageDescriptor_t {
    Time_t timestamp;
    int32_t drone_id = 0;
    int64_t msg_id = 0, frame_id = 0;
    int32_t image_desc_size = 0;
    std::vector<float> image_desc;
    Pose_t pose_drone, camera_extrinsic;
    bool prevent_adding_db = false;
    int32_t landmark_num = 0;
    std::vector<Point2d_t> landmarks_2d_norm, landmarks_2d;
    std::vector<Point3d_t> landmarks_3d;
    std::vector<int8_t> landmarks_flag;
    int32_t feature_descriptor_size = 0;
    std::vector<float> feature_descriptor;
    int32_t direction = 0;
    int getEncodedSize() const { return 0; }
};
struct ImageDescriptorHeader_t {
    Time_t timestamp;
    int32_t drone_id = 0;
    int64_t msg_id = 0, frame_id = 0;
    int32_t image_desc_size = 0;
    std::vector<float> image_desc;
    Pose_t pose_drone, camera_extrinsic;
    bool prevent_adding_db = false;
    int32_t feature_num = 0, direction = 0;
    int getEncodedSize() const { return 0; }
};
struct LandmarkDescriptor_t {
    int64_t msg_id = 0, header_id = 0;
    int32_t landmark_id = 0, drone_id = 0, desc_len = 0;
    int8_t landmark_flag = 0;
    Point2d_t landmark_2d_norm, landmark_2d;
    Point3d_t landmark_3d;
    std::vector<float> feature_descriptor;
    int getEncodedSize() const { return 0; }
};
struct FisheyeFrameDescriptor_t {
    int32_t image_num = 0;
    Time_t timestamp;
    std::vector<ImageDescriptor_t> images;
    int64_t msg_id = 0;
    Pose_t pose_drone;
    int32_t landmark_num = 0, drone_id = 0;
};
struct LoopEdge_t { int64_t id = 0; int32_t drone_id_a = 0, drone_id_b = 0; };
struct LoopEdge { int64_t id = 0; int32_t drone_id_a = 0, drone_id_b = 0; };          // the ROS message of the same name
inline LoopEdge_t toLCMLoopEdge(const LoopEdge& e) { LoopEdge_t o; o.id = e.id; o.drone_id_a = e.drone_id_a; o.drone_id_b = e.drone_id_b; return o; }
inline ImageDescriptor_t generate_null_img_desc() { ImageDescriptor_t d; d.landmark_num = 0; return d; }
}  // namespace swarm_msgs

namespace ref_net {
// what a LoopNet published, in order
struct Published {
    std::string channel;
    swarm_msgs::ImageDescriptorHeader_t header;
    swarm_msgs::LandmarkDescriptor_t landmark;
    swarm_msgs::ImageDescriptor_t image;
};
inline std::vector<Published>& sink() { static std::vector<Published> s; return s; }
inline void capture(const std::string& ch, const swarm_msgs::ImageDescriptorHeader_t* m) { Published p; p.channel = ch; p.header = *m; sink().push_back(p); }
inline void capture(const std::string& ch, const swarm_msgs::LandmarkDescriptor_t* m) { Published p; p.channel = ch; p.landmark = *m; sink().push_back(p); }
inline void capture(const std::string& ch, const swarm_msgs::ImageDescriptor_t* m) { Published p; p.channel = ch; p.image = *m; sink().push_back(p); }
inline void capture(const std::string& ch, const swarm_msgs::LoopEdge_t*) { Published p; p.channel = ch; sink().push_back(p); }
}  // namespace ref_net

namespace lcm {
struct ReceiveBuffer {};
class LCM {
public:
    explicit LCM(const std::string&) {}
    bool good() const { return true; }
    int handle() { return 0; }
    template <class M, class C>
    void subscribe(const std::string&, void (C::*)(const ReceiveBuffer*, const std::string&, const M*), C*) {}
    template <class M>
    int publish(const std::string& ch, const M* m) { ref_net::capture(ch, m); return 0; }
};
}  // namespace lcm
