// params_pin.cpp -- TEST INFRASTRUCTURE.  The parameter block of the reference's SwarmLoop::Init (/root/reference/swarm_loop/src/swarm_loop.cpp:205-270:
// its local declarations and the 50 nh.param<T>(name, variable, default) calls; extracted by oracle/Makefile into oracle/_ref/swarmloop_params.inc) compiled
// VERBATIM against a stand-in ros::NodeHandle, with the reference's own loop_defines.h (included from where it lies) and its own loop_params.cpp (compiled as
// a second translation unit: the globals the block sets).  Pins omni::SwarmLoopParams (host/swarm_loop_params.hpp):
//   params_pin defaults            -> one line per call, in call order:  name <TAB> type <TAB> default      (what the table must hold)
//   params_pin apply < typed.tsv   -> the parameter server's content on stdin (name <TAB> I|D|B|S <TAB> value, as an independent reader -- Python's xml.etree
//                                     + PyYAML, the library roslaunch uses -- took it from a launch file); the block runs with roscpp's param<T> conversions
//                                     (ros::param::getParamImpl: double <- int | double, int <- int | double rounded, bool <- boolean, string <- string; anything
//                                     else: the default stays) and every variable it set is printed:  name <TAB> value
#include <cmath>
#include <cstdio>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "swarm_loop/loop_defines.h"             // the reference's own header (-I <reference>/swarm_loop/include)

double TRIANGLE_THRES;                            // loop_cam.cpp:12 (the one global of the block that loop_params.cpp does not define)

struct Stored { char type; std::string text; };
static std::map<std::string, Stored> g_server;
struct Call { std::string name, type, def; };
static std::vector<Call> g_calls;

static std::string show(int v) { return std::to_string(v); }
static std::string show(bool v) { return v ? "true" : "false"; }
static std::string show(double v) { char b[64]; snprintf(b, sizeof(b), "%.17g", v); return b; }
static std::string show(const std::string& v) { return v; }
static const char* tname(int*) { return "I"; }
static const char* tname(bool*) { return "B"; }
static const char* tname(double*) { return "D"; }
static const char* tname(std::string*) { return "S"; }
static bool fetch(const Stored& s, int& v) {
    if (s.type == 'I') { v = (int)std::stol(s.text); return true; }
    if (s.type == 'D') { double d = std::stod(s.text); d = std::fmod(d, 1.0) < 0.5 ? std::floor(d) : std::ceil(d); v = (int)d; return true; }      // roscpp param.cpp getParamImpl(int)
    return false;
}
static bool fetch(const Stored& s, double& v) {
    if (s.type == 'I') { v = (double)(int)std::stol(s.text); return true; }
    if (s.type == 'D') { v = std::stod(s.text); return true; }
    return false;
}
static bool fetch(const Stored& s, bool& v) { if (s.type != 'B') return false; v = s.text == "true"; return true; }
static bool fetch(const Stored& s, std::string& v) { if (s.type != 'S') return false; v = s.text; return true; }

namespace ros {
struct NodeHandle {
    template <class T> bool param(const std::string& name, T& var, const T& def) const {
        g_calls.push_back({name, tname((T*)nullptr), show(def)});
        auto it = g_server.find(name);
        if (it != g_server.end() && fetch(it->second, var)) return true;
        var = def;
        return false;
    }
};
}  // namespace ros
namespace cv { inline void setNumThreads(int) {} }

class SwarmLoop {                                   // swarm_loop.h:25-34, 83-93: the members the block sets
public:
    bool debug_image = false;
    double min_movement_keyframe = 0.3;
    int self_id = 0;
    CameraConfig camera_configuration;
    bool enable_pub_remote_frame, enable_pub_local_frame, enable_sub_remote_frame, send_img, send_whole_img_desc;
    double max_freq = 1.0, recv_msg_duration = 0.5, superpoint_thres = 0.012;
    int superpoint_max_num = 200;
    std::map<std::string, std::string> locals;      // the block's local strings, copied out behind it
    void Init(ros::NodeHandle& nh) {
#include REF_SWARMLOOP_PARAMS
        locals = {{"lcm_uri", _lcm_uri}, {"camera_config_path", camera_config_path}, {"superpoint_model_path", superpoint_model_path}, {"netvlad_model_path", netvlad_model_path},
                  {"vins_config_path", vins_config_path}, {"pca_comp_path", _pca_comp_path}, {"pca_mean_path", _pca_mean_path}};
        (void)IMAGE0_TOPIC; (void)_camconfig;
    }
};

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "defaults";
    if (mode == "apply") {
        std::string line;
        while (std::getline(std::cin, line)) {
            const size_t a = line.find('\t'), b = line.find('\t', a + 1);
            if (a == std::string::npos || b == std::string::npos) continue;
            g_server[line.substr(0, a)] = {line[a + 1], line.substr(b + 1)};
        }
    }
    ros::NodeHandle nh;
    SwarmLoop s;
    s.Init(nh);
    if (mode == "defaults") {
        for (const auto& c : g_calls) std::cout << c.name << "\t" << c.type << "\t" << c.def << "\n";
        return 0;
    }
    // every variable of the block, under the name of the parameter that feeds it (swarm_loop.cpp:215-270)
    auto out = [](const char* n, const std::string& v) { std::cout << n << "\t" << v << "\n"; };
    out("self_id", show(s.self_id)); out("is_4dof", show(is_4dof)); out("min_movement_keyframe", show(s.min_movement_keyframe)); out("nonkeyframe_waitsec", show(ACCEPT_NONKEYFRAME_WAITSEC));
    out("lcm_uri", s.locals["lcm_uri"]); out("init_loop_min_feature_num", show(INIT_MODE_MIN_LOOP_NUM)); out("match_index_dist", show(MATCH_INDEX_DIST));
    out("min_loop_feature_num", show(MIN_LOOP_NUM)); out("min_match_per_dir", show(MIN_MATCH_PRE_DIR)); out("jpg_quality", show(JPG_QUALITY)); out("accept_min_3d_pts", show(ACCEPT_MIN_3D_PTS));
    out("inter_drone_init_frames", show(inter_drone_init_frames)); out("enable_lk", show(ENABLE_LK_LOOP_DETECTION)); out("enable_pub_remote_frame", show(s.enable_pub_remote_frame));
    out("enable_pub_local_frame", show(s.enable_pub_local_frame)); out("enable_sub_remote_frame", show(s.enable_sub_remote_frame)); out("send_img", show(s.send_img));
    out("is_pc_replay", show(IS_PC_REPLAY)); out("send_whole_img_desc", show(s.send_whole_img_desc)); out("send_all_features", show(SEND_ALL_FEATURES));
    out("query_thres", show(INNER_PRODUCT_THRES)); out("init_query_thres", show(INIT_MODE_PRODUCT_THRES)); out("max_freq", show(s.max_freq)); out("recv_msg_duration", show(s.recv_msg_duration));
    out("superpoint_thres", show(s.superpoint_thres)); out("superpoint_max_num", show(s.superpoint_max_num)); out("detector_match_thres", show(DETECTOR_MATCH_THRES));
    out("lower_cam_as_main", show(LOWER_CAM_AS_MAIN)); out("output_raw_superpoint_desc", show(OUTPUT_RAW_SUPERPOINT_DESC));
    out("odometry_consistency_threshold", show(odometry_consistency_threshold)); out("pos_covariance_per_meter", show(pos_covariance_per_meter));
    out("yaw_covariance_per_meter", show(yaw_covariance_per_meter)); out("triangle_thres", show(TRIANGLE_THRES)); out("debug_no_rejection", show(DEBUG_NO_REJECT));
    out("depth_far_thres", show(DEPTH_FAR_THRES)); out("depth_near_thres", show(DEPTH_NEAR_THRES)); out("loop_cov_pos", show(loop_cov_pos)); out("loop_cov_ang", show(loop_cov_ang));
    out("min_direction_loop", show(MIN_DIRECTION_LOOP)); out("width", show(width)); out("height", show(height)); out("camera_configuration", show((int)s.camera_configuration));
    out("vins_config_path", s.locals["vins_config_path"]); out("pca_comp_path", s.locals["pca_comp_path"]); out("pca_mean_path", s.locals["pca_mean_path"]);
    out("camera_config_path", s.locals["camera_config_path"]); out("superpoint_model_path", s.locals["superpoint_model_path"]); out("netvlad_model_path", s.locals["netvlad_model_path"]);
    out("debug_image", show(s.debug_image)); out("output_path", OUTPUT_PATH);
    return 0;
}
