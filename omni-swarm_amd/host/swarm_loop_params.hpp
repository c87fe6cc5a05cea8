// swarm_loop_params.hpp -- the parameter surface of the reference's node without ROS: every parameter SwarmLoop::Init reads
// (swarm_loop/src/swarm_loop.cpp:215-270: 50 nh.param<T>(name, variable, default) calls), with the reference's names, types and defaults, as ONE table;
// a loader for the reference's own launch files (swarm_loop/launch/*.launch: <arg>, $(arg ..), the node's inline <rosparam> block and its <param> tags, in
// roslaunch's order: later settings win); and the mapping onto this build's host objects (KeyframePipeline::Config, KeyframeIntake, LoopGeometry,
// LoopDetectorCore, LoopNetWire), so that a deployment keeps its launch files when the node's inside is swapped.
// Pinned to the reference's text: tests/cpp/params_pin.cpp compiles the nh.param block verbatim against a recording NodeHandle and compares names, types,
// defaults and -- fed the reference's four launch files as roslaunch + roscpp would deliver them (an independent Python reading: xml.etree + PyYAML, the
// library roslaunch itself uses) -- every variable the block sets (tests/test_cpp_host.py).
// Typing matters: a <rosparam> value has the type YAML 1.1 gives it, and roscpp's param<T> keeps the DEFAULT when the stored type does not convert to T.
// The reference's own files hit that: `loop_cov_pos: 1e-2` (nodelet-sfisheye.launch:45, node-sfisheye.launch:45) is a STRING in YAML 1.1 (a float needs
// a '.'), so those nodes run with the default 0.013, not 0.01 -- reproduced here (type_mismatches lists such parameters).
// Parameters the hot path has no use for (topics, LCM uri, JPEG quality, visualisation) are carried and reported, not interpreted.
#pragma once
#include <cmath>
#include <cstdlib>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace omni {

// X(type tag, C++ type, ROS parameter name, default)   -- in the order of swarm_loop.cpp:215-270
#define OMNI_SWARM_LOOP_PARAMS(X)                                                                                     \
    X(I, int, self_id, -1)                                                                                            \
    X(B, bool, is_4dof, true)                                                                                         \
    X(D, double, min_movement_keyframe, 0.3)                                                                          \
    X(D, double, nonkeyframe_waitsec, 5.0)                                                                            \
    X(S, std::string, lcm_uri, "udpm://224.0.0.251:7667?ttl=1")                                                       \
    X(I, int, init_loop_min_feature_num, 10)                                                                          \
    X(I, int, match_index_dist, 10)                                                                                   \
    X(I, int, min_loop_feature_num, 15)                                                                               \
    X(I, int, min_match_per_dir, 15)                                                                                  \
    X(I, int, jpg_quality, 50)                                                                                        \
    X(I, int, accept_min_3d_pts, 50)                                                                                  \
    X(I, int, inter_drone_init_frames, 50)                                                                            \
    X(B, bool, enable_lk, true)                                                                                       \
    X(B, bool, enable_pub_remote_frame, false)                                                                        \
    X(B, bool, enable_pub_local_frame, false)                                                                         \
    X(B, bool, enable_sub_remote_frame, false)                                                                        \
    X(B, bool, send_img, false)                                                                                       \
    X(B, bool, is_pc_replay, false)                                                                                   \
    X(B, bool, send_whole_img_desc, false)                                                                            \
    X(B, bool, send_all_features, false)                                                                              \
    X(D, double, query_thres, 0.6)                                                                                    \
    X(D, double, init_query_thres, 0.3)                                                                               \
    X(D, double, max_freq, 1.0)                                                                                       \
    X(D, double, recv_msg_duration, 0.5)                                                                              \
    X(D, double, superpoint_thres, 0.012)                                                                             \
    X(I, int, superpoint_max_num, 200)                                                                                \
    X(D, double, detector_match_thres, 0.9)                                                                           \
    X(B, bool, lower_cam_as_main, false)                                                                              \
    X(B, bool, output_raw_superpoint_desc, false)                                                                     \
    X(D, double, odometry_consistency_threshold, 2.0)                                                                 \
    X(D, double, pos_covariance_per_meter, 0.01)                                                                      \
    X(D, double, yaw_covariance_per_meter, 0.003)                                                                     \
    X(D, double, triangle_thres, 0.006)                                                                               \
    X(B, bool, debug_no_rejection, false)                                                                             \
    X(D, double, depth_far_thres, 10.0)                                                                               \
    X(D, double, depth_near_thres, 0.3)                                                                               \
    X(D, double, loop_cov_pos, 0.013)                                                                                 \
    X(D, double, loop_cov_ang, 2.5e-04)                                                                               \
    X(I, int, min_direction_loop, 3)                                                                                  \
    X(I, int, width, 400)                                                                                             \
    X(I, int, height, 208)                                                                                            \
    X(I, int, camera_configuration, 1)                                                                                \
    X(S, std::string, vins_config_path, "")                                                                           \
    X(S, std::string, pca_comp_path, "")                                                                              \
    X(S, std::string, pca_mean_path, "")                                                                              \
    X(S, std::string, camera_config_path, "/home/xuhao/swarm_ws/src/VINS-Fusion-gpu/config/vi_car/cam0_mei.yaml")    \
    X(S, std::string, superpoint_model_path, "")                                                                      \
    X(S, std::string, netvlad_model_path, "")                                                                         \
    X(B, bool, debug_image, false)                                                                                    \
    X(S, std::string, output_path, "")

struct SwarmLoopParams {
#define X(tag, type, name, def) type name = def;
    OMNI_SWARM_LOOP_PARAMS(X)
#undef X

    struct Field { const char* name; char type; /* 'I' int, 'B' bool, 'D' double, 'S' string */ };
    static const std::vector<Field>& fields() {
        static const std::vector<Field> f = {
#define X(tag, type, name, def) {#name, #tag[0]},
            OMNI_SWARM_LOOP_PARAMS(X)
#undef X
        };
        return f;
    }

    // a value as the ROS parameter server holds it: XmlRpc int / double / boolean / string ('N' = YAML null: roslaunch does not set it)
    struct Value { char type = 'S'; long i = 0; double d = 0; bool b = false; std::string s; };
    // YAML 1.1 plain-scalar resolution as PyYAML's SafeLoader does it (what roslaunch runs on a <rosparam> block)
    static Value yaml_scalar(const std::string& raw) {
        Value v;
        std::string t = raw;
        if (t.size() >= 2 && ((t.front() == '"' && t.back() == '"') || (t.front() == '\'' && t.back() == '\''))) { v.type = 'S'; v.s = t.substr(1, t.size() - 2); return v; }
        v.s = t;
        if (t.empty() || t == "~" || t == "null" || t == "Null" || t == "NULL") { v.type = 'N'; return v; }
        for (const char* w : {"yes", "Yes", "YES", "true", "True", "TRUE", "on", "On", "ON"}) if (t == w) { v.type = 'B'; v.b = true; return v; }
        for (const char* w : {"no", "No", "NO", "false", "False", "FALSE", "off", "Off", "OFF"}) if (t == w) { v.type = 'B'; v.b = false; return v; }
        size_t k = 0;
        const bool neg = t[0] == '-';
        if (t[0] == '-' || t[0] == '+') k = 1;
        const std::string body = t.substr(k);
        auto strip_ = [](std::string x) { std::string o; for (char c : x) if (c != '_') o += c; return o; };
        auto all_of = [](const std::string& x, const char* set) { return !x.empty() && x.find_first_not_of(set) == std::string::npos; };
        // int: 0b..., 0x..., 0[0-7_]+, 0 | [1-9][0-9_]*, sexagesimal [1-9][0-9_]*(:[0-5]?[0-9])+
        if (body.size() > 2 && body[0] == '0' && body[1] == 'b' && all_of(body.substr(2), "01_")) { v.type = 'I'; v.i = std::strtol(strip_(body.substr(2)).c_str(), nullptr, 2); if (neg) v.i = -v.i; return v; }
        if (body.size() > 2 && body[0] == '0' && body[1] == 'x' && all_of(body.substr(2), "0123456789abcdefABCDEF_")) { v.type = 'I'; v.i = std::strtol(strip_(body.substr(2)).c_str(), nullptr, 16); if (neg) v.i = -v.i; return v; }
        if (body == "0" || (body.size() > 1 && body[0] == '0' && all_of(body.substr(1), "01234567_"))) { v.type = 'I'; v.i = std::strtol(strip_(body).c_str(), nullptr, 8); if (neg) v.i = -v.i; return v; }
        if (body[0] >= '1' && body[0] <= '9' && all_of(body, "0123456789_")) { v.type = 'I'; v.i = std::strtol(strip_(body).c_str(), nullptr, 10); if (neg) v.i = -v.i; return v; }
        if (body[0] >= '1' && body[0] <= '9' && all_of(body, "0123456789_:") && body.find(':') != std::string::npos && body.back() != ':') {
            long acc = 0; size_t a = 0; bool ok = true;
            while (a <= body.size()) {
                size_t b = body.find(':', a); if (b == std::string::npos) b = body.size();
                const std::string part = strip_(body.substr(a, b - a));
                if (part.empty() || (a > 0 && (part.size() > 2 || std::atoi(part.c_str()) > 59))) { ok = false; break; }
                acc = acc * 60 + std::atol(part.c_str());
                a = b + 1;
            }
            if (ok) { v.type = 'I'; v.i = neg ? -acc : acc; return v; }
        }
        // float: [0-9][0-9_]*\.[0-9_]*([eE][-+][0-9]+)? | \.[0-9_]+([eE][-+][0-9]+)? | .inf | .nan   (no '.', no float: `1e-2` is a string)
        if (body == ".inf" || body == ".Inf" || body == ".INF") { v.type = 'D'; v.d = neg ? -HUGE_VAL : HUGE_VAL; return v; }
        if (t == ".nan" || t == ".NaN" || t == ".NAN") { v.type = 'D'; v.d = std::nan(""); return v; }
        {
            size_t e = body.find_first_of("eE");
            const std::string mant = body.substr(0, e), ex = e == std::string::npos ? std::string() : body.substr(e + 1);
            const size_t dot = mant.find('.');
            bool ok = dot != std::string::npos && mant.find('.', dot + 1) == std::string::npos;
            if (ok) {
                const std::string ip = mant.substr(0, dot), fp = mant.substr(dot + 1);
                if (ip.empty()) ok = all_of(fp, "0123456789_");                                   // \.[0-9_]+
                else ok = isdigit((unsigned char)ip[0]) && all_of(ip, "0123456789_") && (fp.empty() || all_of(fp, "0123456789_"));
            }
            if (ok && e != std::string::npos) ok = ex.size() >= 2 && (ex[0] == '-' || ex[0] == '+') && all_of(ex.substr(1), "0123456789");
            if (ok) { v.type = 'D'; v.d = std::strtod(strip_(t).c_str(), nullptr); return v; }
        }
        v.type = 'S';
        return v;
    }
    // <param name value type>: roslaunch's convert_value
    static Value launch_value(const std::string& name, const std::string& value, const std::string& type) {
        Value v; v.s = value;
        std::string lo = value; for (char& c : lo) c = (char)tolower((unsigned char)c);
        if (type == "str" || type == "string") { v.type = 'S'; return v; }
        if (type == "int") { v.type = 'I'; v.i = to_int(name, value); return v; }
        if (type == "double") { v.type = 'D'; v.d = to_double(name, value); return v; }
        if (type == "bool" || type == "boolean") { v.type = 'B'; v.b = to_bool(name, lo); return v; }
        if (type == "auto" || type.empty()) {
            char* end = nullptr;
            if (value.find('.') != std::string::npos) { const double d = std::strtod(value.c_str(), &end); if (end != value.c_str() && *end == '\0') { v.type = 'D'; v.d = d; return v; } }
            else { const long i = std::strtol(value.c_str(), &end, 10); if (end != value.c_str() && *end == '\0') { v.type = 'I'; v.i = i; return v; } }
            if (lo == "true" || lo == "false") { v.type = 'B'; v.b = lo == "true"; return v; }
            v.type = 'S';
            return v;
        }
        throw std::invalid_argument("launch file: <param name=\"" + name + "\"> has the unsupported type " + type);
    }
    static bool to_bool(const std::string& name, const std::string& v) {
        if (v == "true" || v == "True" || v == "TRUE" || v == "1") return true;
        if (v == "false" || v == "False" || v == "FALSE" || v == "0") return false;
        throw std::invalid_argument("parameter " + name + ": '" + v + "' is not a bool");
    }
    static double to_double(const std::string& name, const std::string& v) {
        char* end = nullptr;
        const double d = std::strtod(v.c_str(), &end);
        if (end == v.c_str() || *end != '\0') throw std::invalid_argument("parameter " + name + ": '" + v + "' is not a number");
        return d;
    }
    static int to_int(const std::string& name, const std::string& v) {
        char* end = nullptr;
        const long d = std::strtol(v.c_str(), &end, 10);
        if (end == v.c_str() || *end != '\0') throw std::invalid_argument("parameter " + name + ": '" + v + "' is not an int");
        return (int)d;
    }
    // `name` as text (a command-line override, a test): parsed as the parameter's own type; false when the node has no such parameter
    bool set(const std::string& pname, const std::string& v) {
#define X(tag, type, name, def) if (pname == #name) { assign(name, #name, v); return true; }
        OMNI_SWARM_LOOP_PARAMS(X)
#undef X
        return false;
    }
    // `name` as the parameter server holds it, through roscpp's param<T> conversions (ros::param::getParamImpl): double <- int | double; int <- int | double
    // (rounded half up); bool <- boolean only; string <- string only.  0 = the node has no such parameter, 1 = assigned, 2 = stored type does not convert:
    // the DEFAULT stays (what nh.param<T> does)
    int set_typed(const std::string& pname, const Value& v) {
#define X(tag, type, name, def) if (pname == #name) return conv(name, v) ? 1 : 2;
        OMNI_SWARM_LOOP_PARAMS(X)
#undef X
        return 0;
    }
    std::string get(const std::string& pname) const {
#define X(tag, type, name, def) if (pname == #name) return str(name);
        OMNI_SWARM_LOOP_PARAMS(X)
#undef X
        throw std::invalid_argument("no parameter named " + pname);
    }
    // ---- onto this build's host objects (templates: this header stays free of the HIP-facing ones) -----------------------------------------------
    // KeyframePipeline::Config: what LoopCam's and LoopDetector's constructors receive (swarm_loop.cpp:310-317) + the thresholds the pipeline hands on
    template <class Config> void to_pipeline_config(Config& c) const {
        c.width = width; c.height = height; c.thres = (float)superpoint_thres; c.max_num = superpoint_max_num; c.self_id = self_id;
        c.inner_product_thres = query_thres; c.init_mode_product_thres = init_query_thres; c.match_index_dist = match_index_dist;
        c.min_loop_num = min_loop_feature_num; c.min_direction_loop = min_direction_loop; c.triangle_thres = triangle_thres; c.accept_min_3d_pts = accept_min_3d_pts;
        c.camera_configuration = camera_configuration; c.depth_near = depth_near_thres; c.depth_far = depth_far_thres;
        c.pca_comp = pca_comp_path; c.pca_mean = pca_mean_path;
    }
    template <class Intake> void to_intake(Intake& i) const {              // swarm_loop.cpp:217-218, 238 -> KeyframeIntake
        i.max_freq = max_freq; i.min_movement_keyframe = min_movement_keyframe; i.accept_nonkeyframe_waitsec = nonkeyframe_waitsec;
    }
    template <class Detector> void to_detector(Detector& d) const {        // loop_params.cpp's globals as LoopDetectorCore's members
        d.INNER_PRODUCT_THRES = query_thres; d.INIT_MODE_PRODUCT_THRES = init_query_thres; d.MATCH_INDEX_DIST = match_index_dist; d.MIN_LOOP_NUM = min_loop_feature_num;
        d.MIN_DIRECTION_LOOP = min_direction_loop; d.inter_drone_init_frames = inter_drone_init_frames; d.stereo_fisheye = camera_configuration == 1;
    }
    template <class Geometry> void to_geometry(Geometry& g) const {        // ... as LoopGeometry's
        g.MIN_LOOP_NUM = min_loop_feature_num; g.INIT_MODE_MIN_LOOP_NUM = init_loop_min_feature_num; g.MIN_MATCH_PRE_DIR = min_match_per_dir; g.MIN_DIRECTION_LOOP = min_direction_loop;
        g.MAX_DIRS = camera_configuration == 1 ? 4 : 1;                     // swarm_loop.cpp:275-281
        g.is_4dof = is_4dof; g.debug_no_reject = debug_no_rejection; g.loop_cov_pos = loop_cov_pos; g.loop_cov_ang = loop_cov_ang;
        g.odometry_consistency_threshold = odometry_consistency_threshold; g.self_id = self_id;
    }
    template <class Wire> void to_wire(Wire& w) const {                    // LoopNet(_lcm_uri, send_img, send_whole_img_desc, recv_msg_duration), swarm_loop.cpp:309
        w.recv_period = recv_msg_duration; w.MIN_DIRECTION_LOOP = min_direction_loop; w.SEND_ALL_FEATURES = send_all_features;
    }

    std::vector<std::string> type_mismatches;      // parameters a launch file set with a type param<T> refuses: they kept their defaults
    std::vector<std::string> unknown_parameters;   // names a launch file set that the node never reads (the reference's files: enable_pub_remote_img, ...)

    // ---- the reference's launch files ----------------------------------------------------------------------------------------------------------------
    // roslaunch as far as swarm_loop/launch/*.launch use it: <arg name default|value>, $(arg x) and $(find pkg) substitution, if= / unless= on nodes, the
    // node's <rosparam> block (flat `key: value` YAML with # comments) and <param name value>; settings are applied in document order.  node_name: the
    // <node name=...> whose private parameters are read ("swarm_loop" in all four files); args: command-line overrides (`roslaunch f.launch self_id:=2`).
    static SwarmLoopParams from_launch(const std::string& xml, const std::string& node_name = "swarm_loop", const std::map<std::string, std::string>& args = {},
                                       const std::function<std::string(const std::string&)>& find_pkg = nullptr) {
        SwarmLoopParams p;
        std::map<std::string, std::string> argv = args;
        std::map<std::string, std::string> declared;
        bool found = false;
        size_t pos = 0;
        auto subst = [&](std::string s) {
            for (;;) {
                const size_t a = s.find("$(");
                if (a == std::string::npos) return s;
                const size_t b = s.find(')', a);
                if (b == std::string::npos) throw std::invalid_argument("launch file: unterminated $( in '" + s + "'");
                const std::string inner = s.substr(a + 2, b - a - 2);
                const size_t sp = inner.find(' ');
                const std::string verb = inner.substr(0, sp), what = sp == std::string::npos ? "" : trim(inner.substr(sp + 1));
                std::string rep;
                if (verb == "arg") {
                    auto it = declared.find(what);
                    if (it == declared.end()) throw std::invalid_argument("launch file: $(arg " + what + ") before its <arg>");
                    rep = it->second;
                } else if (verb == "find") {
                    rep = find_pkg ? find_pkg(what) : "$(find " + what + ")";
                    if (!find_pkg) { s = s.substr(0, a) + "\x01" + s.substr(a + 1); continue; }      // kept verbatim: hidden from the loop, restored below
                } else throw std::invalid_argument("launch file: $(" + verb + " ...) is not supported");
                s = s.substr(0, a) + rep + s.substr(b + 1);
            }
        };
        auto restore = [](std::string s) { for (char& c : s) if (c == '\x01') c = '$'; return s; };
        while (true) {
            Tag t;
            if (!next_tag(xml, pos, t)) break;
            if (t.name == "arg") {
                const std::string name = t.attr("name");
                std::string v;
                if (t.has("value")) v = restore(subst(t.attr("value")));
                else if (argv.count(name)) v = argv[name];
                else if (t.has("default")) v = restore(subst(t.attr("default")));
                else throw std::invalid_argument("launch file: <arg name=\"" + name + "\"> has no value");
                declared[name] = v;
            } else if (t.name == "node" && !t.closing && t.attr("name") == node_name) {
                if (t.has("if") && !to_bool("if", restore(subst(t.attr("if"))))) { skip_element(xml, pos, t); continue; }
                if (t.has("unless") && to_bool("unless", restore(subst(t.attr("unless"))))) { skip_element(xml, pos, t); continue; }
                found = true;
                if (t.self_closing) continue;
                // the node's children, in order
                while (true) {
                    Tag c;
                    if (!next_tag(xml, pos, c)) throw std::invalid_argument("launch file: <node> is not closed");
                    if (c.closing && c.name == "node") break;
                    if (c.name == "param") {
                        const std::string n = c.attr("name");
                        p.note(n, p.set_typed(n, launch_value(n, restore(subst(c.attr("value"))), c.has("type") ? c.attr("type") : std::string("auto"))));
                    } else if (c.name == "rosparam" && !c.self_closing) {
                        const size_t end = xml.find("</rosparam>", pos);
                        if (end == std::string::npos) throw std::invalid_argument("launch file: <rosparam> is not closed");
                        apply_yaml(p, restore(subst(xml.substr(pos, end - pos))));
                        pos = end + 11;
                    }
                }
            }
        }
        if (!found) throw std::invalid_argument("launch file: no <node name=\"" + node_name + "\">");
        return p;
    }

private:
    static std::string str(int v) { return std::to_string(v); }
    static std::string str(bool v) { return v ? "true" : "false"; }
    static std::string str(double v) { char b[64]; snprintf(b, sizeof(b), "%.17g", v); return b; }
    static std::string str(const std::string& v) { return v; }
    static void assign(int& dst, const char* n, const std::string& v) { dst = to_int(n, v); }
    static void assign(bool& dst, const char* n, const std::string& v) { dst = to_bool(n, v); }
    static void assign(double& dst, const char* n, const std::string& v) { dst = to_double(n, v); }
    static void assign(std::string& dst, const char*, const std::string& v) { dst = v; }
    static std::string trim(const std::string& s) {
        const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    }
    struct Tag {
        std::string name;
        std::vector<std::pair<std::string, std::string>> attrs;
        bool closing = false, self_closing = false;
        bool has(const std::string& k) const { for (auto& a : attrs) if (a.first == k) return true; return false; }
        std::string attr(const std::string& k) const {
            for (auto& a : attrs) if (a.first == k) return a.second;
            throw std::invalid_argument("launch file: <" + name + "> has no attribute " + k);
        }
    };
    // the next tag at or after pos (comments, the XML declaration and text are skipped); pos moves behind it
    static bool next_tag(const std::string& x, size_t& pos, Tag& t) {
        for (;;) {
            const size_t a = x.find('<', pos);
            if (a == std::string::npos) return false;
            if (x.compare(a, 4, "<!--") == 0) { const size_t e = x.find("-->", a); if (e == std::string::npos) return false; pos = e + 3; continue; }
            if (x.compare(a, 2, "<?") == 0) { const size_t e = x.find("?>", a); if (e == std::string::npos) return false; pos = e + 2; continue; }
            size_t i = a + 1;
            t = Tag();
            if (i < x.size() && x[i] == '/') { t.closing = true; ++i; }
            while (i < x.size() && (isalnum((unsigned char)x[i]) || x[i] == '_' || x[i] == '-' || x[i] == ':')) t.name += x[i++];
            for (;;) {
                while (i < x.size() && isspace((unsigned char)x[i])) ++i;
                if (i >= x.size()) return false;
                if (x[i] == '>') { pos = i + 1; return true; }
                if (x[i] == '/' && i + 1 < x.size() && x[i + 1] == '>') { t.self_closing = true; pos = i + 2; return true; }
                std::string k;
                while (i < x.size() && x[i] != '=' && !isspace((unsigned char)x[i]) && x[i] != '>') k += x[i++];
                while (i < x.size() && isspace((unsigned char)x[i])) ++i;
                if (i >= x.size() || x[i] != '=') throw std::invalid_argument("launch file: attribute " + k + " of <" + t.name + "> has no value");
                ++i;
                while (i < x.size() && isspace((unsigned char)x[i])) ++i;
                if (i >= x.size() || (x[i] != '"' && x[i] != '\'')) throw std::invalid_argument("launch file: attribute " + k + " is not quoted");
                const char q = x[i++];
                std::string v;
                while (i < x.size() && x[i] != q) v += x[i++];
                ++i;
                t.attrs.emplace_back(k, v);
            }
        }
    }
    static void skip_element(const std::string& x, size_t& pos, const Tag& open) {
        if (open.self_closing) return;
        int depth = 1;
        Tag t;
        while (depth > 0 && next_tag(x, pos, t))
            if (t.name == open.name) depth += t.closing ? -1 : (t.self_closing ? 0 : 1);
    }
    // the <rosparam> blocks of the reference's launch files: one `key: value` per line, `#` comments (also trailing), blank lines
    static void apply_yaml(SwarmLoopParams& p, const std::string& text) {
        size_t a = 0;
        while (a <= text.size()) {
            size_t b = text.find('\n', a);
            if (b == std::string::npos) b = text.size();
            std::string line = text.substr(a, b - a);
            a = b + 1;
            const size_t h = line.find('#');
            if (h != std::string::npos && (h == 0 || isspace((unsigned char)line[h - 1]))) line = line.substr(0, h);
            line = trim(line);
            if (line.empty()) continue;
            const size_t c = line.find(": ");
            const size_t c2 = (c == std::string::npos && line.back() == ':') ? line.size() - 1 : c;
            if (c2 == std::string::npos) throw std::invalid_argument("launch file: <rosparam> line '" + line + "' is not `key: value`");
            const std::string key = trim(line.substr(0, c2)), val = c2 + 1 < line.size() ? trim(line.substr(c2 + 1)) : std::string();
            const Value v = yaml_scalar(val);
            if (v.type != 'N') p.note(key, p.set_typed(key, v));
        }
    }
    void note(const std::string& name, int rc) {
        auto drop = [&](std::vector<std::string>& l) { for (size_t i = 0; i < l.size(); ++i) if (l[i] == name) { l.erase(l.begin() + i); break; } };
        drop(type_mismatches); drop(unknown_parameters);       // a later setting of the same name decides
        if (rc == 0) unknown_parameters.push_back(name);
        if (rc == 2) type_mismatches.push_back(name);
    }
    static bool conv(int& dst, const Value& v) {
        if (v.type == 'I') { dst = (int)v.i; return true; }
        if (v.type == 'D') { double d = v.d; d = std::fmod(d, 1.0) < 0.5 ? std::floor(d) : std::ceil(d); dst = (int)d; return true; }
        return false;
    }
    static bool conv(double& dst, const Value& v) {
        if (v.type == 'I') { dst = (double)(int)v.i; return true; }
        if (v.type == 'D') { dst = v.d; return true; }
        return false;
    }
    static bool conv(bool& dst, const Value& v) { if (v.type != 'B') return false; dst = v.b; return true; }
    static bool conv(std::string& dst, const Value& v) { if (v.type != 'S') return false; dst = v.s; return true; }
};

}  // namespace omni
