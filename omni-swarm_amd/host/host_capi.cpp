// host_capi.cpp -> lib/libomni_host.so: C entry points over omni::KeyframePipeline (keyframe_pipeline.hpp) so that a launcher written
// in any language (bench.py via ctypes; a ROS nodelet via plain linking) can run the C++ key-frame host loop in-process.
// Plain g++ -- no HIP headers: everything below the adapters goes through the C ABI of libomni_hip.so (include/omni_hip.h).
#include <cstring>
#include <string>

#include "fisheye_flatten.hpp"
#include "keyframe_pipeline.hpp"
#include "loop_net_wire.hpp"
#include "swarm_loop_params.hpp"
#include "omni_host.h"      // include/: the declarations of everything below (a mismatch is a compile error)

namespace {
thread_local std::string g_err;
}

extern "C" {

struct omni_pipeline { omni::KeyframePipeline* p; };

const char* omni_pipeline_last_error(void) { return g_err.c_str(); }

// weights: OMNW1 files (tools/export_weights.py), PCA: the reference's two CSV files (superpoint_tensorrt.cpp:110-111)
omni_pipeline* omni_pipeline_create(int device, const char* sp_weights, const char* pca_comp_csv, const char* pca_mean_csv, const char* vlad_weights,
                                    int width, int height, float thres, int max_num, int precision, int microbatch, int pipelines, int storage,
                                    int self_id, double inner_product_thres, double init_mode_product_thres, int match_index_dist, int min_loop_num,
                                    int min_direction_loop, int geometry) {
    try {
        omni::KeyframePipeline::Config c;
        c.device = device; c.sp_weights = sp_weights; c.pca_comp = pca_comp_csv ? pca_comp_csv : ""; c.pca_mean = pca_mean_csv ? pca_mean_csv : "";
        c.vlad_weights = vlad_weights; c.width = width; c.height = height; c.thres = thres; c.max_num = max_num; c.precision = precision;
        c.microbatch = microbatch; c.pipelines = pipelines; c.storage = storage; c.self_id = self_id;
        c.inner_product_thres = inner_product_thres; c.init_mode_product_thres = init_mode_product_thres; c.match_index_dist = match_index_dist;
        c.min_loop_num = min_loop_num; c.min_direction_loop = min_direction_loop; c.geometry = geometry != 0;
        c.cx = width / 2.0; c.cy = height / 2.0; c.fx = c.fy = width / 2.0;      // 90 degree horizontal field of view
        return new omni_pipeline{new omni::KeyframePipeline(c)};
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// CameraConfig::PINHOLE_DEPTH (launch/realsense.launch; BASELINE.json configs[0]): one gray image + one depth image per key frame, pinhole model
// fx fy cx cy, landmarks where depth_near < depth < depth_far (metres).  The other arguments as omni_pipeline_create; micro-batch s of run() is
// [microbatch][height][width] u8; the depth images come in through omni_pipeline_set_depth.
omni_pipeline* omni_pipeline_create_pinhole_depth(int device, const char* sp_weights, const char* pca_comp_csv, const char* pca_mean_csv, const char* vlad_weights,
                                                  int width, int height, float thres, int max_num, int precision, int microbatch, int pipelines, int storage,
                                                  int self_id, double inner_product_thres, double init_mode_product_thres, int match_index_dist, int min_loop_num,
                                                  int min_direction_loop, int geometry, double fx, double fy, double cx, double cy, double depth_near,
                                                  double depth_far, int accept_min_3d_pts) {
    try {
        omni::KeyframePipeline::Config c;
        c.device = device; c.sp_weights = sp_weights; c.pca_comp = pca_comp_csv ? pca_comp_csv : ""; c.pca_mean = pca_mean_csv ? pca_mean_csv : "";
        c.vlad_weights = vlad_weights; c.width = width; c.height = height; c.thres = thres; c.max_num = max_num; c.precision = precision;
        c.microbatch = microbatch; c.pipelines = pipelines; c.storage = storage; c.self_id = self_id;
        c.inner_product_thres = inner_product_thres; c.init_mode_product_thres = init_mode_product_thres; c.match_index_dist = match_index_dist;
        c.min_loop_num = min_loop_num; c.min_direction_loop = min_direction_loop; c.geometry = geometry != 0;
        c.camera_configuration = 2; c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy; c.depth_near = depth_near; c.depth_far = depth_far; c.accept_min_3d_pts = accept_min_3d_pts;
        return new omni_pipeline{new omni::KeyframePipeline(c)};
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// depth images (u16 millimetres, [n][height][width]) of key frames first_msg_id .. first_msg_id + n - 1; BORROWED until the run that uses them returns
int omni_pipeline_set_depth(omni_pipeline* h, int64_t first_msg_id, int64_t n, const uint16_t* depth) {
    try { h->p->set_depth(first_msg_id, depth, n); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

void omni_pipeline_destroy(omni_pipeline* h) {
    if (!h) return;
    delete h->p;
    delete h;
}

int omni_pipeline_preload(omni_pipeline* h, const float* rows, int64_t n) {
    try { h->p->preload(rows, n); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

int64_t omni_pipeline_db_rows(omni_pipeline* h) { return h->p->db_rows(); }

// collective over all ranks (one process per GPU): switch the database to the row-sharded index (omni_shard_*, RCCL inside libomni_hip.so);
// unique_id = omni_shard_unique_id() of rank 0.  Call before preload / run; preload then takes THIS rank's rows.
int omni_pipeline_attach_shard(omni_pipeline* h, int rank, int world, const char* unique_id) {
    try { h->p->attach_shard(rank, world, unique_id); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// see omni::KeyframePipeline::run; *hits = loop candidates found.  Returns 0 on success.
int omni_pipeline_run(omni_pipeline* h, int n_keyframes, int64_t first_msg_id, const uint8_t* const* pool, int n_pool, int first_slot,
                      const uint8_t* tail, int from_host, int* hits) {
    try {
        const int n = h->p->run(n_keyframes, first_msg_id, pool, n_pool, first_slot, tail, from_host != 0);
        if (hits) *hits = n;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// the streaming intake (omni::KeyframePipeline::push_keyframe / flush): one key frame as SwarmLoop::VIOKF_callback hands it on -- images[0..n_images) host
// pointers (up cameras, then down cameras; PINHOLE_DEPTH: the one gray image), pose7 = position xyz + quaternion wxyz, depth (PINHOLE_DEPTH, may be
// NULL) borrowed until the key frame's unit is finished.  *hits += loop candidates found by the units the call finished.
int omni_pipeline_push_keyframe(omni_pipeline* h, const uint8_t* const* images, int stride, int64_t msg_id, double stamp, const double* pose7,
                                int prevent_adding_db, const uint16_t* depth, int* hits) {
    try {
        omni::KeyframePipeline::KeyframeIn k;
        k.images = images; k.stride = stride; k.msg_id = msg_id; k.stamp = stamp; k.prevent_adding_db = prevent_adding_db != 0; k.depth = depth;
        if (pose7) { for (int i = 0; i < 3; ++i) k.pose_drone.position[i] = pose7[i]; for (int i = 0; i < 4; ++i) k.pose_drone.quat_wxyz[i] = pose7[3 + i]; }
        const int n = h->p->push_keyframe(k);
        if (hits) *hits += n;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// the latency bound of the streaming intake (omni::KeyframePipeline::poll): call it from a timer or after every push; never waits for a CNN unit.
// *hits += loop candidates found.
int omni_pipeline_poll(omni_pipeline* h, int* hits) {
    try { const int n = h->p->poll(); if (hits) *hits += n; return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// max_wait_ms: a partly filled micro-batch older than this leaves for the GPU at the next push / poll (< 0: only flush() sends it; default 50);
// dispatch_when_idle: a key frame that arrives while no unit is in flight goes at once, as a unit of one (default on)
int omni_pipeline_set_latency(omni_pipeline* h, double max_wait_ms, int dispatch_when_idle) {
    try { h->p->set_latency(max_wait_ms, dispatch_when_idle != 0); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// units in flight (the `pipelines` argument, or the library's default for the precision when that was <= 0) and how the last run() ordered them
// (0: the units' kernels take turns; 1 / 2: oldest first, omni_cam_order_after)
int omni_pipeline_units(omni_pipeline* h, int* units_oldest_first) {
    if (units_oldest_first) *units_oldest_first = h->p->last_run_fifo();
    return h->p->pipelines();
}
// sharded mode: device microseconds of the two all-gathers of every exchange unit since the last reset, out[i][2] = {new rows, per-shard top-k lists};
// returns how many exist (writes <= max)
int omni_pipeline_get_exchange_us(omni_pipeline* h, float* out, int max, int reset) {
    const auto& v = h->p->exchange_us();
    const int n = (int)v.size();
    for (int i = 0; i < n && i < max; ++i) { out[2 * i] = v[(size_t)i].first; out[2 * i + 1] = v[(size_t)i].second; }
    if (reset) h->p->clear_exchange_us();
    return n;
}
int omni_pipeline_flush(omni_pipeline* h, int* hits) {
    try { const int n = h->p->flush(); if (hits) *hits += n; return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// allocates whatever a later omni_pipeline_run(h, n_keyframes, ...) would allocate on first use (the unit for a partial micro-batch)
int omni_pipeline_prepare(omni_pipeline* h, int n_keyframes) {
    try { h->p->prepare(n_keyframes); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// must be called before the first run: switches the geometric verification stage on (host/loop_geometry.hpp) with a pinhole model for the
// flattened views.  omni_pipeline_geometry_stats: candidates handed to compute_loop / loop edges accepted so far.
int omni_pipeline_geometry_stats(omni_pipeline* h, int* compute_loop_calls, int* edges) {
    if (compute_loop_calls) *compute_loop_calls = h->p->geometry_calls();
    if (edges) *edges = (int)h->p->edges().size();
    return 0;
}

// odometry poses of the key frames msg_id = first_msg_id .. first_msg_id + n - 1: poses7[i] = position xyz + quaternion wxyz
int omni_pipeline_set_poses(omni_pipeline* h, int64_t first_msg_id, int64_t n, const double* poses7) {
    try { h->p->set_poses(first_msg_id, poses7, n); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// loop candidates so far, [i][4] = {new key frame, old key frame, direction_new, direction_old}; returns how many exist (writes <= max)
int omni_pipeline_get_candidates(omni_pipeline* h, int64_t* out, int max) {
    const auto& c = h->p->candidates();
    for (int i = 0; i < (int)c.size() && i < max; ++i) { out[4 * i] = c[i].new_msg_id; out[4 * i + 1] = c[i].old_msg_id; out[4 * i + 2] = c[i].dir_new; out[4 * i + 3] = c[i].dir_old; }
    return (int)c.size();
}

// accepted loop edges so far (swarm_msgs::LoopEdge as compute_loop fills it, loop_detector.cpp:789-811), [i][12] = {keyframe_id_a, keyframe_id_b,
// drone_id_a, drone_id_b, pnp_inlier_num, relative position xyz, relative attitude quaternion wxyz}; returns how many exist (writes <= max)
int omni_pipeline_get_edges(omni_pipeline* h, double* out, int max) {
    const auto& e = h->p->edges();
    for (int i = 0; i < (int)e.size() && i < max; ++i) {
        double* o = out + 12 * i;
        o[0] = (double)e[i].keyframe_id_a; o[1] = (double)e[i].keyframe_id_b; o[2] = e[i].drone_id_a; o[3] = e[i].drone_id_b; o[4] = e[i].pnp_inlier_num;
        o[5] = e[i].relative_pose.pos.x; o[6] = e[i].relative_pose.pos.y; o[7] = e[i].relative_pose.pos.z;
        o[8] = e[i].relative_pose.att.w; o[9] = e[i].relative_pose.att.x; o[10] = e[i].relative_pose.att.y; o[11] = e[i].relative_pose.att.z;
    }
    return (int)e.size();
}

// per-micro-batch latencies (ms, upload start -> detector / geometry done) recorded since the last call with reset != 0; returns how many exist
int omni_pipeline_get_latencies(omni_pipeline* h, double* out, int max, int reset) {
    const auto& l = h->p->latencies_ms();
    const int n = (int)l.size();
    for (int i = 0; i < n && i < max; ++i) out[i] = l[i];
    if (reset) h->p->clear_latencies();
    return n;
}

// FisheyeUndist's undistortion maps (host/fisheye_flatten.hpp).  mei = {xi, k1, k2, p1, p2, gamma1, gamma2, u0, v0}.  Call with maps == NULL to get
// n_views / view_w / view_h (arrays of >= 5), then with maps[v] pointing at view_w[v] * view_h[v] * 2 floats each.
int omni_fisheye_maps(const double* mei, int img_width, double fov_deg, int cam_id, int* n_views, int* view_w, int* view_h, float* const* maps) {
    try {
        omni::MeiCamera c;
        c.xi = mei[0]; c.k1 = mei[1]; c.k2 = mei[2]; c.p1 = mei[3]; c.p2 = mei[4]; c.gamma1 = mei[5]; c.gamma2 = mei[6]; c.u0 = mei[7]; c.v0 = mei[8];
        if (!maps) {          // sizes only: no map is generated
            omni::FlattenMaps m;
            double side_fov = (fov_deg - 180) * M_PI / 180.0;
            if (side_fov < 0) side_fov = 0;
            const int sh = (int)(2 * (img_width / 2.0) * std::tan(side_fov / 2));
            *n_views = sh > 0 ? 5 : 1;
            view_w[0] = view_h[0] = img_width;
            for (int v = 1; v < *n_views; ++v) { view_w[v] = img_width; view_h[v] = sh; }
            return 0;
        }
        const omni::FlattenMaps m = omni::generate_all_undist_maps(c, (unsigned)img_width, fov_deg, cam_id);
        *n_views = (int)m.w.size();
        for (size_t v = 0; v < m.w.size(); ++v) { view_w[v] = m.w[v]; view_h[v] = m.h[v]; std::memcpy(maps[v], m.xy[v].data(), m.xy[v].size() * 4); }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// The reference's launch files without ROS (host/swarm_loop_params.hpp): launch_xml = the text of a swarm_loop/launch/*.launch file, node_name = the node
// whose private parameters are read (NULL: "swarm_loop"), args = "name:=value" overrides separated by newlines (NULL: none).  out receives one line per
// parameter of SwarmLoop::Init (swarm_loop.cpp:215-270), "name=value", then "#mismatch name" for every parameter the file set with a type nh.param<T> refuses
// (its default stays) and "#unknown name" for every name the node never reads.  Returns 0, 1 on error (omni_pipeline_last_error), 2 when out is too small.
int omni_swarm_params_from_launch(const char* launch_xml, const char* node_name, const char* args, char* out, int cap) {
    try {
        std::map<std::string, std::string> av;
        if (args) {
            std::string a(args);
            size_t i = 0;
            while (i < a.size()) {
                size_t j = a.find('\n', i);
                if (j == std::string::npos) j = a.size();
                const std::string item = a.substr(i, j - i);
                const size_t k = item.find(":=");
                if (k != std::string::npos) av[item.substr(0, k)] = item.substr(k + 2);
                i = j + 1;
            }
        }
        const omni::SwarmLoopParams p = omni::SwarmLoopParams::from_launch(launch_xml, node_name ? node_name : "swarm_loop", av);
        std::string o;
        for (const auto& f : omni::SwarmLoopParams::fields()) o += std::string(f.name) + "=" + p.get(f.name) + "\n";
        for (const auto& m : p.type_mismatches) o += "#mismatch " + m + "\n";
        for (const auto& m : p.unknown_parameters) o += "#unknown " + m + "\n";
        if ((int)o.size() + 1 > cap) { g_err = "omni_swarm_params_from_launch: output buffer too small"; return 2; }
        std::memcpy(out, o.c_str(), o.size() + 1);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// A pipeline configured by one of the reference's launch files: image size, SuperPoint threshold / max_num, camera configuration, depth thresholds, PCA
// files (when the launch file's paths exist as written; pca_comp_csv / pca_mean_csv override them), every detector and geometry threshold, self_id.  What a
// launch file cannot know stays an argument: the weight files of THIS build (the launch files name TensorRT engines), precision, micro-batch, units in
// flight, row storage, whether the geometry stage runs.  PINHOLE_DEPTH takes fx fy cx cy from intrinsics4 (NULL: 90 degree field of view, as omni_pipeline_create).
omni_pipeline* omni_pipeline_create_from_launch(int device, const char* launch_xml, const char* node_name, const char* args, const char* sp_weights, const char* vlad_weights,
                                                const char* pca_comp_csv, const char* pca_mean_csv, int precision, int microbatch, int pipelines, int storage, int geometry,
                                                const double* intrinsics4) {
    try {
        std::map<std::string, std::string> av;
        if (args) {
            std::string a(args);
            size_t i = 0;
            while (i < a.size()) {
                size_t j = a.find('\n', i);
                if (j == std::string::npos) j = a.size();
                const std::string item = a.substr(i, j - i);
                const size_t k = item.find(":=");
                if (k != std::string::npos) av[item.substr(0, k)] = item.substr(k + 2);
                i = j + 1;
            }
        }
        const omni::SwarmLoopParams p = omni::SwarmLoopParams::from_launch(launch_xml, node_name ? node_name : "swarm_loop", av);
        if (p.camera_configuration != 1 && p.camera_configuration != 2)
            throw std::runtime_error("camera_configuration " + std::to_string(p.camera_configuration) + ": STEREO_FISHEYE (1) and PINHOLE_DEPTH (2) are built");
        omni::KeyframePipeline::Config c;
        p.to_pipeline_config(c);
        c.device = device; c.sp_weights = sp_weights; c.vlad_weights = vlad_weights; c.precision = precision; c.microbatch = microbatch; c.pipelines = pipelines;
        c.storage = storage; c.geometry = geometry != 0;
        if (pca_comp_csv) c.pca_comp = pca_comp_csv;
        if (pca_mean_csv) c.pca_mean = pca_mean_csv;
        if (intrinsics4) { c.fx = intrinsics4[0]; c.fy = intrinsics4[1]; c.cx = intrinsics4[2]; c.cy = intrinsics4[3]; }
        else { c.cx = c.width / 2.0; c.cy = c.height / 2.0; c.fx = c.fy = c.width / 2.0; }
        auto* pl = new omni::KeyframePipeline(c);
        try { pl->apply_params(p); } catch (...) { delete pl; throw; }
        return new omni_pipeline{pl};
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// the table itself: one line per parameter, "name<TAB>I|B|D|S<TAB>default" (swarm_loop.cpp:215-270's order); 0, or 2 when out is too small
int omni_swarm_params_table(char* out, int cap) {
    const omni::SwarmLoopParams p;
    std::string o;
    for (const auto& f : omni::SwarmLoopParams::fields()) o += std::string(f.name) + "\t" + f.type + "\t" + p.get(f.name) + "\n";
    if ((int)o.size() + 1 > cap) { g_err = "omni_swarm_params_table: output buffer too small"; return 2; }
    std::memcpy(out, o.c_str(), o.size() + 1);
    return 0;
}
// ... and onto a pipeline that exists: the thresholds of the detector and the geometry stage (the constructor arguments -- image size, SuperPoint threshold,
// camera configuration -- are the caller's, from the same parameters)
int omni_pipeline_apply_launch(omni_pipeline* h, const char* launch_xml, const char* node_name) {
    try { h->p->apply_params(omni::SwarmLoopParams::from_launch(launch_xml, node_name ? node_name : "swarm_loop")); return 0; }
    catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// omni::KeyframePipeline::host_times: the host thread's milliseconds per unit in {enqueue, wait for the GPU, build messages, detector step, geometry hand-over}
int omni_pipeline_host_times(omni_pipeline* h, double* out5, int reset) { return h->p->host_times(out5, reset != 0); }

int omni_pipeline_sync(omni_pipeline* h) {
    try { h->p->sync(); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

}  // extern "C"
