// loop_geometry.hpp -- the geometric verification stage of LoopDetector (swarm_loop/src/loop_detector.cpp), host side, f64:
//
//   LoopGeometry::compute_correspond_features (image pair)   :539-624  BFMatcher(L2, crossCheck) -> 3-D-flag filter (:574) -> homography-RANSAC mask (:589-598)
//   LoopGeometry::compute_correspond_features (frame pair)   :431-537  direction pairing, MIN_MATCH_PRE_DIR count, rotation into the main direction
//   LoopGeometry::compute_relative_pose                      :355-413  PnP-RANSAC, PnPRestoCamPose, DeltaPose, RPerror, pnp_result_verify
//   LoopGeometry::compute_loop                               :627-836  the LoopEdge the back end consumes (visualisation left out)
//   LoopGeometry::check_loop_odometry_consistency            :295-315  Mahalanobis gate against the ego-motion trajectory (pluggable source)
//   fill_stereo_landmarks                                    loop_cam.cpp:397-444 on the messages of one direction
//   fill_depth_landmarks                                     loop_cam.cpp:260-304 (PINHOLE_DEPTH: landmarks from the depth image)
//   fill_image_descriptor / stamp_image_descriptor / finish_frame_descriptor   loop_cam.cpp:525-585, 362-374, 178-217: the messages themselves
//
// The descriptor matcher is a callback: omni::BFMatcherL2X (the HIP matcher, bit-identical to cv::BFMatcher's cross-check semantics) in the
// product, a CPU matcher in the host-only tests.  Numerical core: geometry.hpp (see its header for what is restated from OpenCV 3.4 and where
// the spec deviates).  Plug into LoopDetectorCore with  det.compute_loop = geo.as_callback(on_loop);
#pragma once
#include <functional>

#include "geometry.hpp"
#include "omni_swarm.hpp"

namespace omni {

inline geom::Pose to_pose(const PoseMsg& m) {
    return {{m.position[0], m.position[1], m.position[2]}, geom::Quat{m.quat_wxyz[0], m.quat_wxyz[1], m.quat_wxyz[2], m.quat_wxyz[3]}.normalized()};
}
inline PoseMsg to_msg(const geom::Pose& p) {
    PoseMsg m;
    m.position[0] = p.pos.x; m.position[1] = p.pos.y; m.position[2] = p.pos.z;
    m.quat_wxyz[0] = p.att.w; m.quat_wxyz[1] = p.att.x; m.quat_wxyz[2] = p.att.y; m.quat_wxyz[3] = p.att.z;
    return m;
}

struct LoopEdge {                                // swarm_msgs::LoopEdge as compute_loop fills it (:789-811)
    int64_t id = 0, keyframe_id_a = 0, keyframe_id_b = 0;
    int drone_id_a = 0, drone_id_b = 0, pnp_inlier_num = 0;
    double ts_a = 0, ts_b = 0;
    geom::Pose relative_pose, self_pose_a, self_pose_b;
    double pos_cov[3] = {0, 0, 0}, ang_cov[3] = {0, 0, 0};
};

// What LoopCam::extractor_img_desc_deepnet returns for one image (loop_cam.cpp:525-585, the TensorRT branch), from the networks' outputs: the
// key points (pixels), their descriptors, the global descriptor when the image is the main camera's (!superpoint_mode; `global_desc` null
// otherwise), the key points lifted by the camera model and stored as FLOATS (:558-566), a zero landmark and a zero flag per key point.
// `lift` = cam->liftProjective followed by the division by z (camodocal is un-vendored: the camera model stays with the caller).
// `desc` may be null (a message whose descriptors were already consumed on the device, e.g. the down camera's).
inline void fill_image_descriptor(ImageDescriptor& im, const float* kps_xy, int n, const float* desc, int desc_dim, const float* global_desc, int global_dim,
                                  const std::function<geom::Vec2(const Point2f&)>& lift) {
    im.landmark_num = n;
    im.landmarks_2d.resize((size_t)n);
    for (int k = 0; k < n; ++k) im.landmarks_2d[(size_t)k] = {kps_xy[2 * k], kps_xy[2 * k + 1]};
    if (desc) im.feature_descriptor.assign(desc, desc + (size_t)n * desc_dim); else im.feature_descriptor.clear();
    if (global_desc) im.image_desc.assign(global_desc, global_desc + global_dim); else im.image_desc.clear();
    im.landmarks_2d_norm.resize((size_t)n);
    for (int k = 0; k < n; ++k) { const geom::Vec2 q = lift(im.landmarks_2d[(size_t)k]); im.landmarks_2d_norm[(size_t)k] = {(float)q.x, (float)q.y}; }
    im.landmarks_3d.assign((size_t)n, Point3f{});
    im.landmarks_flag.assign((size_t)n, 0);
}
// The per-image fields generate_stereo_image_descriptor stamps on both messages of a direction (loop_cam.cpp:362-374)
inline void stamp_image_descriptor(ImageDescriptor& im, double stamp, int self_id, const PoseMsg& camera_extrinsic, const PoseMsg& pose_drone, int64_t keyframe_id) {
    im.timestamp = stamp; im.drone_id = self_id; im.camera_extrinsic = camera_extrinsic; im.pose_drone = pose_drone; im.frame_id = keyframe_id;
}
// The frame-level fields of LoopCam::on_flattened_images (loop_cam.cpp:178-217) once f.images holds the per-direction messages
inline void finish_frame_descriptor(FisheyeFrameDescriptor& f, double stamp, int64_t keyframe_id, const PoseMsg& pose_drone, int self_id) {
    f.timestamp = stamp;
    for (size_t i = 0; i < f.images.size(); ++i) f.images[i].direction = (int)i;
    f.image_num = (int)f.images.size();
    f.msg_id = keyframe_id;
    f.pose_drone = pose_drone;
    f.landmark_num = 0;
    for (const ImageDescriptor& im : f.images) f.landmark_num += im.landmark_num;
    f.drone_id = self_id;
}

// generate_stereo_image_descriptor's triangulation (loop_cam.cpp:397-444) on one direction's up / down messages: fills landmarks_3d and
// landmarks_flag of both from the up<->down match list (ids_up / ids_down as omni_cam returns them).  Needs pose_drone and camera_extrinsic.
// The reference lifts the matched PIXELS again, in double, for the triangulation (cam->liftProjective, :403-407) -- the float
// landmarks_2d_norm of the message (extractor_img_desc_deepnet :558-566) are not what it triangulates from: pass the camera's lifting as
// `lift` (camodocal is un-vendored: it stays with the caller); without it the message's float points are used.  Skipped, as in the
// reference (:385), unless the up image has more than accept_min_3d_pts key points.
inline int fill_stereo_landmarks(ImageDescriptor& up, ImageDescriptor& down, const int* ids_up, const int* ids_down, int n_matches, double triangle_thres,
                                 int accept_min_3d_pts, const std::function<geom::Vec2(const Point2f&)>* lift = nullptr) {
    auto init = [](ImageDescriptor& im) { im.landmarks_3d.assign(im.landmarks_2d.size(), Point3f{}); im.landmarks_flag.assign(im.landmarks_2d.size(), 0); };
    init(up); init(down);
    if ((int)up.landmarks_2d.size() <= accept_min_3d_pts) return 0;
    std::vector<geom::Vec2> nu(up.landmarks_2d.size()), nd(down.landmarks_2d.size());
    for (size_t i = 0; i < nu.size(); ++i) nu[i] = lift ? (*lift)(up.landmarks_2d[i]) : geom::Vec2{up.landmarks_2d_norm[i].x, up.landmarks_2d_norm[i].y};
    for (size_t i = 0; i < nd.size(); ++i) nd[i] = lift ? (*lift)(down.landmarks_2d[i]) : geom::Vec2{down.landmarks_2d_norm[i].x, down.landmarks_2d_norm[i].y};
    std::vector<geom::Vec3> l3u, l3d;
    std::vector<uint8_t> fu, fd;
    const int count = geom::stereo_landmarks(to_pose(up.pose_drone), to_pose(up.camera_extrinsic), to_pose(down.camera_extrinsic), nu, nd, ids_up, ids_down,
                                             n_matches, triangle_thres, l3u, fu, l3d, fd);
    for (size_t i = 0; i < l3u.size(); ++i) { up.landmarks_3d[i] = {(float)l3u[i].x, (float)l3u[i].y, (float)l3u[i].z}; up.landmarks_flag[i] = fu[i]; }
    for (size_t i = 0; i < l3d.size(); ++i) { down.landmarks_3d[i] = {(float)l3d[i].x, (float)l3d[i].y, (float)l3d[i].z}; down.landmarks_flag[i] = fd[i]; }
    return count;
}

// generate_gray_depth_image_descriptor's landmarks (loop_cam.cpp:260-304; CameraConfig::PINHOLE_DEPTH, the camera mode of launch/realsense.launch):
// every key point whose depth-image value (u16 millimetres, `depth` with `depth_stride` ELEMENTS per row, depth_w x depth_h) lies strictly between
// depth_near and depth_far (metres; swarm_loop.cpp:252-253: 0.3 / 10.0) gets landmarks_3d = pose_drone * camera_extrinsic * (lifted pixel * depth) and
// flag 1; the message must carry pose_drone and camera_extrinsic (stamp_image_descriptor).  As in the reference: nothing at all unless the image has MORE
// than accept_min_3d_pts key points (:266-270); the pixel gate is the reference's literal 640 x 480 with INCLUSIVE upper bounds (:280); the depth is read at
// the pixel cv::Mat::at<ushort>(Point2f) addresses, i.e. the coordinates rounded half to even (cv::Point_<int>(Point_<float>) = saturate_cast = cvRound).
// Deviation: a key point the gate lets through but which lies outside THIS depth image (x = 640 on a 640-wide image: the reference reads past the row)
// gets no landmark.  `lift` = cam->liftProjective (camodocal is un-vendored: the camera model stays with the caller), applied to the pixel in double
// (:289) and divided by z (:291).  Returns the number of landmarks set (count_3d).
inline int fill_depth_landmarks(ImageDescriptor& im, const uint16_t* depth, int depth_stride, int depth_w, int depth_h, double depth_near, double depth_far,
                                int accept_min_3d_pts, const std::function<geom::Vec2(const Point2f&)>& lift) {
    im.landmarks_3d.assign(im.landmarks_2d.size(), Point3f{});
    im.landmarks_flag.assign(im.landmarks_2d.size(), 0);
    if ((int)im.landmarks_2d.size() <= accept_min_3d_pts) return 0;
    const geom::Pose pose_cam = to_pose(im.pose_drone) * to_pose(im.camera_extrinsic);
    int count = 0;
    for (size_t i = 0; i < im.landmarks_2d.size(); ++i) {
        const Point2f pt = im.landmarks_2d[i];
        if (pt.x < 0 || pt.x > 640 || pt.y < 0 || pt.y > 480) continue;
        const long px = std::lrint((double)pt.x), py = std::lrint((double)pt.y);
        if (px < 0 || px >= depth_w || py < 0 || py >= depth_h) continue;
        const double dep = depth[(size_t)py * depth_stride + (size_t)px] / 1000.0;
        if (dep > depth_near && dep < depth_far) {
            const geom::Vec2 n = lift(pt);
            const geom::Vec3 w = pose_cam.pos + pose_cam.att * geom::Vec3{n.x * dep, n.y * dep, dep};     // Swarm::Pose * point
            im.landmarks_3d[i] = {(float)w.x, (float)w.y, (float)w.z};
            im.landmarks_flag[i] = 1;
            ++count;
        }
    }
    return count;
}

class LoopGeometry {
public:
    // launch parameters (swarm_loop.cpp:221-250) and loop_defines.h constants
    int MIN_LOOP_NUM = 15, INIT_MODE_MIN_LOOP_NUM = 10, MIN_MATCH_PRE_DIR = 15, MIN_DIRECTION_LOOP = 3, MAX_DIRS = 4;
    bool is_4dof = true, debug_no_reject = false;
    double loop_cov_pos = 0.05, loop_cov_ang = 0.05, odometry_consistency_threshold = 2.0;
    int self_id = 0;
    int64_t MAX_LOOP_ID = 100000000;
    geom::VerifyParams verify;
    // cv::BFMatcher(NORM_L2, true).match(query n x 64, train m x 64)
    std::function<void(const float* q, int nq, const float* t, int nt, int dim, std::vector<DMatch>& out)> match;
    // ego_motion_traj.get_relative_pose_by_ts(ts_a, ts_b) -> (relative pose, 6x6 covariance diagonal as [pos3, ang3]); unset = gate passes
    std::function<bool(double ts_a, double ts_b, geom::Pose& rel, double cov6[6])> relative_odometry;
    int loop_count = 0;

    struct Correspondence {
        std::vector<geom::Vec2> new_norm_2d, old_norm_2d;
        std::vector<geom::Vec3> new_3d, old_3d;
        std::vector<std::vector<int>> new_idx, old_idx;
        std::vector<int> dirs_new, dirs_old;
    };

    // :539-624 (the USE_FUNDMENTAL branch, loop_detector.cpp:8).  Returns false when fewer than 4 flagged matches exist -- the vectors then
    // keep the unfiltered matches, and the caller (like the reference's) uses them regardless of the return value.
    bool compute_correspond_features(const ImageDescriptor& nw, const ImageDescriptor& old, std::vector<geom::Vec2>& new_norm_2d, std::vector<geom::Vec3>& new_3d,
                                     std::vector<int>& new_idx, std::vector<geom::Vec2>& old_norm_2d, std::vector<geom::Vec3>& old_3d, std::vector<int>& old_idx) const {
        const int dim = nw.landmarks_2d.empty() ? 64 : (int)(nw.feature_descriptor.size() / nw.landmarks_2d.size());
        std::vector<DMatch> matches;
        match(nw.feature_descriptor.data(), (int)nw.landmarks_2d.size(), old.feature_descriptor.data(), (int)old.landmarks_2d.size(), dim, matches);
        std::vector<geom::Vec2> old_2d, new_2d;
        for (const DMatch& m : matches) {
            const int now_id = m.queryIdx, old_id = m.trainIdx;
            if (now_id >= (int)nw.landmarks_flag.size() || !nw.landmarks_flag[now_id]) continue;               // :574
            new_2d.push_back({nw.landmarks_2d[now_id].x, nw.landmarks_2d[now_id].y});
            old_2d.push_back({old.landmarks_2d[old_id].x, old.landmarks_2d[old_id].y});
            new_idx.push_back(now_id); old_idx.push_back(old_id);
            new_3d.push_back({nw.landmarks_3d[now_id].x, nw.landmarks_3d[now_id].y, nw.landmarks_3d[now_id].z});
            new_norm_2d.push_back({nw.landmarks_2d_norm[now_id].x, nw.landmarks_2d_norm[now_id].y});
            const Point3f o3 = old_id < (int)old.landmarks_3d.size() ? old.landmarks_3d[old_id] : Point3f{};
            old_3d.push_back({o3.x, o3.y, o3.z});
            old_norm_2d.push_back({old.landmarks_2d_norm[old_id].x, old.landmarks_2d_norm[old_id].y});
        }
        if (old_2d.size() < 4) return false;
        std::vector<uint8_t> mask;
        geom::find_homography_ransac(old_2d, new_2d, 3.0, mask);                                              // :590
        auto reduce = [&](auto& v) { size_t j = 0; for (size_t i = 0; i < v.size(); ++i) if (mask[i]) v[j++] = v[i]; v.resize(j); };
        reduce(new_idx); reduce(old_idx); reduce(new_3d); reduce(new_norm_2d); reduce(old_3d); reduce(old_norm_2d);
        return true;
    }

    // :431-537.  Points of every direction pair are rotated onto the unit sphere of the MAIN direction's camera.
    bool compute_correspond_features(const FisheyeFrameDescriptor& nw, const FisheyeFrameDescriptor& old, int main_dir_new, int main_dir_old, Correspondence& c) const {
        for (int d = main_dir_new; d < main_dir_new + MAX_DIRS; ++d) {
            const int dir_new = d % MAX_DIRS, dir_old = ((main_dir_old - main_dir_new + MAX_DIRS) % MAX_DIRS + d) % MAX_DIRS;
            if (dir_new < (int)nw.images.size() && dir_old < (int)old.images.size() && old.images[dir_old].landmark_num > 0 && nw.images[dir_new].landmark_num > 0) {
                c.dirs_new.push_back(dir_new); c.dirs_old.push_back(dir_old);
            }
        }
        if (main_dir_new >= (int)nw.images.size() || main_dir_old >= (int)old.images.size()) return false;
        const geom::Quat main_quat_new = to_pose(nw.images[main_dir_new].camera_extrinsic).att, main_quat_old = to_pose(old.images[main_dir_old].camera_extrinsic).att;
        int matched_dir_count = 0;
        for (size_t i = 0; i < c.dirs_new.size(); ++i) {
            const int dir_new = c.dirs_new[i], dir_old = c.dirs_old[i];
            std::vector<geom::Vec2> n2, o2;
            std::vector<geom::Vec3> n3, o3;
            std::vector<int> ni, oi;
            compute_correspond_features(nw.images[dir_new], old.images[dir_old], n2, n3, ni, o2, o3, oi);
            if ((int)n3.size() >= MIN_MATCH_PRE_DIR) ++matched_dir_count;
            c.new_3d.insert(c.new_3d.end(), n3.begin(), n3.end());
            c.old_3d.insert(c.old_3d.end(), o3.begin(), o3.end());
            c.new_idx.push_back(ni); c.old_idx.push_back(oi);
            const geom::Quat dq_new = main_quat_new.inverse() * to_pose(nw.images[dir_new].camera_extrinsic).att;
            const geom::Quat dq_old = main_quat_old.inverse() * to_pose(old.images[dir_old].camera_extrinsic).att;
            for (auto& p : o2) c.old_norm_2d.push_back(geom::rotate_pt_norm2d(p, dq_old));
            for (auto& p : n2) c.new_norm_2d.push_back(geom::rotate_pt_norm2d(p, dq_new));
        }
        return !c.new_norm_2d.empty() && matched_dir_count >= MIN_DIRECTION_LOOP;
    }

    // :355-413: the OLD main camera is located against the NEW frame's 3-D landmarks
    int compute_relative_pose(const std::vector<geom::Vec3>& matched_3d_now, const std::vector<geom::Vec2>& matched_2d_norm_old, const geom::Pose& old_extrinsic,
                              const geom::Pose& drone_pose_now, const geom::Pose& drone_pose_old, geom::Pose& DP_old_to_new, bool init_mode, int& inlier_num) const {
        geom::Rt rt;
        std::vector<int> inliers;
        const bool ok = geom::solve_pnp_ransac(matched_3d_now, matched_2d_norm_old, init_mode ? 1000 : 100, 3.0, 0.99, rt, inliers);
        if (!ok) return 0;
        const geom::Pose p_cam_old_in_new = geom::pnp_res_to_cam_pose(rt);
        const geom::Pose p_drone_old_in_new = p_cam_old_in_new * old_extrinsic.inverse();
        DP_old_to_new = geom::Pose::DeltaPose(p_drone_old_in_new, drone_pose_now, is_4dof);
        const double rperr = geom::rp_error(p_drone_old_in_new, drone_pose_old, drone_pose_now);
        inlier_num = (int)inliers.size();
        // the reference has ONE MIN_LOOP_NUM / INIT_MODE_MIN_LOOP_NUM (launch parameters, swarm_loop.cpp:221-226) for the feature-count gates
        // and for the inlier gate of pnp_result_verify (loop_detector.cpp:317-334): the members of this class are the single source
        geom::VerifyParams vp = verify;
        vp.min_loop_num = MIN_LOOP_NUM; vp.init_mode_min_loop_num = INIT_MODE_MIN_LOOP_NUM;
        return geom::pnp_result_verify(true, init_mode, inlier_num, rperr, DP_old_to_new, vp) ? 1 : 0;
    }

    // :295-315
    bool check_loop_odometry_consistency(const LoopEdge& e) const {
        if (e.drone_id_a != e.drone_id_b || debug_no_reject || !relative_odometry) return true;       // inter-drone loops are not gated
        geom::Pose odom;
        double cov[6];
        if (!relative_odometry(e.ts_a, e.ts_b, odom, cov)) return true;
        const geom::Pose dp = geom::Pose::DeltaPose(e.relative_pose, odom, false);
        // log map (translation, rotation vector) and the squared Mahalanobis distance under the summed diagonal covariances
        const geom::Quat q = dp.att.w < 0 ? geom::Quat{-dp.att.w, -dp.att.x, -dp.att.y, -dp.att.z} : dp.att;
        const double vn = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z), ang = 2 * std::atan2(vn, q.w), k = vn > 1e-12 ? ang / vn : 2.0;
        const double v[6] = {dp.pos.x, dp.pos.y, dp.pos.z, k * q.x, k * q.y, k * q.z};
        double md = 0;
        for (int i = 0; i < 6; ++i) md += v[i] * v[i] / (cov[i] + (i < 3 ? e.pos_cov[i] : e.ang_cov[i - 3]));
        return md <= odometry_consistency_threshold;
    }

    // :627-836.  `old` must be a frame of the self drone (it provides the 2-D side); returns true and fills ret for an accepted loop.
    // compute_loop = compute_loop_core (everything but the running edge id: const, safe to run for several candidates at once on copies of this
    // object that differ only in `match`) + number_edge (the id `self_id * MAX_LOOP_ID + loop_count`, :804, in the order the loops are accepted)
    bool compute_loop_core(const FisheyeFrameDescriptor& nw, const FisheyeFrameDescriptor& old, int main_dir_new, int main_dir_old, LoopEdge& ret, bool init_mode,
                           Correspondence* out_corr = nullptr) const {
        if (nw.landmark_num < MIN_LOOP_NUM) return false;
        Correspondence c;
        bool success = compute_correspond_features(nw, old, main_dir_new, main_dir_old, c);
        geom::Pose DP_old_to_new;
        int inlier_num = 0;
        if (success) {
            if ((int)c.new_norm_2d.size() > MIN_LOOP_NUM || (init_mode && (int)c.new_norm_2d.size() > INIT_MODE_MIN_LOOP_NUM))
                success = compute_relative_pose(c.new_3d, c.old_norm_2d, to_pose(old.images[main_dir_old].camera_extrinsic), to_pose(nw.pose_drone),
                                                to_pose(old.pose_drone), DP_old_to_new, init_mode, inlier_num) != 0;
            else success = false;
        }
        if (out_corr) *out_corr = c;
        if (!success) return false;
        ret.relative_pose = DP_old_to_new;
        ret.drone_id_a = old.drone_id; ret.ts_a = old.timestamp; ret.drone_id_b = nw.drone_id; ret.ts_b = nw.timestamp;
        ret.self_pose_a = to_pose(old.pose_drone); ret.self_pose_b = to_pose(nw.pose_drone);
        ret.keyframe_id_a = old.msg_id; ret.keyframe_id_b = nw.msg_id;
        for (int i = 0; i < 3; ++i) { ret.pos_cov[i] = loop_cov_pos; ret.ang_cov[i] = loop_cov_ang; }
        ret.pnp_inlier_num = inlier_num;
        return check_loop_odometry_consistency(ret);
    }
    void number_edge(LoopEdge& e) { e.id = (int64_t)self_id * MAX_LOOP_ID + loop_count; ++loop_count; }
    bool compute_loop(const FisheyeFrameDescriptor& nw, const FisheyeFrameDescriptor& old, int main_dir_new, int main_dir_old, LoopEdge& ret, bool init_mode,
                      Correspondence* out_corr = nullptr) {
        if (!compute_loop_core(nw, old, main_dir_new, main_dir_old, ret, init_mode, out_corr)) return false;
        number_edge(ret);
        return true;
    }

    // the callback LoopDetectorCore::compute_loop expects; on_loop receives every accepted edge (LoopDetector::on_loop_connection, :838-840)
    std::function<bool(const FisheyeFrameDescriptor&, const FisheyeFrameDescriptor&, int, int, bool)> as_callback(std::function<void(const LoopEdge&)> on_loop) {
        return [this, on_loop](const FisheyeFrameDescriptor& nw, const FisheyeFrameDescriptor& old, int dn, int dold, bool init_mode) {
            LoopEdge e;
            if (!compute_loop(nw, old, dn, dold, e, init_mode)) return false;
            if (on_loop) on_loop(e);
            return true;
        };
    }
};

}  // namespace omni
