// keyframe_intake.hpp -- which camera frames become key frames: the intake of SwarmLoop in front of the CNN front end
// (swarm_loop/src/swarm_loop.cpp), restated without ROS:
//   push_images            flatten_raw_callback / stereo_images_callback / depth_images_callback   :55-98   the queue of frames waiting for their odometry
//   find_images_raw        find_images_raw                                                          :32-53   oldest queued frame within 1 ms of an odometry stamp
//   odometry               odometry_callback            (every VIO odometry message)                :100-112
//   odometry_keyframe      odometry_keyframe_callback   (VIO key-frame poses)                       :114-122
//   nonkeyframe            VIOnonKF_callback                                                        :124-138
//   keyframe               VIOKF_callback: rate limit, prevent_adding_db, the "CNN not ready" exit  :140-170
// Pinned to the reference's text by tests/cpp/ingest_pin.cpp (oracle/Makefile compiles those functions verbatim against stand-in ROS types) on
// jittered stamp streams with drops: tests/test_geometry_cpu.py::test_keyframe_intake_is_pinned_to_the_reference_text.
//
// Frame: whatever the caller queues per camera frame (image pointers, extrinsics ...); the intake only needs its stamp and lets the caller's
// `extract` do the rest.  Times are seconds as double (ros::Time::toSec()).  One deviation, stated: ros::Time subtracts in integer nanoseconds,
// here the doubles are subtracted -- the two agree unless a difference sits within 1e-9 of one of the thresholds.
#pragma once
#include <cmath>
#include <deque>
#include <functional>
#include <mutex>

#include "omni_swarm.hpp"

namespace omni {

template <class Frame>
class KeyframeIntake {
public:
    struct Queued { double stamp = 0; PoseMsg pose_drone; Frame frame; };
    // launch parameters (swarm_loop.cpp:216-217,238) and loop_defines.h:34
    double max_freq = 1.0, min_movement_keyframe = 0.3, accept_nonkeyframe_waitsec = 5.0;
    static constexpr double INIT_ACCEPT_NONKEYFRAME_WAITSEC = 1.0;
    // LoopCam::on_flattened_images + everything behind it (broadcast, LoopDetector::on_image_recv, node_frame): called for every frame that passes
    // the gates with the frame, its odometry pose and `nonkeyframe` && moved less than min_movement_keyframe (= FisheyeFrameDescriptor_t::prevent_adding_db,
    // :156); returns the frame's landmark_num -- 0 means "the networks gave nothing" and the frame does not count as a key frame (:158-161).
    // With an asynchronous front end (KeyframePipeline::push_keyframe) return a positive number at once: the two differ only for frames without a
    // single key point, which the detector drops anyway (loop_detector.cpp:60-68).
    std::function<int(const Queued&, bool prevent_adding_db)> extract;

    // a camera frame with its stamp (the image callbacks, :55-98)
    void push_images(double stamp, Frame f) {
        std::lock_guard<std::mutex> lk(mu_);
        Queued q; q.stamp = stamp; q.frame = std::move(f);
        queue_.push_back(std::move(q));
    }
    // odometry_callback (:100-112)
    void odometry(double stamp, const PoseMsg& pose) {
        if (stamp - last_invoke_ < accept_nonkeyframe_waitsec) return;
        Queued q;
        if (find_images_raw(stamp, pose, q) && q.stamp > 1000) nonkeyframe(q);
    }
    // odometry_keyframe_callback (:114-122); false: no camera frame waits for this key-frame pose (the reference warns)
    bool odometry_keyframe(double stamp, const PoseMsg& pose) {
        Queued q;
        if (find_images_raw(stamp, pose, q) && q.stamp > 1000) { keyframe(q, false); return true; }
        return false;
    }

    bool received_image() const { return received_image_; }
    double last_invoke() const { return last_invoke_; }
    double last_kftime() const { return last_kftime_; }
    size_t queued() const { std::lock_guard<std::mutex> lk(mu_); return queue_.size(); }

private:
    // find_images_raw (:32-53): frames more than 1 ms older than the odometry are dropped; the front frame is taken if it is within 1 ms
    bool find_images_raw(double stamp, const PoseMsg& pose, Queued& out) {
        std::lock_guard<std::mutex> lk(mu_);
        while (!queue_.empty() && stamp - queue_.front().stamp > 1e-3) queue_.pop_front();
        if (!queue_.empty() && std::fabs(stamp - queue_.front().stamp) < 1e-3) {
            out = std::move(queue_.front());
            queue_.pop_front();
            out.pose_drone = pose;
            return true;
        }
        return false;                                   // (the reference returns a frame stamped 0, which fails its `> 1000` test)
    }
    // VIOnonKF_callback (:124-138)
    void nonkeyframe(const Queued& q) {
        if (!received_image_ && q.stamp - last_kftime_ > INIT_ACCEPT_NONKEYFRAME_WAITSEC) { keyframe(q, false); return; }
        if (q.stamp - last_kftime_ > accept_nonkeyframe_waitsec) keyframe(q, true);
    }
    // VIOKF_callback (:140-170)
    void keyframe(const Queued& q, bool nonkeyframe) {
        const double dx = last_pos_[0] - q.pose_drone.position[0], dy = last_pos_[1] - q.pose_drone.position[1], dz = last_pos_[2] - q.pose_drone.position[2];
        const double dpos = std::sqrt(dx * dx + dy * dy + dz * dz);
        if (q.stamp - last_invoke_ < 1 / max_freq) return;
        last_invoke_ = q.stamp;
        last_kftime_ = q.stamp;
        const int landmark_num = extract ? extract(q, nonkeyframe && dpos < min_movement_keyframe) : 0;
        if (landmark_num == 0) return;                  // "Null img desc, CNN no ready"
        received_image_ = true;
        for (int k = 0; k < 3; ++k) last_pos_[k] = q.pose_drone.position[k];
    }

    mutable std::mutex mu_;
    std::deque<Queued> queue_;
    bool received_image_ = false;
    double last_invoke_ = 0, last_kftime_ = 0;
    double last_pos_[3] = {10000, 10000, 10000};        // swarm_loop.h:30
};

}  // namespace omni
