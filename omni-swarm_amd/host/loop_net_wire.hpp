// loop_net_wire.hpp -- the wire side of LoopNet (swarm_loop/src/loop_net.cpp), transport-agnostic: how one key frame leaves a drone as
// packets and how packets from other drones become FisheyeFrameDescriptor messages for LoopDetector::on_image_recv.
//
//   LoopNetWire::broadcast_fisheye_desc / broadcast_img_desc   loop_net.cpp:19-120   one "VIOKF_HEADER" packet per image + one "VIOKF_LANDMARKS"
//                                                                                    packet per landmark with a 3-D point (or all, SEND_ALL_FEATURES)
//   LoopNetWire::on_packet -> header / landmark reassembly     loop_net.cpp:187-229, 300-324
//   LoopNetWire::scan_recv_packets                             loop_net.cpp:231-298  image complete or recv_period elapsed -> image; >= MIN_DIRECTION_LOOP
//                                                                                    images or 2 x recv_period elapsed -> frame_desc_callback
//   LoopNetWire::image_desc_callback                           loop_net.cpp:143-174  frames keyed by the image's msg_id (sic), null images for the
//                                                                                    directions not (yet) received (generate_null_img_desc)
//
// The reference publishes LCM messages generated from swarm_msgs' .lcm files, which are un-vendored: the exact field order and the 8-byte LCM
// type fingerprints cannot be reproduced here.  The encoding below follows LCM's rules (fingerprint first, big-endian scalars, every
// variable-length array preceded by its own length member) over the fields loop_net.cpp reads and writes, with fingerprints of our own:
// two builds of THIS library interoperate; byte-level interop with a reference drone needs the .lcm definitions.
// Pinned: tests/cpp/wire_pin.cpp compiles the reference's own LoopNet (class + loop_net.cpp, extracted at build time) against stand-in ROS /
// LCM / swarm_msgs types and runs it next to this class: same messages out of broadcast_fisheye_desc, same frames out of frame_desc_callback
// under complete, lossy and shuffled delivery.
// Deviations (both asserted by that test): message ids come from a per-object counter mixed with the drone id instead of rand() + nsec
// (loop_net.cpp:28,81): deterministic; a header always activates its image, also when a landmark packet overtook it (see on_header).
#pragma once
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <set>

#include "omni_swarm.hpp"

namespace omni {
namespace wire {

struct Writer {
    std::vector<uint8_t> b;
    void u8(uint8_t v) { b.push_back(v); }
    void i32(int32_t v) { for (int s = 24; s >= 0; s -= 8) b.push_back((uint8_t)((uint32_t)v >> s)); }
    void i64(int64_t v) { for (int s = 56; s >= 0; s -= 8) b.push_back((uint8_t)((uint64_t)v >> s)); }
    void f32(float v) { uint32_t u; std::memcpy(&u, &v, 4); i32((int32_t)u); }
    void f64(double v) { uint64_t u; std::memcpy(&u, &v, 8); i64((int64_t)u); }
    void pose(const PoseMsg& p) { for (double v : p.position) f64(v); for (double v : p.quat_wxyz) f64(v); }
};
struct Reader {
    const uint8_t* p; size_t n, o = 0; bool ok = true;
    Reader(const uint8_t* data, size_t len) : p(data), n(len) {}
    bool need(size_t k) { if (o + k > n) { ok = false; return false; } return true; }
    uint8_t u8() { return need(1) ? p[o++] : 0; }
    int32_t i32() { if (!need(4)) return 0; uint32_t v = 0; for (int i = 0; i < 4; ++i) v = (v << 8) | p[o++]; return (int32_t)v; }
    int64_t i64() { if (!need(8)) return 0; uint64_t v = 0; for (int i = 0; i < 8; ++i) v = (v << 8) | p[o++]; return (int64_t)v; }
    float f32() { uint32_t u = (uint32_t)i32(); float v; std::memcpy(&v, &u, 4); return v; }
    double f64() { uint64_t u = (uint64_t)i64(); double v; std::memcpy(&v, &u, 8); return v; }
    PoseMsg pose() { PoseMsg m; for (double& v : m.position) v = f64(); for (double& v : m.quat_wxyz) v = f64(); return m; }
};

constexpr int64_t FP_HEADER = 0x4f4d4e4948445231ll;      // "OMNIHDR1"
constexpr int64_t FP_LANDMARK = 0x4f4d4e494c4d4b31ll;    // "OMNILMK1"
constexpr const char* CH_HEADER = "VIOKF_HEADER";        // loop_net.cpp:12
constexpr const char* CH_LANDMARKS = "VIOKF_LANDMARKS";  // loop_net.cpp:13

struct ImageDescriptorHeader {                           // ImageDescriptorHeader_t as loop_net.cpp:51-63 fills it
    double timestamp = 0; int drone_id = 0, feature_num = 0, direction = 0; bool prevent_adding_db = false;
    int64_t msg_id = 0, frame_id = 0;
    std::vector<float> image_desc; PoseMsg pose_drone, camera_extrinsic;
};
struct LandmarkDescriptor {                              // LandmarkDescriptor_t as loop_net.cpp:70-85 fills it
    int landmark_id = 0, drone_id = 0, landmark_flag = 0;
    Point2f landmark_2d_norm{0, 0}, landmark_2d{0, 0}; Point3f landmark_3d;
    std::vector<float> feature_descriptor; int64_t msg_id = 0, header_id = 0;
};

inline std::vector<uint8_t> encode(const ImageDescriptorHeader& h) {
    Writer w;
    w.i64(FP_HEADER); w.f64(h.timestamp); w.i32(h.drone_id); w.i64(h.msg_id); w.i64(h.frame_id); w.i32(h.feature_num); w.i32(h.direction);
    w.u8(h.prevent_adding_db ? 1 : 0); w.pose(h.pose_drone); w.pose(h.camera_extrinsic);
    w.i32((int32_t)h.image_desc.size());
    for (float v : h.image_desc) w.f32(v);
    return std::move(w.b);
}
inline bool decode(const uint8_t* p, size_t n, ImageDescriptorHeader& h) {
    Reader r(p, n);
    if (r.i64() != FP_HEADER) return false;
    h.timestamp = r.f64(); h.drone_id = r.i32(); h.msg_id = r.i64(); h.frame_id = r.i64(); h.feature_num = r.i32(); h.direction = r.i32();
    h.prevent_adding_db = r.u8() != 0; h.pose_drone = r.pose(); h.camera_extrinsic = r.pose();
    const int32_t k = r.i32();
    if (!r.ok || k < 0 || (size_t)k * 4 > n) return false;
    h.image_desc.resize(k);
    for (auto& v : h.image_desc) v = r.f32();
    return r.ok && r.o == n;
}
inline std::vector<uint8_t> encode(const LandmarkDescriptor& l) {
    Writer w;
    w.i64(FP_LANDMARK); w.i64(l.msg_id); w.i64(l.header_id); w.i32(l.drone_id); w.i32(l.landmark_id); w.i32(l.landmark_flag);
    w.f32(l.landmark_2d_norm.x); w.f32(l.landmark_2d_norm.y); w.f32(l.landmark_2d.x); w.f32(l.landmark_2d.y);
    w.f32(l.landmark_3d.x); w.f32(l.landmark_3d.y); w.f32(l.landmark_3d.z);
    w.i32((int32_t)l.feature_descriptor.size());
    for (float v : l.feature_descriptor) w.f32(v);
    return std::move(w.b);
}
inline bool decode(const uint8_t* p, size_t n, LandmarkDescriptor& l) {
    Reader r(p, n);
    if (r.i64() != FP_LANDMARK) return false;
    l.msg_id = r.i64(); l.header_id = r.i64(); l.drone_id = r.i32(); l.landmark_id = r.i32(); l.landmark_flag = r.i32();
    l.landmark_2d_norm.x = r.f32(); l.landmark_2d_norm.y = r.f32(); l.landmark_2d.x = r.f32(); l.landmark_2d.y = r.f32();
    l.landmark_3d.x = r.f32(); l.landmark_3d.y = r.f32(); l.landmark_3d.z = r.f32();
    const int32_t k = r.i32();
    if (!r.ok || k < 0 || (size_t)k * 4 > n) return false;
    l.feature_descriptor.resize(k);
    for (auto& v : l.feature_descriptor) v = r.f32();
    return r.ok && r.o == n;
}

}  // namespace wire

class LoopNetWire {
public:
    static constexpr int FEATURE_DESC_SIZE = 64;         // loop_defines.h:67
    static constexpr size_t IMAGE_DESC_SIZE = 4096;      // MobileNetVLAD's output (loop_defines.h)
    static constexpr int MAX_FEATURES = 4096;            // sanity bound on a header's feature_num (the reference sends <= 200)
    double orphan_timeout = 10.0;                        // landmarks whose header never arrives are forgotten after this many seconds
    double recv_period = 0.5;                            // loop_net.h:33
    int MIN_DIRECTION_LOOP = 3;
    bool SEND_ALL_FEATURES = false;
    // false = the reference's key (frame_hash = image.msg_id, loop_net.cpp:144: every image opens a frame of its own, so a received fisheye
    // frame never holds more than one direction); true = key by (drone, frame_id) so that the 4 directions of a key frame meet again
    bool group_by_frame_id = false;
    // transport: publish(channel, bytes) is called for every outgoing packet; feed incoming packets to on_packet()
    std::function<void(const char* channel, const std::vector<uint8_t>& bytes)> publish;
    std::function<void(const FisheyeFrameDescriptor&)> frame_desc_callback;      // -> LoopDetector::on_image_recv
    std::function<void(int drone_id, float rate)> msg_recv_rate_callback = [](int, float) {};
    explicit LoopNetWire(int self_id) : self_id_(self_id) {}

    // :19-26
    void broadcast_fisheye_desc(FisheyeFrameDescriptor& f) { for (auto& img : f.images) if (img.landmark_num > 0) broadcast_img_desc(img); }

    // :28-101 (the header + landmark path; the optional whole-ImageDescriptor_t publish of :103-118 is debug transport and left out)
    int broadcast_img_desc(ImageDescriptor& img) {
        img.msg_id = next_id();
        sent_message.insert(img.msg_id);
        int feature_num = 0;
        for (int i = 0; i < img.landmark_num; ++i) if (i < (int)img.landmarks_flag.size() && img.landmarks_flag[i] > 0) ++feature_num;
        wire::ImageDescriptorHeader h;
        h.timestamp = img.timestamp; h.drone_id = img.drone_id; h.image_desc = img.image_desc; h.pose_drone = img.pose_drone;
        h.camera_extrinsic = img.camera_extrinsic; h.prevent_adding_db = img.prevent_adding_db; h.msg_id = img.msg_id; h.frame_id = img.frame_id;
        // the landmarks WITH a 3-D point, also under SEND_ALL_FEATURES (loop_net.cpp:62): the receiver then closes the image after that many
        // packets and drops the rest -- the reference's behaviour, kept so that its drones and ours read each other's headers the same way
        h.feature_num = feature_num; h.direction = img.direction;
        size_t bytes = 0;
        { auto b = wire::encode(h); bytes += b.size(); publish(wire::CH_HEADER, b); }
        for (int i = 0; i < img.landmark_num; ++i) {
            const int flag = i < (int)img.landmarks_flag.size() ? img.landmarks_flag[i] : 0;
            if (!(flag > 0 || SEND_ALL_FEATURES)) continue;
            wire::LandmarkDescriptor lm;
            lm.landmark_id = i; lm.landmark_2d_norm = img.landmarks_2d_norm[i]; lm.landmark_2d = img.landmarks_2d[i];
            lm.landmark_3d = i < (int)img.landmarks_3d.size() ? img.landmarks_3d[i] : Point3f{}; lm.landmark_flag = flag; lm.drone_id = img.drone_id;
            lm.feature_descriptor.assign(img.feature_descriptor.begin() + (size_t)i * FEATURE_DESC_SIZE, img.feature_descriptor.begin() + (size_t)(i + 1) * FEATURE_DESC_SIZE);
            lm.msg_id = next_id(); lm.header_id = img.msg_id;
            auto b = wire::encode(lm); bytes += b.size(); publish(wire::CH_LANDMARKS, b);
        }
        return (int)bytes;
    }

    // lcm.subscribe handlers (:9-13): returns false for packets that do not parse
    // Array lengths come from the network: a landmark must carry exactly FEATURE_DESC_SIZE floats and a header a 4096-float global descriptor
    // (or none) -- LoopDetector reads that many from image_desc / feature_descriptor without looking at the vector's size -- and the direction
    // must index the 4 images of a frame.  Packets of any other shape (another version, corruption that kept the framing) are dropped.
    bool on_packet(const char* channel, const uint8_t* data, size_t n, double now) {
        if (std::strcmp(channel, wire::CH_HEADER) == 0) {
            wire::ImageDescriptorHeader h;
            if (!wire::decode(data, n, h)) return false;
            if (!(h.image_desc.size() == IMAGE_DESC_SIZE || h.image_desc.empty()) || h.feature_num < 0 || h.feature_num > MAX_FEATURES || h.direction < 0 || h.direction >= 4) return false;
            on_header(h, now);
            return true;
        }
        if (std::strcmp(channel, wire::CH_LANDMARKS) == 0) {
            wire::LandmarkDescriptor l;
            if (!wire::decode(data, n, l)) return false;
            if (l.feature_descriptor.size() != (size_t)FEATURE_DESC_SIZE) return false;
            on_landmark(l, now);
            return true;
        }
        return false;
    }

    // :231-298; call it periodically too (the reference runs it from every landmark packet, :323)
    void scan_recv_packets(double tnow) {
        std::vector<int64_t> finish_recv;
        for (int64_t id : active_receving_msg) {
            ImageDescriptor& im = received_images[id];
            if (tnow - msg_header_recv_time[id] > recv_period || im.landmark_num == (int)im.landmarks_2d.size()) {
                const float rate = im.landmark_num > 0 ? (float)im.landmarks_2d.size() / (float)im.landmark_num : 1.f;
                im.landmark_num = (int)im.landmarks_2d.size();
                finish_recv.push_back(id);
                msg_recv_rate_callback(im.drone_id, rate);
            }
        }
        for (int64_t id : finish_recv) { blacklist.insert(id); active_receving_msg.erase(id); }
        for (int64_t id : finish_recv) {
            ImageDescriptor& im = received_images[id];
            im.landmark_num = (int)im.landmarks_2d.size();
            if (!im.landmarks_2d.empty()) image_desc_callback(im);
            received_images.erase(id);
            msg_header_recv_time.erase(id);
            orphan_first_seen.erase(id);
        }
        // entries created by landmarks whose header never came (lost, rejected, or sent by somebody else's bug) must not pile up
        for (auto it = orphan_first_seen.begin(); it != orphan_first_seen.end();) {
            if (msg_header_recv_time.count(it->first)) { it = orphan_first_seen.erase(it); continue; }      // the header arrived meanwhile
            if (tnow - it->second > orphan_timeout) { received_images.erase(it->first); blacklist.insert(it->first); it = orphan_first_seen.erase(it); continue; }
            ++it;
        }
        std::vector<int64_t> finish_frames;
        for (int64_t hash : active_receving_frames) {
            int count = 0;
            for (auto& im : received_frames[hash].images) if (im.landmark_num > 0) ++count;
            if (tnow - frame_header_recv_time[hash] > 2.0 * recv_period || count >= MIN_DIRECTION_LOOP) finish_frames.push_back(hash);
        }
        for (int64_t hash : finish_frames) {
            FisheyeFrameDescriptor& f = received_frames[hash];
            active_receving_frames.erase(hash);
            f.landmark_num = 0;
            for (auto& im : f.images) f.landmark_num += im.landmark_num;
            if (frame_desc_callback) frame_desc_callback(f);
            received_frames.erase(hash);
            frame_header_recv_time.erase(hash);
        }
    }

    bool msg_blocked(int64_t id) const { return blacklist.count(id) || sent_message.count(id); }     // loop_net.h:85-87
    int pending_images() const { return (int)received_images.size(); }

private:
    // :187-229
    void on_header(const wire::ImageDescriptorHeader& m, double now) {
        if (msg_blocked(m.msg_id)) return;
        msg_header_recv_time[m.msg_id] = now;
        // the reference marks the image active only when the header CREATES the entry (:197-201): a landmark packet that overtakes its header
        // (on_landmark creates the entry, :308-311) leaves the image inactive for ever.  Fixed here: the header always activates it.
        active_receving_msg.insert(m.msg_id);
        ImageDescriptor& t = received_images[m.msg_id];
        t.timestamp = m.timestamp; t.drone_id = m.drone_id; t.image_desc = m.image_desc; t.pose_drone = m.pose_drone; t.camera_extrinsic = m.camera_extrinsic;
        t.landmark_num = m.feature_num; t.frame_id = m.frame_id; t.msg_id = m.msg_id; t.prevent_adding_db = m.prevent_adding_db; t.direction = m.direction;
    }
    // :300-324
    void on_landmark(const wire::LandmarkDescriptor& m, double now) {
        if (msg_blocked(m.header_id)) return;
        if (!received_images.count(m.header_id)) orphan_first_seen.emplace(m.header_id, now);
        ImageDescriptor& t = received_images[m.header_id];          // a landmark may overtake its header: the entry is created here (:308-311)
        t.landmarks_2d_norm.push_back(m.landmark_2d_norm); t.landmarks_2d.push_back(m.landmark_2d); t.landmarks_3d.push_back(m.landmark_3d);
        t.landmarks_flag.push_back((uint8_t)m.landmark_flag);
        t.feature_descriptor.insert(t.feature_descriptor.end(), m.feature_descriptor.begin(), m.feature_descriptor.begin() + FEATURE_DESC_SIZE);
        scan_recv_packets(now);
    }
    // :143-174
    void image_desc_callback(const ImageDescriptor& image) {
        const int64_t frame_hash = group_by_frame_id ? (((int64_t)image.drone_id << 48) ^ image.frame_id) : image.msg_id;      // (sic) see group_by_frame_id
        auto it = received_frames.find(frame_hash);
        if (it == received_frames.end()) {
            FisheyeFrameDescriptor f;
            f.image_num = 4; f.timestamp = image.timestamp;
            for (int i = 0; i < f.image_num; ++i) {
                if (i != image.direction) { ImageDescriptor null_img; null_img.landmark_num = 0; f.images.push_back(null_img); }   // generate_null_img_desc
                else f.images.push_back(image);
            }
            f.msg_id = image.frame_id; f.pose_drone = image.pose_drone; f.landmark_num = 0; f.drone_id = image.drone_id;
            received_frames[frame_hash] = f;
            frame_header_recv_time[frame_hash] = msg_header_recv_time[image.msg_id];
            active_receving_frames.insert(frame_hash);
        } else if (image.direction >= 0 && image.direction < (int)it->second.images.size()) {
            it->second.images[image.direction] = image;
        }
    }
    int64_t next_id() { return ((int64_t)self_id_ << 40) | (++counter_); }

    int self_id_;
    int64_t counter_ = 0;
    std::set<int64_t> sent_message, blacklist, active_receving_msg, active_receving_frames;
    std::map<int64_t, ImageDescriptor> received_images;
    std::map<int64_t, FisheyeFrameDescriptor> received_frames;
    std::map<int64_t, double> msg_header_recv_time, frame_header_recv_time, orphan_first_seen;
};

}  // namespace omni
