// Pins omni::LoopNetWire (omni-swarm_amd/host/loop_net_wire.hpp) to the TEXT of the reference's LoopNet: swarm_loop/include/swarm_loop/loop_net.h
// (class LoopNet) and swarm_loop/src/loop_net.cpp are extracted at build time by oracle/Makefile (oracle/_ref/loopnet_hdr.inc, loopnet_src.inc:
// git-ignored, nothing of the reference enters the repository) and compiled VERBATIM into this program against the stand-ins of
// oracle/ref_build/loopnet_shim.h (ROS, LCM, the un-vendored swarm_msgs structs).  Both classes then see the same key frames and the same packet
// schedules:
//   SEND     broadcast_fisheye_desc: the same sequence of header / landmark messages, field for field (message ids: same structure -- the
//            reference draws them with rand())
//   RECV     on_img_desc_header_recevied / on_landmark_recevied / scan_recv_packets vs on_packet: the same FisheyeFrameDescriptor sequence out of
//            frame_desc_callback under complete, lossy (landmarks, headers) and shuffled delivery with the reference's time-outs
//   OVERTAKE a landmark that arrives before its header: the reference never activates that image (loop_net.cpp:197-201 vs :308-311) and never
//            delivers it; LoopNetWire does -- the one deliberate difference, asserted as such
// usage: wire_pin <seed>      prints one line per check, exit code 0 = all equal
#include "../../oracle/ref_build/loopnet_shim.h"

#include <ctime>
using namespace swarm_msgs;
#include REF_LOOPNET_HDR
#include REF_LOOPNET_SRC

#undef FEATURE_DESC_SIZE            // loop_defines.h's macro; LoopNetWire has a constant of that name
#include "../../omni-swarm_amd/host/loop_net_wire.hpp"

#include <algorithm>

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 6364136223846793005ull + 1442695040888963407ull) {}
    uint32_t u32() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }
    int below(int n) { return (int)(u32() % (uint32_t)n); }
    float unit() { return (float)(u32() & 0xffffff) / (float)0x1000000; }
};

int g_fail = 0;
#define CHECK(cond, what) do { if (!(cond)) { if (g_fail < 20) std::printf("MISMATCH %s (%s:%d)\n", what, __FILE__, __LINE__); ++g_fail; } } while (0)

// one key frame, the same content in the reference's and in the product's types
void make_frame(Rng& rng, int drone, int64_t frame_id, FisheyeFrameDescriptor_t& fr, omni::FisheyeFrameDescriptor& fp) {
    fr = FisheyeFrameDescriptor_t(); fp = omni::FisheyeFrameDescriptor();
    fr.image_num = 4; fp.image_num = 4;
    fr.msg_id = frame_id; fp.msg_id = frame_id;
    fr.drone_id = drone; fp.drone_id = drone;
    fr.timestamp.sec = 1000 + (int)frame_id; fr.timestamp.nsec = 1000 * rng.below(1000000);
    fp.timestamp = fr.timestamp.sec + 1e-9 * fr.timestamp.nsec;
    for (int k = 0; k < 3; ++k) { fr.pose_drone.position[k] = rng.unit() * 10; fp.pose_drone.position[k] = fr.pose_drone.position[k]; }
    for (int k = 0; k < 4; ++k) { fr.pose_drone.orientation[k] = rng.unit(); fp.pose_drone.quat_wxyz[k] = fr.pose_drone.orientation[k]; }
    for (int d = 0; d < 4; ++d) {
        ImageDescriptor_t ir; omni::ImageDescriptor ip;
        const int n = (d == (int)(frame_id % 4)) ? 0 : 3 + rng.below(30);              // one direction without landmarks
        ir.timestamp = fr.timestamp; ip.timestamp = fp.timestamp;
        ir.drone_id = drone; ip.drone_id = drone; ir.frame_id = frame_id; ip.frame_id = frame_id; ir.direction = d; ip.direction = d;
        ir.prevent_adding_db = (rng.below(4) == 0); ip.prevent_adding_db = ir.prevent_adding_db;
        ir.pose_drone = fr.pose_drone; ip.pose_drone = fp.pose_drone;
        for (int k = 0; k < 3; ++k) { ir.camera_extrinsic.position[k] = rng.unit(); ip.camera_extrinsic.position[k] = ir.camera_extrinsic.position[k]; }
        for (int k = 0; k < 4; ++k) { ir.camera_extrinsic.orientation[k] = rng.unit(); ip.camera_extrinsic.quat_wxyz[k] = ir.camera_extrinsic.orientation[k]; }
        ir.image_desc_size = 4096; ir.image_desc.resize(4096);
        for (auto& v : ir.image_desc) v = rng.unit() - 0.5f;
        ip.image_desc = ir.image_desc;
        ir.landmark_num = n; ip.landmark_num = n;
        for (int i = 0; i < n; ++i) {
            Point2d_t a, b; Point3d_t c;
            a.x = rng.unit(); a.y = rng.unit(); b.x = rng.unit() * 600; b.y = rng.unit() * 480;
            const int flag = rng.below(3) != 0;
            if (flag) { c.x = rng.unit() * 5; c.y = rng.unit() * 5; c.z = rng.unit() * 5; }
            ir.landmarks_2d_norm.push_back(a); ir.landmarks_2d.push_back(b); ir.landmarks_3d.push_back(c); ir.landmarks_flag.push_back((int8_t)flag);
            ip.landmarks_2d_norm.push_back({a.x, a.y}); ip.landmarks_2d.push_back({b.x, b.y});
            omni::Point3f c3; c3.x = c.x; c3.y = c.y; c3.z = c.z;
            ip.landmarks_3d.push_back(c3); ip.landmarks_flag.push_back((uint8_t)flag);
            for (int k = 0; k < 64; ++k) { const float v = rng.unit() - 0.5f; ir.feature_descriptor.push_back(v); ip.feature_descriptor.push_back(v); }
        }
        ir.feature_descriptor_size = (int)ir.feature_descriptor.size();
        fr.images.push_back(ir); fp.images.push_back(ip);
    }
}

bool same_pose(const Pose_t& a, const omni::PoseMsg& b) {
    for (int k = 0; k < 3; ++k) if (a.position[k] != b.position[k]) return false;
    for (int k = 0; k < 4; ++k) if (a.orientation[k] != b.quat_wxyz[k]) return false;
    return true;
}
double secs(const Time_t& t) { return t.sec + 1e-9 * t.nsec; }

struct ProductPacket { std::string channel; std::vector<uint8_t> bytes; };

// ---- SEND: the two senders emit the same messages ---------------------------------------------------------------------------------------
void check_send(Rng& rng, bool send_all) {
    SEND_ALL_FEATURES = send_all;
    FisheyeFrameDescriptor_t fr; omni::FisheyeFrameDescriptor fp;
    make_frame(rng, 3, 77, fr, fp);
    LoopNet ref("", false, false);
    ref_net::sink().clear();
    ref.broadcast_fisheye_desc(fr);
    const std::vector<ref_net::Published> sent = ref_net::sink();
    omni::LoopNetWire w(3);
    w.SEND_ALL_FEATURES = send_all;
    std::vector<ProductPacket> pk;
    w.publish = [&](const char* ch, const std::vector<uint8_t>& b) { pk.push_back({ch, b}); };
    w.broadcast_fisheye_desc(fp);
    CHECK(sent.size() == pk.size(), "number of messages");
    int64_t ref_hdr = 0, prod_hdr = 0;
    std::set<int64_t> prod_ids;
    for (size_t i = 0; i < sent.size() && i < pk.size(); ++i) {
        CHECK(sent[i].channel == pk[i].channel, "channel order");
        if (sent[i].channel == "VIOKF_HEADER") {
            omni::wire::ImageDescriptorHeader h;
            CHECK(omni::wire::decode(pk[i].bytes.data(), pk[i].bytes.size(), h), "header decodes");
            const ImageDescriptorHeader_t& r = sent[i].header;
            CHECK(h.drone_id == r.drone_id && h.frame_id == r.frame_id && h.feature_num == r.feature_num && h.direction == r.direction, "header scalars");
            CHECK(h.prevent_adding_db == r.prevent_adding_db && h.timestamp == secs(r.timestamp), "header flags / time");
            CHECK(h.image_desc == r.image_desc && same_pose(r.pose_drone, h.pose_drone) && same_pose(r.camera_extrinsic, h.camera_extrinsic), "header arrays");
            ref_hdr = r.msg_id; prod_hdr = h.msg_id;
            CHECK(prod_ids.insert(h.msg_id).second, "header id unique");
        } else if (sent[i].channel == "VIOKF_LANDMARKS") {
            omni::wire::LandmarkDescriptor l;
            CHECK(omni::wire::decode(pk[i].bytes.data(), pk[i].bytes.size(), l), "landmark decodes");
            const LandmarkDescriptor_t& r = sent[i].landmark;
            CHECK(l.landmark_id == r.landmark_id && l.drone_id == r.drone_id && l.landmark_flag == r.landmark_flag, "landmark scalars");
            CHECK(l.landmark_2d.x == r.landmark_2d.x && l.landmark_2d.y == r.landmark_2d.y && l.landmark_2d_norm.x == r.landmark_2d_norm.x &&
                  l.landmark_2d_norm.y == r.landmark_2d_norm.y, "landmark 2-D");
            CHECK(l.landmark_3d.x == r.landmark_3d.x && l.landmark_3d.y == r.landmark_3d.y && l.landmark_3d.z == r.landmark_3d.z, "landmark 3-D");
            CHECK(l.feature_descriptor == r.feature_descriptor && (int)l.feature_descriptor.size() == r.desc_len, "landmark descriptor");
            CHECK(r.header_id == ref_hdr && l.header_id == prod_hdr, "landmark -> header link");
            CHECK(prod_ids.insert(l.msg_id).second, "landmark id unique");
        } else {
            CHECK(false, "unexpected channel");
        }
    }
    std::printf("SEND all=%d messages %zu\n", (int)send_all, sent.size());
    SEND_ALL_FEATURES = false;
}

// ---- RECV: the same packets, the same schedule -> the same frames -------------------------------------------------------------------------
struct Delivery { double t; int idx; };

std::vector<uint8_t> to_product(const ref_net::Published& p) {
    if (p.channel == "VIOKF_HEADER") {
        omni::wire::ImageDescriptorHeader h;
        const ImageDescriptorHeader_t& r = p.header;
        h.timestamp = secs(r.timestamp); h.drone_id = r.drone_id; h.feature_num = r.feature_num; h.direction = r.direction;
        h.prevent_adding_db = r.prevent_adding_db; h.msg_id = r.msg_id; h.frame_id = r.frame_id; h.image_desc = r.image_desc;
        for (int k = 0; k < 3; ++k) { h.pose_drone.position[k] = r.pose_drone.position[k]; h.camera_extrinsic.position[k] = r.camera_extrinsic.position[k]; }
        for (int k = 0; k < 4; ++k) { h.pose_drone.quat_wxyz[k] = r.pose_drone.orientation[k]; h.camera_extrinsic.quat_wxyz[k] = r.camera_extrinsic.orientation[k]; }
        return omni::wire::encode(h);
    }
    omni::wire::LandmarkDescriptor l;
    const LandmarkDescriptor_t& r = p.landmark;
    l.landmark_id = r.landmark_id; l.drone_id = r.drone_id; l.landmark_flag = r.landmark_flag; l.msg_id = r.msg_id; l.header_id = r.header_id;
    l.landmark_2d_norm = {r.landmark_2d_norm.x, r.landmark_2d_norm.y}; l.landmark_2d = {r.landmark_2d.x, r.landmark_2d.y};
    l.landmark_3d.x = r.landmark_3d.x; l.landmark_3d.y = r.landmark_3d.y; l.landmark_3d.z = r.landmark_3d.z;
    l.feature_descriptor = r.feature_descriptor;
    return omni::wire::encode(l);
}

bool same_image(const ImageDescriptor_t& r, const omni::ImageDescriptor& p) {
    if (r.landmark_num != p.landmark_num) return false;
    if (r.landmark_num == 0 && r.landmarks_2d.empty() && p.landmarks_2d.empty() && r.image_desc.empty() && p.image_desc.empty()) return true;   // null image
    if (r.drone_id != p.drone_id || r.direction != p.direction || r.msg_id != p.msg_id || r.frame_id != p.frame_id || r.prevent_adding_db != p.prevent_adding_db) return false;
    if (secs(r.timestamp) != p.timestamp || r.image_desc != p.image_desc || r.feature_descriptor != p.feature_descriptor) return false;
    if (!same_pose(r.pose_drone, p.pose_drone) || !same_pose(r.camera_extrinsic, p.camera_extrinsic)) return false;
    if (r.landmarks_2d.size() != p.landmarks_2d.size() || r.landmarks_3d.size() != p.landmarks_3d.size() || r.landmarks_flag.size() != p.landmarks_flag.size()) return false;
    for (size_t i = 0; i < r.landmarks_2d.size(); ++i) {
        if (r.landmarks_2d[i].x != p.landmarks_2d[i].x || r.landmarks_2d[i].y != p.landmarks_2d[i].y) return false;
        if (r.landmarks_2d_norm[i].x != p.landmarks_2d_norm[i].x || r.landmarks_2d_norm[i].y != p.landmarks_2d_norm[i].y) return false;
        if (r.landmarks_3d[i].x != p.landmarks_3d[i].x || r.landmarks_3d[i].y != p.landmarks_3d[i].y || r.landmarks_3d[i].z != p.landmarks_3d[i].z) return false;
        if ((int)r.landmarks_flag[i] != (int)p.landmarks_flag[i]) return false;
    }
    return true;
}
bool same_frame(const FisheyeFrameDescriptor_t& r, const omni::FisheyeFrameDescriptor& p) {
    if (r.image_num != p.image_num || r.msg_id != p.msg_id || r.drone_id != p.drone_id || r.landmark_num != p.landmark_num) return false;
    if (secs(r.timestamp) != p.timestamp || !same_pose(r.pose_drone, p.pose_drone) || r.images.size() != p.images.size()) return false;
    for (size_t i = 0; i < r.images.size(); ++i) if (!same_image(r.images[i], p.images[i])) return false;
    return true;
}

struct Outcome { std::vector<FisheyeFrameDescriptor_t> ref; std::vector<omni::FisheyeFrameDescriptor> prod; std::vector<float> ref_rate, prod_rate; };

Outcome run_schedule(const std::vector<ref_net::Published>& msgs, const std::vector<Delivery>& plan) {
    Outcome o;
    LoopNet rr("", false, false, 0.5);
    rr.frame_desc_callback = [&](const FisheyeFrameDescriptor_t& f) { o.ref.push_back(f); };
    rr.msg_recv_rate_callback = [&](const int, float rate) { o.ref_rate.push_back(rate); };
    omni::LoopNetWire wr(9);
    wr.recv_period = 0.5; wr.MIN_DIRECTION_LOOP = MIN_DIRECTION_LOOP;
    wr.frame_desc_callback = [&](const omni::FisheyeFrameDescriptor& f) { o.prod.push_back(f); };
    wr.msg_recv_rate_callback = [&](int, float rate) { o.prod_rate.push_back(rate); };
    for (const Delivery& d : plan) {
        ref_net::now_ref() = d.t;
        const ref_net::Published& p = msgs[d.idx];
        if (p.channel == "VIOKF_HEADER") rr.on_img_desc_header_recevied(nullptr, p.channel, &p.header);
        else rr.on_landmark_recevied(nullptr, p.channel, &p.landmark);
        const std::vector<uint8_t> b = to_product(p);
        CHECK(wr.on_packet(p.channel.c_str(), b.data(), b.size(), d.t), "product accepts the packet");
    }
    return o;
}

void compare(const char* name, const Outcome& o) {
    CHECK(o.ref.size() == o.prod.size(), name);
    for (size_t i = 0; i < o.ref.size() && i < o.prod.size(); ++i) CHECK(same_frame(o.ref[i], o.prod[i]), name);
    CHECK(o.ref_rate == o.prod_rate, name);
    size_t lm = 0;
    for (auto& f : o.ref) lm += (size_t)f.landmark_num;
    std::printf("RECV %s frames %zu landmarks %zu rates %zu\n", name, o.ref.size(), lm, o.ref_rate.size());
}

void check_recv(Rng& rng) {
    // three drones' key frames through the REFERENCE sender; a last small frame at the end flushes the time-outs in both receivers
    std::vector<ref_net::Published> msgs;
    {
        LoopNet rs("", false, false);
        ref_net::sink().clear();
        for (int k = 0; k < 8; ++k) {
            FisheyeFrameDescriptor_t fr; omni::FisheyeFrameDescriptor fp;
            make_frame(rng, 1 + k % 3, 100 + k, fr, fp);
            rs.broadcast_fisheye_desc(fr);
        }
        msgs = ref_net::sink();
    }
    const int n = (int)msgs.size();
    // image boundaries: a header opens an image
    std::vector<int> first;
    for (int i = 0; i < n; ++i) if (msgs[i].channel == "VIOKF_HEADER") first.push_back(i);
    first.push_back(n);
    const int n_img = (int)first.size() - 1;
    // the reference scans for time-outs only when a landmark of a not yet finished image arrives (loop_net.cpp:323): the last images of the
    // stream are held back and delivered later, one per call, so that the pending time-outs of the others fire
    int tails = 0;
    const int body_end_img = n_img - 4;
    auto flush_tail = [&](std::vector<Delivery>& plan, double t) {
        const int im = n_img - 1 - (tails++ % 4);
        for (int i = first[im]; i < first[im + 1]; ++i) plan.push_back({t + 0.001 * (i - first[im]), i});
    };
    {   // complete, in order, 1 ms apart
        std::vector<Delivery> plan; tails = 0;
        for (int i = 0; i < first[body_end_img]; ++i) plan.push_back({10.0 + 0.001 * i, i});
        flush_tail(plan, 20.0); flush_tail(plan, 21.0); flush_tail(plan, 23.0); flush_tail(plan, 26.0);
        compare("in-order", run_schedule(msgs, plan));
    }
    {   // every fifth landmark lost; images complete by recv_period, frames by 2 x recv_period
        std::vector<Delivery> plan; tails = 0;
        double t = 10.0;
        for (int i = 0; i < first[body_end_img]; ++i) {
            t += 0.004;
            if (msgs[i].channel == "VIOKF_LANDMARKS" && rng.below(5) == 0) continue;
            plan.push_back({t, i});
        }
        flush_tail(plan, t + 0.3); flush_tail(plan, t + 0.8); flush_tail(plan, t + 1.6); flush_tail(plan, t + 3.0);
        compare("landmarks-lost", run_schedule(msgs, plan));
    }
    {   // the headers of two images lost: their landmarks never become an image, in either implementation
        std::vector<Delivery> plan; tails = 0;
        double t = 10.0;
        for (int i = 0; i < first[body_end_img]; ++i) {
            t += 0.002;
            if (i == first[1] || i == first[5]) continue;
            plan.push_back({t, i});
        }
        flush_tail(plan, t + 0.7); flush_tail(plan, t + 2.0); flush_tail(plan, t + 15.0); flush_tail(plan, t + 30.0);
        compare("headers-lost", run_schedule(msgs, plan));
    }
    {   // landmark packets of every image shuffled among themselves (the header stays first), images interleaved two by two
        std::vector<Delivery> plan; tails = 0;
        double t = 10.0;
        for (int im = 0; im < body_end_img; im += 2) {
            std::vector<int> a, b;
            for (int i = first[im]; i < first[im + 1]; ++i) a.push_back(i);
            if (im + 1 < body_end_img) for (int i = first[im + 1]; i < first[im + 2]; ++i) b.push_back(i);
            for (std::vector<int>* v : {&a, &b})
                for (size_t k = v->size(); k > 2; --k) std::swap((*v)[k - 1], (*v)[1 + rng.below((int)k - 1)]);
            size_t ia = 0, ib = 0;
            while (ia < a.size() || ib < b.size()) {
                if (ia < a.size()) plan.push_back({t += 0.001, a[ia++]});
                if (ib < b.size()) plan.push_back({t += 0.001, b[ib++]});
            }
        }
        // whatever the pairing left out, then the flush
        std::set<int> seen;
        for (auto& d : plan) seen.insert(d.idx);
        for (int i = 0; i < first[body_end_img]; ++i) if (!seen.count(i)) plan.push_back({t += 0.001, i});
        flush_tail(plan, t + 5.0); flush_tail(plan, t + 7.0); flush_tail(plan, t + 9.0); flush_tail(plan, t + 12.0);
        compare("shuffled", run_schedule(msgs, plan));
    }
    {   // one landmark overtakes its header: the reference never delivers that image, LoopNetWire does; every other frame is the same
        std::vector<Delivery> plan; tails = 0;
        double t = 10.0;
        const int victim = 2;
        for (int i = 0; i < first[body_end_img]; ++i) {
            if (i == first[victim]) { plan.push_back({t += 0.001, i + 1}); plan.push_back({t += 0.001, i}); ++i; continue; }
            plan.push_back({t += 0.001, i});
        }
        flush_tail(plan, t + 5.0); flush_tail(plan, t + 7.0); flush_tail(plan, t + 9.0); flush_tail(plan, t + 12.0);
        const Outcome o = run_schedule(msgs, plan);
        const int64_t victim_id = msgs[first[victim]].header.msg_id;
        bool ref_has = false, prod_has = false;
        for (auto& f : o.ref) for (auto& im : f.images) if (im.msg_id == victim_id && im.landmark_num > 0) ref_has = true;
        for (auto& f : o.prod) for (auto& im : f.images) if (im.msg_id == victim_id && im.landmark_num > 0) prod_has = true;
        CHECK(!ref_has && prod_has, "overtaken header: reference drops the image, LoopNetWire keeps it");
        CHECK(o.prod.size() == o.ref.size() + 1, "overtaken header: exactly one frame more");
        size_t j = 0;
        for (size_t i = 0; i < o.prod.size(); ++i) {
            bool is_victim = false;
            for (auto& im : o.prod[i].images) if (im.msg_id == victim_id && im.landmark_num > 0) is_victim = true;
            if (is_victim) continue;
            CHECK(j < o.ref.size() && same_frame(o.ref[j], o.prod[i]), "overtaken header: the other frames");
            ++j;
        }
        std::printf("OVERTAKE reference frames %zu product frames %zu\n", o.ref.size(), o.prod.size());
    }
}

}  // namespace

int main(int argc, char** argv) {
    const uint64_t seed = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1;
    Rng rng(seed);
    check_send(rng, false);
    check_send(rng, true);
    check_recv(rng);
    std::printf("%s %d\n", g_fail ? "FAILED" : "OK", g_fail);
    return g_fail ? 1 : 0;
}
