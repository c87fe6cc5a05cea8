// Pins the MESSAGES the key-frame front end hands to the detector -- omni::fill_image_descriptor / stamp_image_descriptor /
// finish_frame_descriptor (omni-swarm_amd/host/loop_geometry.hpp: what KeyframePipeline::finish() builds from the cam unit's result block) and
// the fisheye mask rows (omni_fisheye_mask_rows, include/omni_hip.h: what every kernel that reads the gray image blanks) -- to the TEXT of the
// reference: LoopCam::on_flattened_images (swarm_loop/src/loop_cam.cpp:178-229), generate_stereo_image_descriptor (:341-523),
// extractor_img_desc_deepnet (:525-634, the USE_TENSORRT branch), triangulatePoint and match_HFNet_local_features are extracted at build time by
// oracle/Makefile (oracle/_ref/loopcam_*.inc: git-ignored) and compiled VERBATIM into this program against the stand-ins of
// oracle/ref_build/loopgeo_shim.h and tests/cpp/loopcam_common.h.  The whole chain runs: on_flattened_images -> generate_stereo_image_descriptor
// -> extractor_img_desc_deepnet -> the two networks.
//
// Stand-ins, not pinned: the two networks (hooks that return the prepared key points / descriptors / global descriptor of the image they are
// given -- and record the image AS THEY SEE IT), camodocal's liftProjective (a pinhole), cv::BFMatcher (the oracle's restatement), JacobiSVD,
// Swarm::Pose, CVPoints2LCM / toLCMTime / fromROSPose (element-wise converters of the un-vendored swarm_msgs).
// Pinned: which rows of which images are blanked, and that BOTH networks see the blanked image (the cv::Mat passed by value shares its pixels);
// which image gets a global descriptor (the main camera's only); the float lifting of the key points; the zero landmarks / flags the
// triangulation starts from; every stamp of the per-image messages (time, drone id, extrinsic, pose, frame id) and of the frame (time, image
// count, msg id, pose, landmark total, drone id, directions).
//
// stdin: "frame" commands (see tests/test_geometry_cpu.py);  stdout: PROD / REF lines
#include <fstream>

#include "../../include/omni_hip.h"
#include "loopcam_common.h"

struct TicToc { double toc() const { return 0.0; } };
namespace swarm_msgs {
inline void CVPoints2LCM(const std::vector<cv::Point2f>& in, std::vector<Point2d_t>& out) {
    out.clear();
    for (auto& p : in) { Point2d_t q; q.x = p.x; q.y = p.y; out.push_back(q); }
}
}  // namespace swarm_msgs
struct SuperPointNet {
    std::function<void(const cv::Mat&, std::vector<cv::Point2f>&, std::vector<float>&)> fn;
    void inference(const cv::Mat& img, std::vector<cv::Point2f>& kps, std::vector<float>& desc) { fn(img, kps, desc); }
};
struct NetVLADNet {
    std::function<std::vector<float>(const cv::Mat&)> fn;
    std::vector<float> inference(const cv::Mat& img) { return fn(img); }
};

class LoopCam {
public:
    int self_id = 0, kf_count = 0;
    CameraConfig camera_configuration = CameraConfig::STEREO_FISHEYE;
    bool send_img = false, show = false;
    CameraPtr cam = nullptr;
    SuperPointNet superpoint_net;
    NetVLADNet netvlad_net;
    std::ofstream fsp;
    void encode_image(const cv::Mat&, ImageDescriptor_t&) {}
    void match_HFNet_local_features(std::vector<cv::Point2f>& pts_up, std::vector<cv::Point2f>& pts_down, std::vector<float> _desc_up, std::vector<float> _desc_down,
                                    std::vector<int>& ids_up, std::vector<int>& ids_down);
    ImageDescriptor_t extractor_img_desc_deepnet(ros::Time stamp, cv::Mat img, bool superpoint_mode);
    ImageDescriptor_t generate_stereo_image_descriptor(const StereoFrame& msg, cv::Mat& img, const int& vcam_id, cv::Mat& _show);
    ImageDescriptor_t generate_gray_depth_image_descriptor(const StereoFrame& msg, cv::Mat& img, const int& vcam_id, cv::Mat& _show);
    FisheyeFrameDescriptor_t on_flattened_images(const StereoFrame& msg, std::vector<cv::Mat>& imgs);
};

#define USE_TENSORRT
#include REF_LOOPGEO_PARAMS                      // loop_params.cpp: ACCEPT_MIN_3D_PTS, LOWER_CAM_AS_MAIN, OUTPUT_RAW_SUPERPOINT_DESC, ...
double TRIANGLE_THRES;                           // (defined next to its rosparam in swarm_loop.cpp, which is not part of this program)
#include REF_LOOPCAM_TRI                         // loop_cam.cpp:73-106    triangulatePoint
#include REF_LOOPCAM_MATCH                       // loop_cam.cpp:141-175   match_HFNet_local_features
#include REF_LOOPCAM_FRAME                       // loop_cam.cpp:178-229   on_flattened_images
#include REF_LOOPCAM_DEPTH                       // loop_cam.cpp:231-339   generate_gray_depth_image_descriptor (CameraConfig::PINHOLE_DEPTH)
#include REF_LOOPCAM_STEREO                      // loop_cam.cpp:341-523   generate_stereo_image_descriptor
#include REF_LOOPCAM_EXTRACT                     // loop_cam.cpp:525-634   extractor_img_desc_deepnet

// ---------------------------------------------------------------------------------------------------------------- the harness
namespace og = omni::geom;

static og::Pose read_pose() { og::Pose p; std::cin >> p.pos.x >> p.pos.y >> p.pos.z >> p.att.w >> p.att.x >> p.att.y >> p.att.z; return p; }
static geometry_msgs::Pose to_ros(const og::Pose& p) {
    geometry_msgs::Pose o;
    o.position.x = p.pos.x; o.position.y = p.pos.y; o.position.z = p.pos.z;
    o.orientation.w = p.att.w; o.orientation.x = p.att.x; o.orientation.y = p.att.y; o.orientation.z = p.att.z;
    return o;
}
static void matcher(const float* q, int nq, const float* t, int nt, int dim, std::vector<omni::DMatch>& out) {
    out.clear();
    if (nq <= 0 || nt <= 0) return;
    std::vector<int> qi(nq), ti(nq);
    std::vector<float> dd(nq);
    const int n = oracle_bf_match(q, nq, t, nt, dim, 0, qi.data(), ti.data(), dd.data());
    for (int i = 0; i < n; ++i) out.push_back({qi[i], ti[i], dd[i]});
}
static uint64_t checksum(const unsigned char* p, int H, int W, size_t step) {      // order-sensitive (FNV-1a over the pixels, row by row)
    uint64_t h = 1469598103934665603ull;
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) { h ^= p[(size_t)y * step + x]; h *= 1099511628211ull; }
    return h;
}
struct Net { std::vector<omni::Point2f> kps; std::vector<float> desc, gdesc; };     // what the networks answer for one image

template <typename Img> static void print_image(const char* tag, int idx, const Img& im, int dir, int drone, long long frame_id, double ts, double ext_x, double ext_qw,
                                                double pose_x, size_t n_gdesc, double gsum, double dsum) {
    std::printf("%s IMG %d dir %d drone %d frame %lld ts %.9f ext %.17g %.17g pose %.17g n %d sizes %zu %zu %zu %zu %zu gd %zu %.9g ds %.9g :", tag, idx, dir, drone, frame_id, ts, ext_x,
                ext_qw, pose_x, (int)im.landmark_num, im.landmarks_2d.size(), im.landmarks_2d_norm.size(), im.landmarks_3d.size(), im.landmarks_flag.size(),
                im.feature_descriptor.size(), n_gdesc, gsum, dsum);
    for (size_t i = 0; i < im.landmarks_2d.size(); ++i)
        std::printf(" %.9g %.9g %.9g %.9g %d %.9g %.9g %.9g", im.landmarks_2d[i].x, im.landmarks_2d[i].y, im.landmarks_2d_norm[i].x, im.landmarks_2d_norm[i].y,
                    (int)im.landmarks_flag[i], im.landmarks_3d[i].x, im.landmarks_3d[i].y, im.landmarks_3d[i].z);
    std::printf("\n");
}
static double fsum(const std::vector<float>& v) { double s = 0; for (size_t i = 0; i < v.size(); ++i) s += (double)v[i] * (double)(1 + i % 7); return s; }

int main() {
    cv::hooks().bf_match_l2_crosscheck = [](const cv::Mat& q, const cv::Mat& t, std::vector<cv::DMatch>& out) {
        std::vector<omni::DMatch> m;
        matcher(reinterpret_cast<const float*>(q.data), q.rows, reinterpret_cast<const float*>(t.data), t.rows, q.cols, m);
        out.clear();
        for (auto& x : m) out.push_back(cv::DMatch(x.queryIdx, x.trainIdx, x.distance));
    };
    std::string cmd;
    while (std::cin >> cmd) {
        if (cmd == "depthframe") {
            // One PINHOLE_DEPTH key frame (the camera mode of launch/realsense.launch): ONE gray image + its depth image.  Product: fill_image_descriptor,
            // stamp_image_descriptor, fill_depth_landmarks, finish_frame_descriptor; reference: on_flattened_images -> generate_gray_depth_image_descriptor
            // -> extractor_img_desc_deepnet (no rows are blanked in this mode: loop_cam.cpp:536 only masks STEREO_FISHEYE).
            int H, W, gdim, accept_min, self_id; long long kf_id; double stamp, dnear, dfar; PinholeCam cam; unsigned seed;
            std::cin >> H >> W >> gdim >> accept_min >> dnear >> dfar >> self_id >> kf_id >> stamp >> cam.fx >> cam.fy >> cam.cx >> cam.cy >> seed;
            const og::Pose pd = read_pose(), ext = read_pose();
            Net nt;
            int n; std::cin >> n;
            nt.kps.resize(n); nt.desc.resize((size_t)n * 64);
            for (auto& k : nt.kps) std::cin >> k.x >> k.y;
            for (auto& v : nt.desc) std::cin >> v;
            nt.gdesc.resize(gdim);
            for (auto& v : nt.gdesc) std::cin >> v;
            std::vector<unsigned char> pix((size_t)H * W);
            for (auto& v : pix) { seed = seed * 1664525u + 1013904223u; v = (unsigned char)(1 + (seed >> 24) % 255); }
            // depth in millimetres: a quarter of the pixels 0 (no return), the rest spread from 0.1 m to 14 m -- both thresholds are crossed; the key points'
            // own pixels get values right at, just below and just above the two thresholds in turn
            std::vector<unsigned short> dep((size_t)H * W);
            for (auto& v : dep) { seed = seed * 1664525u + 1013904223u; v = (seed >> 30) == 0 ? 0 : (unsigned short)(100 + (seed >> 8) % 13900); }
            const unsigned short edge[6] = {(unsigned short)std::lrint(dnear * 1000), (unsigned short)(std::lrint(dnear * 1000) + 1), (unsigned short)(std::lrint(dnear * 1000) - 1),
                                            (unsigned short)std::lrint(dfar * 1000), (unsigned short)(std::lrint(dfar * 1000) - 1), (unsigned short)(std::lrint(dfar * 1000) + 1)};
            for (size_t i = 0; i < nt.kps.size(); i += 3) {
                const long px = std::lrint((double)nt.kps[i].x), py = std::lrint((double)nt.kps[i].y);
                if (px >= 0 && px < W && py >= 0 && py < H) dep[(size_t)py * W + px] = edge[(i / 3) % 6];
            }
            const std::function<og::Vec2(const omni::Point2f&)> lift = [&](const omni::Point2f& p) { return og::Vec2{((double)p.x - cam.cx) / cam.fx, ((double)p.y - cam.cy) / cam.fy}; };
            omni::FisheyeFrameDescriptor pf;
            pf.images.resize(1);
            {
                omni::ImageDescriptor& im = pf.images[0];
                std::vector<float> kx;
                for (auto& k : nt.kps) { kx.push_back(k.x); kx.push_back(k.y); }
                omni::fill_image_descriptor(im, kx.data(), (int)nt.kps.size(), nt.desc.data(), 64, nt.gdesc.data(), gdim, lift);
                omni::stamp_image_descriptor(im, stamp, self_id, omni::to_msg(ext), omni::to_msg(pd), kf_id);
                const int c3 = omni::fill_depth_landmarks(im, dep.data(), W, W, H, dnear, dfar, accept_min, lift);
                std::printf("PROD COUNT3D %d\n", c3);
            }
            omni::finish_frame_descriptor(pf, stamp, kf_id, omni::to_msg(pd), self_id);
            std::printf("PROD FRAME ts %.9f image_num %d msg_id %lld pose %.17g %.17g landmark_num %d drone %d\n", pf.timestamp, pf.image_num, (long long)pf.msg_id, pf.pose_drone.position[0],
                        pf.pose_drone.quat_wxyz[0], pf.landmark_num, pf.drone_id);
            {
                const omni::ImageDescriptor& im = pf.images[0];
                print_image("PROD", 0, im, im.direction, im.drone_id, (long long)im.frame_id, im.timestamp, im.camera_extrinsic.position[0], im.camera_extrinsic.quat_wxyz[0],
                            im.pose_drone.position[0], im.image_desc.size(), fsum(im.image_desc), fsum(im.feature_descriptor));
            }
            std::printf("PROD PIX %llu\n", (unsigned long long)checksum(pix.data(), H, W, (size_t)W));       // no rows blanked in this mode
            ACCEPT_MIN_3D_PTS = accept_min; LOWER_CAM_AS_MAIN = false; OUTPUT_RAW_SUPERPOINT_DESC = false; DEPTH_NEAR_THRES = dnear; DEPTH_FAR_THRES = dfar;
            LoopCam lc;
            lc.cam = &cam; lc.self_id = self_id; lc.camera_configuration = CameraConfig::PINHOLE_DEPTH;
            StereoFrame msg;
            msg.stamp = ros::Time(stamp); msg.keyframe_id = kf_id; msg.pose_drone = to_ros(pd);
            msg.left_images.push_back(cv::Mat(H, W, CV_8U, pix.data()));
            msg.depth_images.push_back(cv::Mat(H, W, CV_16U, dep.data()));
            msg.left_extrisincs.push_back(to_ros(ext));
            int sp_calls = 0, vlad_calls = 0;
            uint64_t seen_sp = 0, seen_vlad = 0;
            lc.superpoint_net.fn = [&](const cv::Mat& img, std::vector<cv::Point2f>& kps, std::vector<float>& desc) {
                ++sp_calls; seen_sp = checksum(img.data, img.rows, img.cols, img.step);
                kps.clear();
                for (auto& k : nt.kps) kps.push_back(cv::Point2f(k.x, k.y));
                desc = nt.desc;
            };
            lc.netvlad_net.fn = [&](const cv::Mat& img) { ++vlad_calls; seen_vlad = checksum(img.data, img.rows, img.cols, img.step); return nt.gdesc; };
            std::vector<cv::Mat> imgs;
            const FisheyeFrameDescriptor_t rf = lc.on_flattened_images(msg, imgs);
            int rc3 = 0;
            for (auto fl : rf.images[0].landmarks_flag) rc3 += fl ? 1 : 0;
            std::printf("REF COUNT3D %d\n", rc3);
            std::printf("REF FRAME ts %.9f image_num %d msg_id %lld pose %.17g %.17g landmark_num %d drone %d\n", rf.timestamp.sec + 1e-9 * rf.timestamp.nsec, (int)rf.image_num,
                        (long long)rf.msg_id, rf.pose_drone.position[0], rf.pose_drone.orientation[0], (int)rf.landmark_num, (int)rf.drone_id);
            {
                const ImageDescriptor_t& im = rf.images[0];
                print_image("REF", 0, im, im.direction, im.drone_id, (long long)im.frame_id, im.timestamp.sec + 1e-9 * im.timestamp.nsec, im.camera_extrinsic.position[0],
                            im.camera_extrinsic.orientation[0], im.pose_drone.position[0], im.image_desc.size(), fsum(im.image_desc), fsum(im.feature_descriptor));
                if (im.image_desc_size != (int)im.image_desc.size() || im.feature_descriptor_size != (int)im.feature_descriptor.size() || im.image_size != 0) std::printf("REF SIZES-DIFFER 0\n");
            }
            std::printf("REF PIX %llu\n", (unsigned long long)checksum(msg.left_images[0].data, H, W, msg.left_images[0].step));
            std::printf("REF CALLS %d %d SEEN %llu %llu\n", sp_calls, vlad_calls, (unsigned long long)seen_sp, (unsigned long long)seen_vlad);
            continue;
        }
        if (cmd != "frame") { std::fprintf(stderr, "unknown command %s\n", cmd.c_str()); return 2; }
        int H, W, n_dirs, gdim, accept_min, self_id; long long kf_id; double stamp, thres; PinholeCam cam; unsigned seed;
        std::cin >> H >> W >> n_dirs >> gdim >> accept_min >> thres >> self_id >> kf_id >> stamp >> cam.fx >> cam.fy >> cam.cx >> cam.cy >> seed;
        const og::Pose pd = read_pose();
        std::vector<og::Pose> eu(n_dirs), ed(n_dirs);
        std::vector<Net> up(n_dirs), down(n_dirs);
        for (int d = 0; d < n_dirs; ++d) {
            eu[d] = read_pose(); ed[d] = read_pose();
            for (Net* nt : {&up[d], &down[d]}) {
                int n; std::cin >> n;
                nt->kps.resize(n); nt->desc.resize((size_t)n * 64);
                for (auto& k : nt->kps) std::cin >> k.x >> k.y;
                for (auto& v : nt->desc) std::cin >> v;
            }
            up[d].gdesc.resize(gdim);
            for (auto& v : up[d].gdesc) std::cin >> v;
        }
        // the gray images (random pixels, never 0 so that a blanked pixel is told apart): [up | down] per direction
        std::vector<std::vector<unsigned char>> pix(2 * n_dirs, std::vector<unsigned char>((size_t)H * W));
        for (auto& p : pix) for (auto& v : p) { seed = seed * 1664525u + 1013904223u; v = (unsigned char)(1 + (seed >> 24) % 255); }

        // ---- product: what KeyframePipeline::finish() does with the cam unit's result block, and what the kernels blank ---------------------------
        const std::function<og::Vec2(const omni::Point2f&)> lift = [&](const omni::Point2f& p) { return og::Vec2{((double)p.x - cam.cx) / cam.fx, ((double)p.y - cam.cy) / cam.fy}; };
        omni::FisheyeFrameDescriptor pf;
        std::vector<omni::ImageDescriptor> pdown(n_dirs);
        pf.images.resize(n_dirs);
        for (int d = 0; d < n_dirs; ++d) {
            omni::ImageDescriptor& im = pf.images[d];
            std::vector<float> kx;
            auto flat = [&](const Net& nt) { kx.clear(); for (auto& k : nt.kps) { kx.push_back(k.x); kx.push_back(k.y); } return kx.data(); };
            omni::fill_image_descriptor(im, flat(up[d]), (int)up[d].kps.size(), up[d].desc.data(), 64, up[d].gdesc.data(), gdim, lift);
            omni::stamp_image_descriptor(im, stamp, self_id, omni::to_msg(eu[d]), omni::to_msg(pd), kf_id);
            omni::fill_image_descriptor(pdown[d], flat(down[d]), (int)down[d].kps.size(), nullptr, 64, nullptr, 0, lift);
            omni::stamp_image_descriptor(pdown[d], stamp, self_id, omni::to_msg(ed[d]), omni::to_msg(pd), kf_id);
            std::vector<omni::DMatch> m;                              // (the cam unit's up <-> down match list)
            matcher(up[d].desc.data(), (int)up[d].kps.size(), down[d].desc.data(), (int)down[d].kps.size(), 64, m);
            std::vector<int> iu, idn;
            for (auto& x : m) { iu.push_back(x.queryIdx); idn.push_back(x.trainIdx); }
            omni::fill_stereo_landmarks(im, pdown[d], iu.data(), idn.data(), (int)m.size(), thres, accept_min, &lift);
        }
        omni::finish_frame_descriptor(pf, stamp, kf_id, omni::to_msg(pd), self_id);
        std::printf("PROD FRAME ts %.9f image_num %d msg_id %lld pose %.17g %.17g landmark_num %d drone %d\n", pf.timestamp, pf.image_num, (long long)pf.msg_id, pf.pose_drone.position[0],
                    pf.pose_drone.quat_wxyz[0], pf.landmark_num, pf.drone_id);
        for (int d = 0; d < n_dirs; ++d) {
            const omni::ImageDescriptor& im = pf.images[d];
            print_image("PROD", d, im, im.direction, im.drone_id, (long long)im.frame_id, im.timestamp, im.camera_extrinsic.position[0], im.camera_extrinsic.quat_wxyz[0],
                        im.pose_drone.position[0], im.image_desc.size(), fsum(im.image_desc), fsum(im.feature_descriptor));
        }
        int r0, r1;
        omni_fisheye_mask_rows(H, 1, &r0, &r1);
        std::printf("PROD PIX");
        for (auto p : pix) {                                          // (a copy) what every kernel reads: rows [r0, r1) as zeros
            for (int y = r0; y < r1; ++y) std::memset(p.data() + (size_t)y * W, 0, (size_t)W);
            std::printf(" %llu", (unsigned long long)checksum(p.data(), H, W, (size_t)W));
        }
        std::printf("\n");

        // ---- reference text -------------------------------------------------------------------------------------------------------------------------
        ACCEPT_MIN_3D_PTS = accept_min; TRIANGLE_THRES = thres; LOWER_CAM_AS_MAIN = false; OUTPUT_RAW_SUPERPOINT_DESC = false;
        LoopCam lc;
        lc.cam = &cam; lc.self_id = self_id;
        StereoFrame msg;
        msg.stamp = ros::Time(stamp); msg.keyframe_id = kf_id; msg.pose_drone = to_ros(pd);
        for (int d = 0; d < n_dirs; ++d) {
            msg.left_images.push_back(cv::Mat(H, W, CV_8U, pix[2 * d].data()));
            msg.right_images.push_back(cv::Mat(H, W, CV_8U, pix[2 * d + 1].data()));
            msg.left_extrisincs.push_back(to_ros(eu[d])); msg.right_extrisincs.push_back(to_ros(ed[d]));
        }
        auto which = [&](const cv::Mat& img) {                       // the image a network was handed, by its pixel buffer (cv::Mat copies share it)
            for (int d = 0; d < n_dirs; ++d) {
                if (img.data == msg.left_images[d].data) return 2 * d;
                if (img.data == msg.right_images[d].data) return 2 * d + 1;
            }
            std::fprintf(stderr, "a network was handed an unknown image\n"); std::abort();
        };
        std::vector<uint64_t> seen_sp(2 * n_dirs, 0), seen_vlad(2 * n_dirs, 0);
        lc.superpoint_net.fn = [&](const cv::Mat& img, std::vector<cv::Point2f>& kps, std::vector<float>& desc) {
            const int w = which(img);
            seen_sp[w] = checksum(img.data, img.rows, img.cols, img.step);
            const Net& nt = (w & 1) ? down[w / 2] : up[w / 2];
            kps.clear();
            for (auto& k : nt.kps) kps.push_back(cv::Point2f(k.x, k.y));
            desc = nt.desc;
        };
        lc.netvlad_net.fn = [&](const cv::Mat& img) {
            const int w = which(img);
            seen_vlad[w] = checksum(img.data, img.rows, img.cols, img.step);
            return (w & 1) ? std::vector<float>(gdim, -1.f) : up[w / 2].gdesc;     // (the down camera must never get here: its answer would show)
        };
        std::vector<cv::Mat> imgs;
        const FisheyeFrameDescriptor_t rf = lc.on_flattened_images(msg, imgs);
        std::printf("\nREF FRAME ts %.9f image_num %d msg_id %lld pose %.17g %.17g landmark_num %d drone %d\n", rf.timestamp.sec + 1e-9 * rf.timestamp.nsec, (int)rf.image_num,
                    (long long)rf.msg_id, rf.pose_drone.position[0], rf.pose_drone.orientation[0], (int)rf.landmark_num, (int)rf.drone_id);
        for (size_t d = 0; d < rf.images.size(); ++d) {
            const ImageDescriptor_t& im = rf.images[d];
            print_image("REF", (int)d, im, im.direction, im.drone_id, (long long)im.frame_id, im.timestamp.sec + 1e-9 * im.timestamp.nsec, im.camera_extrinsic.position[0],
                        im.camera_extrinsic.orientation[0], im.pose_drone.position[0], im.image_desc.size(), fsum(im.image_desc), fsum(im.feature_descriptor));
            if (im.image_desc_size != (int)im.image_desc.size() || im.feature_descriptor_size != (int)im.feature_descriptor.size() || im.image_size != 0) std::printf("REF SIZES-DIFFER %zu\n", d);
        }
        std::printf("REF PIX");
        for (int i = 0; i < 2 * n_dirs; ++i) {
            const cv::Mat& m = (i & 1) ? msg.right_images[i / 2] : msg.left_images[i / 2];
            std::printf(" %llu", (unsigned long long)checksum(m.data, m.rows, m.cols, m.step));
        }
        std::printf("\nREF SEEN-SP");
        for (auto v : seen_sp) std::printf(" %llu", (unsigned long long)v);
        std::printf("\nREF SEEN-VLAD");
        for (auto v : seen_vlad) std::printf(" %llu", (unsigned long long)v);
        std::printf("\nREF KF-COUNT %d\n", lc.kf_count);
    }
    return 0;
}
