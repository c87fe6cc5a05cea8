// Pins the stereo half of the key-frame front end -- omni::fill_stereo_landmarks / omni::geom::stereo_landmarks / triangulate_point
// (omni-swarm_amd/host/loop_geometry.hpp, geometry.hpp) -- to the TEXT of the reference: triangulatePoint (swarm_loop/src/loop_cam.cpp:73-106),
// LoopCam::match_HFNet_local_features (:141-175) and LoopCam::generate_stereo_image_descriptor (:341-523) are extracted at build time by
// oracle/Makefile (oracle/_ref/loopcam_*.inc: git-ignored) and compiled VERBATIM into this program against the stand-ins of
// oracle/ref_build/loopgeo_shim.h (ROS, cv::, swarm_msgs) and the small Eigen stand-in below (fixed-size matrices, the block / row / comma
// expressions those functions use; plain loops).  loop_defines.h and loop_params.cpp are the reference's own.
//
// Stand-ins, not pinned: the feature extractor (extractor_img_desc_deepnet: a hook that returns the prepared up / down descriptors -- key
// points, 64-d descriptors, zero landmarks, lifted points), camodocal's liftProjective (a pinhole here), cv::BFMatcher (the oracle's restatement,
// the same function the product side uses), JacobiSVD's V (the eigenvectors of A^T A from the product's Jacobi routine: the sign and the
// order of the last column are what the text uses), Swarm::Pose.
// Pinned: the construction of the two projection matrices and of the design matrix, the homogeneous division, the error measure and its
// normalisation, the acceptance test (err > TRIANGLE_THRES || z < 0), which key points get a landmark and a flag (both images), the
// ACCEPT_MIN_3D_PTS early return, the match bookkeeping (ids_up / ids_down), that the pixel is lifted AGAIN in double for the triangulation.
//
// stdin: "stereo" commands (see tests/test_geometry_cpu.py);  stdout: PROD / REF lines
#include "loopcam_common.h"

class LoopCam {
public:
    int self_id = 0;
    CameraConfig camera_configuration = CameraConfig::STEREO_FISHEYE;
    bool send_img = false, show = false;
    CameraPtr cam = nullptr;
    std::function<ImageDescriptor_t(const cv::Mat& img, bool superpoint_mode)> extractor;
    ImageDescriptor_t extractor_img_desc_deepnet(ros::Time, cv::Mat img, bool superpoint_mode) { return extractor(img, superpoint_mode); }
    void encode_image(const cv::Mat&, ImageDescriptor_t&) {}
    void match_HFNet_local_features(std::vector<cv::Point2f>& pts_up, std::vector<cv::Point2f>& pts_down, std::vector<float> _desc_up, std::vector<float> _desc_down,
                                    std::vector<int>& ids_up, std::vector<int>& ids_down);
    ImageDescriptor_t generate_stereo_image_descriptor(const StereoFrame& msg, cv::Mat& img, const int& vcam_id, cv::Mat& _show);
};

#include REF_LOOPGEO_PARAMS                      // loop_params.cpp: ACCEPT_MIN_3D_PTS, LOWER_CAM_AS_MAIN, ...
double TRIANGLE_THRES;                           // (defined next to its rosparam in swarm_loop.cpp, which is not part of this program)
#include REF_LOOPCAM_TRI                         // loop_cam.cpp:73-106   triangulatePoint
#include REF_LOOPCAM_MATCH                       // loop_cam.cpp:141-175  match_HFNet_local_features
#include REF_LOOPCAM_STEREO                      // loop_cam.cpp:341-523  generate_stereo_image_descriptor

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

int main() {
    cv::hooks().bf_match_l2_crosscheck = [](const cv::Mat& q, const cv::Mat& t, std::vector<cv::DMatch>& out) {
        std::vector<omni::DMatch> m;
        matcher(reinterpret_cast<const float*>(q.data), q.rows, reinterpret_cast<const float*>(t.data), t.rows, q.cols, m);
        out.clear();
        for (auto& x : m) out.push_back(cv::DMatch(x.queryIdx, x.trainIdx, x.distance));
    };
    std::string cmd;
    while (std::cin >> cmd) {
        if (cmd != "stereo") { std::fprintf(stderr, "unknown command %s\n", cmd.c_str()); return 2; }
        const og::Pose pd = read_pose(), eu = read_pose(), ed = read_pose();
        int nu, nd, accept_min; double thres; PinholeCam cam;
        std::cin >> nu >> nd >> thres >> accept_min >> cam.fx >> cam.fy >> cam.cx >> cam.cy;
        omni::ImageDescriptor up, down;
        auto read_img = [&](omni::ImageDescriptor& im, int n, const og::Pose& ext) {
            im.landmark_num = n; im.landmarks_2d.resize(n); im.feature_descriptor.resize((size_t)n * 64);
            for (int i = 0; i < n; ++i) std::cin >> im.landmarks_2d[i].x >> im.landmarks_2d[i].y;
            for (auto& v : im.feature_descriptor) std::cin >> v;
            im.pose_drone = omni::to_msg(pd); im.camera_extrinsic = omni::to_msg(ext);
            im.landmarks_2d_norm.resize(n);                           // the message field: float (extractor_img_desc_deepnet :558-566)
            for (int i = 0; i < n; ++i) im.landmarks_2d_norm[i] = {(float)((im.landmarks_2d[i].x - cam.cx) / cam.fx), (float)((im.landmarks_2d[i].y - cam.cy) / cam.fy)};
        };
        read_img(up, nu, eu); read_img(down, nd, ed);
        // ---- product: the matcher's list, then fill_stereo_landmarks with the pipeline's lift
        std::vector<omni::DMatch> m;
        matcher(up.feature_descriptor.data(), nu, down.feature_descriptor.data(), nd, 64, m);
        std::vector<int> iu, idn;
        for (auto& x : m) { iu.push_back(x.queryIdx); idn.push_back(x.trainIdx); }
        const std::function<og::Vec2(const omni::Point2f&)> lift = [&](const omni::Point2f& p) { return og::Vec2{(p.x - cam.cx) / cam.fx, (p.y - cam.cy) / cam.fy}; };
        omni::ImageDescriptor pu = up, pdn = down;
        const int pc = omni::fill_stereo_landmarks(pu, pdn, iu.data(), idn.data(), (int)m.size(), thres, accept_min, &lift);
        std::printf("PROD %d", pc);
        for (int i = 0; i < nu; ++i) std::printf(" %d %.9g %.9g %.9g", (int)pu.landmarks_flag[i], pu.landmarks_3d[i].x, pu.landmarks_3d[i].y, pu.landmarks_3d[i].z);
        std::printf(" |");
        for (int i = 0; i < nd; ++i) std::printf(" %d %.9g %.9g %.9g", (int)pdn.landmarks_flag[i], pdn.landmarks_3d[i].x, pdn.landmarks_3d[i].y, pdn.landmarks_3d[i].z);
        std::printf("\n");
        // ---- reference text
        ACCEPT_MIN_3D_PTS = accept_min; TRIANGLE_THRES = thres; LOWER_CAM_AS_MAIN = false;
        auto to_desc = [&](const omni::ImageDescriptor& im) {       // what extractor_img_desc_deepnet returns (:525-585): key points, descriptors, zero landmarks
            ImageDescriptor_t d;
            d.landmark_num = im.landmark_num; d.feature_descriptor = im.feature_descriptor;
            for (int i = 0; i < im.landmark_num; ++i) {
                Point2d_t p; p.x = im.landmarks_2d[i].x; p.y = im.landmarks_2d[i].y; d.landmarks_2d.push_back(p);
                Point2d_t q; q.x = im.landmarks_2d_norm[i].x; q.y = im.landmarks_2d_norm[i].y; d.landmarks_2d_norm.push_back(q);
                d.landmarks_3d.push_back(Point3d_t()); d.landmarks_flag.push_back(0);
            }
            return d;
        };
        LoopCam lc;
        lc.cam = &cam; lc.self_id = 1;
        StereoFrame msg;
        msg.stamp = ros::Time(12.5); msg.keyframe_id = 77; msg.pose_drone = to_ros(pd);
        cv::Mat tag_up(1, 1, CV_32F), tag_down(2, 1, CV_32F);          // the "images": told apart by their row count
        msg.left_images.push_back(tag_up); msg.right_images.push_back(tag_down);
        msg.left_extrisincs.push_back(to_ros(eu)); msg.right_extrisincs.push_back(to_ros(ed));
        ImageDescriptor_t ref_down;
        lc.extractor = [&](const cv::Mat& img, bool) { return img.rows == 1 ? to_desc(up) : to_desc(down); };
        cv::Mat img, show;
        const ImageDescriptor_t ru = lc.generate_stereo_image_descriptor(msg, img, 0, show);
        LOWER_CAM_AS_MAIN = true;                                    // the same call again hands back the DOWN descriptor (:517-521)
        const ImageDescriptor_t rd = lc.generate_stereo_image_descriptor(msg, img, 0, show);
        int rc = 0;
        for (auto f : ru.landmarks_flag) rc += f != 0;
        std::printf("\nREF %d", rc);                                 // (the reference text prints its own progress without a line end)
        for (int i = 0; i < nu; ++i) std::printf(" %d %.9g %.9g %.9g", (int)ru.landmarks_flag[i], ru.landmarks_3d[i].x, ru.landmarks_3d[i].y, ru.landmarks_3d[i].z);
        std::printf(" |");
        // (below ACCEPT_MIN_3D_PTS the function returns the UP descriptor whatever LOWER_CAM_AS_MAIN says, :385-389: print what came back)
        const bool early = nu <= accept_min;
        for (size_t i = 0; !early && i < rd.landmarks_flag.size(); ++i) std::printf(" %d %.9g %.9g %.9g", (int)rd.landmarks_flag[i], rd.landmarks_3d[i].x, rd.landmarks_3d[i].y, rd.landmarks_3d[i].z);
        for (int i = 0; early && i < nd; ++i) std::printf(" 0 0 0 0");
        std::printf("\n");
        std::printf("META %d %lld %d %.9f\n", ru.drone_id, (long long)ru.frame_id, (int)(ru.camera_extrinsic.position[0] == eu.pos.x), ru.timestamp.sec + 1e-9 * ru.timestamp.nsec);
    }
    return 0;
}
