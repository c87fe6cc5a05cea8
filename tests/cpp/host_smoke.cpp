// Exercises the C++ host adapters (omni-swarm_amd/host/omni_swarm.hpp) exactly the way LoopCam / LoopDetector call the
// reference classes; tests/test_gpu_cpp_host.py builds it with g++, runs it on the GPU box and compares the printed
// results with the oracle.  Usage:
//   host_smoke sp.omnw comp.csv mean.csv vlad.omnw image.u8 W H db.f32 N query.f32 descA.f32 nA descB.f32 nB stream.bin
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "../../omni-swarm_amd/host/omni_swarm.hpp"

template <typename T>
static std::vector<T> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
    std::vector<T> v((size_t)f.tellg() / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), v.size() * sizeof(T));
    return v;
}

int main(int argc, char** argv) {
    if (argc < 16) { std::fprintf(stderr, "usage: see source\n"); return 2; }
    const int W = std::atoi(argv[6]), H = std::atoi(argv[7]);
    omni::Context ctx(0);
    // --- LoopCam::extractor_img_desc_deepnet (loop_cam.cpp:541-556)
    Swarm::SuperPointHIP superpoint_net(ctx, argv[1], argv[2], argv[3], W, H, 0.015f, 200, false, OMNI_PREC_F32);
    Swarm::MobileNetVLADHIP netvlad_net(ctx, argv[4], W, H);
    auto img = slurp<uint8_t>(argv[5]);
    std::vector<omni::Point2f> features;
    std::vector<float> feature_descriptor;
    superpoint_net.inference(img.data(), W, features, feature_descriptor);
    std::printf("SP_N %zu %d\n", features.size(), superpoint_net.desc_dim());
    std::printf("SP_KPS");
    for (auto& p : features) std::printf(" %d %d", (int)p.x, (int)p.y);
    std::printf("\nSP_DESC");
    for (float v : feature_descriptor) std::printf(" %.9g", v);
    auto image_desc = netvlad_net.inference(img.data(), W);
    std::printf("\nVLAD");
    for (float v : image_desc) std::printf(" %.9g", v);
    // --- LoopDetector::add_to_database / query_from_database (loop_detector.cpp:166,213)
    auto db = slurp<float>(argv[8]);
    const int N = std::atoi(argv[9]);
    auto q = slurp<float>(argv[10]);
    omni::IndexFlatIP local_index(ctx, 4096);
    for (int i = 0; i < N; ++i) local_index.add(1, db.data() + (size_t)i * 4096);
    float distances[1000] = {0};
    omni::IndexFlatIP::idx_t labels[1000];
    const int search_num = 5 + 10;
    local_index.search(1, q.data(), search_num, distances, labels);
    std::printf("\nIP_NTOTAL %lld\nIP_I", (long long)local_index.ntotal);
    for (int i = 0; i < search_num; ++i) std::printf(" %lld", (long long)labels[i]);
    std::printf("\nIP_D");
    for (int i = 0; i < search_num; ++i) std::printf(" %.9g", distances[i]);
    // --- cv::BFMatcher(NORM_L2, true).match (loop_cam.cpp:147-150)
    auto a = slurp<float>(argv[11]);
    auto b = slurp<float>(argv[13]);
    omni::BFMatcherL2X bfmatcher(ctx);
    std::vector<omni::DMatch> matches;
    bfmatcher.match(a.data(), std::atoi(argv[12]), b.data(), std::atoi(argv[14]), 64, matches);
    std::printf("\nBF");
    for (auto& m : matches) std::printf(" %d %d %.9g", m.queryIdx, m.trainIdx, m.distance);
    // --- LoopDetector::on_image_recv over a descriptor stream
    auto s = slurp<float>(argv[15]);      // per frame: msg_id drone_id landmark_num prevent | 4 x (landmark_num, 4096 floats)
    omni::LoopDetectorCore det(ctx, 1);
    det.INNER_PRODUCT_THRES = 0.6; det.INIT_MODE_PRODUCT_THRES = 0.3; det.MATCH_INDEX_DIST = 5; det.MIN_LOOP_NUM = 30;
    det.MIN_DIRECTION_LOOP = 3; det.inter_drone_init_frames = 3;
    det.compute_loop = [](const omni::FisheyeFrameDescriptor& n, const omni::FisheyeFrameDescriptor& o, int, int, bool) {
        return (n.msg_id + o.msg_id) % 3 != 0;
    };
    const size_t per = 4 + 4 * (1 + 4096);
    std::printf("\nDET");
    for (size_t off = 0; off + per <= s.size(); off += per) {
        omni::FisheyeFrameDescriptor f;
        f.msg_id = (int64_t)s[off]; f.drone_id = (int)s[off + 1]; f.landmark_num = (int)s[off + 2]; f.prevent_adding_db = s[off + 3] != 0;
        for (int d = 0; d < 4; ++d) {
            omni::ImageDescriptor im;
            const float* p = s.data() + off + 4 + d * 4097;
            im.drone_id = f.drone_id; im.landmark_num = (int)p[0];
            im.image_desc.assign(p + 1, p + 1 + 4096);
            f.images.push_back(std::move(im));
        }
        auto r = det.on_image_recv(f);
        std::printf(" %lld %d %d %d %lld %d %d", (long long)f.msg_id, (int)r.added, (int)r.queried, r.image_id, (long long)r.old_msg_id,
                    r.direction_old, (int)r.loop);
    }
    // --- the same stream through on_images_recv_batch, 5 frames per call (one host synchronisation each): identical decisions
    {
        omni::LoopDetectorCore detb(ctx, 1);
        detb.INNER_PRODUCT_THRES = 0.6; detb.INIT_MODE_PRODUCT_THRES = 0.3; detb.MATCH_INDEX_DIST = 5; detb.MIN_LOOP_NUM = 30;
        detb.MIN_DIRECTION_LOOP = 3; detb.inter_drone_init_frames = 3;
        detb.compute_loop = det.compute_loop;
        std::vector<omni::FisheyeFrameDescriptor> chunk;
        std::printf("\nDETB");
        auto flush = [&]() {
            auto rs = detb.on_images_recv_batch(chunk);
            for (size_t i = 0; i < rs.size(); ++i)
                std::printf(" %lld %d %d %d %lld %d %d", (long long)chunk[i].msg_id, (int)rs[i].added, (int)rs[i].queried, rs[i].image_id,
                            (long long)rs[i].old_msg_id, rs[i].direction_old, (int)rs[i].loop);
            chunk.clear();
        };
        for (size_t off = 0; off + per <= s.size(); off += per) {
            omni::FisheyeFrameDescriptor f;
            f.msg_id = (int64_t)s[off]; f.drone_id = (int)s[off + 1]; f.landmark_num = (int)s[off + 2]; f.prevent_adding_db = s[off + 3] != 0;
            for (int d = 0; d < 4; ++d) {
                omni::ImageDescriptor im;
                const float* p = s.data() + off + 4 + d * 4097;
                im.drone_id = f.drone_id; im.landmark_num = (int)p[0];
                im.image_desc.assign(p + 1, p + 1 + 4096);
                f.images.push_back(std::move(im));
            }
            chunk.push_back(std::move(f));
            if (chunk.size() == 5) flush();
        }
        flush();
    }
    // --- LoopCam::on_flattened_images as one asynchronous unit: one direction, the image as both the up and the down camera
    {
        omni::Context vctx(0);
        Swarm::SuperPointHIP sp2(ctx, argv[1], argv[2], argv[3], W, H, 0.015f, 200, false, OMNI_PREC_F32, 2);
        Swarm::MobileNetVLADHIP vl2(vctx, argv[4], W, H, false, 1);
        omni::LoopCamHIP cam(ctx, sp2, vctx, vl2, 1, 200, W, H);
        const uint8_t* imgs[2] = {img.data(), img.data()};
        cam.enqueue(imgs, W, false);
        const omni_cam_result r = cam.wait();
        int same = r.n_kps[0] == (int)features.size() && r.n_kps[1] == r.n_kps[0];
        for (int i = 0; same && i < r.n_kps[0]; ++i) same = r.kps_xy[2 * i] == features[i].x && r.kps_xy[2 * i + 1] == features[i].y;
        for (size_t i = 0; same && i < feature_descriptor.size(); ++i) same = r.desc[i] == feature_descriptor[i];
        for (int i = 0; same && i < r.global_dim; ++i) same = r.global_desc[i] == image_desc[i];
        int diag = r.n_matches[0] == r.n_kps[0];                 // identical descriptor sets: every point matches itself at distance 0
        for (int i = 0; diag && i < r.n_matches[0]; ++i) diag = r.match_up[i] == i && r.match_down[i] == i && r.match_dist[i] == 0.f;
        std::printf("\nCAM %d %d %d", same, diag, r.n_matches[0]);
    }
    {   // enable_perf = true (superpoint_tensorrt.h:20-28): the reference's timing line of every call (superpoint_tensorrt.cpp:130-162), then the stages
        Swarm::SuperPointHIP sp_perf(ctx, argv[1], argv[2], argv[3], W, H, 0.015f, 200, true, OMNI_PREC_F32);
        std::vector<omni::Point2f> k2;
        std::vector<float> d2;
        std::printf("\n");
        sp_perf.inference(img.data(), W, k2, d2);
        std::printf("PERF_SAME %d", (int)(k2.size() == features.size() && d2 == feature_descriptor));
    }
    std::printf("\nOK\n");
    return 0;
}
