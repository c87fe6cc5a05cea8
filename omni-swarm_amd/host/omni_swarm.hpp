// omni_swarm.hpp -- header-only C++17 host adapters over the C ABI (include/omni_hip.h) with the call surfaces of the
// reference classes, so LoopCam / LoopDetector change only in #include and type names:
//
//   Swarm::SuperPointHIP       <-> Swarm::SuperPointTensorRT      swarm_loop/include/swarm_loop/superpoint_tensorrt.h:12-29
//   Swarm::MobileNetVLADHIP    <-> Swarm::MobileNetVLADTensorRT   swarm_loop/include/swarm_loop/mobilenetvlad_tensorrt.h:6-22
//   omni::IndexFlatIP          <-> faiss::IndexFlatIP             (add / search / ntotal; loop_detector.cpp:166-170,213,232,291)
//   omni::BFMatcherL2X         <-> cv::BFMatcher(NORM_L2, true)   (match; loop_cam.cpp:147-150, loop_detector.cpp:564-567)
//   omni::LoopDetectorCore     <-> LoopDetector's DB + decision rules (loop_detector.cpp:11-287), geometry via callback
//   omni::LoopCamHIP           <-> the CNN + matching work of LoopCam::on_flattened_images (loop_cam.cpp:178-229) as one async unit
//
// Compiles with plain g++ (no HIP, ROS, OpenCV or faiss headers); link with -lomni_hip.  Define OMNI_WITH_OPENCV to get
// the cv::Mat / cv::Point2f / cv::DMatch overloads the reference call sites use verbatim.
// Errors: the reference aborts (assert / NV_CUDA_CHECK); these adapters throw std::runtime_error carrying
// omni_last_error() from constructors and return empty results + set last_status from inference calls.
#pragma once
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/omni_hip.h"
#ifdef OMNI_WITH_OPENCV
#include <opencv2/opencv.hpp>
#endif

namespace omni {

struct Point2f { float x, y; };
struct DMatch { int queryIdx, trainIdx; float distance; };   // cv::DMatch fields used by the reference

inline void check(int rc, const char* what) {
    if (rc != OMNI_OK) throw std::runtime_error(std::string(what) + ": " + omni_last_error());
}

// ---- shared context (one HIP stream); the reference creates one cudaStream per runner (tensorrt_generic.cpp:103) ----
class Context {
public:
    // high_priority: the stream of short work the host waits on (the detector's searches) next to streams that keep the whole GPU busy
    explicit Context(int device_id = 0, bool high_priority = false) : h_(nullptr) {
        // a host built against another include/omni_hip.h (struct layouts, entry points) must not run: checked before the first handle exists
        if (omni_abi_version() != OMNI_ABI_VERSION)
            throw std::runtime_error("libomni_hip.so has ABI version " + std::to_string(omni_abi_version()) + ", this host was built against " + std::to_string(OMNI_ABI_VERSION));
        h_ = high_priority ? omni_ctx_create_priority(device_id, 1) : omni_ctx_create(device_id);
        if (!h_) throw std::runtime_error(std::string("omni_ctx_create: ") + omni_last_error());
    }
    ~Context() { omni_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    omni_ctx* get() const { return h_; }
private:
    omni_ctx* h_;
};

// ---- weight container: "OMNW1" file = named float32 tensors (written by tools/export_weights.py from a state_dict) ----
struct Tensor { std::vector<uint32_t> shape; std::vector<float> data; };
using WeightMap = std::map<std::string, Tensor>;

inline WeightMap load_omnw(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open weight file " + path);
    char magic[8];
    f.read(magic, 8);
    if (std::memcmp(magic, "OMNW1\0\0\0", 8) != 0) throw std::runtime_error(path + ": not an OMNW1 file");
    uint32_t n = 0;
    f.read(reinterpret_cast<char*>(&n), 4);
    WeightMap m;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t len = 0, nd = 0;
        f.read(reinterpret_cast<char*>(&len), 4);
        std::string name(len, '\0');
        f.read(&name[0], len);
        f.read(reinterpret_cast<char*>(&nd), 4);
        Tensor t;
        t.shape.resize(nd);
        size_t cnt = 1;
        for (uint32_t d = 0; d < nd; ++d) { f.read(reinterpret_cast<char*>(&t.shape[d]), 4); cnt *= t.shape[d]; }
        t.data.resize(cnt);
        f.read(reinterpret_cast<char*>(t.data.data()), cnt * 4);
        if (!f) throw std::runtime_error(path + ": truncated");
        m.emplace(std::move(name), std::move(t));
    }
    return m;
}

// components_.csv: one row per component, comma separated; mean_.csv: one value per line
// (load_csv_mat_eigen / load_csv_vec_eigen, superpoint_tensorrt.cpp:14-89)
inline std::vector<float> load_csv_floats(const std::string& path, int* rows = nullptr, int* cols = nullptr) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<float> v;
    std::string line;
    int r = 0, c = 0;
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        std::stringstream ss(line);
        std::string cell;
        int cc = 0;
        while (std::getline(ss, cell, ',')) { if (!cell.empty()) { v.push_back(std::stof(cell)); ++cc; } }
        if (cc) { ++r; c = cc; }
    }
    if (rows) *rows = r;
    if (cols) *cols = c;
    return v;
}

}  // namespace omni

namespace Swarm {

// SuperPointTensorRT(engine_path, pca_comp, pca_mean, width, height, thres, max_num, enable_perf)
class SuperPointHIP {
public:
    SuperPointHIP(omni::Context& ctx, const std::string& weights_path, const std::string& pca_comp_csv, const std::string& pca_mean_csv,
                  int width, int height, float thres = 0.015f, int max_num = 200, bool enable_perf = false,
                  int precision = OMNI_PREC_F16, int max_batch = 1)
        : width_(width), height_(height), max_num_(max_num), max_batch_(max_batch), enable_perf_(enable_perf) {
        static const char* names[OMNI_SP_NUM_LAYERS] = {"conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b",
                                                        "convPa", "convPb", "convDa", "convDb"};
        omni::WeightMap w = omni::load_omnw(weights_path);
        omni_sp_weights sw{};
        for (int i = 0; i < OMNI_SP_NUM_LAYERS; ++i) {
            auto wi = w.find(std::string(names[i]) + ".weight"), bi = w.find(std::string(names[i]) + ".bias");
            if (wi == w.end() || bi == w.end()) throw std::runtime_error(std::string("weights missing for ") + names[i]);
            sw.weight[i] = wi->second.data.data();
            sw.bias[i] = bi->second.data.data();
        }
        std::vector<float> comp, mean;
        int rows = 0, cols = 0;
        if (!pca_comp_csv.empty()) {
            comp = omni::load_csv_floats(pca_comp_csv, &rows, &cols);
            mean = omni::load_csv_floats(pca_mean_csv);
            if (cols != 256 || mean.size() != 256) throw std::runtime_error("PCA files must be [pca_dim x 256] and [256]");
        }
        h_ = omni_sp_create(ctx.get(), &sw, comp.empty() ? nullptr : comp.data(), mean.empty() ? nullptr : mean.data(), rows, width, height,
                            thres, max_num, precision, max_batch);
        if (!h_) throw std::runtime_error(std::string("omni_sp_create: ") + omni_last_error());
        if (enable_perf_) (void)omni_sp_set_perf(h_, 1);
        dim_ = omni_sp_desc_dim(h_);
    }
    ~SuperPointHIP() { omni_sp_destroy(h_); }
    SuperPointHIP(const SuperPointHIP&) = delete;
    SuperPointHIP& operator=(const SuperPointHIP&) = delete;

    // void inference(const cv::Mat& input, std::vector<cv::Point2f>& keypoints, std::vector<float>& local_descriptors)
    // (superpoint_tensorrt.cpp:117-162): keypoints/local_descriptors are cleared first (:120-121); descriptors are n x dim.
    void inference(const uint8_t* gray, int stride, std::vector<omni::Point2f>& keypoints, std::vector<float>& local_descriptors,
                   bool fisheye_mask = false) {
        const auto t_call = std::chrono::steady_clock::now();
        keypoints.clear();
        local_descriptors.clear();
        kps_.resize((size_t)max_num_ * 2);
        desc_.resize((size_t)max_num_ * dim_);
        int n = 0;
        last_status = omni_sp_infer(h_, gray, stride, 1, fisheye_mask ? 1 : 0, kps_.data(), &n, desc_.data(), nullptr);
        if (last_status != OMNI_OK) { std::fprintf(stderr, "[SuperPointHIP] %s\n", omni_last_error()); return; }
        keypoints.reserve(n);
        for (int i = 0; i < n; ++i) keypoints.push_back({kps_[2 * i], kps_[2 * i + 1]});
        local_descriptors.assign(desc_.begin(), desc_.begin() + (size_t)n * dim_);
        if (enable_perf_) {
            // the reference's line (superpoint_tensorrt.cpp:130-162): engine time, from_blob (nothing to wrap here), getKeyPoints + computeDescriptors (one GPU stage
            // here: nms + top-k + describe), the whole call on the host's clock -- then the stages one by one, which the reference cannot see inside its engine
            float st[OMNI_SP_NUM_STAGES] = {};
            if (omni_sp_last_stage_ms(h_, st) == OMNI_OK) {
                float net = 0.f;
                for (int i = 0; i + 1 < OMNI_SP_NUM_STAGES && omni_sp_stage_name(i + 1)[0]; ++i) net += st[i];          // every stage but the last named one
                int last = 0;
                while (last + 1 < OMNI_SP_NUM_STAGES && omni_sp_stage_name(last + 1)[0]) ++last;
                std::printf("Inference Time %.3f from_blob 0 getKeyPoints+computeDescriptors %.3f inference all %.3f features %d desc size %zu\n", net, st[last],
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(), n, local_descriptors.size());
                std::printf("  stages (ms):");
                for (int i = 0; i <= last; ++i) std::printf(" %s %.3f", omni_sp_stage_name(i), st[i]);
                std::printf("\n");
            }
        }
    }
#ifdef OMNI_WITH_OPENCV
    void inference(const cv::Mat& input, std::vector<cv::Point2f>& keypoints, std::vector<float>& local_descriptors) {
        std::vector<omni::Point2f> k;
        if (input.rows != height_ || input.cols != width_ || input.type() != CV_8UC1) { keypoints.clear(); local_descriptors.clear(); last_status = OMNI_ERR_INVALID; return; }
        inference(input.data, (int)input.step, k, local_descriptors);
        keypoints.clear();
        for (auto& p : k) keypoints.emplace_back(p.x, p.y);
    }
#endif
    int desc_dim() const { return dim_; }
    omni_sp* handle() const { return h_; }
    int last_status = OMNI_OK;
private:
    omni_sp* h_ = nullptr;
    int width_, height_, max_num_, max_batch_, dim_ = 0;
    bool enable_perf_;
    std::vector<float> kps_, desc_;
};

// MobileNetVLADTensorRT(engine_path, width, height, enable_perf); std::vector<float> inference(const cv::Mat&)
// ASSUMED architecture: the OMNW1 file carries the layer table as "<layer>.weight/.bias" tensors plus a
// "layers" index tensor [n][4] = (kind, cin, cout, stride) in execution order (tools/export_weights.py).
class MobileNetVLADHIP {
public:
    MobileNetVLADHIP(omni::Context& ctx, const std::string& weights_path, int width, int height, bool enable_perf = false, int max_batch = 1)
        : width_(width), height_(height) {
        (void)enable_perf;
        omni::WeightMap w = omni::load_omnw(weights_path);
        auto need = [&](const std::string& n) -> omni::Tensor& {
            auto it = w.find(n);
            if (it == w.end()) throw std::runtime_error("weights missing: " + n);
            return it->second;
        };
        omni::Tensor& tab = need("layers");
        const int n_layers = (int)tab.shape.at(0);
        std::vector<omni_vlad_layer> layers(n_layers);
        for (int i = 0; i < n_layers; ++i) {
            const std::string base = "layer" + std::to_string(i);
            layers[i].kind = (int)tab.data[i * 4 + 0]; layers[i].cin = (int)tab.data[i * 4 + 1];
            layers[i].cout = (int)tab.data[i * 4 + 2]; layers[i].stride = (int)tab.data[i * 4 + 3];
            layers[i].weight = need(base + ".weight").data.data();
            layers[i].bias = need(base + ".bias").data.data();
        }
        omni_vlad_weights vw{};
        vw.n_layers = n_layers; vw.layers = layers.data();
        vw.n_clusters = (int)need("vlad.clusters").shape.at(0); vw.feat_dim = (int)need("vlad.clusters").shape.at(1);
        vw.out_dim = (int)need("fc.bias").shape.at(0);
        vw.assign_w = need("vlad.assign.weight").data.data(); vw.assign_b = need("vlad.assign.bias").data.data();
        vw.clusters = need("vlad.clusters").data.data(); vw.fc_w = need("fc.weight").data.data(); vw.fc_b = need("fc.bias").data.data();
        out_dim_ = vw.out_dim;
        h_ = omni_vlad_create(ctx.get(), &vw, width, height, max_batch);
        if (!h_) throw std::runtime_error(std::string("omni_vlad_create: ") + omni_last_error());
    }
    ~MobileNetVLADHIP() { omni_vlad_destroy(h_); }
    MobileNetVLADHIP(const MobileNetVLADHIP&) = delete;
    MobileNetVLADHIP& operator=(const MobileNetVLADHIP&) = delete;

    std::vector<float> inference(const uint8_t* gray, int stride, bool fisheye_mask = false) {
        std::vector<float> out(out_dim_);
        last_status = omni_vlad_infer(h_, gray, stride, 1, fisheye_mask ? 1 : 0, out.data());
        if (last_status != OMNI_OK) { std::fprintf(stderr, "[MobileNetVLADHIP] %s\n", omni_last_error()); out.clear(); }
        return out;
    }
#ifdef OMNI_WITH_OPENCV
    std::vector<float> inference(const cv::Mat& input) { return inference(input.data, (int)input.step); }
#endif
    omni_vlad* handle() const { return h_; }
    int out_dim() const { return out_dim_; }
    int last_status = OMNI_OK;
private:
    omni_vlad* h_ = nullptr;
    int width_, height_, out_dim_ = 0;
};

}  // namespace Swarm

namespace omni {

// The CNN + matching part of LoopCam::on_flattened_images (loop_cam.cpp:178-229): SuperPoint on the 2*n_dirs images of one
// fisheye key frame (up cameras, then down cameras), MobileNetVLAD on the n_dirs up images, BFMatcher(L2, crossCheck) up <->
// down per direction -- one asynchronous unit on the GPU (omni_cam_*), results in one pinned host block.  The reference runs
// these 12 engine calls and 4 matches strictly serially; camera lifting / triangulation stay in LoopCam.
//   superpoint_net must have been created with max_batch >= 2*n_dirs (on sp_ctx), netvlad_net with max_batch >= n_dirs (on
//   vlad_ctx; a second Context = a second HIP stream lets the two networks overlap).
class LoopCamHIP {
public:
    // mono: CameraConfig::PINHOLE_DEPTH (loop_cam.cpp:190-194) -- n_dirs images of ONE camera each, no up/down match (omni_cam_create_mono)
    LoopCamHIP(Context& sp_ctx, Swarm::SuperPointHIP& superpoint_net, Context& vlad_ctx, Swarm::MobileNetVLADHIP& netvlad_net, int n_dirs,
               int max_num, int width, int height, bool mono = false)
        : ctx_(sp_ctx), n_(n_dirs), ni_(mono ? n_dirs : 2 * n_dirs), w_(width), h_img_(height) {
        h_ = mono ? omni_cam_create_mono(sp_ctx.get(), superpoint_net.handle(), vlad_ctx.get(), netvlad_net.handle(), n_dirs, max_num, netvlad_net.out_dim())
                  : omni_cam_create(sp_ctx.get(), superpoint_net.handle(), vlad_ctx.get(), netvlad_net.handle(), n_dirs, max_num, netvlad_net.out_dim(),
                                    OMNI_BF_OPENCV);
        if (!h_) throw std::runtime_error(std::string("omni_cam_create: ") + omni_last_error());
        gray_dev_ = static_cast<uint8_t*>(omni_dev_alloc(sp_ctx.get(), (size_t)ni_ * width * height));
        if (!gray_dev_) { omni_cam_destroy(h_); throw std::runtime_error(std::string("omni_dev_alloc: ") + omni_last_error()); }
    }
    ~LoopCamHIP() { omni_cam_destroy(h_); omni_dev_free(ctx_.get(), gray_dev_); if (pinned_) omni_host_free(pinned_); }
    LoopCamHIP(const LoopCamHIP&) = delete;
    LoopCamHIP& operator=(const LoopCamHIP&) = delete;

    // images[0..n_dirs) = up cameras, images[n_dirs..2*n_dirs) = down cameras; each H rows of `stride` bytes (u8 gray).
    // Uploads and enqueues; returns at once.  fisheye_mask zeroes rows [3H/4, H) (loop_cam.cpp:536-539).
    void enqueue(const uint8_t* const* images, int stride, bool fisheye_mask = true) {
        stage_.resize((size_t)ni_ * w_ * h_img_);
        for (int i = 0; i < ni_; ++i)
            for (int y = 0; y < h_img_; ++y) std::memcpy(stage_.data() + ((size_t)i * h_img_ + y) * w_, images[i] + (size_t)y * stride, w_);
        check(omni_memcpy_h2d(ctx_.get(), gray_dev_, stage_.data(), stage_.size()), "LoopCamHIP upload");
        check(omni_cam_enqueue_dev(h_, gray_dev_, w_, fisheye_mask ? 1 : 0), "omni_cam_enqueue_dev");
    }
    // same, without blocking: the images are packed into a pinned block owned by this object and go up as ONE asynchronous copy in front of
    // the kernels (omni_cam_enqueue_host); the host is free again as soon as the memcpy into the pinned block is done
    void enqueue_async(const uint8_t* const* images, int stride, bool fisheye_mask = true) {
        const size_t bytes = (size_t)ni_ * w_ * h_img_;
        if (!pinned_) { pinned_ = static_cast<uint8_t*>(omni_host_alloc(bytes)); if (!pinned_) throw std::runtime_error(std::string("omni_host_alloc: ") + omni_last_error()); }
        for (int i = 0; i < ni_; ++i)
            for (int y = 0; y < h_img_; ++y) std::memcpy(pinned_ + ((size_t)i * h_img_ + y) * w_, images[i] + (size_t)y * stride, w_);
        check(omni_cam_enqueue_host(h_, pinned_, w_, w_, h_img_, fisheye_mask ? 1 : 0), "omni_cam_enqueue_host");
    }
    // images already packed [2*n_dirs][H][W] in (ideally pinned) host memory: no staging copy at all
    void enqueue_host(const uint8_t* gray_host, int stride, bool fisheye_mask = true) {
        check(omni_cam_enqueue_host(h_, gray_host, stride, w_, h_img_, fisheye_mask ? 1 : 0), "omni_cam_enqueue_host");
    }
    // the unit's images as segments of packed host memory (omni_cam_enqueue_host_parts): up cameras' parts, then the down cameras' (none for a mono rig)
    void enqueue_host_parts(const std::vector<const uint8_t*>& up, const std::vector<int>& up_images, const std::vector<const uint8_t*>& down, const std::vector<int>& down_images,
                            bool fisheye_mask = true) {
        check(omni_cam_enqueue_host_parts(h_, up.data(), up_images.data(), (int)up.size(), down.empty() ? nullptr : down.data(), down.empty() ? nullptr : down_images.data(),
                                          (int)down.size(), w_, h_img_, fisheye_mask ? 1 : 0), "omni_cam_enqueue_host_parts");
    }
    void enqueue_dev(const uint8_t* gray_dev, int stride, bool fisheye_mask = true) {      // images already in HBM
        check(omni_cam_enqueue_dev(h_, gray_dev, stride, fisheye_mask ? 1 : 0), "omni_cam_enqueue_dev");
    }
    // the next enqueue on this object starts behind the convolution stack of `earlier`'s last one (omni_cam_order_after)
    void order_after(LoopCamHIP& earlier, int streams) { check(omni_cam_order_after(h_, earlier.h_, streams), "omni_cam_order_after"); }
    // a unit of fewer directions than this object was created for (omni_cam_set_active): the next enqueues read cams * n_dirs images
    void set_active(int n_dirs) { check(omni_cam_set_active(h_, n_dirs), "omni_cam_set_active"); }
    // non-blocking: would wait() return at once?
    bool ready() { int r = 0; check(omni_cam_ready(h_, &r), "omni_cam_ready"); return r != 0; }
    // blocks until the key frame is done; pointers stay valid until the next enqueue on this object
    omni_cam_result wait() {
        omni_cam_result r{};
        check(omni_cam_wait(h_, &r), "omni_cam_wait");
        return r;
    }
private:
    Context& ctx_;
    omni_cam* h_ = nullptr;
    uint8_t* gray_dev_ = nullptr;
    uint8_t* pinned_ = nullptr;
    int n_, ni_, w_, h_img_;                  // ni_: images per unit (2 per direction, or 1: mono)
    std::vector<uint8_t> stage_;
};

// faiss::IndexFlatIP(d): the members LoopDetector touches
class IndexFlatIP {
public:
    using idx_t = int64_t;                       // faiss::Index::idx_t
    IndexFlatIP(Context& ctx, int d, int storage = OMNI_STORE_F32) : d(d), h_(omni_index_create(ctx.get(), d, storage, 0)) {
        if (!h_) throw std::runtime_error(std::string("omni_index_create: ") + omni_last_error());
    }
    ~IndexFlatIP() { omni_index_destroy(h_); }
    IndexFlatIP(const IndexFlatIP&) = delete;
    IndexFlatIP& operator=(const IndexFlatIP&) = delete;
    void add(idx_t n, const float* x) { check(omni_index_add(h_, n, x), "IndexFlatIP::add"); ntotal = omni_index_ntotal(h_); }
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const {
        check(omni_index_search(h_, (int)n, x, (int)k, distances, labels), "IndexFlatIP::search");
    }
    void reset() { check(omni_index_reset(h_), "IndexFlatIP::reset"); ntotal = 0; }
    // shard checkpoint / restore (new: the reference's database lives in RAM only)
    void save(const std::string& path) const { check(omni_index_save(h_, path.c_str()), "IndexFlatIP::save"); }
    void load(const std::string& path) { check(omni_index_load(h_, path.c_str()), "IndexFlatIP::load"); ntotal = omni_index_ntotal(h_); }
    int d;
    idx_t ntotal = 0;                            // public data member, as in faiss
    omni_index* handle() const { return h_; }
private:
    omni_index* h_;
};

// cv::BFMatcher(cv::NORM_L2, /*crossCheck=*/true).match(query, train, matches)
class BFMatcherL2X {
public:
    explicit BFMatcherL2X(Context& ctx, int mode = OMNI_BF_OPENCV) : ctx_(ctx), mode_(mode) {}
    void match(const float* query, int nq, const float* train, int nt, int dim, std::vector<DMatch>& matches) {
        matches.clear();
        std::vector<int> qi(nq > 0 ? nq : 1), ti(nq > 0 ? nq : 1);
        std::vector<float> dd(nq > 0 ? nq : 1);
        int n = 0;
        check(omni_bf_match(ctx_.get(), query, nq, train, nt, dim, mode_, qi.data(), ti.data(), dd.data(), &n), "BFMatcherL2X::match");
        for (int i = 0; i < n; ++i) matches.push_back({qi[i], ti[i], dd[i]});
    }
    // several (query, train) pairs in one GPU round trip; out[p] as match() would return it
    struct Pair { const float* query; int nq; const float* train; int nt; };
    void match_multi(const std::vector<Pair>& pairs, int dim, std::vector<std::vector<DMatch>>& out) {
        out.assign(pairs.size(), {});
        if (pairs.empty()) return;
        int max_n = 1;
        std::vector<const float*> q(pairs.size()), t(pairs.size());
        std::vector<int> nq(pairs.size()), nt(pairs.size()), n(pairs.size());
        for (size_t p = 0; p < pairs.size(); ++p) {
            q[p] = pairs[p].query; t[p] = pairs[p].train; nq[p] = pairs[p].nq; nt[p] = pairs[p].nt;
            max_n = std::max(max_n, std::max(nq[p], nt[p]));
        }
        std::vector<int> qi(pairs.size() * max_n), ti(pairs.size() * max_n);
        std::vector<float> dd(pairs.size() * max_n);
        check(omni_bf_match_multi(ctx_.get(), (int)pairs.size(), q.data(), nq.data(), t.data(), nt.data(), dim, mode_, max_n, qi.data(), ti.data(), dd.data(),
                                  n.data()), "BFMatcherL2X::match_multi");
        for (size_t p = 0; p < pairs.size(); ++p)
            for (int i = 0; i < n[p]; ++i) out[p].push_back({qi[p * max_n + i], ti[p * max_n + i], dd[p * max_n + i]});
    }
#ifdef OMNI_WITH_OPENCV
    void match(const cv::Mat& query, const cv::Mat& train, std::vector<cv::DMatch>& matches) {
        std::vector<DMatch> m;
        match(query.ptr<float>(), query.rows, train.ptr<float>(), train.rows, query.cols, m);
        matches.clear();
        for (auto& x : m) matches.emplace_back(x.queryIdx, x.trainIdx, x.distance);
    }
#endif
private:
    Context& ctx_;
    int mode_;
};

// ---- LoopDetector's database and decision rules (loop_detector.cpp:11-287), POD messages instead of swarm_msgs -------
struct Point3f { float x = 0, y = 0, z = 0; };
struct PoseMsg { double position[3] = {0, 0, 0}; double quat_wxyz[4] = {1, 0, 0, 0}; };     // Pose_t as fromROSPose fills it (loop_cam.cpp:366-368)
// ImageDescriptor_t (swarm_msgs, un-vendored): every field the reference reads or writes on this path -- loop_cam.cpp:529-585 (extractor),
// :362-374,:434-440 (stereo part), loop_net.cpp:51-79,206-218 (wire split / reassembly), loop_detector.cpp:539-603 (matching)
struct ImageDescriptor {
    int drone_id = 0, landmark_num = 0, direction = 0;
    int64_t msg_id = 0, frame_id = 0;
    double timestamp = 0;                        // Time_t (sec, nsec) as seconds
    bool prevent_adding_db = false;
    std::vector<float> image_desc;               // 4096 (image_desc_size)
    std::vector<float> feature_descriptor;       // n x 64 (feature_descriptor_size)
    std::vector<Point2f> landmarks_2d;           // pixel key points
    std::vector<Point2f> landmarks_2d_norm;      // camodocal liftProjective -> (x/z, y/z) (loop_cam.cpp:558-569)
    std::vector<Point3f> landmarks_3d;           // triangulated, drone world frame (loop_cam.cpp:434-440); (0,0,0) when flag == 0
    std::vector<uint8_t> landmarks_flag;         // 1 = has a 3-D point
    PoseMsg pose_drone, camera_extrinsic;
    int image_width = 0, image_height = 0;
    std::vector<uint8_t> image;                  // optional JPEG (encode_image, loop_cam.cpp:49-71), opaque here
};
struct FisheyeFrameDescriptor {                  // FisheyeFrameDescriptor_t
    int64_t msg_id = 0;
    int drone_id = 0, landmark_num = 0, image_num = 0;
    double timestamp = 0;
    bool prevent_adding_db = false;
    PoseMsg pose_drone;
    std::vector<ImageDescriptor> images;
};
struct LoopCandidate { bool found = false; int64_t old_msg_id = -1; int image_id = -1, direction_new = -1, direction_old = -1; double distance = -1; bool added = false, queried = false, loop = false; };

class LoopDetectorCore {
public:
    static constexpr int REMOTE_MAGIN_NUMBER = 1000000;   // loop_detector.h:22
    static constexpr int SEARCH_NEAREST_NUM = 5;          // loop_defines.h:32
    // tunables (swarm_loop.cpp:221-237)
    double INNER_PRODUCT_THRES = 0.6, INIT_MODE_PRODUCT_THRES = 0.3;
    int MATCH_INDEX_DIST = 10, MIN_LOOP_NUM = 15, MIN_DIRECTION_LOOP = 3, inter_drone_init_frames = 50;
    bool stereo_fisheye = true;
    // geometry stage (compute_loop, :627-836) stays on the host: (new, old, dir_new, dir_old, init_mode) -> success
    std::function<bool(const FisheyeFrameDescriptor&, const FisheyeFrameDescriptor&, int, int, bool)> compute_loop;

    LoopDetectorCore(Context& ctx, int self_id, int storage = OMNI_STORE_F32)
        : self_id(self_id), local_index(ctx, 4096, storage), remote_index(ctx, 4096, storage), ctx_(ctx) {}
    ~LoopDetectorCore() { if (batch_buf_) omni_dev_free(ctx_.get(), batch_buf_); if (raw_pinned_) omni_host_free(raw_pinned_); }

    int database_size() const { return replaying_ ? (int)(sim_local_ + sim_remote_) : (int)(local_index.ntotal + remote_index.ntotal); }

    // on_image_recv for several frames in arrival order with ONE host synchronisation instead of ~6 per frame (4 row appends, 1-2
    // searches).  Which rows a frame appends and which searches it runs (:36-98) does not depend on any search RESULT: the batch is
    // planned first, enqueued on the index stream -- all appends, then every search restricted to the rows its frame would have seen
    // (omni_index_search_prefix_dev) -- fetched with one copy, and the decision rules are replayed frame by frame through the unchanged
    // on_image_recv.  rows_dev (optional): the frames' global descriptors, [sum of images][4096] fp32 in frame order, already in HBM
    // (e.g. MobileNetVLAD's output buffer): rows are appended and queried from there, the host copies are not read.
    // rvalue overload: frames that enter the database are MOVED into it (the const& overload copies them, 270 KB per fisheye key frame)
    std::vector<LoopCandidate> on_images_recv_batch(std::vector<FisheyeFrameDescriptor>&& frames, const float* rows_dev = nullptr) {
        begin_images_batch(std::move(frames), rows_dev);
        return end_images_batch();
    }
    std::vector<LoopCandidate> on_images_recv_batch(const std::vector<FisheyeFrameDescriptor>& frames_in, const float* rows_dev = nullptr) {
        begin_batch(&frames_in, false, rows_dev);
        return end_images_batch();
    }
    // The same in two halves, for a host loop that must not wait for the GPU here (KeyframePipeline): begin_images_batch plans the batch, enqueues the
    // appends, the prefix searches and the copy of the result lists into a pinned block, and returns; end_images_batch waits for that copy (long done
    // when it is called one micro-batch later), replays the decision rules and returns the candidates.  One batch at a time; the frames are owned
    // by the detector in between (those that enter the database are moved into it by end_images_batch, the others stay in held_frames() until the
    // next begin).  rows_dev must stay untouched until the appends and searches have read it: order the producer's stream behind ctx's
    // (omni_ctx_order_after) before it writes the buffer again.
    void begin_images_batch(std::vector<FisheyeFrameDescriptor>&& frames, const float* rows_dev = nullptr) {
        if (pb_.active) throw std::logic_error("begin_images_batch: the previous batch was not ended");
        pb_.owned = std::move(frames);
        begin_batch(&pb_.owned, true, rows_dev);
    }
    bool batch_pending() const { return pb_.active; }
    std::vector<FisheyeFrameDescriptor>& held_frames() { return pb_.owned; }

private:
    struct BatchSearch { IndexFlatIP* index; size_t row; int max_index; int64_t n_limit; };
    struct PendingBatch {
        bool active = false, movable = false;
        std::vector<FisheyeFrameDescriptor> owned, cleaned;
        const std::vector<FisheyeFrameDescriptor>* frames = nullptr;
        std::vector<BatchSearch> searches;
        std::vector<std::pair<size_t, size_t>> where;            // byte offsets of I and D per search
        size_t need = 0;
        bool any_add = false;
        int64_t start_local = 0, start_remote = 0;
        float* own_rows = nullptr;
    } pb_;
    char* raw_pinned_ = nullptr;
    size_t raw_pinned_bytes_ = 0;
    void rollback_batch() {
        // a failed append / search / copy: take the rows of this batch out again so that row ids, the id maps and the recency rule of every
        // later frame stay what they were (the Python twin's _rollback)
        (void)omni_index_truncate(local_index.handle(), pb_.start_local);
        (void)omni_index_truncate(remote_index.handle(), pb_.start_remote);
        local_index.ntotal = omni_index_ntotal(local_index.handle());
        remote_index.ntotal = omni_index_ntotal(remote_index.handle());
        if (pb_.own_rows) { omni_dev_free(ctx_.get(), pb_.own_rows); pb_.own_rows = nullptr; }
        row_ids_.clear(); deferred_.clear();
        pb_.active = false;
    }
    void begin_batch(const std::vector<FisheyeFrameDescriptor>* frames_in, bool movable, const float* rows_dev) {
        if (pb_.active) throw std::logic_error("on_images_recv_batch: a batch begun with begin_images_batch is still open");
        // host rows are read as 4096 floats each: an image whose global descriptor has another length (a malformed or other-version packet
        // that got past LoopNet) counts as an image without landmarks -- the reference would read past the vector (loop_detector.cpp:166-170)
        pb_.cleaned.clear();
        pb_.movable = movable;
        if (!rows_dev) {
            bool bad = false;
            for (auto& f : *frames_in) bad = bad || malformed(f);
            if (bad) { pb_.cleaned = *frames_in; for (auto& f : pb_.cleaned) sanitise(f); pb_.movable = false; }
        }
        pb_.frames = pb_.cleaned.empty() ? frames_in : &pb_.cleaned;
        const std::vector<FisheyeFrameDescriptor>& frames = *pb_.frames;
        struct Add { IndexFlatIP* index; size_t row; };
        std::vector<Add> adds;
        std::vector<BatchSearch>& searches = pb_.searches;
        searches.clear();
        int64_t sim_local = local_index.ntotal, sim_remote = remote_index.ntotal;
        std::set<int> nodes = all_nodes;
        row_ids_.clear();
        size_t base = 0, total_rows = 0;
        for (auto& f : frames) total_rows += f.images.size();
        for (auto& f : frames) {                                                  // plan: the gating of on_image_recv on simulated sizes
            const size_t first = base;
            base += f.images.size();
            if (f.images.empty() || (f.drone_id != self_id && sim_local + sim_remote == 0)) continue;
            const bool new_node = nodes.find(f.drone_id) == nodes.end();
            nodes.insert(f.drone_id);
            int dir_count = 0;
            for (auto& img : f.images) if (img.landmark_num > 0) ++dir_count;
            if (dir_count < MIN_DIRECTION_LOOP || f.landmark_num < MIN_LOOP_NUM) continue;
            if (!f.prevent_adding_db || new_node)
                for (size_t i = 0; i < f.images.size(); ++i) if (f.images[i].landmark_num > 0) {
                    if (f.images[i].drone_id == self_id) { adds.push_back({&local_index, first + i}); row_ids_.push_back((int)sim_local++); }
                    else { adds.push_back({&remote_index, first + i}); row_ids_.push_back((int)sim_remote++ + REMOTE_MAGIN_NUMBER); }
                }
            if (sim_local + sim_remote > MATCH_INDEX_DIST || f.drone_id != self_id) {        // init_mode implies a remote drone
                const size_t d = stereo_fisheye ? 1 : 0;
                if (f.images.size() > d && f.images[d].landmark_num > 0) {
                    if (f.images[d].drone_id == self_id) {
                        searches.push_back({&remote_index, first + d, 1, sim_remote});
                        if (!f.prevent_adding_db) searches.push_back({&local_index, first + d, MATCH_INDEX_DIST, sim_local});
                    } else {
                        searches.push_back({&local_index, first + d, 1, sim_local});
                    }
                }
            }
        }
        // enqueue: rows to HBM unless they are there already, appends, prefix searches, one result copy
        pb_.own_rows = nullptr;
        if (!rows_dev && (!adds.empty() || !searches.empty())) {
            std::vector<float> stage(total_rows * 4096, 0.f);
            size_t r = 0;
            for (auto& f : frames) for (auto& img : f.images) { if (img.image_desc.size() == 4096) std::memcpy(&stage[r * 4096], img.image_desc.data(), 4096 * 4); ++r; }
            pb_.own_rows = static_cast<float*>(omni_dev_alloc(ctx_.get(), stage.size() * 4));
            if (!pb_.own_rows) throw std::runtime_error(std::string("omni_dev_alloc: ") + omni_last_error());
            check(omni_memcpy_h2d(ctx_.get(), pb_.own_rows, stage.data(), stage.size() * 4), "on_images_recv_batch upload");
            rows_dev = pb_.own_rows;
        }
        pb_.start_local = local_index.ntotal; pb_.start_remote = remote_index.ntotal;
        pb_.where.assign(searches.size(), {SIZE_MAX, SIZE_MAX});
        pb_.active = true;
        pb_.any_add = !adds.empty();
        try {
        for (size_t a = 0; a < adds.size();) {                                    // consecutive rows of one index go in as one append
            size_t b = a + 1;
            while (b < adds.size() && adds[b].index == adds[a].index && adds[b].row == adds[b - 1].row + 1) ++b;
            check(omni_index_add_dev(adds[a].index->handle(), (int64_t)(b - a), rows_dev + adds[a].row * 4096), "omni_index_add_dev");
            a = b;
        }
        local_index.ntotal = omni_index_ntotal(local_index.handle());
        remote_index.ntotal = omni_index_ntotal(remote_index.handle());
        // searches of one index with one k share ONE pass over that index (omni_index_search_batch_prefix_dev, <= 64 queries per pass):
        // every query still sees only the rows of its own turn, but the database is read once per micro-batch instead of once per frame
        struct Chunk { IndexFlatIP* index; int k; std::vector<size_t> js; size_t off_i, off_d; };
        std::vector<Chunk> chunks;
        size_t need = 0;
        for (size_t j = 0; j < searches.size(); ++j) {
            if (searches[j].n_limit <= 0) continue;
            const int k = SEARCH_NEAREST_NUM + searches[j].max_index;
            Chunk* c = nullptr;
            for (auto& cc : chunks) if (cc.index == searches[j].index && cc.k == k && cc.js.size() < 64) c = &cc;
            if (!c) { chunks.push_back({searches[j].index, k, {}, 0, 0}); c = &chunks.back(); }
            c->js.push_back(j);
        }
        for (auto& c : chunks) { c.off_i = need; c.off_d = need + c.js.size() * c.k * 8; need += c.js.size() * c.k * 12; }
        pb_.need = need;
        if (need) {
            if (batch_buf_bytes_ < need) {
                if (batch_buf_) omni_dev_free(ctx_.get(), batch_buf_);
                batch_buf_ = static_cast<char*>(omni_dev_alloc(ctx_.get(), need));
                if (!batch_buf_) throw std::runtime_error(std::string("omni_dev_alloc: ") + omni_last_error());
                batch_buf_bytes_ = need;
            }
            if (raw_pinned_bytes_ < need) {
                if (raw_pinned_) omni_host_free(raw_pinned_);
                raw_pinned_ = static_cast<char*>(omni_host_alloc(need));
                if (!raw_pinned_) { raw_pinned_bytes_ = 0; throw std::runtime_error(std::string("omni_host_alloc: ") + omni_last_error()); }
                raw_pinned_bytes_ = need;
            }
            for (auto& c : chunks) {
                std::vector<int64_t> rows_idx, limits;
                for (size_t pos = 0; pos < c.js.size(); ++pos) {
                    const BatchSearch& sj = searches[c.js[pos]];
                    rows_idx.push_back((int64_t)sj.row); limits.push_back(sj.n_limit);
                    pb_.where[c.js[pos]] = {c.off_i + pos * c.k * 8, c.off_d + pos * c.k * 4};
                }
                check(omni_index_search_batch_prefix_dev(c.index->handle(), (int)c.js.size(), rows_dev, rows_idx.data(), c.k, limits.data(),
                                                         reinterpret_cast<float*>(batch_buf_ + c.off_d), reinterpret_cast<int64_t*>(batch_buf_ + c.off_i)),
                      "omni_index_search_batch_prefix_dev");
            }
            check(omni_memcpy_d2h_async(ctx_.get(), raw_pinned_, batch_buf_, need), "on_images_recv_batch fetch");      // (end_images_batch waits for it)
        }
        } catch (...) { rollback_batch(); throw; }
    }

public:
    std::vector<LoopCandidate> end_images_batch() {
        if (!pb_.active) throw std::logic_error("end_images_batch without a batch");
        const std::vector<FisheyeFrameDescriptor>& frames = *pb_.frames;
        try {
            if (pb_.need || pb_.any_add || pb_.own_rows) check(omni_ctx_sync(ctx_.get()), "omni_ctx_sync");      // the only synchronisation
        } catch (...) { rollback_batch(); throw; }
        deferred_.clear();
        for (size_t j = 0; j < pb_.searches.size(); ++j) {
            const BatchSearch& sj = pb_.searches[j];
            Deferred d; d.ntotal = sj.n_limit;
            const int k = SEARCH_NEAREST_NUM + sj.max_index;
            d.D.assign(k, -3.402823466e+38f); d.I.assign(k, -1);
            if (pb_.where[j].first != SIZE_MAX) {
                std::memcpy(d.I.data(), raw_pinned_ + pb_.where[j].first, (size_t)k * 8);
                std::memcpy(d.D.data(), raw_pinned_ + pb_.where[j].second, (size_t)k * 4);
            }
            deferred_.push_back(std::move(d));
        }
        if (pb_.own_rows) { omni_dev_free(ctx_.get(), pb_.own_rows); pb_.own_rows = nullptr; }
        pb_.active = false;
        // replay the decision rules frame by frame on the fetched results
        replaying_ = true; sim_local_ = pb_.start_local; sim_remote_ = pb_.start_remote; next_deferred_ = next_row_id_ = 0;
        std::vector<LoopCandidate> out;
        std::vector<FisheyeFrameDescriptor>* movable = pb_.movable ? &pb_.owned : nullptr;
        try {
            for (size_t fi = 0; fi < frames.size(); ++fi) { move_src_ = movable ? &(*movable)[fi] : nullptr; out.push_back(on_image_recv(frames[fi])); }
            move_src_ = nullptr;
        } catch (...) { replaying_ = false; move_src_ = nullptr; throw; }
        replaying_ = false;
        if (next_deferred_ != deferred_.size() || next_row_id_ != row_ids_.size()) throw std::logic_error("on_images_recv_batch: plan and replay diverged");
        return out;
    }

    LoopCandidate on_image_recv(const FisheyeFrameDescriptor& f_in) {            // :11-137
        if (!replaying_ && malformed(f_in)) { FisheyeFrameDescriptor c = f_in; sanitise(c); return on_image_recv(c); }
        LoopCandidate r;
        if (f_in.images.empty()) return r;
        const int drone_id = f_in.drone_id;
        if (drone_id != self_id && database_size() == 0) return r;                // :36-38
        const bool new_node = all_nodes.find(drone_id) == all_nodes.end();
        all_nodes.insert(drone_id);
        int dir_count = 0;
        for (auto& img : f_in.images) if (img.landmark_num > 0) ++dir_count;
        if (dir_count < MIN_DIRECTION_LOOP) return r;                             // :60-63
        if (f_in.landmark_num < MIN_LOOP_NUM) return r;                           // :65
        bool init_mode = false;
        if (drone_id != self_id) init_mode = inter_drone_loop_count[{drone_id, self_id}] < inter_drone_init_frames;   // :67-72
        const bool nonkeyframe = f_in.prevent_adding_db;
        const FisheyeFrameDescriptor* fp = &f_in;          // re-pointed to the database's copy: the frame may have been MOVED into it
        if (!nonkeyframe || new_node) { fp = add_to_database(f_in); r.added = true; }                               // :89-94
        const FisheyeFrameDescriptor& f = *fp;
        if (database_size() > MATCH_INDEX_DIST || init_mode || drone_id != self_id) {                               // :98
            r.queried = true;
            int direction_new = stereo_fisheye ? 1 : 0, direction_old = -1, image_id = -1;
            double distance = -1;
            const FisheyeFrameDescriptor* old = query_fisheyeframe_from_database(f, init_mode, nonkeyframe, direction_new, direction_old, image_id, distance);
            if (direction_old >= 0 && old) {
                r.found = true; r.old_msg_id = old->msg_id; r.image_id = image_id; r.direction_new = direction_new; r.direction_old = direction_old; r.distance = distance;
                bool success = false;
                if (old->drone_id == self_id) success = compute_loop && compute_loop(f, *old, direction_new, direction_old, init_mode);       // :110-111
                else if (f.drone_id == self_id) success = compute_loop && compute_loop(*old, f, direction_old, direction_new, init_mode);    // :114-115
                if (success) {                                                                                                                // :826-827
                    ++inter_drone_loop_count[{f.drone_id, old->drone_id}];
                    ++inter_drone_loop_count[{old->drone_id, f.drone_id}];
                    r.loop = true;
                }
            }
        }
        return r;
    }

    // an image that claims landmarks but whose global descriptor is not 4096 floats long
    static bool malformed(const FisheyeFrameDescriptor& f) {
        for (auto& img : f.images) if (img.landmark_num > 0 && img.image_desc.size() != 4096) return true;
        return false;
    }
    static void sanitise(FisheyeFrameDescriptor& f) {
        for (auto& img : f.images) if (img.landmark_num > 0 && img.image_desc.size() != 4096) { f.landmark_num -= img.landmark_num; img.landmark_num = 0; }
    }

    int self_id;
    IndexFlatIP local_index, remote_index;
    std::map<int, int64_t> imgid2fisheye;
    std::map<int, int> imgid2dir;
    std::map<int64_t, FisheyeFrameDescriptor> fisheyeframe_database;
    std::map<std::pair<int, int>, int> inter_drone_loop_count;
    std::set<int> all_nodes;

private:
    struct Deferred { std::vector<float> D; std::vector<int64_t> I; int64_t ntotal = 0; };
    Context& ctx_;
    bool replaying_ = false;
    int64_t sim_local_ = 0, sim_remote_ = 0;
    std::vector<Deferred> deferred_;
    std::vector<int> row_ids_;
    size_t next_deferred_ = 0, next_row_id_ = 0;
    char* batch_buf_ = nullptr;
    size_t batch_buf_bytes_ = 0;
    FisheyeFrameDescriptor* move_src_ = nullptr;
    int add_image(const ImageDescriptor& img) {                                   // :164-173
        if (replaying_) {                                                         // on_images_recv_batch: already appended, in this order
            const int row = row_ids_.at(next_row_id_++);
            if (row >= REMOTE_MAGIN_NUMBER) ++sim_remote_; else ++sim_local_;
            return row;
        }
        if (img.drone_id == self_id) { local_index.add(1, img.image_desc.data()); return (int)local_index.ntotal - 1; }
        remote_index.add(1, img.image_desc.data());
        return (int)remote_index.ntotal - 1 + REMOTE_MAGIN_NUMBER;
    }
    const FisheyeFrameDescriptor* add_to_database(const FisheyeFrameDescriptor& f) {      // :150-162; returns the stored frame
        for (size_t i = 0; i < f.images.size(); ++i)
            if (f.images[i].landmark_num > 0) { int index = add_image(f.images[i]); imgid2fisheye[index] = f.msg_id; imgid2dir[index] = (int)i; }
        FisheyeFrameDescriptor& slot = fisheyeframe_database[f.msg_id];
        if (move_src_ && move_src_ == &f) slot = std::move(*move_src_); else slot = f;
        return &slot;
    }
    int query_index(const ImageDescriptor& img, IndexFlatIP& index, bool remote_db, double thres, int max_index, double& distance) {   // :199-242
        float distances[1000] = {0};
        IndexFlatIP::idx_t labels[1000];
        const int index_offset = remote_db ? REMOTE_MAGIN_NUMBER : 0;
        for (auto& l : labels) l = -1;
        const int search_num = SEARCH_NEAREST_NUM + max_index;
        IndexFlatIP::idx_t ntotal = index.ntotal;
        if (replaying_) {                                                         // on_images_recv_batch: the search ran ahead, over the rows of its turn
            const Deferred& d = deferred_.at(next_deferred_++);
            for (int i = 0; i < search_num; ++i) { distances[i] = d.D[i]; labels[i] = d.I[i]; }
            ntotal = d.ntotal;
        } else {
            index.search(1, img.image_desc.data(), search_num, distances, labels);
        }
        int return_msg_id = -1;
        for (int i = 0; i < search_num; ++i) {
            if (labels[i] < 0) continue;
            if (imgid2fisheye.find((int)labels[i] + index_offset) == imgid2fisheye.end()) continue;
            return_msg_id = (int)labels[i] + index_offset;
            if (labels[i] <= ntotal - max_index && distances[i] > thres) { distance = distances[i]; return return_msg_id; }
        }
        return return_msg_id;                                                     // :241 fall-through (sic)
    }
    int query_from_database(const ImageDescriptor& img, bool init_mode, bool nonkeyframe, double& distance) {    // :176-197
        const double thres = init_mode ? INIT_MODE_PRODUCT_THRES : INNER_PRODUCT_THRES;
        if (img.drone_id == self_id) {
            int _id = query_index(img, remote_index, true, thres, 1, distance);
            if (!nonkeyframe) return query_index(img, local_index, false, thres, MATCH_INDEX_DIST, distance);
            else if (_id != -1) return _id;
        } else {
            return query_index(img, local_index, false, thres, 1, distance);
        }
        return -1;
    }
    const FisheyeFrameDescriptor* query_fisheyeframe_from_database(const FisheyeFrameDescriptor& f, bool init_mode, bool nonkeyframe, int direction_new,
                                                                   int& direction_old, int& image_id, double& distance) {   // :245-287
        direction_old = -1;
        if ((int)f.images.size() <= direction_new || f.images[direction_new].landmark_num <= 0) return nullptr;
        distance = -1;
        int id = query_from_database(f.images[direction_new], init_mode, nonkeyframe, distance);
        if (id != -1 && distance > -1) {
            image_id = id;
            direction_old = imgid2dir[id];
            return &fisheyeframe_database[imgid2fisheye[id]];
        }
        return nullptr;
    }
};

}  // namespace omni
