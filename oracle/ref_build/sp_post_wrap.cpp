// TEST INFRASTRUCTURE ONLY (oracle/).  C entry points around the reference's OWN post-processing text -- SuperPointTensorRT::getKeyPoints,
// SuperPointTensorRT::computeDescriptors, pt_conf_comp and NMS2 (swarm_loop/src/superpoint_tensorrt.cpp:164-310: from getKeyPoints to the end
// of the file), extracted at build time into oracle/_ref/sp_post_snippet.inc and compiled VERBATIM (oracle/Makefile, _ref/libref_sp_post.so).
//   * torch:: is the REAL libtorch of this image (the wheel's C++ headers and libtorch_cpu.so): from_blob, the grid arithmetic,
//     grid_sampler(…, 0, 0, 0), norm(desc, 2, 1), div, transpose run as the reference wrote them;
//   * cv:: is the stand-in of nms2_shim.h (OpenCV 3.4 is absent);
//   * Eigen:: is a stand-in for the three expressions of :219-222 (Map, rowwise() - row vector, matrix product; plain loops, k ascending):
//     the PCA projection's summation order is therefore NOT pinned (float32, 256 terms: ~1e-7); everything in front of it is;
//   * the class is reduced to the members those two functions read (width, height, max_num, enable_perf, pca_comp_T, pca_mean); TicToc is a stub.
// USE_PCA is defined at the top of the reference file (:7); REF_SP_NO_PCA builds the variant without it (the 256-d descriptors in front of the
// projection), so that the torch part can be compared bit for bit.
#include <torch/torch.h>

#include <iostream>

#include "nms2_shim.h"

namespace Eigen {
enum { Dynamic = -1, ColMajor = 0, RowMajor = 1 };
template <typename T, int R, int C, int O = ColMajor>
class Matrix {
public:
    Matrix() : r_(0), c_(0) {}
    Matrix(long r, long c) : r_(r), c_(c), d_((size_t)r * c) {}
    long rows() const { return r_; }
    long cols() const { return c_; }
    long size() const { return r_ * c_; }
    T* data() { return d_.data(); }
    const T* data() const { return d_.data(); }
    T& operator()(long i, long j) { return d_[(size_t)i * c_ + j]; }          // row-major storage whatever O says: only this file looks inside
    const T& operator()(long i, long j) const { return d_[(size_t)i * c_ + j]; }
private:
    long r_, c_;
    std::vector<T> d_;
};
typedef Matrix<float, Dynamic, Dynamic> MatrixXf;
typedef Matrix<float, 1, Dynamic> RowVectorXf;
typedef Matrix<float, Dynamic, Dynamic, RowMajor> RowMatrixXf;
template <typename M> class Map;
template <>
class Map<RowMatrixXf> {
public:
    Map(float* p, long r, long c) : p_(p), r_(r), c_(c) {}
    struct Rowwise {
        const Map& m;
        RowMatrixXf operator-(const RowVectorXf& v) const {
            RowMatrixXf o(m.r_, m.c_);
            for (long i = 0; i < m.r_; ++i)
                for (long j = 0; j < m.c_; ++j) o(i, j) = m.p_[i * m.c_ + j] - v(0, j);
            return o;
        }
    };
    Rowwise rowwise() const { return Rowwise{*this}; }
    float* data() { return p_; }
    long rows() const { return r_; }
    long cols() const { return c_; }
private:
    float* p_; long r_, c_;
};
inline RowMatrixXf operator*(const RowMatrixXf& a, const MatrixXf& b) {
    RowMatrixXf o(a.rows(), b.cols());
    for (long i = 0; i < a.rows(); ++i)
        for (long j = 0; j < b.cols(); ++j) {
            float s = 0.f;
            for (long k = 0; k < a.cols(); ++k) s += a(i, k) * b(k, j);
            o(i, j) = s;
        }
    return o;
}
}  // namespace Eigen

struct TicToc { double toc() { return 0.0; } };

void NMS2(std::vector<cv::Point2f> det, cv::Mat conf, std::vector<cv::Point2f>& pts, int border, int dist_thresh, int img_width, int img_height, int max_num);

class SuperPointTensorRT {
public:
    Eigen::MatrixXf pca_comp_T;
    Eigen::RowVectorXf pca_mean;
    int width = 0, height = 0;
    float thres = 0.015f;
    bool enable_perf = false;
    int max_num = 200;
    void getKeyPoints(const cv::Mat& prob, float threshold, std::vector<cv::Point2f>& keypoints);
    void computeDescriptors(const torch::Tensor& mProb, const torch::Tensor& mDesc, const std::vector<cv::Point2f>& keypoints, std::vector<float>& local_descriptors);
};

#ifndef REF_SP_NO_PCA
#define USE_PCA                                  // superpoint_tensorrt.cpp:7
#endif
#include REF_SP_POST_SNIPPET

#ifdef REF_SP_NO_PCA
#define ENTRY(name) name##_nopca
#else
#define ENTRY(name) name
#endif

// prob [H][W] -> key points (x, y) in the reference's order; returns their number
extern "C" int ENTRY(ref_sp_get_keypoints)(const float* prob, int W, int H, float thres, int max_num, float* out_xy) {
    SuperPointTensorRT sp;
    sp.width = W; sp.height = H; sp.max_num = max_num;
    cv::Mat p(H, W, CV_32F);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) p.at<float>(y, x) = prob[(size_t)y * W + x];
    std::vector<cv::Point2f> kps;
    sp.getKeyPoints(p, thres, kps);
    for (size_t i = 0; i < kps.size(); ++i) { out_xy[2 * i] = kps[i].x; out_xy[2 * i + 1] = kps[i].y; }
    return (int)kps.size();
}

// desc [1][256][Hc][Wc], key points (x, y) -> n x d_out floats (d_out = 64 with the PCA, 256 without); returns the number of floats written
extern "C" int ENTRY(ref_sp_compute_descriptors)(const float* desc, int Hc, int Wc, const float* kps_xy, int n, int W, int H, const float* pca_comp /* [d_out][256] */,
                                                 const float* pca_mean /* [256] */, int d_out, float* out) {
    SuperPointTensorRT sp;
    sp.width = W; sp.height = H;
#ifndef REF_SP_NO_PCA
    sp.pca_comp_T = Eigen::MatrixXf(256, d_out);                    // pca_comp_T = load_csv_mat_eigen(_pca_comp).transpose()  (:110)
    for (int j = 0; j < d_out; ++j)
        for (int c = 0; c < 256; ++c) sp.pca_comp_T(c, j) = pca_comp[(size_t)j * 256 + c];
    sp.pca_mean = Eigen::RowVectorXf(1, 256);
    for (int c = 0; c < 256; ++c) sp.pca_mean(0, c) = pca_mean[c];
#else
    (void)pca_comp; (void)pca_mean; (void)d_out;
#endif
    std::vector<cv::Point2f> kps(n);
    for (int i = 0; i < n; ++i) kps[i] = cv::Point2f(kps_xy[2 * i], kps_xy[2 * i + 1]);
    const torch::Tensor md = torch::from_blob(const_cast<float*>(desc), {1, 256, Hc, Wc}, torch::kFloat);
    std::vector<float> local;
    sp.computeDescriptors(torch::Tensor(), md, kps, local);
    for (size_t i = 0; i < local.size(); ++i) out[i] = local[i];
    return (int)local.size();
}
