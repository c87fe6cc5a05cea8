// omni_flatten_*: fisheye -> virtual pinhole views ("flattening") on the GPU: the cv::cuda::remap(INTER_LINEAR) calls of
// FisheyeUndist::undist_all_cuda (swarm_localization/test/fisheye_undist.hpp:57-90; VINS-Fisheye runs the same class in front of swarm_loop,
// SURVEY.md 8f rank 4).  The undistortion maps (generateAllUndistMap, :118-186: one float (x, y) source coordinate per output pixel and
// view) are made on the host (host/fisheye_flatten.hpp) and live in HBM; one launch remaps a batch of fisheye images into all views,
// written back to back so that the result can be handed to omni_cam_enqueue_dev / omni_sp_enqueue_dev without leaving the GPU.
// Interpolation = cv::cuda's LinearFilter with BORDER_CONSTANT(0): floor, four taps weighted (x2-x)(y2-y) ... in float, saturate_cast<uchar>
// (round half to even); the products and sums are rounded one by one (no FMA contraction) so that the numpy oracle reproduces the bytes.
// OpenCV is un-vendored: PARITY UNPINNED.  HBM-bound: 1 output byte + ~4 gathered source bytes + 8 map bytes per pixel.
#include "common.h"

struct omni_flatten {
    omni_ctx* ctx = nullptr;
    int src_w = 0, src_h = 0, n_views = 0;
    std::vector<int> vw, vh;
    std::vector<int64_t> out_off;          // byte offset of view v inside one image's output block
    int64_t out_bytes = 0;                 // per source image
    float* maps = nullptr;                 // all views back to back, [h][w][2]
    int* meta = nullptr;                   // per view: w, h, map offset (in float2), out offset
    std::mutex mu;
};

namespace omni {

__global__ void __launch_bounds__(256)
flatten_remap_kernel(const uint8_t* __restrict__ src, int src_stride, int src_w, int src_h, int64_t src_image_bytes, const float2* __restrict__ maps,
                     const int* __restrict__ meta, int n_views, uint8_t* __restrict__ out, int64_t out_image_bytes) {
    // every product and sum below must be rounded on its own: this file is compiled with -ffp-contract=off (Makefile; HIP's __fmul_rn / __fadd_rn
    // are plain operators that hipcc would otherwise fuse into FMAs -- measured: 1 pixel in a million off by one)
    const int v = blockIdx.y, b = blockIdx.z;
    const int w = meta[4 * v], h = meta[4 * v + 1];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w * h) return;
    const float2 m = maps[meta[4 * v + 2] + i];
    const uint8_t* s = src + (int64_t)b * src_image_bytes;
    const int x1 = (int)floorf(m.x), y1 = (int)floorf(m.y), x2 = x1 + 1, y2 = y1 + 1;
    auto at = [&](int y, int x) -> float { return (x >= 0 && x < src_w && y >= 0 && y < src_h) ? (float)s[(int64_t)y * src_stride + x] : 0.f; };
    const float ax2 = __fsub_rn((float)x2, m.x), ax1 = __fsub_rn(m.x, (float)x1), ay2 = __fsub_rn((float)y2, m.y), ay1 = __fsub_rn(m.y, (float)y1);
    float acc = __fmul_rn(at(y1, x1), __fmul_rn(ax2, ay2));
    acc = __fadd_rn(acc, __fmul_rn(at(y1, x2), __fmul_rn(ax1, ay2)));
    acc = __fadd_rn(acc, __fmul_rn(at(y2, x1), __fmul_rn(ax2, ay1)));
    acc = __fadd_rn(acc, __fmul_rn(at(y2, x2), __fmul_rn(ax1, ay1)));
    const float r = rintf(acc);
    out[(int64_t)b * out_image_bytes + meta[4 * v + 3] + i] = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}

}  // namespace omni

extern "C" {

omni_flatten* omni_flatten_create(omni_ctx* ctx, int src_width, int src_height, int n_views, const int* view_w, const int* view_h, const float* const* map_xy) {
    if (!ctx || !view_w || !view_h || !map_xy || n_views < 1 || n_views > 16 || src_width < 1 || src_height < 1) { omni::set_error("bad argument"); return nullptr; }
    (void)hipSetDevice(ctx->device);
    omni_flatten* f = new omni_flatten();
    f->ctx = ctx; f->src_w = src_width; f->src_h = src_height; f->n_views = n_views;
    std::vector<int> meta(4 * n_views);
    int64_t map_px = 0;
    for (int v = 0; v < n_views; ++v) {
        if (view_w[v] < 1 || view_h[v] < 1 || !map_xy[v]) { omni::set_error("bad view %d", v); delete f; return nullptr; }
        f->vw.push_back(view_w[v]); f->vh.push_back(view_h[v]); f->out_off.push_back(f->out_bytes);
        meta[4 * v] = view_w[v]; meta[4 * v + 1] = view_h[v]; meta[4 * v + 2] = (int)map_px; meta[4 * v + 3] = (int)f->out_bytes;
        map_px += (int64_t)view_w[v] * view_h[v];
        f->out_bytes += (int64_t)view_w[v] * view_h[v];
    }
    bool ok = hipMalloc((void**)&f->maps, (size_t)map_px * 8) == hipSuccess && hipMalloc((void**)&f->meta, meta.size() * 4) == hipSuccess;
    for (int v = 0; ok && v < n_views; ++v)
        ok = hipMemcpyAsync(f->maps + 2 * (int64_t)meta[4 * v + 2], map_xy[v], (size_t)view_w[v] * view_h[v] * 8, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(f->meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (!ok) { omni::set_error("omni_flatten_create: device allocation / upload failed"); omni_flatten_destroy(f); return nullptr; }
    return f;
}

void omni_flatten_destroy(omni_flatten* f) {
    if (!f) return;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->stream);
    if (f->maps) (void)hipFree(f->maps);
    if (f->meta) (void)hipFree(f->meta);
    delete f;
}

int64_t omni_flatten_out_bytes(const omni_flatten* f) { return f ? f->out_bytes : -1; }

int omni_flatten_enqueue_dev(omni_flatten* f, const uint8_t* src_dev, int src_stride, int batch, uint8_t* out_dev) {
    OMNI_REQUIRE(f && src_dev && out_dev && batch >= 1 && src_stride >= f->src_w, OMNI_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(f->mu);
    (void)hipSetDevice(f->ctx->device);
    int max_px = 0;
    for (int v = 0; v < f->n_views; ++v) max_px = std::max(max_px, f->vw[v] * f->vh[v]);
    hipLaunchKernelGGL(omni::flatten_remap_kernel, dim3(omni::cdiv(max_px, 256), f->n_views, batch), dim3(256), 0, f->ctx->stream, src_dev, src_stride,
                       f->src_w, f->src_h, (int64_t)src_stride * f->src_h, reinterpret_cast<const float2*>(f->maps), f->meta, f->n_views, out_dev, f->out_bytes);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

}  // extern "C"
