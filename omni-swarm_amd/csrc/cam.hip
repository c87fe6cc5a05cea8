// omni_cam_*: the CNN + matching part of one fisheye key frame as ONE asynchronous unit -- the device-side work of
//   LoopCam::on_flattened_images / generate_stereo_image_descriptor / extractor_img_desc_deepnet / match_HFNet_local_features
//   (swarm_loop/src/loop_cam.cpp:178-229, 341-523, 525-585, 141-174).
// The reference runs 8 SuperPoint + 4 MobileNetVLAD engine calls and 4 BFMatcher calls strictly one after another, each
// with its own H2D / D2H and a blocking stream sync.  Here enqueue() puts on the GPU, without any host
// synchronisation:  SuperPoint(2n images) -> BF cross-check of the n up/down descriptor sets   on the SuperPoint stream,
//                   MobileNetVLAD(n images)                                                    on the MobileNetVLAD stream,
// plus the D2H copies of every result into ONE pinned host block, and records one event per stream.  wait() blocks on the
// two events and hands out pointers into the pinned block (valid until the next enqueue on this handle).  Several handles
// (each with its own SuperPoint / MobileNetVLAD instance) keep several key frames in flight.
#include "common.h"

struct omni_cam {
    omni_sp* sp = nullptr;
    omni_vlad* vlad = nullptr;
    omni_ctx *c1 = nullptr, *c2 = nullptr;
    int n = 0, n_cap = 0, cams = 2, M = 0, D = 0, out_dim = 0, bf_mode = 0, W = 0, H = 0;   // cams: 2 = up + down camera per direction, 1 = one camera (no stereo match)
      // n: directions of the NEXT enqueue (omni_cam_set_active), n_cap: what the handle and its networks were created for
      // W x H: the size the SuperPoint handle (and MobileNetVLAD) was created for
    int *d_qidx = nullptr, *d_tidx = nullptr, *d_nm = nullptr;
    float* d_dist = nullptr;
    const float *kps_dev = nullptr, *desc_dev = nullptr, *sc_dev = nullptr, *g_dev = nullptr;
    const int* n_dev = nullptr;
    char* host = nullptr;
    size_t off_kps = 0, off_n = 0, off_desc = 0, off_sc = 0, off_g = 0, off_q = 0, off_t = 0, off_d = 0, off_nm = 0, host_bytes = 0;
    hipEvent_t e1 = nullptr, e2 = nullptr, e_up = nullptr;
    uint8_t* d_gray = nullptr;        // staging for omni_cam_enqueue_host: the key frame's images, rows packed to `width`
    size_t d_gray_bytes = 0;
    bool pending = false;
    std::mutex mu;
};

extern "C" {

static omni_cam* cam_create(omni_ctx* sp_ctx, omni_sp* sp, omni_ctx* vlad_ctx, omni_vlad* vlad, int n_dirs, int cams, int max_num, int global_dim, int bf_mode);

omni_cam* omni_cam_create(omni_ctx* sp_ctx, omni_sp* sp, omni_ctx* vlad_ctx, omni_vlad* vlad, int n_dirs, int max_num, int global_dim,
                          int bf_mode) {
    return cam_create(sp_ctx, sp, vlad_ctx, vlad, n_dirs, 2, max_num, global_dim, bf_mode);
}

// CameraConfig::PINHOLE_DEPTH (loop_cam.cpp:190-194, generate_gray_depth_image_descriptor :231-339): ONE camera per image, both networks on every
// image, no up/down match; the landmarks come from the depth image on the host (loop_geometry.hpp fill_depth_landmarks)
omni_cam* omni_cam_create_mono(omni_ctx* sp_ctx, omni_sp* sp, omni_ctx* vlad_ctx, omni_vlad* vlad, int n_images, int max_num, int global_dim) {
    return cam_create(sp_ctx, sp, vlad_ctx, vlad, n_images, 1, max_num, global_dim, 0);
}

static omni_cam* cam_create(omni_ctx* sp_ctx, omni_sp* sp, omni_ctx* vlad_ctx, omni_vlad* vlad, int n_dirs, int cams, int max_num, int global_dim,
                            int bf_mode) {
    if (!sp_ctx || !sp || !vlad_ctx || !vlad) { omni::set_error("null handle"); return nullptr; }
    if (n_dirs < 1 || n_dirs > 64 || max_num < 1 || max_num > 1024 || global_dim < 1) { omni::set_error("bad n_dirs/max_num/global_dim"); return nullptr; }
    if (sp_ctx->device != vlad_ctx->device) { omni::set_error("SuperPoint and MobileNetVLAD contexts are on different devices"); return nullptr; }
    (void)hipSetDevice(sp_ctx->device);
    omni_cam* c = new omni_cam();
    c->sp = sp; c->vlad = vlad; c->c1 = sp_ctx; c->c2 = vlad_ctx; c->n = c->n_cap = n_dirs; c->cams = cams; c->M = max_num; c->D = omni_sp_desc_dim(sp);
    c->out_dim = global_dim; c->bf_mode = bf_mode;
    (void)omni_sp_image_size(sp, &c->W, &c->H);
    const size_t n = n_dirs, M = max_num, D = c->D, ni = (size_t)cams * n;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o = 0;
    c->off_kps = o; o += al(ni * M * 2 * 4);
    c->off_n = o;   o += al(ni * 4);
    c->off_desc = o; o += al(ni * M * D * 4);
    c->off_sc = o;  o += al(ni * M * 4);
    c->off_g = o;   o += al(n * (size_t)global_dim * 4);
    c->off_q = o;   o += al(n * M * 4);
    c->off_t = o;   o += al(n * M * 4);
    c->off_d = o;   o += al(n * M * 4);
    c->off_nm = o;  o += al(n * 4);
    c->host_bytes = o;
    bool ok = hipHostMalloc((void**)&c->host, o, hipHostMallocDefault) == hipSuccess &&
              hipMalloc((void**)&c->d_qidx, n * M * 4) == hipSuccess && hipMalloc((void**)&c->d_tidx, n * M * 4) == hipSuccess &&
              hipMalloc((void**)&c->d_dist, n * M * 4) == hipSuccess && hipMalloc((void**)&c->d_nm, n * 4) == hipSuccess &&
              hipEventCreateWithFlags(&c->e1, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->e2, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->e_up, hipEventDisableTiming) == hipSuccess &&
              omni_sp_dev_outputs(sp, &c->kps_dev, &c->n_dev, &c->desc_dev, &c->sc_dev) == OMNI_OK &&
              omni_vlad_dev_output(vlad, &c->g_dev) == OMNI_OK;
    if (!ok) { omni::set_error("omni_cam_create: allocation failed"); omni_cam_destroy(c); return nullptr; }
    memset(c->host, 0, o);
    return c;
}

void omni_cam_destroy(omni_cam* c) {
    if (!c) return;
    (void)hipSetDevice(c->c1->device);
    (void)hipStreamSynchronize(c->c1->stream);
    (void)hipStreamSynchronize(c->c2->stream);
    void* ptrs[] = {c->d_qidx, c->d_tidx, c->d_dist, c->d_nm, c->d_gray};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->host) (void)hipHostFree(c->host);
    if (c->e1) (void)hipEventDestroy(c->e1);
    if (c->e2) (void)hipEventDestroy(c->e2);
    if (c->e_up) (void)hipEventDestroy(c->e_up);
    delete c;
}

static int cam_enqueue_locked(omni_cam* c, const uint8_t* gray_dev, int stride, int fisheye_mask);

int omni_cam_enqueue_dev(omni_cam* c, const uint8_t* gray_dev, int stride, int fisheye_mask) {
    omni::TraceRange trace_range("omni_cam_enqueue_dev");
    OMNI_REQUIRE(c && gray_dev, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->c1->device);
    return cam_enqueue_locked(c, gray_dev, stride, fisheye_mask);
}

int omni_cam_enqueue_host(omni_cam* c, const uint8_t* gray_host, int stride, int width, int height, int fisheye_mask) {
    omni::TraceRange trace_range("omni_cam_enqueue_host (upload + unit)");
    OMNI_REQUIRE(c && gray_host, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(width > 0 && height > 0 && stride >= width, OMNI_ERR_INVALID, "bad image geometry %dx%d stride %d", width, height, stride);
    // the networks read cams * n images of THEIR size from the staging buffer: any other size would run them past its end
    OMNI_REQUIRE(width == c->W && height == c->H, OMNI_ERR_INVALID, "omni_cam_enqueue_host: images are %dx%d but the networks were created for %dx%d", width, height, c->W, c->H);
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->c1->device);
    const size_t need = (size_t)c->cams * c->n * width * height;
    if (c->d_gray_bytes < need) {
        (void)hipStreamSynchronize(c->c1->stream);
        (void)hipStreamSynchronize(c->c2->stream);
        if (c->d_gray) (void)hipFree(c->d_gray);
        c->d_gray = nullptr; c->d_gray_bytes = 0;
        OMNI_HIP_TRY(hipMalloc((void**)&c->d_gray, need));
        c->d_gray_bytes = need;
    }
    // the reference uploads one image per engine call and blocks (tensorrt_generic.cpp:58-75); here the key frame's 2n images go up as one
    // asynchronous copy on the SuperPoint stream (pinned source: the copy engine runs it next to the other pipelines' kernels) and the
    // MobileNetVLAD stream waits for it on the device
    // (two copies for a stereo rig: MobileNetVLAD only reads the up cameras' images -- the first half -- and starts as soon as they are up, while the
    // down cameras' half is still on the bus; SuperPoint's stream carries both copies and so waits for all of it)
    // (packed rows -- stride == width, what the key-frame pipeline hands over -- go up as plain 1-D copies: 56 GB/s against 50 GB/s for the 2-D form of the same
    // bytes, tools/probes/h2d_probe.hip)
    const size_t rows_up = (size_t)c->n * height, rows_all = (size_t)c->cams * c->n * height;
    auto upload = [&](size_t row0, size_t rows) -> hipError_t {
        if (stride == width) return hipMemcpyAsync(c->d_gray + row0 * width, gray_host + row0 * width, rows * width, hipMemcpyHostToDevice, c->c1->stream);
        return hipMemcpy2DAsync(c->d_gray + row0 * width, (size_t)width, gray_host + row0 * stride, (size_t)stride, (size_t)width, rows, hipMemcpyHostToDevice, c->c1->stream);
    };
    OMNI_HIP_TRY(upload(0, rows_up));
    OMNI_HIP_TRY(hipEventRecord(c->e_up, c->c1->stream));
    OMNI_HIP_TRY(hipStreamWaitEvent(c->c2->stream, c->e_up, 0));
    if (rows_all > rows_up) OMNI_HIP_TRY(upload(rows_up, rows_all - rows_up));
    return cam_enqueue_locked(c, c->d_gray, width, fisheye_mask);
}

// The same from SEGMENTS of host memory: the up cameras' n images are the concatenation of n_up parts (up[i]: up_images[i] images, rows packed), the down
// cameras' likewise (n_down = 0 for a mono handle).  What a key-frame loop needs to cut a run of key frames into units of ITS choice out of blocks laid out
// for another unit size (KeyframePipeline::run): every part is one asynchronous 1-D copy; MobileNetVLAD starts behind the up cameras' parts.
int omni_cam_enqueue_host_parts(omni_cam* c, const uint8_t* const* up, const int* up_images, int n_up, const uint8_t* const* down, const int* down_images, int n_down,
                                int width, int height, int fisheye_mask) {
    omni::TraceRange trace_range("omni_cam_enqueue_host_parts (upload + unit)");
    OMNI_REQUIRE(c && up && up_images && n_up > 0 && (n_down == 0 || (down && down_images)), OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(width == c->W && height == c->H, OMNI_ERR_INVALID, "omni_cam_enqueue_host_parts: images are %dx%d but the networks were created for %dx%d", width, height, c->W, c->H);
    int nu = 0, nd = 0;
    for (int i = 0; i < n_up; ++i) { OMNI_REQUIRE(up[i] && up_images[i] > 0, OMNI_ERR_INVALID, "omni_cam_enqueue_host_parts: empty part"); nu += up_images[i]; }
    for (int i = 0; i < n_down; ++i) { OMNI_REQUIRE(down[i] && down_images[i] > 0, OMNI_ERR_INVALID, "omni_cam_enqueue_host_parts: empty part"); nd += down_images[i]; }
    OMNI_REQUIRE(nu == c->n && nd == (c->cams - 1) * c->n, OMNI_ERR_INVALID, "omni_cam_enqueue_host_parts: %d + %d images for a unit of %d x %d", nu, nd, c->cams, c->n);
    std::lock_guard<std::mutex> lk(c->mu);
    (void)hipSetDevice(c->c1->device);
    const size_t img = (size_t)width * height, need = (size_t)c->cams * c->n * img;
    if (c->d_gray_bytes < need) {
        (void)hipStreamSynchronize(c->c1->stream);
        (void)hipStreamSynchronize(c->c2->stream);
        if (c->d_gray) (void)hipFree(c->d_gray);
        c->d_gray = nullptr; c->d_gray_bytes = 0;
        OMNI_HIP_TRY(hipMalloc((void**)&c->d_gray, need));
        c->d_gray_bytes = need;
    }
    size_t at = 0;
    for (int i = 0; i < n_up; ++i) { OMNI_HIP_TRY(hipMemcpyAsync(c->d_gray + at, up[i], up_images[i] * img, hipMemcpyHostToDevice, c->c1->stream)); at += up_images[i] * img; }
    OMNI_HIP_TRY(hipEventRecord(c->e_up, c->c1->stream));
    OMNI_HIP_TRY(hipStreamWaitEvent(c->c2->stream, c->e_up, 0));
    for (int i = 0; i < n_down; ++i) { OMNI_HIP_TRY(hipMemcpyAsync(c->d_gray + at, down[i], down_images[i] * img, hipMemcpyHostToDevice, c->c1->stream)); at += down_images[i] * img; }
    return cam_enqueue_locked(c, c->d_gray, width, fisheye_mask);
}

static int cam_enqueue_locked(omni_cam* c, const uint8_t* gray_dev, int stride, int fisheye_mask) {
    const int n = c->n, M = c->M, D = c->D, ni = c->cams * c->n;
    int rc;
    // images 0..n-1 = "up" (main) camera of each direction, n..2n-1 = "down" camera (loop_cam.cpp:350-351)
    if ((rc = omni_sp_enqueue_dev(c->sp, gray_dev, stride, ni, fisheye_mask))) return rc;
    if ((rc = omni_vlad_enqueue_dev(c->vlad, gray_dev, stride, n, fisheye_mask))) return rc;       // main camera only (:553-556)
    // match_HFNet_local_features: up = query, down = train (:147-150); pair p = direction p
    if (c->cams == 2 &&
        (rc = omni_bf_match_batched_dev(c->c1, n, M, D, c->bf_mode, c->desc_dev, (int64_t)M * D, c->n_dev,
                                        c->desc_dev + (size_t)n * M * D, (int64_t)M * D, c->n_dev + n, c->d_qidx, c->d_tidx, c->d_dist, c->d_nm)))
        return rc;
    hipStream_t s1 = c->c1->stream, s2 = c->c2->stream;
    char* h = c->host;
    OMNI_HIP_TRY(hipMemcpyAsync(h + c->off_kps, c->kps_dev, (size_t)ni * M * 2 * 4, hipMemcpyDeviceToHost, s1));
    OMNI_HIP_TRY(hipMemcpyAsync(h + c->off_n, c->n_dev, (size_t)ni * 4, hipMemcpyDeviceToHost, s1));
    OMNI_HIP_TRY(hipMemcpyAsync(h + c->off_desc, c->desc_dev, (size_t)ni * M * D * 4, hipMemcpyDeviceToHost, s1));
    OMNI_HIP_TRY(hipMemcpyAsync(h + c->off_sc, c->sc_dev, (size_t)ni * M * 4, hipMemcpyDeviceToHost, s1));
    if (c->cams == 2) {                                                       // (one camera: n_matches stays 0, as the block was zeroed at creation)
        OMNI_HIP_TRY(hipMemcpyAsync(h + c->off_q, c->d_qidx, (size_t)n * M * 4, hipMemcpyDeviceToHost, s1));
        OMNI_HIP_TRY(hipMemcpyAsync(h + c->off_t, c->d_tidx, (size_t)n * M * 4, hipMemcpyDeviceToHost, s1));
        OMNI_HIP_TRY(hipMemcpyAsync(h + c->off_d, c->d_dist, (size_t)n * M * 4, hipMemcpyDeviceToHost, s1));
        OMNI_HIP_TRY(hipMemcpyAsync(h + c->off_nm, c->d_nm, (size_t)n * 4, hipMemcpyDeviceToHost, s1));
    }
    OMNI_HIP_TRY(hipEventRecord(c->e1, s1));
    OMNI_HIP_TRY(hipMemcpyAsync(h + c->off_g, c->g_dev, (size_t)n * c->out_dim * 4, hipMemcpyDeviceToHost, s2));
    OMNI_HIP_TRY(hipEventRecord(c->e2, s2));
    c->pending = true;
    return OMNI_OK;
}

int omni_cam_order_after(omni_cam* later, omni_cam* earlier, int streams) {
    OMNI_REQUIRE(later && earlier, OMNI_ERR_INVALID, "null argument");
    if (later == earlier || streams <= 0) return OMNI_OK;
    OMNI_REQUIRE(later->c1->device == earlier->c1->device, OMNI_ERR_INVALID, "omni_cam_order_after: two units of one device");
    (void)hipSetDevice(later->c1->device);
    hipEvent_t ev = omni_sp_convs_event(earlier->sp);
    OMNI_REQUIRE(ev, OMNI_ERR_INVALID, "omni_cam_order_after: no event");
    OMNI_HIP_TRY(hipStreamWaitEvent(later->c1->stream, ev, 0));          // (an event that was never recorded does not block)
    if (streams >= 2 && later->c2 != later->c1) OMNI_HIP_TRY(hipStreamWaitEvent(later->c2->stream, ev, 0));
    return OMNI_OK;
}

// a unit smaller than the handle was created for (a partly filled micro-batch that must not wait any longer): the next enqueues read cams * n_dirs
// images -- up cameras first, the down cameras right behind them -- and every array of omni_cam_result has that leading dimension
int omni_cam_set_active(omni_cam* c, int n_dirs) {
    OMNI_REQUIRE(c, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    OMNI_REQUIRE(n_dirs >= 1 && n_dirs <= c->n_cap, OMNI_ERR_INVALID, "omni_cam_set_active: %d directions, the handle holds 1..%d", n_dirs, c->n_cap);
    OMNI_REQUIRE(!c->pending, OMNI_ERR_INVALID, "omni_cam_set_active with a unit in flight (omni_cam_wait first)");
    c->n = n_dirs;
    return OMNI_OK;
}

// non-blocking: *ready = 1 when omni_cam_wait would return at once (or nothing is pending)
int omni_cam_ready(omni_cam* c, int* ready) {
    OMNI_REQUIRE(c && ready, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    *ready = 1;
    if (!c->pending) return OMNI_OK;
    (void)hipSetDevice(c->c1->device);
    for (hipEvent_t e : {c->e1, c->e2}) {
        const hipError_t r = hipEventQuery(e);
        if (r == hipErrorNotReady) { *ready = 0; return OMNI_OK; }
        if (r != hipSuccess) { omni::set_error("hipEventQuery failed: %s", hipGetErrorString(r)); return OMNI_ERR_HIP; }
    }
    return OMNI_OK;
}

int omni_cam_wait(omni_cam* c, omni_cam_result* out) {
    omni::TraceRange trace_range("omni_cam_wait");
    OMNI_REQUIRE(c && out, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    OMNI_REQUIRE(c->pending, OMNI_ERR_INVALID, "omni_cam_wait without a pending omni_cam_enqueue_dev");
    (void)hipSetDevice(c->c1->device);
    OMNI_HIP_TRY(hipEventSynchronize(c->e1));
    OMNI_HIP_TRY(hipEventSynchronize(c->e2));
    c->pending = false;
    const char* h = c->host;
    out->n_dirs = c->n; out->max_num = c->M; out->desc_dim = c->D; out->global_dim = c->out_dim; out->n_images = c->cams * c->n;
    out->kps_xy = reinterpret_cast<const float*>(h + c->off_kps);
    out->n_kps = reinterpret_cast<const int*>(h + c->off_n);
    out->desc = reinterpret_cast<const float*>(h + c->off_desc);
    out->scores = reinterpret_cast<const float*>(h + c->off_sc);
    out->global_desc = reinterpret_cast<const float*>(h + c->off_g);
    out->match_up = reinterpret_cast<const int*>(h + c->off_q);
    out->match_down = reinterpret_cast<const int*>(h + c->off_t);
    out->match_dist = reinterpret_cast<const float*>(h + c->off_d);
    out->n_matches = reinterpret_cast<const int*>(h + c->off_nm);
    return OMNI_OK;
}

}  // extern "C"
