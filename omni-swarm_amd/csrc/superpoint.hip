// omni_sp_*: drop-in for SuperPointTensorRT (swarm_loop/include/swarm_loop/superpoint_tensorrt.h:12-29,
// swarm_loop/src/superpoint_tensorrt.cpp:91-230) including the TensorRT engine it wraps (the graph of
// swarm_loop/superpoint.ipynb:135-205) and the runner plumbing of tensorrt_generic.cpp:14-120.
//
// HBM layout per handle (sized for max_batch images, all NHWC, element type = precision):
//   a1a [B][H][W][64]   a1b [B][H/2][W/2][64]   a2a  a2b [B][H/4][W/4][64]   a3a [..][128]  a3b [B][H/8][W/8][128]
//   a4a a4b [..][128]   heads [B][Hc][Wc][512] (cPa | cDa, one fused N=512 conv)   draw [B][Hc][Wc][256] fp32
//   semi [B][H][W] fp32
#include <algorithm>
#include <vector>

#include "config.h"
#include "conv.h"
#include "sp_post.h"

namespace {
struct LayerDef { const char* name; int cin, cout, ks; };
const LayerDef kLayers[OMNI_SP_NUM_LAYERS] = {
    {"conv1a", 1, 64, 3},  {"conv1b", 64, 64, 3},   {"conv2a", 64, 64, 3},   {"conv2b", 64, 64, 3},
    {"conv3a", 64, 128, 3}, {"conv3b", 128, 128, 3}, {"conv4a", 128, 128, 3}, {"conv4b", 128, 128, 3},
    {"convPa", 128, 256, 3}, {"convPb", 256, 65, 1}, {"convDa", 128, 256, 3}, {"convDb", 256, 256, 1}};
enum { L1A = 0, L1B, L2A, L2B, L3A, L3B, L4A, L4B, LPA, LPB, LDA, LDB };
// profiling stages
enum { ST_CONV1A = 0, ST_CONV1B, ST_CONV2A, ST_CONV2B, ST_CONV3A, ST_CONV3B, ST_CONV4A, ST_CONV4B, ST_HEADS_A,
       ST_DET_TAIL, ST_DESC_TAIL, ST_POST, ST_COUNT };
const char* kStageNames[OMNI_SP_NUM_STAGES] = {
    "conv1a", "conv1b+pool", "conv2a", "conv2b+pool", "conv3a", "conv3b+pool", "conv4a", "conv4b", "convPa|convDa",
    "convPb+softmax+d2s", "convDb+l2norm", "nms+topk+describe", "", "", "", ""};
}  // namespace

struct omni_sp {
    omni_ctx* ctx = nullptr;
    omni::Config cfg;                        // the switches as they stood when the handle was created (config.h)
    int W = 0, H = 0, Hc = 0, Wc = 0, max_num = 0, max_batch = 0, precision = 0, pca_dim = 0, desc_dim = 256;
    float thres = 0.f;
    size_t esz = 4;
    // weights
    void* wpk[OMNI_SP_NUM_LAYERS] = {};     // packed MFMA weights (L1B..L4B, LDB) ; heads_a fused in wpk[LPA]
    float* bias[OMNI_SP_NUM_LAYERS] = {};
    float* bias_s[OMNI_SP_NUM_LAYERS] = {};  // OMNI_PREC_SPLIT: conv_split_act_scale() * bias (the split layers write scaled activations)
    float winv[OMNI_SP_NUM_LAYERS] = {};     // OMNI_PREC_SPLIT: 2^-k of the layer's weight scaling
    float* w1a = nullptr;                    // [64][9]
    float* wPbT = nullptr;                   // [256][65]
    float *wPbA = nullptr, *wPbDust = nullptr; // convPb in MFMA A-fragment order + the dustbin row
    void* wPbA16 = nullptr;                    // convPb as split-fp16 A fragments (detector_head_mfma16_kernel); OMNI_DET16=0 keeps the f32 MFMA head
    bool det16 = true;
    bool fused_cand = true;                  // getKeyPoints' threshold inside the detector head's epilogue (OMNI_SP_FUSED_CAND)
    // fp16 path: descriptors are computed only at the four coarse cells around each key point (convdb_sparse_sample) and the dense map `draw`
    // is produced on demand (omni_sp_get_dense) -- OMNI_SP_SPARSE_DESC=0 keeps the dense map in every forward pass (A/B, parity tests)
    bool sparse_desc = true;
    bool sparse_da = true;                   // ... and convDa itself only there too (conv_c128_sparse); OMNI_SP_SPARSE_DA=0: convDa stays dense
    void* headsP = nullptr;                  // [B][Hc][Wc][256] cPa alone (sparse_da passes)
    bool heads_full = true;                  // `heads` holds the fused layer of the last pass (false: OMNI_PREC_SPLIT ran cPa alone + convDa at the key points)
    void* da_compact = nullptr;              // [B][max_num][4][256] fp16: cDa at the corner cells of the key points
    float *cx32 = nullptr, *cy32 = nullptr;  // fp32 paths: the gathered cDa rows / their convDb + norm, [ceil8(B * max_num * 4)][256] each
    bool dense_valid = false, dense_possible = false;   // `draw` holds / `heads` can still produce the dense map of the last forward pass
    int last_batch = 0;
    void* wDbFrag = nullptr;                    // convDb as register-resident fp16 A fragments (fused convDb + L2 norm, fp16 path)
    void* wDbFragHi = nullptr; void* wDbFragLo = nullptr;      // OMNI_PREC_SPLIT: the same as (hi, lo) pairs (convdb_l2norm_split); OMNI_SP_SPLIT_DB=0: exact-f32 convDb
    float* bias_heads = nullptr;             // [512]
    float* lut = nullptr;
    uint16_t* w1a_frag = nullptr;            // conv1a split-fp16 A fragments (fused conv1a+conv1b, fp16 path)
    uint32_t* lut_hl = nullptr;              // u8 -> (half hi, half lo) table
    bool fuse1a = false;
    bool split_fuse1a = false;               // OMNI_PREC_SPLIT: conv1a is built inside conv1b's kernel (OMNI_SPLIT_FUSE1A=0: the separate conv1a_split pass)
    // OMNI_PREC_SPLIT, Winograd F(2x2,3x3) kernels (conv_wino.hip) for the cin = 64 layers: bit 0 = conv1b (needs the conv1a fusion), 1 = conv2a, 2 = conv2b, 3 = conv3a (64 -> 128: two output-channel groups)
    // (OMNI_SPLIT_WINO).  Between two Winograd layers the activation frame is raw-32 instead of split-64 (same geometry, same bytes): raw_1b / raw_2a say
    // what the LAST pass left in a1b / a2a; a_tmp: the converted input of a Winograd layer behind a direct one (mixed configurations only)
    int wino = 0;
    void* wpk_w[OMNI_SP_NUM_LAYERS] = {};
    float winv_w[OMNI_SP_NUM_LAYERS] = {};
    bool raw_1b = false, raw_2a = false, raw_2b = false;
    void* a_tmp = nullptr;
    bool mask_skip_cal_fused = false;        // ... and which of the two the mask's constant region was calibrated with
    float* pca_compT = nullptr;
    float* pca_mean = nullptr;
    // activations
    void *a1a = nullptr, *a1b = nullptr, *a2a = nullptr, *a2b = nullptr, *a3a = nullptr, *a3b = nullptr, *a4a = nullptr,
         *a4b = nullptr, *heads = nullptr;
    float *draw = nullptr, *semi = nullptr;
    uint8_t* gray_stage = nullptr;           // device copy for host-pointer entry points
    omni::SpPostBuffers pb = {};
    omni::HostBuf hstage;
    omni::DevBuf dense_tmp;
    hipEvent_t ev[OMNI_SP_NUM_STAGES + 1] = {};
    hipEvent_t ev_convs = nullptr;           // recorded behind the last CU-filling kernel of a pass (the detector head): what omni_cam_order_after waits for
    bool perf = false, perf_valid = false;    // omni_sp_set_perf: every pass records its stage events (omni_sp_last_stage_ms)
    int conv_variant = 0;                    // OMNI_CONV_V1=1: generic conv kernel everywhere, 2: v2 persistent kernel (A/B and debugging)
    // The constant region of the fisheye mask (fp16 path; OMNI_SP_MASK_SKIP=0 switches it off).  LoopCam blanks rows [3H/4, 3H/4 + H/4) of every image before the
    // network sees it (loop_cam.cpp:536-539): a few pixels inside that band -- one per 3x3 convolution, doubling with every pool -- every
    // activation is ONE vector per layer, whatever the image (its whole receptive field is zeros; the arithmetic of a pixel does not depend on
    // where it is).  The vectors are read once from a pass over an all-zero image (sp_calibrate_mask_skip) and written once into the rectangle
    // of CONV_TH x CONV_TW tiles that lies inside the region, in every image slot of the activation buffers; the persistent cin = 64 kernel then
    // leaves those tiles out of its walk (ConvArgs::skip_*).  Results are bit-identical to the dense pass (tests/test_gpu_mask_skip.py).  A pass
    // without the mask (or on another path) overwrites the rectangles: the next masked pass calibrates again.
    struct MaskSkip {
        int ty0 = 0, ty1 = 0, tx0 = 0, tx1 = 0;      // tile rectangle in the layer's conv-output tile grid (before the pool)
        int oy0 = 0, oy1 = 0, ox0 = 0, ox1 = 0;      // the same rectangle in the layer's output map (after the pool)
        int oh = 0, ow = 0, oc = 0;                  // output map: rows, cols, channels
        int pix_bytes = 0;                           // its layout: bytes per pixel, per row, per image, offset of pixel (0, 0) (fp16: NHWC; split: framed split-64)
        int64_t row_bytes = 0, img_bytes = 0, org_bytes = 0;
        void* vec = nullptr;                         // [pix_bytes]: the constant
        void** map = nullptr;                        // the activation buffer
        double frac = 0.0;                           // the rectangle's share of the layer's tiles (omni_sp_stage_tiles_left_out)
    };
    MaskSkip mskip[6];                       // conv1a (OMNI_PREC_SPLIT, unfused, only), conv1b (+pool), conv2a, conv2b (+pool), conv3a, conv3b (+pool; OMNI_PREC_SPLIT only)
    bool mask_skip = false, mask_skip_ready = false, mask_skip_calibrating = false;
    uint8_t* zero_gray = nullptr;
    size_t zero_gray_bytes = 0;
    std::mutex mu;
};

namespace omni {

static int dev_upload(void** dst, const void* src, size_t bytes, hipStream_t st) {
    OMNI_HIP_TRY(hipMalloc(dst, bytes));
    OMNI_HIP_TRY(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st));
    OMNI_HIP_TRY(hipStreamSynchronize(st));
    return OMNI_OK;
}

// Where every layer's output is constant under the fisheye mask, and the tile rectangle inside it (see omni_sp::MaskSkip): pure integer arithmetic
// on (H, W) and the kernels' tile shapes; k[0] = conv1a (OMNI_PREC_SPLIT only), k[1..4] = conv1b, conv2a, conv2b, conv3a
static void sp_mask_skip_rects(int H, int W, bool split, omni_sp::MaskSkip (&ks)[6]) {
    int m0, m1;
    omni_fisheye_mask_rows(H, 1, &m0, &m1);
    int h = H, w = W;
    for (auto& k : ks) k = omni_sp::MaskSkip{};
    // conv1a's output is relu(bias) on the rows whose three input rows are blanked (the zero padding below the image counts as blanked), in
    // every column (the padding left and right of the image is zeros too)
    int a = m0 + 1, b = (m1 == h) ? h - 1 : m1 - 2, c = 0, d = w - 1;
    if (split && b >= a) {                                 // conv1a_split: 8-row tile rows, the whole width
        omni_sp::MaskSkip& k = ks[0];
        k.ty0 = (a + 7) / 8; k.ty1 = (b + 1) / 8; k.tx0 = 0; k.tx1 = (w + 31) / 32;
        if (k.ty1 <= k.ty0) k.ty0 = k.ty1 = k.tx0 = k.tx1 = 0;
        k.oy0 = k.ty0 * 8; k.oy1 = k.ty1 * 8 < h ? k.ty1 * 8 : h; k.ox0 = 0; k.ox1 = k.ty1 > k.ty0 ? w : 0;
        k.oh = h; k.ow = w; k.oc = 64;
        k.frac = (double)(k.ty1 - k.ty0) / ((h + 7) / 8);
    }
    const bool pool[5] = {true, false, true, false, true};
    const int chans[5] = {64, 64, 64, 128, 128};
    static_assert(CONV_TW == 32, "tile width");
    const int TW = 32;
    for (int i = 0; i < 5; ++i) {                          // conv1b, conv2a, conv2b, conv3a, conv3b (cin = 128: the split kernel's 2 x 32 tiles, the fp16 register-stationary kernel's 6 x 32)
        const int TH = split ? (i == 4 ? 2 : 4) : (i == 4 ? conv_rs_pool_tile_rows() : CONV_TH);  // the kernels' output tiles (conv_split.hip: 4 x 32 / 2 x 32, conv.hip: CONV_TH x CONV_TW / RS_TH x RS_TW)
        a += 1; b -= 1; c += 1; d -= 1;                    // a 3x3 convolution (zero padding is NOT the constant): one pixel in from every side
        omni_sp::MaskSkip& k = ks[1 + i];
        if (b < a || d < c) break;                         // nothing constant from here on
        k.ty0 = (a + TH - 1) / TH; k.ty1 = (b + 1) / TH; k.tx0 = (c + TW - 1) / TW; k.tx1 = (d + 1) / TW;
        if (k.ty1 <= k.ty0 || k.tx1 <= k.tx0) k.ty0 = k.ty1 = k.tx0 = k.tx1 = 0;
        k.frac = (double)(k.ty1 - k.ty0) * (k.tx1 - k.tx0) / ((double)((h + TH - 1) / TH) * ((w + TW - 1) / TW));
        const int f = pool[i] ? 2 : 1;
        k.oy0 = k.ty0 * TH / f; k.oy1 = k.ty1 * TH / f; k.ox0 = k.tx0 * TW / f; k.ox1 = k.tx1 * TW / f;
        if (pool[i]) { a = (a + 1) / 2; b = (b - 1) >> 1; c = (c + 1) / 2; d = (d - 1) >> 1; h /= 2; w /= 2; }      // pooled pixel r = conv pixels 2r, 2r + 1
        k.oh = h; k.ow = w; k.oc = chans[i];
    }
}

static int sp_plan_mask_skip(omni_sp* s) {
    s->mask_skip = false;
    const bool split = s->precision == OMNI_PREC_SPLIT;
    if (s->conv_variant != 0 || s->precision == OMNI_PREC_F32) return OMNI_OK;
    if (!s->cfg[split ? CFG_SP_MASK_SKIP_SPLIT : CFG_SP_MASK_SKIP]) return OMNI_OK;       // = 0: the dense pass (A/B, tests)
    sp_mask_skip_rects(s->H, s->W, split, s->mskip);
    void** maps[6] = {&s->a1a, &s->a1b, &s->a2a, &s->a2b, &s->a3a, &s->a3b};
    for (int i = 0; i < 6; ++i) {
        omni_sp::MaskSkip& k = s->mskip[i];
        k.map = maps[i];
        if (k.oc == 0) continue;
        if (split) {
            k.pix_bytes = k.oc * 4;
            k.row_bytes = (int64_t)split_frame_w(k.ow) * k.pix_bytes;
            k.img_bytes = (int64_t)split_frame_bytes(k.oh, k.ow, k.oc);
            k.org_bytes = k.row_bytes + k.pix_bytes;
        } else {
            k.pix_bytes = k.oc * 2; k.row_bytes = (int64_t)k.ow * k.pix_bytes; k.img_bytes = k.row_bytes * k.oh; k.org_bytes = 0;
        }
        if (k.ty1 > k.ty0) {
            OMNI_HIP_TRY(hipMalloc(&k.vec, (size_t)k.pix_bytes * 4));                // (x 4: conv2a as a Winograd layer keeps one vector per position in the 2 x 2 tile)
            s->mask_skip = true;
        }
    }
    return OMNI_OK;
}

static int sp_init(omni_sp* s, const omni_sp_weights* w, const float* pca_comp, const float* pca_mean) {
    hipStream_t st = s->ctx->stream;
    int rc;
    // biases, conv1a, convPb (fp32 always)
    for (int l = 0; l < OMNI_SP_NUM_LAYERS; ++l)
        if ((rc = dev_upload((void**)&s->bias[l], w->bias[l], (size_t)kLayers[l].cout * 4, st))) return rc;
    if ((rc = dev_upload((void**)&s->w1a, w->weight[L1A], 64 * 9 * 4, st))) return rc;
    {
        std::vector<float> t(256 * 65);
        for (int c = 0; c < 65; ++c)
            for (int k = 0; k < 256; ++k) t[(size_t)k * 65 + c] = w->weight[LPB][(size_t)c * 256 + k];
        if ((rc = dev_upload((void**)&s->wPbT, t.data(), t.size() * 4, st))) return rc;
        std::vector<float> wa(16384), wdst(256);
        detector_pack_weights(t.data(), wa.data(), wdst.data());
        if ((rc = dev_upload((void**)&s->wPbA, wa.data(), wa.size() * 4, st))) return rc;
        if ((rc = dev_upload((void**)&s->wPbDust, wdst.data(), wdst.size() * 4, st))) return rc;
        std::vector<uint16_t> w16(2 * 2 * 16 * 64 * 8);
        detector_pack_weights16(t.data(), w16.data());
        if ((rc = dev_upload(&s->wPbA16, w16.data(), w16.size() * 2, st))) return rc;
        s->det16 = s->cfg[CFG_DET16] != 0;
        s->fused_cand = s->cfg[CFG_SP_FUSED_CAND] != 0;
        s->sparse_desc = s->cfg[CFG_SP_SPARSE_DESC] != 0;
        s->sparse_da = s->sparse_desc && s->cfg[CFG_SP_SPARSE_DA] != 0;
    }
    {
        std::vector<float> bh(512);
        memcpy(bh.data(), w->bias[LPA], 256 * 4);
        memcpy(bh.data() + 256, w->bias[LDA], 256 * 4);
        if ((rc = dev_upload((void**)&s->bias_heads, bh.data(), 512 * 4, st))) return rc;
    }
    {   // cv::Mat::convertTo(CV_32F, 1/255.0) (superpoint_tensorrt.cpp:127): OpenCV 3.4 scales 8-bit sources in float (cvt_32f): float(u8) * float(1/255.0)
        float lut[256];
        const volatile float alpha = (float)(1.0 / 255.0);                    // (volatile: one fp32 multiply, no contraction / folding in double)
        for (int i = 0; i < 256; ++i) lut[i] = (float)i * alpha;
        if ((rc = dev_upload((void**)&s->lut, lut, sizeof(lut), st))) return rc;
    }
    if (s->precision == OMNI_PREC_F16) {
        std::vector<uint16_t> db(65536);
        convdb_pack_weights(w->weight[LDB], db.data());
        if ((rc = dev_upload(&s->wDbFrag, db.data(), db.size() * 2, st))) return rc;
    }
    if (s->precision == OMNI_PREC_SPLIT && s->cfg[CFG_SP_SPLIT_DB] != 0) {
        std::vector<uint16_t> hi(65536), lo(65536);
        convdb_pack_weights_split(w->weight[LDB], hi.data(), lo.data());
        if ((rc = dev_upload(&s->wDbFragHi, hi.data(), hi.size() * 2, st))) return rc;
        if ((rc = dev_upload(&s->wDbFragLo, lo.data(), lo.size() * 2, st))) return rc;
    }
    if (s->precision == OMNI_PREC_SPLIT) {     // conv1a inside conv1b's kernel (conv1ab_split_fused): its weights x the activation scale, the u8 table
        std::vector<uint16_t> fr(2048);
        conv1a_split_pack_fused(w->weight[L1A], w->bias[L1A], fr.data());
        if ((rc = dev_upload((void**)&s->w1a_frag, fr.data(), fr.size() * 2, st))) return rc;
        uint32_t lh[256];
        conv1a_make_split_lut(lh);
        if ((rc = dev_upload((void**)&s->lut_hl, lh, sizeof(lh), st))) return rc;
        s->split_fuse1a = s->cfg[CFG_SPLIT_FUSE1A] != 0;
    }
    if (s->precision == OMNI_PREC_F16) {
        std::vector<uint16_t> fr(2048);
        if (s->cfg[CFG_PP_U8]) {              // operands straight from the bytes: no table (lut_hl stays null)
            conv1a_pack_u8_weights(w->weight[L1A], w->bias[L1A], fr.data());
            if ((rc = dev_upload((void**)&s->w1a_frag, fr.data(), fr.size() * 2, st))) return rc;
        } else {
            conv1a_pack_split_weights(w->weight[L1A], w->bias[L1A], fr.data());
            if ((rc = dev_upload((void**)&s->w1a_frag, fr.data(), fr.size() * 2, st))) return rc;
            uint32_t lh[256];
            conv1a_make_split_lut(lh);
            if ((rc = dev_upload((void**)&s->lut_hl, lh, sizeof(lh), st))) return rc;
        }
    }
    // packed MFMA weights
    auto pack_upload = [&](int l, const float* w_oihw, int cin, int cout, int ks) -> int {
        const size_t n = conv_packed_elems(cin, cout, ks);
        if (s->precision == OMNI_PREC_SPLIT && ks == 3) {
            std::vector<uint16_t> p(n * 2);
            s->winv[l] = conv_pack_weights_split(w_oihw, cin, cout, p.data());
            return dev_upload(&s->wpk[l], p.data(), n * 4, st);
        }
        if (s->precision == OMNI_PREC_F16) {
            std::vector<__half> p(n);
            conv_pack_weights_f16(w_oihw, cin, cout, ks, p.data());
            return dev_upload(&s->wpk[l], p.data(), n * 2, st);
        }
        std::vector<float> p(n);
        conv_pack_weights_f32(w_oihw, cin, cout, ks, p.data());
        return dev_upload(&s->wpk[l], p.data(), n * 4, st);
    };
    for (int l : {L1B, L2A, L2B, L3A, L3B, L4A, L4B, LDB})
        if ((rc = pack_upload(l, w->weight[l], kLayers[l].cin, kLayers[l].cout, kLayers[l].ks))) return rc;
    {   // convPa | convDa fused along the output-channel axis: one N = 512 conv over the shared conv4b activations
        const size_t per = (size_t)256 * 128 * 9;
        std::vector<float> wh(per * 2);
        memcpy(wh.data(), w->weight[LPA], per * 4);
        memcpy(wh.data() + per, w->weight[LDA], per * 4);
        if ((rc = pack_upload(LPA, wh.data(), 128, 512, 3))) return rc;
    }
    if (s->precision == OMNI_PREC_SPLIT) {
        s->wino = s->cfg[CFG_SPLIT_WINO];
        if (s->H % 8 != 0 || s->W % 8 != 0) s->wino &= 7;                         // conv3a: even H / 4, W / 4 (F(2x2,3x3) tiles)
        if (s->H % 4 != 0 || s->W % 4 != 0) s->wino &= 1;                         // conv2a / conv2b: even H / 2, W / 2
        if (s->H % 2 != 0 || s->W % 2 != 0 || !s->split_fuse1a) s->wino &= ~1;
        for (int l : {L1B, L2A, L2B, L3A}) {
            if (!(s->wino & (l == L1B ? 1 : l == L2A ? 2 : l == L2B ? 4 : 8))) continue;
            const int co = kLayers[l].cout;
            std::vector<uint16_t> p((size_t)64 * co * 16 * 2);
            s->winv_w[l] = conv_pack_weights_wino(w->weight[l], 64, co, p.data());
            if ((rc = dev_upload(&s->wpk_w[l], p.data(), p.size() * 2, st))) return rc;
        }
    }
    if (s->precision == OMNI_PREC_SPLIT) {
        const float S = conv_split_act_scale();
        for (int l : {L1B, L2A, L2B, L3A, L3B, L4A, L4B}) {
            std::vector<float> b(kLayers[l].cout);
            for (int c = 0; c < kLayers[l].cout; ++c) b[c] = S * w->bias[l][c];
            if ((rc = dev_upload((void**)&s->bias_s[l], b.data(), b.size() * 4, st))) return rc;
        }
    }
    if (pca_comp) {
        std::vector<float> t((size_t)256 * s->pca_dim);
        for (int j = 0; j < s->pca_dim; ++j)
            for (int c = 0; c < 256; ++c) t[(size_t)c * s->pca_dim + j] = pca_comp[(size_t)j * 256 + c];
        if ((rc = dev_upload((void**)&s->pca_compT, t.data(), t.size() * 4, st))) return rc;
        if ((rc = dev_upload((void**)&s->pca_mean, pca_mean, 256 * 4, st))) return rc;
    }
    // activations
    const size_t B = s->max_batch, H = s->H, W = s->W, e = s->esz;
    {
        // OMNI_PREC_SPLIT: every map in its zero frame (conv.h), zeroed here once
        struct Act { void** p; size_t h, w, c; };
        const Act acts[] = {{&s->a1a, H, W, 64},         {&s->a1b, H / 2, W / 2, 64},  {&s->a2a, H / 2, W / 2, 64},  {&s->a2b, H / 4, W / 4, 64},
                            {&s->a3a, H / 4, W / 4, 128}, {&s->a3b, H / 8, W / 8, 128}, {&s->a4a, H / 8, W / 8, 128}, {&s->a4b, H / 8, W / 8, 128}};
        for (const Act& a : acts) {
            if (s->precision == OMNI_PREC_SPLIT) {
                const size_t bytes = B * split_frame_bytes((int)a.h, (int)a.w, (int)a.c);
                OMNI_HIP_TRY(hipMalloc(a.p, bytes));
                OMNI_HIP_TRY(hipMemsetAsync(*a.p, 0, bytes, st));
            } else {
                OMNI_HIP_TRY(hipMalloc(a.p, B * a.h * a.w * a.c * e));
            }
        }
        OMNI_HIP_TRY(hipStreamSynchronize(st));
    }
    OMNI_HIP_TRY(hipMalloc(&s->heads, B * (H / 8) * (W / 8) * 512 * e));
    if (s->precision == OMNI_PREC_F16) {
        OMNI_HIP_TRY(hipMalloc(&s->headsP, B * (H / 8) * (W / 8) * 256 * e));
        OMNI_HIP_TRY(hipMalloc(&s->da_compact, B * (size_t)s->max_num * 4 * 256 * 2));
    }
    if (s->precision != OMNI_PREC_F16) {
        const size_t rows = ((B * (size_t)s->max_num * 4) + 7) & ~(size_t)7;
        OMNI_HIP_TRY(hipMalloc((void**)&s->cx32, rows * 256 * 4));
        OMNI_HIP_TRY(hipMalloc((void**)&s->cy32, rows * 256 * 4));
        OMNI_HIP_TRY(hipMemsetAsync(s->cx32, 0, rows * 256 * 4, st));          // rows of key points that do not exist are never written (and never read back)
        OMNI_HIP_TRY(hipStreamSynchronize(st));
    }
    if (s->precision == OMNI_PREC_SPLIT) OMNI_HIP_TRY(hipMalloc(&s->headsP, B * (H / 8) * (W / 8) * 256 * 4));     // cPa alone, fp32 (sparse convDa passes)
    OMNI_HIP_TRY(hipMalloc((void**)&s->draw, B * (H / 8) * (W / 8) * 256 * 4));
    OMNI_HIP_TRY(hipMalloc((void**)&s->semi, B * H * W * 4));
    OMNI_HIP_TRY(hipMalloc((void**)&s->gray_stage, B * H * W));
    // post-processing buffers
    const size_t hw = H * W, M = s->max_num;
    OMNI_HIP_TRY(hipMalloc((void**)&s->pb.cand, B * hw * 4));
    OMNI_HIP_TRY(hipMalloc((void**)&s->pb.cand_masks, B * hw * 16));
    OMNI_HIP_TRY(hipMalloc((void**)&s->pb.counters, B * 4 * 4));
    OMNI_HIP_TRY(hipMalloc((void**)&s->pb.surv_keys, B * hw * 8));
    OMNI_HIP_TRY(hipMalloc((void**)&s->pb.raw_desc, B * M * 256 * 4));
    OMNI_HIP_TRY(hipMalloc((void**)&s->pb.norm_partial, B * 8 * 256 * 4));
    OMNI_HIP_TRY(hipMalloc((void**)&s->pb.kps_xy, B * M * 2 * 4));
    OMNI_HIP_TRY(hipMalloc((void**)&s->pb.scores, B * M * 4));
    OMNI_HIP_TRY(hipMalloc((void**)&s->pb.n_kps, B * 4));
    OMNI_HIP_TRY(hipMalloc((void**)&s->pb.desc_out, B * M * s->desc_dim * 4));
    OMNI_HIP_TRY(hipMemsetAsync(s->pb.n_kps, 0, B * 4, st));
    OMNI_HIP_TRY(hipMemsetAsync(s->pb.kps_xy, 0, B * M * 2 * 4, st));
    OMNI_HIP_TRY(hipMemsetAsync(s->pb.desc_out, 0, B * M * s->desc_dim * 4, st));
    OMNI_HIP_TRY(hipMemsetAsync(s->pb.scores, 0, B * M * 4, st));
    OMNI_HIP_TRY(hipMalloc((void**)&s->pb.cand_bits, B * (H / 8) * (W / 8) * 2 * 4));
    s->pb.pca_compT = s->pca_compT;
    s->pb.pca_mean = s->pca_mean;
    for (int i = 0; i <= OMNI_SP_NUM_STAGES; ++i) OMNI_HIP_TRY(hipEventCreate(&s->ev[i]));
    OMNI_HIP_TRY(hipEventCreateWithFlags(&s->ev_convs, hipEventDisableTiming));
    OMNI_HIP_TRY(hipStreamSynchronize(st));
    return sp_plan_mask_skip(s);
}

static int sp_forward(omni_sp* s, const uint8_t* gray_dev, int stride, int batch, int fisheye_mask, bool with_events, bool run_post);
// One dense pass over an all-zero image with the mask on; every planned layer's constant is read from the middle of its rectangle and written
// into that rectangle of every image slot of the layer's activation buffer.
static int sp_calibrate_mask_skip(omni_sp* s, int stride) {
    hipStream_t st = s->ctx->stream;
    const size_t need = (size_t)stride * s->H;
    if (s->zero_gray_bytes < need) {
        if (s->zero_gray) (void)hipFree(s->zero_gray);
        s->zero_gray = nullptr; s->zero_gray_bytes = 0;
        OMNI_HIP_TRY(hipMalloc((void**)&s->zero_gray, need));
        s->zero_gray_bytes = need;
        OMNI_HIP_TRY(hipMemsetAsync(s->zero_gray, 0, need, st));
    }
    s->mask_skip_calibrating = true;
    int rc = sp_forward(s, s->zero_gray, stride, 1, 1, false, false);
    s->mask_skip_calibrating = false;
    if (rc) return rc;
    s->mask_skip_cal_fused = s->fuse1a;
    for (const omni_sp::MaskSkip& k : s->mskip) {
        if (k.ty1 <= k.ty0 || (k.map == &s->a1a && s->fuse1a)) continue;
        if (s->precision == OMNI_PREC_SPLIT && ((k.map == &s->a2a && (s->wino & 2)) || (k.map == &s->a3a && (s->wino & 8)))) {     // an unpooled Winograd layer: constant per position in the 2 x 2 output tile
            if ((rc = conv_read_pixels2x2_bytes(st, *k.map, k.row_bytes, k.org_bytes, k.pix_bytes, ((k.oy0 + k.oy1) / 2) & ~1, ((k.ox0 + k.ox1) / 2) & ~1, k.vec))) return rc;
            if ((rc = conv_fill_rect2x2_bytes(st, *k.map, s->max_batch, k.img_bytes, k.row_bytes, k.org_bytes, k.pix_bytes, k.oy0, k.oy1, k.ox0, k.ox1, k.vec))) return rc;
            continue;
        }
        if ((rc = conv_read_pixel_bytes(st, *k.map, k.row_bytes, k.org_bytes, k.pix_bytes, (k.oy0 + k.oy1) / 2, (k.ox0 + k.ox1) / 2, k.vec))) return rc;
        if ((rc = conv_fill_rect_bytes(st, *k.map, s->max_batch, k.img_bytes, k.row_bytes, k.org_bytes, k.pix_bytes, k.oy0, k.oy1, k.ox0, k.ox1, k.vec))) return rc;
    }
    s->mask_skip_ready = true;
    return OMNI_OK;
}

static SpPostParams post_params(const omni_sp* s) {
    SpPostParams p;
    p.width = s->W; p.height = s->H; p.thres = s->thres; p.max_num = s->max_num; p.dist_thresh = 4; p.pca_dim = s->pca_dim;
    return p;
}

// Enqueue the whole network + post-processing for `batch` HBM-resident images.  ev != nullptr records an event before
// every stage (profiling only).
static int sp_forward(omni_sp* s, const uint8_t* gray_dev, int stride, int batch, int fisheye_mask, bool with_events,
                      bool run_post) {
    hipStream_t st = s->ctx->stream;
    const int H = s->H, W = s->W, P = s->precision;
    int rc, stage = 0;
    if ((rc = s->ctx->ensure_zero_page())) return rc;
    const bool aligned4 = stride % 4 == 0 && ((uintptr_t)gray_dev & 3) == 0;
    const bool fuse1a = ((P == OMNI_PREC_F16 && s->conv_variant == 0) || (P == OMNI_PREC_SPLIT && s->split_fuse1a)) && aligned4;   // else: separate conv1a
    // the constant region of the fisheye mask (omni_sp::MaskSkip): only on the production path (conv1a fused into conv1b)
    const bool use_skip = s->mask_skip && fisheye_mask && (fuse1a || P == OMNI_PREC_SPLIT) && !s->mask_skip_calibrating;
    if (!use_skip && !s->mask_skip_calibrating) s->mask_skip_ready = false;          // this pass overwrites the filled rectangles
    if (use_skip && s->mask_skip_ready && s->mask_skip_cal_fused != fuse1a) s->mask_skip_ready = false;      // (conv1a's own rectangle is only filled by an unfused calibration)
    if (use_skip && !s->mask_skip_ready && (rc = sp_calibrate_mask_skip(s, stride))) return rc;
    auto mark = [&]() -> int { if (with_events) OMNI_HIP_TRY(hipEventRecord(s->ev[stage], st)); ++stage; return OMNI_OK; };
    // OMNI_PREC_SPLIT: which of the cin = 64 layers run as Winograd kernels in THIS pass, and the frame format between them
    const bool w1b = P == OMNI_PREC_SPLIT && (s->wino & 1) && fuse1a, w2a = P == OMNI_PREC_SPLIT && (s->wino & 2) != 0, w2b = P == OMNI_PREC_SPLIT && (s->wino & 4) != 0,
               w3a = P == OMNI_PREC_SPLIT && (s->wino & 8) != 0;
    auto skip_of = [&](int l, ConvArgs& a) {
        const int i = l == L1B ? 1 : l == L2A ? 2 : l == L2B ? 3 : l == L3A ? 4 : l == L3B ? 5 : -1;
        if (use_skip && i >= 0) { a.skip_ty0 = s->mskip[i].ty0; a.skip_ty1 = s->mskip[i].ty1; a.skip_tx0 = s->mskip[i].tx0; a.skip_tx1 = s->mskip[i].tx1; }
    };
    auto conv = [&](int l, const void* in, void* out, const float* bias, int h, int w, int cin, int cout, int ks, bool relu,
                    bool pool, bool out_f32) -> int {
        ConvArgs a;
        a.in = in; a.out = out; a.w_packed = s->wpk[l]; a.bias = bias; a.batch = batch; a.H = h; a.W = w; a.cin = cin;
        a.cout = cout; a.ksize = ks; a.relu = relu; a.pool = pool; a.out_f32 = out_f32;
        a.n_cu = s->ctx->prop.multiProcessorCount; a.zero_page = s->ctx->zero_page; a.variant = s->conv_variant;
        skip_of(l, a);
        if (P == OMNI_PREC_SPLIT) {     // split-64 activations in and (unless out_f32) out; the scaled bias goes with scaled outputs
            a.split_inv = s->winv[l];
            if (!out_f32) a.bias = s->bias_s[l];
            return conv_split(st, a);
        }
        return conv_mfma(st, P, a);
    };
    const int PH = P == OMNI_PREC_SPLIT ? OMNI_PREC_F32 : P;      // the heads' tails: OMNI_PREC_SPLIT hands them fp32 activations
    if ((rc = mark())) return rc;
    s->fuse1a = fuse1a;
    if (P == OMNI_PREC_SPLIT && !fuse1a) { if ((rc = conv1a_split(st, gray_dev, stride, batch, H, W, fisheye_mask, s->w1a, s->bias[L1A], s->lut, s->a1a, use_skip ? s->mskip[0].ty0 : 0, use_skip ? s->mskip[0].ty1 : 0))) return rc; }
    else if (!s->fuse1a) { if ((rc = conv1a_direct(st, P, gray_dev, stride, batch, H, W, fisheye_mask, s->w1a, s->bias[L1A], s->lut, s->a1a))) return rc; }
    if ((rc = mark())) return rc;
    if (s->fuse1a) {   // conv1a is computed inside conv1b's kernel: the conv1a activation tensor is never materialised
        ConvArgs a;
        a.in = nullptr; a.out = s->a1b; a.w_packed = s->wpk[L1B]; a.bias = s->bias[L1B]; a.batch = batch; a.H = H; a.W = W; a.cin = 64;
        a.cout = 64; a.ksize = 3; a.relu = true; a.pool = true; a.out_f32 = false; a.n_cu = s->ctx->prop.multiProcessorCount; a.zero_page = s->ctx->zero_page;
        skip_of(L1B, a);
        if (P == OMNI_PREC_SPLIT && w1b) {
            a.w_packed = s->wpk_w[L1B]; a.split_inv = s->winv_w[L1B]; a.bias = s->bias_s[L1B];
            if ((rc = conv1ab_wino_fused(st, a, gray_dev, stride, fisheye_mask, s->w1a_frag, s->lut_hl, /*out_split=*/!w2a))) return rc;
        } else if (P == OMNI_PREC_SPLIT) {
            a.split_inv = s->winv[L1B]; a.bias = s->bias_s[L1B];
            if ((rc = conv1ab_split_fused(st, a, gray_dev, stride, fisheye_mask, s->w1a_frag, s->lut_hl))) return rc;
        } else if ((rc = conv1ab_fused(st, a, gray_dev, stride, fisheye_mask, reinterpret_cast<const _Float16*>(s->w1a_frag), s->bias[L1A], s->lut_hl))) return rc;
    } else if ((rc = conv(L1B, s->a1a, s->a1b, s->bias[L1B], H, W, 64, 64, 3, true, true, false))) return rc;
    if ((rc = mark())) return rc;
    // a Winograd layer: raw-32 input (converted into a_tmp when the layer before it wrote split-64: mixed configurations), raw-32 or split-64 output
    auto wino_layer = [&](int l, const void* in, bool in_raw, void* out, int h, int w, bool pool, bool out_split) -> int {
        if (!in_raw) {
            if (!s->a_tmp) OMNI_HIP_TRY(hipMalloc(&s->a_tmp, (size_t)s->max_batch * split_frame_bytes(H / 2, W / 2, 64)));
            if ((rc = split_to_raw32(st, in, s->a_tmp, batch, 64, h, w))) return rc;
            in = s->a_tmp;
        }
        ConvArgs a;
        a.in = in; a.out = out; a.w_packed = s->wpk_w[l]; a.bias = s->bias_s[l]; a.batch = batch; a.H = h; a.W = w; a.cin = 64; a.cout = kLayers[l].cout; a.ksize = 3;
        a.relu = true; a.pool = pool; a.out_f32 = false; a.n_cu = s->ctx->prop.multiProcessorCount; a.split_inv = s->winv_w[l];
        skip_of(l, a);
        return conv_wino(st, a, out_split);
    };
    const bool raw_1b = w1b && w2a, raw_2a = w2a && w2b, raw_2b = w2b && w3a;
    if (P == OMNI_PREC_SPLIT) { s->raw_1b = raw_1b; s->raw_2a = raw_2a; s->raw_2b = raw_2b; }
    if (w2a) { if ((rc = wino_layer(L2A, s->a1b, raw_1b, s->a2a, H / 2, W / 2, false, !w2b))) return rc; }
    else if ((rc = conv(L2A, s->a1b, s->a2a, s->bias[L2A], H / 2, W / 2, 64, 64, 3, true, false, false))) return rc;
    if ((rc = mark())) return rc;
    if (w2b) { if ((rc = wino_layer(L2B, s->a2a, raw_2a, s->a2b, H / 2, W / 2, true, !w3a))) return rc; }
    else if ((rc = conv(L2B, s->a2a, s->a2b, s->bias[L2B], H / 2, W / 2, 64, 64, 3, true, true, false))) return rc;
    if ((rc = mark())) return rc;
    if (w3a) { if ((rc = wino_layer(L3A, s->a2b, raw_2b, s->a3a, H / 4, W / 4, false, true))) return rc; }
    else if ((rc = conv(L3A, s->a2b, s->a3a, s->bias[L3A], H / 4, W / 4, 64, 128, 3, true, false, false))) return rc;
    if ((rc = mark())) return rc;
    if ((rc = conv(L3B, s->a3a, s->a3b, s->bias[L3B], H / 4, W / 4, 128, 128, 3, true, true, false))) return rc;
    if ((rc = mark())) return rc;
    if ((rc = conv(L4A, s->a3b, s->a4a, s->bias[L4A], H / 8, W / 8, 128, 128, 3, true, false, false))) return rc;
    if ((rc = mark())) return rc;
    if ((rc = conv(L4B, s->a4a, s->a4b, s->bias[L4B], H / 8, W / 8, 128, 128, 3, true, false, false))) return rc;
    if ((rc = mark())) return rc;
    const bool sparse = s->precision == OMNI_PREC_F16 && s->conv_variant == 0 && s->sparse_desc && run_post;
    // fp32 / split paths: the exact-f32 convDb + norm likewise only at the cells around the key points (sp_post.hip), the dense map on demand
    const bool sparse32 = s->precision != OMNI_PREC_F16 && s->conv_variant == 0 && s->sparse_desc && run_post && s->cx32;
    const bool sparse_da32 = sparse32 && P == OMNI_PREC_SPLIT && s->sparse_da && s->headsP;      // convDa itself at those cells only (conv_split_c128_sparse)
    const bool sparse_da = (sparse && s->sparse_da && s->headsP) || sparse_da32;
    // the detector branch needs cPa everywhere; cDa (output channels 256-511 of the fused heads layer) is only read around the key points
    const void* cpa = sparse_da ? s->headsP : s->heads;
    const int cpa_stride = sparse_da ? 256 : 512;
    if (sparse_da) { if ((rc = conv(LPA, s->a4b, s->headsP, s->bias_heads, H / 8, W / 8, 128, 256, 3, true, false, P == OMNI_PREC_SPLIT))) return rc; }
    else if ((rc = conv(LPA, s->a4b, s->heads, s->bias_heads, H / 8, W / 8, 128, 512, 3, true, false, P == OMNI_PREC_SPLIT))) return rc;
    if ((rc = mark())) return rc;
    // the head thresholds its own output into the candidate lists when the post-processing follows (superpoint_tensorrt.cpp:167-173 inside the epilogue)
    DetCand dc;
    const bool cand_fused = run_post && s->fused_cand && s->conv_variant != 1;
    if (cand_fused) { dc.thres = s->thres; dc.bits = s->pb.cand_bits; }
    if (s->conv_variant == 1) { if ((rc = detector_head(st, PH, s->heads, 512, 0, batch, s->Hc, s->Wc, s->wPbT, s->bias[LPB], s->semi))) return rc; }
    else if ((P == OMNI_PREC_F16 || P == OMNI_PREC_SPLIT) && s->det16) {
        // fp16: exact operands, split weights; OMNI_PREC_SPLIT: the heads layer's fp32 output split on the fly as well (three terms: fp32-class logits)
        if ((rc = detector_head_mfma16(st, P == OMNI_PREC_F16 ? OMNI_PREC_F16 : OMNI_PREC_F32, cpa, cpa_stride, 0, batch, s->Hc, s->Wc, s->wPbA16, s->wPbDust, s->bias[LPB], s->semi,
                                       s->ctx->prop.multiProcessorCount, dc))) return rc;
    } else if ((rc = detector_head_mfma(st, PH, cpa, cpa_stride, 0, batch, s->Hc, s->Wc, s->wPbA, s->wPbDust, s->bias[LPB], s->semi,
                                        s->ctx->prop.multiProcessorCount, dc))) return rc;
    if ((rc = mark())) return rc;
    OMNI_HIP_TRY(hipEventRecord(s->ev_convs, st));        // the convolution stack and the detector head are enqueued: what follows are small grids
    s->dense_valid = !sparse && !sparse32; s->dense_possible = true; s->last_batch = batch;
    s->heads_full = !sparse_da32;
    if (sparse || sparse32) {
        // nothing here: convDb runs inside the post-processing, at the key points only
    } else if (s->precision == OMNI_PREC_F16 && s->conv_variant == 0) {
        // convDb + descriptor L2 norm in one HBM pass (channels [256,512) = cDa of the fused heads buffer, pixel stride 512)
        if ((rc = convdb_l2norm(st, s->ctx, (const char*)s->heads + (size_t)256 * s->esz, 512, s->wDbFrag, s->bias[LDB], s->draw,
                                (int64_t)batch * s->Hc * s->Wc))) return rc;
    } else {
        // convDb reads channels [256,512) (cDa) of the fused heads buffer: input pointer offset by 256 channels,
        // pixel stride 512
        ConvArgs a;
        a.in = (const char*)s->heads + (size_t)256 * s->esz; a.out = s->draw; a.w_packed = s->wpk[LDB]; a.bias = s->bias[LDB];
        a.batch = batch; a.H = s->Hc; a.W = s->Wc; a.cin = 256; a.cout = 256; a.ksize = 1; a.relu = false; a.pool = false;
        a.out_f32 = true; a.in_cstride = 512;
        if ((rc = conv_mfma(st, PH, a))) return rc;
        if ((rc = l2norm_channels(st, s->draw, (int64_t)batch * s->Hc * s->Wc))) return rc;
    }
    if ((rc = mark())) return rc;
    if (run_post) {
        SpSparseDesc sd;
        if (sparse) { sd.ctx = s->ctx; sd.in_f16 = (const char*)s->heads + (size_t)256 * s->esz; sd.in_cstride = 512; sd.wfrag = s->wDbFrag; sd.bias = s->bias[LDB]; }
        if (sparse_da32) { sd.ctx = s->ctx; sd.a4b_split = s->a4b; sd.da_w = s->wpk[LPA]; sd.da_bias = s->bias_heads; sd.da_g32_first = 8; sd.da_inv = s->winv[LPA];
                           sd.wdb_hi = s->wDbFragHi; sd.wdb_lo = s->wDbFragLo; }
        else if (sparse_da) { sd.a4b = s->a4b; sd.da_w = s->wpk[LPA]; sd.da_bias = s->bias_heads; sd.da_g32_first = 8; sd.da_compact = s->da_compact; }
        if (sparse32) {
            sd.cda_f32 = reinterpret_cast<const float*>(s->heads) + 256; sd.in_cstride = 512; sd.wdb_f32 = s->wpk[LDB]; sd.bias = s->bias[LDB];
            sd.cx = s->cx32; sd.cy = s->cy32; sd.n_cu = s->ctx->prop.multiProcessorCount; sd.zero_page = s->ctx->zero_page;
        }
        sd.cand_fused = cand_fused;
        if ((rc = sp_postprocess(st, post_params(s), s->pb, s->semi, s->draw, batch, sd))) return rc;
    }
    if ((rc = mark())) return rc;
    return OMNI_OK;
}

// the dense head activations and descriptor map of the LAST forward pass, when it sampled its descriptors sparsely: the fused heads layer over
// every cell (conv4b's output is still in HBM) + convDb + L2 norm
static int sp_make_dense(omni_sp* s) {
    hipStream_t st = s->ctx->stream;
    int rc;
    if (s->precision != OMNI_PREC_F16) {       // the heads layer's fp32 output is still in HBM: dense convDb + norm from it
        if (!s->heads_full) {                  // (OMNI_PREC_SPLIT with convDa at the key points only: the fused layer over every cell first)
            ConvArgs h;
            h.in = s->a4b; h.out = s->heads; h.w_packed = s->wpk[LPA]; h.bias = s->bias_heads; h.batch = s->last_batch; h.H = s->Hc; h.W = s->Wc; h.cin = 128;
            h.cout = 512; h.ksize = 3; h.relu = true; h.pool = false; h.out_f32 = true; h.split_inv = s->winv[LPA];
            h.n_cu = s->ctx->prop.multiProcessorCount; h.zero_page = s->ctx->zero_page; h.variant = s->conv_variant;
            if ((rc = conv_split(st, h))) return rc;
            s->heads_full = true;
        }
        ConvArgs a;
        a.in = (const char*)s->heads + (size_t)256 * 4; a.out = s->draw; a.w_packed = s->wpk[LDB]; a.bias = s->bias[LDB];
        a.batch = s->last_batch; a.H = s->Hc; a.W = s->Wc; a.cin = 256; a.cout = 256; a.ksize = 1; a.relu = false; a.pool = false;
        a.out_f32 = true; a.in_cstride = 512;
        if ((rc = conv_mfma(st, OMNI_PREC_F32, a))) return rc;
        if ((rc = l2norm_channels(st, s->draw, (int64_t)s->last_batch * s->Hc * s->Wc))) return rc;
        s->dense_valid = true;
        return OMNI_OK;
    }
    ConvArgs a;
    a.in = s->a4b; a.out = s->heads; a.w_packed = s->wpk[LPA]; a.bias = s->bias_heads; a.batch = s->last_batch; a.H = s->Hc; a.W = s->Wc; a.cin = 128;
    a.cout = 512; a.ksize = 3; a.relu = true; a.pool = false; a.out_f32 = false;
    a.n_cu = s->ctx->prop.multiProcessorCount; a.zero_page = s->ctx->zero_page; a.variant = s->conv_variant;
    if ((rc = conv_mfma(st, s->precision, a))) return rc;
    if ((rc = convdb_l2norm(st, s->ctx, (const char*)s->heads + (size_t)256 * s->esz, 512, s->wDbFrag, s->bias[LDB], s->draw,
                            (int64_t)s->last_batch * s->Hc * s->Wc))) return rc;
    s->dense_valid = true;
    return OMNI_OK;
}

static int sp_fetch_locked(omni_sp* s, int batch, float* kps_xy, int* n_kps, float* desc, float* scores) {
    hipStream_t st = s->ctx->stream;
    const size_t M = s->max_num, D = s->desc_dim;
    const size_t b_kps = (size_t)batch * M * 2 * 4, b_n = (size_t)batch * 4, b_desc = (size_t)batch * M * D * 4, b_sc = (size_t)batch * M * 4;
    int rc;
    if ((rc = s->hstage.ensure(b_kps + b_n + b_desc + b_sc))) return rc;
    char* h = s->hstage.as<char>();
    OMNI_HIP_TRY(hipMemcpyAsync(h, s->pb.kps_xy, b_kps, hipMemcpyDeviceToHost, st));
    OMNI_HIP_TRY(hipMemcpyAsync(h + b_kps, s->pb.n_kps, b_n, hipMemcpyDeviceToHost, st));
    OMNI_HIP_TRY(hipMemcpyAsync(h + b_kps + b_n, s->pb.desc_out, b_desc, hipMemcpyDeviceToHost, st));
    OMNI_HIP_TRY(hipMemcpyAsync(h + b_kps + b_n + b_desc, s->pb.scores, b_sc, hipMemcpyDeviceToHost, st));
    OMNI_HIP_TRY(hipStreamSynchronize(st));
    if (kps_xy) memcpy(kps_xy, h, b_kps);
    if (n_kps) memcpy(n_kps, h + b_kps, b_n);
    if (desc) memcpy(desc, h + b_kps + b_n, b_desc);
    if (scores) memcpy(scores, h + b_kps + b_n + b_desc, b_sc);
    return OMNI_OK;
}

static int upload_gray(omni_sp* s, const uint8_t* gray_host, int stride, int batch) {
    // pack rows to a dense [batch][H][W] device image (stride = W)
    const size_t n = (size_t)batch * s->H * s->W;
    int rc;
    if ((rc = s->hstage.ensure(n))) return rc;
    uint8_t* h = s->hstage.as<uint8_t>();
    for (int b = 0; b < batch; ++b)
        for (int y = 0; y < s->H; ++y)
            memcpy(h + ((size_t)b * s->H + y) * s->W, gray_host + ((size_t)b * s->H + y) * stride, s->W);
    OMNI_HIP_TRY(hipMemcpyAsync(s->gray_stage, h, n, hipMemcpyHostToDevice, s->ctx->stream));
    OMNI_HIP_TRY(hipStreamSynchronize(s->ctx->stream));   // hstage is reused by fetch
    return OMNI_OK;
}

}  // namespace omni

extern "C" {

omni_sp* omni_sp_create(omni_ctx* ctx, const omni_sp_weights* w, const float* pca_comp, const float* pca_mean, int pca_dim,
                        int width, int height, float thres, int max_num, int precision, int max_batch) {
    if (!ctx || !w) { omni::set_error("null ctx/weights"); return nullptr; }
    for (int l = 0; l < OMNI_SP_NUM_LAYERS; ++l)
        if (!w->weight[l] || !w->bias[l]) { omni::set_error("weights for layer %s missing", kLayers[l].name); return nullptr; }
    if (width <= 0 || height <= 0 || width % 8 || height % 8) {
        omni::set_error("width=%d height=%d must be positive multiples of 8 (the reference asserts the engine size, superpoint_tensorrt.cpp:122)", width, height);
        return nullptr;
    }
    if (precision != OMNI_PREC_F32 && precision != OMNI_PREC_F16 && precision != OMNI_PREC_SPLIT) { omni::set_error("bad precision %d", precision); return nullptr; }
    if (max_num < 1 || max_num > 1024 || max_batch < 1 || max_batch > 256) { omni::set_error("max_num=%d (1..1024) / max_batch=%d (1..256) out of range", max_num, max_batch); return nullptr; }
    if (pca_comp && (!pca_mean || pca_dim < 1 || pca_dim > 256)) { omni::set_error("bad PCA arguments"); return nullptr; }
    if ((size_t)width * height / 16 * 4 + 16 > 160 * 1024) { omni::set_error("image %dx%d exceeds the in-LDS NMS plane", width, height); return nullptr; }
    (void)hipSetDevice(ctx->device);
    omni_sp* s = new omni_sp();
    s->ctx = ctx; s->W = width; s->H = height; s->Hc = height / 8; s->Wc = width / 8; s->thres = thres; s->max_num = max_num;
    s->max_batch = max_batch; s->precision = precision; s->esz = precision == OMNI_PREC_F16 ? 2 : 4;
    s->pca_dim = pca_comp ? pca_dim : 0; s->desc_dim = pca_comp ? pca_dim : 256;
    if (omni::config_resolve(&s->cfg) != OMNI_OK) { delete s; return nullptr; }
    s->conv_variant = s->cfg[omni::CFG_CONV_V1];
#ifndef OMNI_TEST_VARIANTS
    if (s->conv_variant != 0) {
        omni::set_error("OMNI_CONV_V1=%d: the reference variants of the fp16 convolutions are only built into the test library (omni-swarm_amd/lib_test/, make -C omni-swarm_amd test-variants)", s->conv_variant);
        delete s;
        return nullptr;
    }
#endif
    if (omni::sp_init(s, w, pca_comp, pca_mean) != OMNI_OK) { omni_sp_destroy(s); return nullptr; }
    return s;
}

void omni_sp_destroy(omni_sp* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->a_tmp) (void)hipFree(s->a_tmp);
    for (int l = 0; l < OMNI_SP_NUM_LAYERS; ++l) { if (s->wpk_w[l]) (void)hipFree(s->wpk_w[l]); if (s->wpk[l]) (void)hipFree(s->wpk[l]); if (s->bias[l]) (void)hipFree(s->bias[l]); if (s->bias_s[l]) (void)hipFree(s->bias_s[l]); }
    void* ptrs[] = {s->wPbA16, s->wDbFragHi, s->wDbFragLo, s->w1a, s->w1a_frag, s->lut_hl, s->wPbT, s->wPbA, s->wPbDust, s->wDbFrag, s->bias_heads, s->lut, s->pca_compT, s->pca_mean, s->a1a, s->a1b, s->a2a, s->a2b, s->a3a, s->a3b,
                    s->a4a, s->a4b, s->heads, s->headsP, s->da_compact, s->cx32, s->cy32, s->draw, s->semi, s->gray_stage, s->pb.cand, s->pb.cand_bits, s->pb.cand_masks, s->pb.counters, s->pb.surv_keys,
                    s->pb.raw_desc, s->pb.norm_partial, s->pb.kps_xy, s->pb.scores, s->pb.n_kps, s->pb.desc_out};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (s->zero_gray) (void)hipFree(s->zero_gray);
    for (auto& k : s->mskip) if (k.vec) (void)hipFree(k.vec);
    s->hstage.release(); s->dense_tmp.release();
    for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
    if (s->ev_convs) (void)hipEventDestroy(s->ev_convs);
    delete s;
}

int omni_sp_desc_dim(const omni_sp* s) { return s ? s->desc_dim : -1; }

int omni_sp_image_size(const omni_sp* s, int* width, int* height) {
    OMNI_REQUIRE(s, OMNI_ERR_INVALID, "null handle");
    if (width) *width = s->W;
    if (height) *height = s->H;
    return OMNI_OK;
}

int omni_sp_enqueue_dev(omni_sp* s, const uint8_t* gray_dev, int stride, int batch, int fisheye_mask) {
    omni::TraceRange trace_range("SuperPoint enqueue (convolutions + heads + post-processing)");
    OMNI_REQUIRE(s && gray_dev, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(batch >= 1 && batch <= s->max_batch, OMNI_ERR_CAPACITY, "batch=%d outside [1,%d]", batch, s->max_batch);
    OMNI_REQUIRE(stride >= s->W, OMNI_ERR_INVALID, "stride=%d < width=%d", stride, s->W);
    std::lock_guard<std::mutex> lk(s->mu);
    (void)hipSetDevice(s->ctx->device);
    s->perf_valid = s->perf;
    return omni::sp_forward(s, gray_dev, stride, batch, fisheye_mask, s->perf, true);
}

int omni_sp_fetch(omni_sp* s, int batch, float* kps_xy, int* n_kps, float* desc, float* scores) {
    OMNI_REQUIRE(s, OMNI_ERR_INVALID, "null handle");
    OMNI_REQUIRE(batch >= 1 && batch <= s->max_batch, OMNI_ERR_CAPACITY, "batch=%d outside [1,%d]", batch, s->max_batch);
    std::lock_guard<std::mutex> lk(s->mu);
    (void)hipSetDevice(s->ctx->device);
    return omni::sp_fetch_locked(s, batch, kps_xy, n_kps, desc, scores);
}

int omni_sp_infer(omni_sp* s, const uint8_t* gray_host, int stride, int batch, int fisheye_mask, float* kps_xy, int* n_kps,
                  float* desc, float* scores) {
    omni::TraceRange trace_range("SuperPoint inference (upload, network, post-processing, download)");
    OMNI_REQUIRE(s && gray_host && kps_xy && n_kps && desc, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(batch >= 1 && batch <= s->max_batch, OMNI_ERR_CAPACITY, "batch=%d outside [1,%d]", batch, s->max_batch);
    OMNI_REQUIRE(stride >= s->W, OMNI_ERR_INVALID, "stride=%d < width=%d", stride, s->W);
    std::lock_guard<std::mutex> lk(s->mu);
    (void)hipSetDevice(s->ctx->device);
    int rc;
    if ((rc = omni::upload_gray(s, gray_host, stride, batch))) return rc;
    s->perf_valid = s->perf;
    if ((rc = omni::sp_forward(s, s->gray_stage, s->W, batch, fisheye_mask, s->perf, true))) return rc;
    return omni::sp_fetch_locked(s, batch, kps_xy, n_kps, desc, scores);
}

int omni_sp_dev_outputs(omni_sp* s, const float** kps_xy_dev, const int** n_kps_dev, const float** desc_dev, const float** scores_dev) {
    OMNI_REQUIRE(s, OMNI_ERR_INVALID, "null handle");
    if (kps_xy_dev) *kps_xy_dev = s->pb.kps_xy;
    if (n_kps_dev) *n_kps_dev = s->pb.n_kps;
    if (desc_dev) *desc_dev = s->pb.desc_out;
    if (scores_dev) *scores_dev = s->pb.scores;
    return OMNI_OK;
}

int omni_sp_get_dense(omni_sp* s, int batch, float* semi_host, float* desc_host) {
    OMNI_REQUIRE(s, OMNI_ERR_INVALID, "null handle");
    OMNI_REQUIRE(batch >= 1 && batch <= s->max_batch, OMNI_ERR_CAPACITY, "batch=%d outside [1,%d]", batch, s->max_batch);
    std::lock_guard<std::mutex> lk(s->mu);
    (void)hipSetDevice(s->ctx->device);
    hipStream_t st = s->ctx->stream;
    int rc;
    if (semi_host) OMNI_HIP_TRY(hipMemcpyAsync(semi_host, s->semi, (size_t)batch * s->H * s->W * 4, hipMemcpyDeviceToHost, st));
    if (desc_host) {
        if (!s->dense_valid) {
            // the last forward pass sampled its descriptors without the dense map: produce it now from the head activations still in HBM
            OMNI_REQUIRE(s->dense_possible && batch <= s->last_batch, OMNI_ERR_INVALID, "no forward pass of >= %d images to take the dense descriptors from", batch);
            if ((rc = omni::sp_make_dense(s))) return rc;
        }
        const size_t n = (size_t)batch * 256 * s->Hc * s->Wc;
        if ((rc = s->dense_tmp.ensure(n * 4))) return rc;
        if ((rc = omni::nhwc_to_nchw(st, s->draw, s->dense_tmp.as<float>(), batch, 256, s->Hc * s->Wc))) return rc;
        OMNI_HIP_TRY(hipMemcpyAsync(desc_host, s->dense_tmp.p, n * 4, hipMemcpyDeviceToHost, st));
    }
    OMNI_HIP_TRY(hipStreamSynchronize(st));
    return OMNI_OK;
}

int omni_sp_postprocess_dense(omni_sp* s, const float* semi_host, const float* desc_host, int batch, float* kps_xy, int* n_kps,
                              float* desc, float* scores) {
    OMNI_REQUIRE(s && semi_host && desc_host, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(batch >= 1 && batch <= s->max_batch, OMNI_ERR_CAPACITY, "batch=%d outside [1,%d]", batch, s->max_batch);
    std::lock_guard<std::mutex> lk(s->mu);
    (void)hipSetDevice(s->ctx->device);
    hipStream_t st = s->ctx->stream;
    int rc;
    const size_t n = (size_t)batch * 256 * s->Hc * s->Wc;
    if ((rc = s->dense_tmp.ensure(n * 4))) return rc;
    OMNI_HIP_TRY(hipMemcpyAsync(s->semi, semi_host, (size_t)batch * s->H * s->W * 4, hipMemcpyHostToDevice, st));
    OMNI_HIP_TRY(hipMemcpyAsync(s->dense_tmp.p, desc_host, n * 4, hipMemcpyHostToDevice, st));
    if ((rc = omni::nchw_to_nhwc(st, s->dense_tmp.as<float>(), s->draw, batch, 256, s->Hc * s->Wc))) return rc;
    s->dense_valid = true; s->dense_possible = false; s->last_batch = batch;          // `draw` / `semi` now hold the caller's maps
    omni::SpSparseDesc sd;
    sd.cand_from_list = s->fused_cand;       // threshold + window masks as the pipeline makes them (OMNI_SP_FUSED_CAND=0: sp_cand_kernel)
    if ((rc = omni::sp_postprocess(st, omni::post_params(s), s->pb, s->semi, s->draw, batch, sd))) return rc;
    return omni::sp_fetch_locked(s, batch, kps_xy, n_kps, desc, scores);
}

int omni_sp_debug_layer(omni_sp* s, const char* name, int batch, float* out_nchw_host, int* C, int* Hl, int* Wl) {
    OMNI_REQUIRE(s && name, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(batch >= 1 && batch <= s->max_batch, OMNI_ERR_CAPACITY, "batch=%d outside [1,%d]", batch, s->max_batch);
    struct Ent { const char* n; const void* p; int c, div, prec; };
    const int P = s->precision;
    const Ent tab[] = {{"conv1a", s->a1a, 64, 1, P},   {"conv1b", s->a1b, 64, 2, P},   {"conv2a", s->a2a, 64, 2, P},
                       {"conv2b", s->a2b, 64, 4, P},   {"conv3a", s->a3a, 128, 4, P},  {"conv3b", s->a3b, 128, 8, P},
                       {"conv4a", s->a4a, 128, 8, P},  {"conv4b", s->a4b, 128, 8, P},  {"heads", s->heads, 512, 8, P == OMNI_PREC_SPLIT ? OMNI_PREC_F32 : P},
                       {"desc", s->draw, 256, 8, OMNI_PREC_F32}};
    for (const Ent& e : tab) {
        if (strcmp(e.n, name) != 0) continue;
        if (e.p == s->a1a && s->fuse1a) { omni::set_error("conv1a is fused into conv1b on this path and not materialised (OMNI_CONV_V1=3 keeps it)"); return OMNI_ERR_INVALID; }
        if ((e.p == s->heads || e.p == s->draw) && !s->dense_valid) {
            OMNI_REQUIRE(s->dense_possible && batch <= s->last_batch, OMNI_ERR_INVALID, "no forward pass of >= %d images to take layer %s from", batch, name);
            std::lock_guard<std::mutex> lk(s->mu);
            (void)hipSetDevice(s->ctx->device);
            int rc = omni::sp_make_dense(s);
            if (rc) return rc;
        }
        const int h = s->H / e.div, w = s->W / e.div;
        if (C) *C = e.c;
        if (Hl) *Hl = h;
        if (Wl) *Wl = w;
        if (!out_nchw_host) return OMNI_OK;
        std::lock_guard<std::mutex> lk(s->mu);
        (void)hipSetDevice(s->ctx->device);
        const size_t n = (size_t)batch * e.c * h * w;
        int rc;
        if ((rc = s->dense_tmp.ensure(n * 4))) return rc;
        if (e.prec == OMNI_PREC_SPLIT && ((e.p == s->a1b && s->raw_1b) || (e.p == s->a2a && s->raw_2a) || (e.p == s->a2b && s->raw_2b))) {      // a raw-32 frame between two Winograd layers
            if ((rc = omni::raw32_to_nchw_f32(s->ctx->stream, e.p, s->dense_tmp.as<float>(), batch, e.c, h, w))) return rc;
        } else if (e.prec == OMNI_PREC_SPLIT) { if ((rc = omni::split_to_nchw_f32(s->ctx->stream, e.p, s->dense_tmp.as<float>(), batch, e.c, h, w))) return rc; }
        else if ((rc = omni::nhwc_any_to_nchw_f32(s->ctx->stream, e.prec, e.p, s->dense_tmp.as<float>(), batch, e.c, h * w))) return rc;
        OMNI_HIP_TRY(hipMemcpyAsync(out_nchw_host, s->dense_tmp.p, n * 4, hipMemcpyDeviceToHost, s->ctx->stream));
        OMNI_HIP_TRY(hipStreamSynchronize(s->ctx->stream));
        return OMNI_OK;
    }
    omni::set_error("unknown layer '%s'", name);
    return OMNI_ERR_INVALID;
}

// (internal, cam.hip) the event a pass records behind its convolution stack
hipEvent_t omni_sp_convs_event(omni_sp* s) { return s ? s->ev_convs : nullptr; }

const char* omni_sp_stage_name(int stage) { return (stage >= 0 && stage < OMNI_SP_NUM_STAGES) ? kStageNames[stage] : ""; }

double omni_sp_stage_flops(const omni_sp* s, int stage) {
    if (!s) return 0.0;
    const double H = s->H, W = s->W;
    auto c = [](double h, double w, double cin, double cout, double k) { return 2.0 * h * w * cin * cout * k * k; };
    switch (stage) {
        case ST_CONV1A: return c(H, W, 1, 64, 3);
        case ST_CONV1B: return c(H, W, 64, 64, 3);
        case ST_CONV2A: case ST_CONV2B: return c(H / 2, W / 2, 64, 64, 3);
        case ST_CONV3A: return c(H / 4, W / 4, 64, 128, 3);
        case ST_CONV3B: return c(H / 4, W / 4, 128, 128, 3);
        case ST_CONV4A: case ST_CONV4B: return c(H / 8, W / 8, 128, 128, 3);
        case ST_HEADS_A: return c(H / 8, W / 8, 128, 512, 3);
        case ST_DET_TAIL: return c(H / 8, W / 8, 256, 65, 1);
        case ST_DESC_TAIL: return c(H / 8, W / 8, 256, 256, 1);
        default: return 0.0;
    }
}

int omni_sp_mask_skip_plan(int width, int height, int precision, int layer, int* rect, double* frac) {
    OMNI_REQUIRE(width > 0 && height > 0 && layer >= 0 && layer < 6, OMNI_ERR_INVALID, "bad argument");
    omni_sp::MaskSkip ks[6];
    if (precision != OMNI_PREC_F32) omni::sp_mask_skip_rects(height, width, precision == OMNI_PREC_SPLIT, ks);
    if (rect) { rect[0] = ks[layer].ty0; rect[1] = ks[layer].ty1; rect[2] = ks[layer].tx0; rect[3] = ks[layer].tx1; }
    if (frac) *frac = ks[layer].ty1 > ks[layer].ty0 ? ks[layer].frac : 0.0;
    return OMNI_OK;
}

double omni_sp_stage_tiles_left_out(const omni_sp* s, int stage) {
    if (!s || !s->mask_skip) return 0.0;
    const int i = stage == ST_CONV1A ? 0 : stage == ST_CONV1B ? 1 : stage == ST_CONV2A ? 2 : stage == ST_CONV2B ? 3 : stage == ST_CONV3A ? 4 : stage == ST_CONV3B ? 5 : -1;
    return i < 0 ? 0.0 : s->mskip[i].frac;
}

// enable_perf of the reference's runners (superpoint_tensorrt.cpp:130-162 prints the engine time and the post-processing time of every call): with perf on, every
// pass records an event in front of each stage (a dozen hipEventRecord: microseconds of host time) and omni_sp_last_stage_ms returns the LAST pass's stage times
int64_t omni_sp_pack_constants(int which, const float* w, const float* bias, int cout, uint16_t* out, int64_t out_halfs, float* scale) {
    if (!w || !out || !scale) { omni::set_error("omni_sp_pack_constants: null argument"); return -2; }
    if (which == 0) {
        if (!bias || out_halfs < 2048) { omni::set_error("omni_sp_pack_constants: conv1a needs a bias and 2048 halfs"); return -2; }
        omni::conv1a_pack_u8_weights(w, bias, out);
        *scale = 1.f;
        return 2048;
    }
    if (which == 1) {
        const int64_t need = (int64_t)64 * cout * 32;
        if (cout < 64 || cout % 64 || out_halfs < need) { omni::set_error("omni_sp_pack_constants: cout %d, %lld halfs", cout, (long long)out_halfs); return -2; }
        *scale = omni::conv_pack_weights_wino(w, 64, cout, out);
        return need;
    }
    omni::set_error("omni_sp_pack_constants: which = %d", which);
    return -2;
}

int omni_sp_set_perf(omni_sp* s, int on) {
    OMNI_REQUIRE(s, OMNI_ERR_INVALID, "null handle");
    std::lock_guard<std::mutex> lk(s->mu);
    s->perf = on != 0;
    if (!s->perf) s->perf_valid = false;
    return OMNI_OK;
}
int omni_sp_last_stage_ms(omni_sp* s, float* stage_ms) {
    OMNI_REQUIRE(s && stage_ms, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    OMNI_REQUIRE(s->perf_valid, OMNI_ERR_INVALID, "omni_sp_last_stage_ms: no pass was run with omni_sp_set_perf on");
    (void)hipSetDevice(s->ctx->device);
    OMNI_HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    for (int i = 0; i < OMNI_SP_NUM_STAGES; ++i) stage_ms[i] = 0.f;
    for (int i = 0; i < ST_COUNT; ++i) OMNI_HIP_TRY(hipEventElapsedTime(&stage_ms[i], s->ev[i], s->ev[i + 1]));
    return OMNI_OK;
}

int omni_sp_profile(omni_sp* s, const uint8_t* gray_dev, int stride, int batch, int reps, float* stage_ms) {
    const int prof_mask = s ? s->cfg[omni::CFG_SP_PROFILE_MASK] : 0;   // stage times with the fisheye mask on (as the key-frame pipeline runs)
    OMNI_REQUIRE(s && gray_dev && stage_ms && reps >= 1, OMNI_ERR_INVALID, "bad argument");
    OMNI_REQUIRE(batch >= 1 && batch <= s->max_batch, OMNI_ERR_CAPACITY, "batch=%d outside [1,%d]", batch, s->max_batch);
    std::lock_guard<std::mutex> lk(s->mu);
    (void)hipSetDevice(s->ctx->device);
    for (int i = 0; i < OMNI_SP_NUM_STAGES; ++i) stage_ms[i] = 0.f;
    // the MEDIAN over the repetitions: the first passes after an idle stretch run at a lower clock (their launches are 10-20 % longer in a kernel
    // trace of the same run); the median is the launch duration a kernel trace of the timed loop shows
    std::vector<float> all((size_t)reps * ST_COUNT);
    for (int r = 0; r < reps; ++r) {
        int rc = omni::sp_forward(s, gray_dev, stride, batch, prof_mask, true, true);
        if (rc) return rc;
        OMNI_HIP_TRY(hipStreamSynchronize(s->ctx->stream));
        for (int i = 0; i < ST_COUNT; ++i) {
            float ms = 0.f;
            OMNI_HIP_TRY(hipEventElapsedTime(&ms, s->ev[i], s->ev[i + 1]));
            all[(size_t)i * reps + r] = ms;
        }
    }
    for (int i = 0; i < ST_COUNT; ++i) {
        float* v = all.data() + (size_t)i * reps;
        std::sort(v, v + reps);
        stage_ms[i] = (reps & 1) ? v[reps / 2] : 0.5f * (v[reps / 2 - 1] + v[reps / 2]);
    }
    return OMNI_OK;
}

}  // extern "C"
