// Convolution kernels of the SuperPoint graph (swarm_loop/superpoint.ipynb:143-205) for gfx950.
//
// Activations are NHWC (channels innermost) so that an MFMA B-fragment -- 8 consecutive input channels of one
// pixel at one filter tap -- is a single 16-byte LDS read.  Implicit GEMM with
//     M = output channels (A operand = pre-packed weights), N = pixels (B operand = LDS-staged input tile),
//     K = taps x input channels,
// so each lane of the 32x32 accumulator fragment holds ONE pixel and 16 output channels in runs of 4: bias + ReLU
// (+ 2x2 max-pool by two cross-lane exchanges) fuse into the epilogue and stores are 8/16-byte NHWC writes.
//   OMNI_PREC_F16: v_mfma_f32_32x32x16_f16, fp16 activations/weights, fp32 accumulate.
//   OMNI_PREC_F32: v_mfma_f32_32x32x2_f32 (exact f32, an fmaf chain), fp32 activations/weights: the parity mode.
#pragma once
#include "common.h"

namespace omni {

#define CONV_TH 8          // output tile rows  (4 waves x 2 rows)
#define CONV_TW 32         // output tile cols  (2 fragments x 16)
#define RS_TH 6            // the register-stationary cin = 128 kernel's output tile (plain orientation): 6 rows x 32 columns
#define RS_TW 32
int conv_rs_pool_tile_rows();   // rows of the pooled cin = 128 fp16 layer's tile grid (ConvArgs::skip_* of conv3b): 4 (OMNI_CONV_RS=2) or RS_TH
struct RsSkip { int act, n_above, n_upto, y0, y1, x0, w, bw, xcd /* OMNI_CONV_XCD: xcd_block_id() */; };      // conv3x3_c128_rs_kernel: the tiles of an image that run (ConvArgs::skip_*)
#define CONV_COUT_TILE 64  // output channels per workgroup (2 fragments x 32)
#define CONV_CIN_CHUNK 64  // input channels staged per pass

// Packed weight size in elements for one conv layer (ksize 1 or 3), cin % 64 == 0, cout % 64 == 0.
static inline size_t conv_packed_elems(int cin, int cout, int ksize) { return (size_t)cin * cout * ksize * ksize; }

// Host-side packing OIHW fp32 -> fragment order [cout_tile][cin_chunk][tap][k-group][m-frag][lane][elems]
// (see conv.hip for the exact element map).  out must hold conv_packed_elems() elements of the precision's type.
void conv_pack_weights_f16(const float* w_oihw, int cin, int cout, int ksize, __half* out);
void conv_pack_weights_f32(const float* w_oihw, int cin, int cout, int ksize, float* out);

struct ConvArgs {
    const void* in;        // NHWC [B][H][W][cin]
    void* out;             // NHWC [B][Ho][Wo][cout]  (Ho,Wo = H/2,W/2 when pool)
    const void* w_packed;
    const float* bias;     // [cout] fp32
    int batch, H, W, cin, cout, ksize;
    int in_cstride = 0;    // channels per pixel of the input buffer (0 = cin); > cin reads a channel slice
    bool relu, pool;
    bool out_f32;          // write fp32 regardless of the compute precision (final 1x1 descriptor conv)
    int n_cu = 0;          // CUs on the device (> 0 enables the persistent cin=64 fp16 kernel)
    const void* zero_page = nullptr;   // omni_ctx::zero_page (the persistent LDS-DMA kernels need it; null = generic kernels only)
    float split_inv = 0.f; // OMNI_PREC_SPLIT: 2^-k of conv_pack_weights_split (the epilogue's factor)
    int variant = 0;       // test hook (OMNI_CONV_V1): 0 = best kernel per layer, 1 = generic kernel everywhere, 2 = v2 persistent kernel,
                           // 3 = v3 ping-pong kernel without the conv1a fusion
    // Tile rectangle [skip_ty0, skip_ty1) x [skip_tx0, skip_tx1) of the CONV_TH x CONV_TW output-tile grid of every image whose results ALREADY
    // stand in `out` (the caller filled them: see superpoint.hip, "constant region of the fisheye mask"): the persistent cin = 64 fp16 kernel
    // leaves those tiles out of its tile walk; every other kernel ignores the hint and recomputes them (same values).  Empty = nothing to skip.
    int skip_ty0 = 0, skip_ty1 = 0, skip_tx0 = 0, skip_tx1 = 0;
};
int conv_mfma(hipStream_t stream, int precision, const ConvArgs& a);
// fp16 NHWC maps: `vec` (C halfs) <- pixel (y, x) of image 0;  every pixel of rows [y0, y1) x cols [x0, x1) of `batch` images <- `vec`
int conv_read_pixel_f16(hipStream_t stream, const void* map, int Ho, int Wo, int C, int y, int x, void* vec);
int conv_fill_rect_f16(hipStream_t stream, void* map, int batch, int Ho, int Wo, int C, int y0, int y1, int x0, int x1, const void* vec);
// the same for any map of `pix_bytes`-byte pixels (a multiple of 16) laid out as image stride / row stride / origin of pixel (0, 0), all in bytes
// (framed split-64 maps: origin = one row + one pixel into the frame)
int conv_read_pixel_bytes(hipStream_t stream, const void* map, int64_t row_bytes, int64_t org_bytes, int pix_bytes, int y, int x, void* vec);
int conv_fill_rect_bytes(hipStream_t stream, void* map, int batch, int64_t img_bytes, int64_t row_bytes, int64_t org_bytes, int pix_bytes, int y0, int y1, int x0, int x1,
                         const void* vec);

// OMNI_PREC_SPLIT (conv_split.hip): 3x3 conv, cin 64 / 128, on the fp16 matrix cores with every operand split into hi + lo halfs (fp32-class).
// Activations are "split-64" NHWC: per pixel and block of 64 channels [hi x 64 | lo x 64] halfs, values x conv_split_act_scale().
// a.w_packed / a.split_inv from conv_pack_weights_split (cin * cout * 9 * 2 halfs); a.bias = out_f32 ? bias : act_scale * bias;
// a.out = out_f32 ? NHWC fp32 (true values) : split-64.
float conv_pack_weights_split(const float* w_oihw, int cin, int cout, uint16_t* out);
float conv_split_act_scale();
int conv_split(hipStream_t stream, const ConvArgs& a);
// skip_tr0 / skip_tr1: the 8-row tile rows [skip_tr0, skip_tr1) already stand in out_split (the mask's constant band) and are not computed
int conv1a_split(hipStream_t stream, const uint8_t* gray, int stride, int batch, int H, int W, int fisheye_mask, const float* w, const float* bias,
                 const float* u8_lut, void* out_split, int skip_tr0 = 0, int skip_tr1 = 0);
int split_to_nchw_f32(hipStream_t stream, const void* in_split, float* out, int batch, int C, int H, int W);   // test hook
// conv1a + conv1b + ReLU + 2x2 max-pool in one launch (OMNI_PREC_SPLIT): conv1b's halo tiles are built from the u8 image inside the kernel (conv1a on the
// matrix cores with split operands); a = the conv1b layer (a.in unused); w1a_frag from conv1a_split_pack_fused, lut_hl from conv1a_make_split_lut
void conv1a_split_pack_fused(const float* w /*[64][9]*/, const float* bias /*[64]*/, uint16_t* frag /*[2048]*/);
int conv1ab_split_fused(hipStream_t stream, const ConvArgs& a, const uint8_t* gray, int gstride, int fisheye_mask, const void* w1a_frag, const uint32_t* lut_hl);
// OMNI_PREC_SPLIT, Winograd F(2x2,3x3) form of the cin = 64 layers (conv_wino.hip): a.in = "raw-32" frames (the zero frame below, 64 fp32 channels per pixel,
// values x conv_split_act_scale()); a.w_packed / a.split_inv from conv_pack_weights_wino (cin * cout * 16 * 2 halfs); a.bias = act scale x bias;
// a.out = raw-32 frames (next layer: Winograd) or, out_split, split-64 frames (next layer: a direct kernel)
float conv_pack_weights_wino(const float* w_oihw, int cin, int cout, uint16_t* out);
int conv_wino(hipStream_t stream, const ConvArgs& a, bool out_split);
int conv1ab_wino_fused(hipStream_t stream, const ConvArgs& a, const uint8_t* gray, int gstride, int fisheye_mask, const void* w1a_frag, const uint32_t* lut_hl, bool out_split);
int split_to_raw32(hipStream_t stream, const void* in_split, void* out_raw, int batch, int C, int H, int W);      // split-64 frames -> raw-32 frames (out of place)
// the mask's constant region behind an unpooled Winograd layer: one vector per position in the 2 x 2 tile (vec: 4 x pix_bytes; y, x even)
int conv_read_pixels2x2_bytes(hipStream_t stream, const void* map, int64_t row_bytes, int64_t org_bytes, int pix_bytes, int y, int x, void* vec);
int conv_fill_rect2x2_bytes(hipStream_t stream, void* map, int batch, int64_t img_bytes, int64_t row_bytes, int64_t org_bytes, int pix_bytes, int y0, int y1, int x0, int x1,
                            const void* vec);
int raw32_to_nchw_f32(hipStream_t stream, const void* in_raw, float* out, int batch, int C, int H, int W);        // test hook
// A split-64 H x W map lives in a zero frame of split_frame_h(H) rows x split_frame_w(W) pixels, pixel (y, x) at row y + 1, column x + 1:
// one pixel of zero padding all round plus the overhang of the last 32-pixel tile in either direction.  The frame must be zeroed once
// (the kernels only ever write the map).
__host__ __device__ inline int split_frame_w(int W) { return ((W + 31) & ~31) + 2; }
__host__ __device__ inline int split_frame_h(int H) { return ((H + 31) & ~31) + 2; }
inline size_t split_frame_bytes(int H, int W, int C) { return (size_t)split_frame_h(H) * split_frame_w(W) * C * 4; }

// conv1a + conv1b + ReLU + 2x2 max-pool fused (fp16 path): conv1a runs on the matrix cores inside conv1b's ping-pong kernel
// with split fp16 operands (conv.hip); a = the conv1b layer (a.in unused).  Host-side packers for its constant inputs.
int conv1ab_fused(hipStream_t stream, const ConvArgs& a, const uint8_t* gray, int gstride, int fisheye_mask, const _Float16* w1a_frag,
                  const float* bias1a, const uint32_t* lut_hl);
void conv1a_pack_split_weights(const float* w /*[64][9]*/, const float* bias /*[64]*/, uint16_t* frag /*[2048]*/);
void conv1a_make_split_lut(uint32_t* lut /*[256]*/);
void conv1a_pack_u8_weights(const float* w /*[64][9]*/, const float* bias /*[64]*/, uint16_t* frag /*[2048]*/);      // for lut_hl = nullptr: operands straight from the bytes

// conv1a: 1 -> 64 channels, 3x3, + ReLU, straight from the u8 image (u8 -> f32 * 1/255 via a 256-entry table that
// reproduces cv::Mat::convertTo(CV_32F, 1/255.0), superpoint_tensorrt.cpp:127; optional fisheye row mask,
// loop_cam.cpp:536-539).  w [64][9] fp32, out NHWC [B][H][W][64] in the precision's type.
int conv1a_direct(hipStream_t stream, int precision, const uint8_t* gray, int stride, int batch, int H, int W,
                  int fisheye_mask, const float* w, const float* bias, const float* u8_lut, void* out);

// Detector head tail: convPb 1x1 (256 -> 65) + softmax over 65 + drop dustbin + 8x8 depth-to-space
// (superpoint.ipynb:184,190-199).  in: NHWC [B][Hc][Wc][in_stride] (channels [in_off, in_off+256) are cPa);
// wT [256][65] fp32 (transposed), bias [65]; semi [B][Hc*8][Wc*8] fp32.
int detector_head(hipStream_t stream, int precision, const void* in, int in_stride, int in_off, int batch, int Hc, int Wc,
                  const float* wT, const float* bias, float* semi);

// The id a persistent convolution kernel derives its (cout group, tile walk) from, with the XCD placement folded in.  Block b runs on XCD b % 8 (observed,
// MI355X_MICROARCH.md "Workgroup dispatch"; for speed only -- any placement gives the same results): XCD x gets the contiguous ids [x n/8, (x+1) n/8), so
// the cout groups of one pixel tile (ids differing by < n_cg) and the tiles next to it sit behind ONE L2 and the input halo is fetched from HBM / MALL once
// instead of once per cout group and per neighbouring tile (round 4's counters: 3.6-8.2x the algorithmic input on the split cin = 128 layers).
#if defined(__HIPCC__)
__device__ __forceinline__ int xcd_block_id(int enable) {
    const int n = (int)gridDim.x, b = (int)blockIdx.x;
    return (enable && (n & 7) == 0) ? (b & 7) * (n >> 3) + (b >> 3) : b;
}
#endif

// getKeyPoints' threshold fused into the head's epilogue (superpoint_tensorrt.cpp:167-173: mask = prob > thres): with bits set, a lane stores the 32
// comparisons of its half cell as ONE word, bits[cell * 2 + hh] (SpPostBuffers::cand_bits: bit i = row i >> 2, column 4 hh + (i & 3) of the 8 x 8 cell) --
// a coalesced 4-byte store, no atomics; sp_mask_kernel turns the bitmap into the candidate lists
struct DetCand { float thres = 0.f; uint32_t* bits = nullptr; };

// Same on the matrix cores (v_mfma_f32_32x32x2_f32 for the 64 kept channels, VALU for the dustbin); weights packed on the host.
void detector_pack_weights(const float* wT /*[256][65]*/, float* wA /*[16384]*/, float* wdust /*[256]*/);
void detector_pack_weights16(const float* wT /*[256][65]*/, uint16_t* wA16 /*[32768]: split-fp16 A fragments*/);
int detector_head_mfma16(hipStream_t stream, int in_precision /* OMNI_PREC_F16: fp16 activations; else fp32 ones, split (hi, lo) on the fly */, const void* in, int in_stride, int in_off, int batch, int Hc, int Wc, const void* wA16, const float* wdust,
                         const float* bias, float* semi, int n_cu, const DetCand& dc = DetCand{});
int detector_head_mfma(hipStream_t stream, int precision, const void* in, int in_stride, int in_off, int batch, int Hc, int Wc,
                       const float* wA, const float* wdust, const float* bias, float* semi, int n_cu, const DetCand& dc = DetCand{});

// desc = desc / ||desc||_2 over the 256 channels of every coarse cell (superpoint.ipynb:187-188); fp32 NHWC in place.
int l2norm_channels(hipStream_t stream, float* desc_nhwc, int64_t n_cells);
// convDb + channel L2 norm fused (fp16 activations): in = NHWC fp16 with pixel stride in_cstride halfs, already offset to the 256 input
// channels; wfrag from convdb_pack_weights; out = [n_pixels][256] f32, unit-norm per pixel
void convdb_pack_weights(const float* w /*[256][256]*/, uint16_t* frag /*[65536]*/);
// OMNI_PREC_SPLIT: the same pass over fp32 rows with split (hi, lo) operands on the fp16 matrix cores (fp32-class); in = [n_pixels][in_cstride] f32
void convdb_pack_weights_split(const float* w /*[256][256]*/, uint16_t* frag_hi /*[65536]*/, uint16_t* frag_lo /*[65536]*/);
int convdb_l2norm_split(hipStream_t stream, const omni_ctx* ctx, const float* in_f32, int in_cstride, const void* wfrag_hi, const void* wfrag_lo, const float* bias,
                        float* out, int64_t n_pixels);
int convdb_l2norm(hipStream_t stream, const omni_ctx* ctx, const void* in_f16, int in_cstride, const void* wfrag, const float* bias,
                  float* out, int64_t n_pixels);

// convDb + L2 norm + bilinear sampling at the four coarse cells around every key point only (the dense map is not produced): raw_desc
// [batch][max_num][256], bit-identical to convdb_l2norm followed by sp_sample_kernel; after sp_nms_kernel on the same stream
// compact: in_f16 = [image][key point][corner][in_cstride] (conv_c128_sparse's output) instead of the coarse map
int convdb_sparse_sample(hipStream_t stream, const omni_ctx* ctx, const void* in_f16, int in_cstride, const void* wfrag, const float* bias, int W,
                         int H, int max_num, const float* kps_xy, const int* n_kps, float* raw_desc, int batch, bool compact);
// 3x3 conv (128 input channels, ReLU, fp16) at the coarse cells around the key points only: output channels [32 g32_first, + 256) of the packed
// layer w_packed / bias, written compactly as [image][key point][corner][out_cstride]; bit-identical to the dense layer at those cells
int conv_c128_sparse(hipStream_t stream, const omni_ctx* ctx, const void* in_f16, const void* w_packed, const float* bias, int Hc, int Wc,
                     int g32_first, int W, int H, int max_num, const float* kps_xy, const int* n_kps, void* out_f16, int out_cstride, int batch);

// OMNI_PREC_SPLIT: convDa (3x3, 128 input channels, ReLU) at the coarse cells around the key points only, from the split-64 frames of conv4b, with the
// fused heads layer's packed weights / split_inv / bias (convDa = g32_first 8): compact rows [image][key point][corner][256] fp32, bit-identical to
// the dense layer's values at those cells (conv_split.hip)
int conv_split_c128_sparse(hipStream_t stream, const omni_ctx* ctx, const void* a4b, const void* w_packed, const float* bias, float split_inv, int Hc, int Wc,
                           int g32_first, int W, int H, int max_num, const float* kps_xy, const int* n_kps, float* out, int batch);

// test hook: NHWC (fp16 or fp32) -> NCHW fp32
int nhwc_any_to_nchw_f32(hipStream_t stream, int precision_of_in, const void* in, float* out, int batch, int C, int HW);

}  // namespace omni
