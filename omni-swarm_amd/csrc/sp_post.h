// SuperPoint post-processing on the GPU (the reference does all of this on the CPU after a 5.8 MB D2H copy):
//   getKeyPoints          swarm_loop/src/superpoint_tensorrt.cpp:164-189
//   NMS2                  swarm_loop/src/superpoint_tensorrt.cpp:237-310
//   computeDescriptors    swarm_loop/src/superpoint_tensorrt.cpp:192-230
#pragma once
#include "common.h"

namespace omni {

struct SpPostParams {
    int width, height;      // image size (multiples of 8)
    float thres;            // prob > thres (strict)                      :167
    int max_num;            // keep at most max_num key points           :305
    int dist_thresh;        // NMS radius, 4                              :183
    int pca_dim;            // 0 = no PCA (desc_dim 256)
};

struct SpPostBuffers {      // all device pointers, sized for max_batch images
    uint32_t* cand_bits;    // [B][H/8 * W/8][2]   prob > thres as a bitmap: word (cell, hh) = rows 0-7 x columns 4 hh .. 4 hh + 3 of the 8 x 8 cell, bit i = (row i >> 2, column i & 3)
    int* cand;              // [B][H*W]            candidate pixel indices (unordered)
    uint64_t* cand_masks;   // [B][H*W][2]         per candidate: earlier / later higher-confidence window masks
    int* counters;          // [B][4]              n_cand, n_surv, n_iter, spare
    uint64_t* surv_keys;    // [B][H*W]            survivor keys (unordered)
    float* raw_desc;        // [B][max_num][256]   sampled descriptors (before the channel normalisation)
    float* norm_partial;    // [B][8][256]         per-channel sums of squares over key-point segments
    // results
    float* kps_xy;          // [B][max_num][2]
    float* scores;          // [B][max_num]
    int* n_kps;             // [B]
    float* desc_out;        // [B][max_num][desc_dim]
    // constants
    const float* pca_compT; // [256][pca_dim]  (pca_comp transposed, as superpoint_tensorrt.cpp:110)
    const float* pca_mean;  // [256]
};

// descriptors sampled without the dense map (fp16 path): convDb + L2 norm only at the cells around the key points (conv.h: convdb_sparse_sample)
struct SpSparseDesc {
    const omni_ctx* ctx = nullptr;
    const void* in_f16 = nullptr;   // cDa: NHWC fp16, already offset to the 256 input channels; null = sample the dense map instead
    int in_cstride = 0;
    const void* wfrag = nullptr;    // convdb_pack_weights
    const float* bias = nullptr;
    // convDa itself only around the key points (conv_c128_sparse): a4b -> da_compact [B][max_num][4][256] fp16, which then replaces in_f16
    const void* a4b = nullptr;      // [B][Hc][Wc][128] fp16; null = in_f16 is the dense cDa map
    const void* da_w = nullptr; const float* da_bias = nullptr; int da_g32_first = 0;
    void* da_compact = nullptr;
    // fp32 variant (OMNI_PREC_F32 / OMNI_PREC_SPLIT): convDb + L2 norm in exact f32 only at the <= 4 * max_num cells the sampler reads.
    // cda_f32: the heads layer's cDa half, NHWC fp32 with pixel stride in_cstride floats; wdb_f32: conv_pack_weights_f32 of convDb; cx / cy:
    // scratch [ceil8(batch * max_num * 4)][256] fp32 each.  Bit-identical to the dense map + sp_sample_kernel (a 1x1 conv and the per-cell norm
    // do not look at neighbours).
    const float* cda_f32 = nullptr; const void* wdb_f32 = nullptr; float* cx = nullptr; float* cy = nullptr;
    int n_cu = 0; const void* zero_page = nullptr;
    // OMNI_PREC_SPLIT: convDa itself only at those cells (conv_split_c128_sparse) -- a4b_split: conv4b's split-64 frames; the rows land in cx directly
    // (cda_f32 is then not read); da_w / da_bias / da_g32_first as above, da_inv = the fused heads layer's split_inv
    const void* a4b_split = nullptr; float da_inv = 0.f;
    // ... and convDb + the norm over those rows with split operands too (convdb_l2norm_split; null = the exact-f32 convolution + l2norm_channels)
    const void* wdb_hi = nullptr; const void* wdb_lo = nullptr;
    // the detector head already thresholded the map (conv.h DetCand): SpPostBuffers::cand_bits is filled; sp_mask_kernel compacts it into the candidate
    // lists and makes the window masks of the candidates only -- sp_cand_kernel, which re-reads the whole heat map through LDS tiles, is not launched
    bool cand_fused = false;
    // a heat map that did not come from the head (omni_sp_postprocess_dense): the same two steps as separate kernels -- sp_thresh_kernel makes the bitmap,
    // sp_mask_kernel lists and masks -- so that every edge case of the post-processing tests runs through the kernel the pipeline uses
    bool cand_from_list = false;
};

// semi: [B][H][W] f32 probability map; desc_nhwc: [B][H/8][W/8][256] f32 (channel-normalised coarse descriptors; unused when sparse.in_f16 is set)
int sp_postprocess(hipStream_t stream, const SpPostParams& p, const SpPostBuffers& b, const float* semi,
                   const float* desc_nhwc, int batch, const SpSparseDesc& sparse = SpSparseDesc{});

// layout helpers between the reference's NCHW binding layout and the internal NHWC
int nchw_to_nhwc(hipStream_t stream, const float* in, float* out, int batch, int C, int HW);
int nhwc_to_nchw(hipStream_t stream, const float* in, float* out, int batch, int C, int HW);

}  // namespace omni
