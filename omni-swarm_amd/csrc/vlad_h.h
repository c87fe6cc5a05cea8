// fp16-operand inverted-residual block kernel of MobileNetVLAD (vlad_h.hip), used by vlad.hip when the handle runs at OMNI_PREC_F16.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace omni {

struct VladHBlockArgs {
    const float* in; float* out;     // fp32 NHWC block input / output (the residual stream stays fp32)
    const void* blob;                // vlad_hblock_pack()
    const float* bp;                 // [cout] projection bias
    int Hi, Wi, Ho, Wo, cin, hid, cout, res, batch;
};

bool vlad_hblock_supported(int cin, int hid, int cout, int stride);
size_t vlad_hblock_blob_bytes(int cin, int hid, int cout);
void vlad_hblock_pack(int cin, int hid, int cout, const float* we, const float* be, const float* wd, const float* bd, const float* wp, void* out);
int launch_vlad_hblock(hipStream_t st, const VladHBlockArgs& a, int stride);

}  // namespace omni

// ---- split-operand variant (vlad_sblock_kernel): fp16 matrix-core operands carried as (hi, lo) pairs, fp32-class results ----
namespace omni {
struct VladSBlockArgs {
    const float* in; float* out;     // fp32 NHWC block input / output
    const void* blob;                // vlad_sblock_pack(): [dw taps + bias, all chunks][per chunk: expand fragments | projection fragments]
    const float* bp;                 // [cout] projection bias
    int Hi, Wi, Ho, Wo, cin, hid, cout, res, batch;
    int n_cu;                        // compute units of the device (persistent grid)
    unsigned m_img, m_tx;            // ceil(2^32 / tiles per image), ceil(2^32 / tiles per row): set by the launcher
    int dbg;                         // OMNI_VLAD_SB_DBG (timing ablations only, WRONG results): bit 0 no result stores, bit 1 no input prefetch
    unsigned long long* trace;       // OMNI_VLAD_SB_TRACE=1: s_memtime stamps of workgroup 0 (debug only), else nullptr
    // the constant region of the fisheye mask (vlad.hip, vlad_plan_mask_skip): tile rows [sk_y0, sk_y1) x tile columns [sk_x0, sk_x0 + sk_w) hold one
    // constant vector, written once, and are left out of the tile walk (sk_y1 <= sk_y0: every tile runs); the rest is set by the launcher
    int sk_y0 = 0, sk_y1 = 0, sk_x0 = 0, sk_w = 0;
    int sk_bw = 0, sk_act = 0, sk_above = 0, sk_upto = 0;      // tile columns left in a band row, tiles of an image that run, ... above the band, ... down to its last row
    unsigned m_act = 0, m_bw = 0;
    int persist;                     // OMNI_VLAD_SB_PERSIST as the handle saw it: 0 = one tile per workgroup, N >= 1 = N x (CUs x resident workgroups) persistent workgroups (read by the launcher only)
};
bool vlad_sblock_supported(int cin, int hid, int cout, int stride);
size_t vlad_sblock_blob_bytes(int cin, int hid, int cout);
void vlad_sblock_pack(int cin, int hid, int cout, const float* we, const float* be, const float* wd, const float* bd, const float* wp, void* out);
int launch_vlad_sblock(hipStream_t st, const VladSBlockArgs& a, int stride);
}  // namespace omni
