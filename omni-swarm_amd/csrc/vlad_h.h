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
