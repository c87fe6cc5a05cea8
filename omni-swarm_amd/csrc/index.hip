// Global-descriptor index: drop-in for faiss::IndexFlatIP as LoopDetector uses it
//   add    : swarm_loop/src/loop_detector.cpp:166,169   (IndexFlatIP::add(1, x))
//   search : swarm_loop/src/loop_detector.cpp:213       (IndexFlatIP::search(1, q, k, D, I))
//   ntotal : swarm_loop/src/loop_detector.cpp:167,170,232,291
//
// HBM layout, rows appended in insertion order (the recency rule `label <= ntotal - max_index`, loop_detector.cpp:232, needs it):
//   fp32 shard: one row-major [capacity][dim] matrix;
//   fp16 shard: "T16" blocks of 16 rows, 16-byte k-chunks interleaved over the rows of a block,
//                 half index(row, k) = (row / 16) * 16 * dim + ((k / 8) * 16 + row % 16) * 8 + k % 8
//               -- exactly the A-operand order of v_mfma_f32_16x16x32_f16 (lane = row % 16 + 16 * (chunk % 4)): every wave instruction
//               of both scan kernels reads ONE contiguous KiB (with row-major rows the matrix-core kernel touched 64 cache lines per
//               instruction and stalled on the L1 tag rate at 4.8 TB/s).
// Search = one streaming scan kernel (HBM-bound: dim*sizeof(elem) bytes per row, read exactly once)
// that writes one 64-bit sortable key per (query,row), then the exact hierarchical top-k of topk.h.
#include "config.h"
#include "common.h"
#include "topk.h"
#include <sys/stat.h>
#include <climits>

struct omni_index {
    omni_ctx* ctx = nullptr;
    int dim = 0;
    int storage = OMNI_STORE_F32;
    int64_t capacity = 0;
    int64_t ntotal = 0;
    int rank = 0, world = 1;
    void* db = nullptr;
    // fp32 shards keep an fp16 MIRROR of the rows in the T16 layout (+50 % memory): a batch of queries is first scored against the mirror in ONE
    // pass on the matrix cores (8 KB per row instead of 16, every query of the batch at once), the best K' candidates per query are re-scored
    // EXACTLY against the fp32 rows, and a per-query certificate (the K'-th mirror score plus the fp16 rounding bound is below the k-th exact
    // score) proves that the result is the exact fp32 top-k -- scores bit-identical to the single-query kernel; a query whose certificate
    // fails is searched again by the exact multi-pass scan.  OMNI_INDEX_MIRROR=0 switches the mirror off.
    void* db16 = nullptr;
    uint32_t* norm_max = nullptr;          // device: bits of the largest squared row norm seen (atomicMax on the bit pattern of a float >= 0)
    omni::DevBuf qbuf, keys_a, keys_b, out_d, out_i, stage, mq_q, mq_inv, cert_keys, cert_flags;
    omni::HostBuf hq, hout, hflags;
    int64_t cert_searches = 0, cert_fallbacks = 0;      // statistics: queries answered through the mirror / of those, re-run exactly
    hipEvent_t scan0 = nullptr, scan1 = nullptr;
    bool scan_timed = false;
    std::mutex mu;
    size_t elem() const { return storage == OMNI_STORE_F16 ? 2 : 4; }
};

namespace omni {

#ifdef OMNI_SCAN_NO_NT
#define NT_LOAD(p) (*(p))
#else
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#endif
#define SCAN_THREADS 256
#define SCAN_WAVES (SCAN_THREADS / 64)
#define SCAN_MAX_QB 8
// per-query row limits of a batched prefix search (omni_index_search_batch_prefix_dev): query q only sees rows [0, v[q]); rows beyond
// get the empty key.  Passed by value (kernel argument segment) so that back-to-back enqueues need no staging buffer.
template <int N> struct ScanLimits { int64_t v[N]; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// fp32 shard: one wave per row at a time, QB queries resident in LDS.  Each lane streams 16 B per load (1 KiB per wave instruction, fully
// coalesced).  keys[q][row] = make_key(dot(q, row), row).
template <typename T, int QB>
__global__ void __launch_bounds__(SCAN_THREADS)
ip_scan_kernel(const T* __restrict__ db, int64_t n_rows, int dim, const float* __restrict__ queries,
               uint64_t* __restrict__ keys, int64_t key_stride, ScanLimits<SCAN_MAX_QB> lim) {
    static_assert(sizeof(T) == 4, "fp16 shards use the T16 layout (ip_scan_t16_kernel / ip_scan_mq_kernel)");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sq = reinterpret_cast<float*>(smem_raw);             // [QB][dim]
    for (int i = threadIdx.x * 4; i < QB * dim; i += SCAN_THREADS * 4)
        *reinterpret_cast<float4*>(sq + i) = *reinterpret_cast<const float4*>(queries + i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t wave_global = (int64_t)blockIdx.x * SCAN_WAVES + wave;
    const int64_t wave_stride = (int64_t)gridDim.x * SCAN_WAVES;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    for (int64_t row = wave_global; row < n_rows; row += wave_stride) {
        const T* rp = db + row * dim;
        float acc[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) acc[q] = 0.f;
#pragma unroll 4
        for (int c = lane * 4; c < dim; c += 64 * 4) {
            // every DB byte is read exactly once per scan: non-temporal loads keep the stream out of the way of the LDS-resident
            // queries' neighbours in L2 (MI355X guide: streamed-once operands measure 6.5-6.8 vs 6.4 TB/s with nt)
            const f32x4_t x = NT_LOAD(reinterpret_cast<const f32x4_t*>(rp + c));
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const float4 qq = *reinterpret_cast<const float4*>(sq + q * dim + c);
                acc[q] = fmaf(x[0], qq.x, acc[q]);
                acc[q] = fmaf(x[1], qq.y, acc[q]);
                acc[q] = fmaf(x[2], qq.z, acc[q]);
                acc[q] = fmaf(x[3], qq.w, acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            float s = wave_sum(acc[q]);
            if (lane == 0) keys[(int64_t)q * key_stride + row] = row < lim.v[q] ? omni_make_key(s, (uint32_t)row) : OMNI_KEY_EMPTY;
        }
    }
}

// fp32 shard, QB >= 4 queries (a micro-batch of key frames against a big database): the query block fills the CU's LDS (QB x 16 KB), so
// only one workgroup = one wave per SIMD fits and the one-row-per-wave kernel above cannot keep enough loads in flight (measured 1.4 TB/s
// at 8 queries x 400k rows).  Here a wave walks R rows at once: R x (unroll) 16-byte loads per lane in flight, and every query float4
// read from LDS feeds R rows (LDS traffic / R).  Per row the lane-local fmaf chain and the wave reduction are those of ip_scan_kernel:
// scores are bit-identical to the single-query path.
template <int QB, int R>
__global__ void __launch_bounds__(SCAN_THREADS)
ip_scan_rows_kernel(const float* __restrict__ db, int64_t n_rows, int dim, const float* __restrict__ queries,
                    uint64_t* __restrict__ keys, int64_t key_stride, ScanLimits<SCAN_MAX_QB> lim) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sq = reinterpret_cast<float*>(smem_raw);             // [QB][dim]
    for (int i = threadIdx.x * 4; i < QB * dim; i += SCAN_THREADS * 4)
        *reinterpret_cast<float4*>(sq + i) = *reinterpret_cast<const float4*>(queries + i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t n_groups = (n_rows + R - 1) / R;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    for (int64_t g = (int64_t)blockIdx.x * SCAN_WAVES + wave; g < n_groups; g += (int64_t)gridDim.x * SCAN_WAVES) {
        const float* rp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = g * R + r;
            rp[r] = db + (row < n_rows ? row : n_rows - 1) * dim + lane * 4;      // rows past the end re-read the last one (not written)
        }
        float acc[R][QB];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < QB; ++q) acc[r][q] = 0.f;
#pragma unroll 2
        for (int c = 0; c < dim; c += 64 * 4) {
            f32x4_t x[R];
#pragma unroll
            for (int r = 0; r < R; ++r) x[r] = NT_LOAD(reinterpret_cast<const f32x4_t*>(rp[r] + c));
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const float4 qq = *reinterpret_cast<const float4*>(sq + q * dim + lane * 4 + c);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[r][q] = fmaf(x[r][0], qq.x, acc[r][q]);
                    acc[r][q] = fmaf(x[r][1], qq.y, acc[r][q]);
                    acc[r][q] = fmaf(x[r][2], qq.z, acc[r][q]);
                    acc[r][q] = fmaf(x[r][3], qq.w, acc[r][q]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = g * R + r;
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const float s = wave_sum(acc[r][q]);
                if (lane == 0 && row < n_rows) keys[(int64_t)q * key_stride + row] = row < lim.v[q] ? omni_make_key(s, (uint32_t)row) : OMNI_KEY_EMPTY;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Batched search (BASELINE config 5: 64 concurrent key frames against an fp16 shard).  The VALU scans are compute bound beyond
// a handful of queries (64 queries x 4096 FMAs per row); here the dot products run on the matrix cores and the kernel is back on the
// HBM roofline.  v_mfma_f32_16x16x32_f16 with the QUERIES as the A operand (16 queries x 32 k, from LDS) and a 16-row block of the
// shard as the B operand: the T16 layout stores a block in exactly that operand order, so k-step s of a block is the contiguous KiB
// at s * 1024, lane-linear, read straight from HBM exactly once (non-temporal, MQ_RING k-steps in flight per wave).  In the C
// fragment a lane then holds ONE row (lane & 15) for 4 queries: key stores are 128-byte lines of 16 consecutive rows.
// The <= 64 queries are split once per search into fp16 hi + lo (q * 2^s = hi + lo, s per query so that max|q| lands in
// [2^13, 2^14): 22 significant bits, products exact in the fp32 accumulator -- the same fp32-class scores as the VALU path)
// and laid out by mq_prep_kernel so that one 256-k slice (64 KB: [hi|lo][64 q][32 x 16 B], chunk position XOR (q & 15) ->
// conflict-free ds_read_b128) is a LINEAR 64 KB copy: the next slice streams into the other LDS buffer by LDS-DMA while the
// current one feeds the MFMAs.  A workgroup = 8 waves x MQ_RT tiles of 16 rows; each wave keeps MQ_RT x 4 accumulator tiles
// (16 rows x 64 queries) across the 16 k-slices of a pass, then writes 64-bit keys for topk.h.
//   algorithmic bytes: rows x dim x 2 (read once) + rows x nq x 8 (keys);  MFMA work 2 x 2 x 64 x dim flop per row ~ 28 % of the
//   matrix peak at 5.3 TB/s, so the bound is HBM.
#define MQ_THREADS 512
#define MQ_WAVES 8
#define MQ_RT 4
#define MQ_NQ 64
#define MQ_KS 256
#define MQ_SLICE_BYTES (2 * MQ_NQ * MQ_KS * 2)
#define MQ_SMEM (2 * MQ_SLICE_BYTES)
#define MQ_RING 4
#define MQ_SLICE_STRIDE (MQ_KS * 2 * 16)     // bytes between consecutive k-slices of a 16-row block (T16 layout): 8 KiB, 1 KiB per k-step
typedef _Float16 mq_half8 __attribute__((ext_vector_type(8)));
typedef float mq_f32x4 __attribute__((ext_vector_type(4)));

// one block per query slot (64): power-of-two scale, hi/lo split, swizzled slice layout; slots >= nq are zero queries
__global__ void __launch_bounds__(256)
mq_prep_kernel(const float* __restrict__ q, int nq, int dim, uint4* __restrict__ qp, float* __restrict__ inv_scale) {
    const int qi = blockIdx.x;
    __shared__ float red[4];
    float m = 0.f;
    if (qi < nq)
        for (int i = threadIdx.x; i < dim; i += 256) m = fmaxf(m, fabsf(q[(int64_t)qi * dim + i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float scale = 1.f, inv = qi < nq ? 1.f : 0.f;
    if (qi < nq && m > 0.f && m < 3.0e38f) {
        int sh = 13 - ilogbf(m);
        sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
        scale = ldexpf(1.f, sh);
        inv = ldexpf(1.f, -sh);
    }
    if (threadIdx.x == 0) inv_scale[qi] = inv;
    for (int c = threadIdx.x; c < dim / 8; c += 256) {
        mq_half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = qi < nq ? q[(int64_t)qi * dim + c * 8 + j] * scale : 0.f;
            const _Float16 h = (_Float16)v;
            hi[j] = h;
            lo[j] = (_Float16)(v - (float)h);
        }
        const int slice = c >> 5, pos = (c & 31) ^ (qi & 15);
        uint4* dst = qp + ((int64_t)(slice * 2) * MQ_NQ + qi) * 32 + pos;
        dst[0] = *reinterpret_cast<uint4*>(&hi);
        dst[MQ_NQ * 32] = *reinterpret_cast<uint4*>(&lo);
    }
}

template <int OFF>
__device__ __forceinline__ void mq_read(uint32_t addr, mq_half8& dst) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ void mq_wait(mq_half8 (&b)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
}
__device__ __forceinline__ void mq_read_step(uint32_t addr, mq_half8 (&b)[8]) {
    // b[2t + hl]: queries 16t .. 16t+15, hl = 0 hi / 1 lo
    mq_read<0 * 8192>(addr, b[0]);          mq_read<32768 + 0 * 8192>(addr, b[1]);
    mq_read<1 * 8192>(addr, b[2]);          mq_read<32768 + 1 * 8192>(addr, b[3]);
    mq_read<2 * 8192>(addr, b[4]);          mq_read<32768 + 2 * 8192>(addr, b[5]);
    mq_read<3 * 8192>(addr, b[6]);          mq_read<32768 + 3 * 8192>(addr, b[7]);
}
// Row loads and their waits are inline asm with hand-counted vmcnt: with an LDS-DMA in flight hipcc turns every vmcnt wait it inserts
// itself into vmcnt(0) (the DMA counts as a pending FLAT access), which would drain the prefetch ring at every k-step.
template <int OFF>
__device__ __forceinline__ void mq_load_a(uint64_t p, mq_half8& dst) {       // non-temporal: every block byte is read once (measured +15 %)
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(dst) : "v"(p), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void mq_wait_a(mq_half8 (&a)[MQ_RT]) {
    static_assert(MQ_RT == 4, "operand list below");
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "i"(N));
}

// k-step S of a slice (a = ring of DB-block fragments, b = the query fragments of this step): ring slot S%4 was filled 4 k-steps ago
// (younger VMEM ops at its wait: 3 slots x 4 loads, plus the 8 DMA instructions issued at the top of the slice for S < 4) and is
// refilled right after its MFMAs with the block data MQ_RING k-steps ahead (next slice / next pass for the second half).  The prologue
// in the kernel issues its loads in the same slot-major order: the counts are only valid for that order.  Offsets: 1 KiB per k-step; the
// instruction's signed 13-bit immediate covers -4096..4095, so the block pointers are biased to the MIDDLE of their 8 KiB slice.
template <int S>
__device__ __forceinline__ void mq_step(uint32_t bx, const uint64_t (&cur)[MQ_RT], const uint64_t (&nxt)[MQ_RT], mq_half8 (&a)[MQ_RING][MQ_RT],
                                        mq_half8 (&b)[8], mq_f32x4 (&acc)[MQ_RT][4]) {
    mq_wait(b);
    mq_wait_a<3 * MQ_RT + (S < 4 ? 8 : 0)>(a[S % MQ_RING]);
#pragma unroll
    for (int hl = 0; hl < 2; ++hl)
#pragma unroll
        for (int i = 0; i < MQ_RT; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[2 * t + hl], a[S % MQ_RING][i], acc[i][t], 0, 0, 0);
    // query fragments of the next k-step: ONE register set (the ring and the accumulators leave no room for two) -- the reads are issued
    // behind the last MFMA that consumes the old fragments (in-order issue; LDS returns >= 64 cycles later), their latency is covered by
    // the other wave of the SIMD, and the matrix pipe is only ~30 % busy at the HBM roofline anyway
    if constexpr (S + 1 < 8) mq_read_step(bx ^ (64 * (S + 1)), b);
#pragma unroll
    for (int i = 0; i < MQ_RT; ++i) {
        if constexpr (S + MQ_RING < 8) mq_load_a<(S + MQ_RING) * 1024 - 4096>(cur[i], a[S % MQ_RING][i]);
        else mq_load_a<(S + MQ_RING - 8) * 1024 - 4096>(nxt[i], a[S % MQ_RING][i]);
    }
    if constexpr (S + 1 < 8) mq_step<S + 1>(bx, cur, nxt, a, b, acc);
}

__global__ void __launch_bounds__(MQ_THREADS, 1)
ip_scan_mq_kernel(const _Float16* __restrict__ db, int64_t n_rows, int dim, const char* __restrict__ qp,
                  const float* __restrict__ inv_scale, int nq, uint64_t* __restrict__ keys, int64_t key_stride, int rotate,
                  ScanLimits<MQ_NQ> lim) {
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, c4 = lane >> 4;
    const int64_t total_tiles = (n_rows + 15) >> 4;
    constexpr int BT = MQ_WAVES * MQ_RT;                          // 32 tiles = 512 rows per block
    const int64_t blocks = (total_tiles + BT - 1) / BT;
    const int passes = (int)((blocks - blockIdx.x + gridDim.x - 1) / gridDim.x);   // blocks blockIdx.x, +gridDim.x, ... (>= 1)
    const int S = dim / MQ_KS;
    const int U = passes * S;                                    // slice units this workgroup walks through
    // Every block walks the k-slices in its own rotation (block index mod S, a function of the ROW only, so a row's score does not
    // depend on the shard size or the launch grid): the workgroups of the chip do not all sit at the same offset of their 128 KB blocks
    // at the same time (worth a few percent with row-major rows, within noise with T16 blocks; OMNI_MQ_ROT=0 switches it off).
    auto blk_of = [&](int p) { return (int64_t)blockIdx.x + (int64_t)p * gridDim.x; };
    auto rot_of = [&](int p) { return rotate ? (int)(blk_of(p) % S) : 0; };

    auto dma = [&](int slice, int buf) {                          // 64 KB linear copy, 8 wave-instructions of 1 KiB per wave
        const char* g = qp + (size_t)slice * MQ_SLICE_BYTES + wave * 8192 + lane * 16;
        char* l = smem_raw + buf * MQ_SLICE_BYTES + wave * 8192;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + j * 1024),
                                             (__attribute__((address_space(3))) void*)(l + j * 1024), 16, 0, 0);
    };
    // block pointers of pass p (T16 layout: a 16-row tile is 16*dim contiguous halfs, k-step s of it the KiB at s*1024, lane-linear);
    // out-of-range tiles are clamped to the last one
    auto row_ptrs = [&](int p, uint64_t (&rp)[MQ_RT]) {
        const int64_t tile0 = blk_of(p) * BT;
#pragma unroll
        for (int i = 0; i < MQ_RT; ++i) {
            int64_t t = tile0 + i * MQ_WAVES + wave;
            t = t < total_tiles ? t : total_tiles - 1;
            rp[i] = (uint64_t)(uintptr_t)(db + t * 16 * dim + lane * 8) + 4096;     // biased to the middle of a slice (see mq_step)
        }
    };

    // query-operand address: query row n = r16 (+16t by immediate), chunk (4s + c4) ^ n -> ((n*512 + ((c4 ^ (n&3)) << 4) + ((n&12) << 4)) ^ (64 s)
    const uint32_t bbase = lds0 + r16 * 512 + ((c4 ^ (r16 & 3)) << 4) + ((r16 & 12) << 4);

    mq_f32x4 acc[MQ_RT][4];
#pragma unroll
    for (int i = 0; i < MQ_RT; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = mq_f32x4{0.f, 0.f, 0.f, 0.f};

    uint64_t cur[MQ_RT];                                          // this unit's rows at its k-slice (the only pointers carried across units)
    row_ptrs(0, cur);
    int rot = rot_of(0);
    dma(rot, 0);
    mq_half8 a[MQ_RING][MQ_RT];
#pragma unroll
    for (int i = 0; i < MQ_RT; ++i) cur[i] += (uint64_t)rot * MQ_SLICE_STRIDE;
#pragma unroll
    for (int i = 0; i < MQ_RT; ++i) mq_load_a<-4096>(cur[i], a[0][i]);      // slot-major, the issue order of mq_step's refills
#pragma unroll
    for (int i = 0; i < MQ_RT; ++i) mq_load_a<-3072>(cur[i], a[1][i]);
#pragma unroll
    for (int i = 0; i < MQ_RT; ++i) mq_load_a<-2048>(cur[i], a[2][i]);
#pragma unroll
    for (int i = 0; i < MQ_RT; ++i) mq_load_a<-1024>(cur[i], a[3][i]);

    int j = 0, p = 0;
    for (int u = 0; u < U; ++u) {
        // slice u sits in buffer u&1 once every wave's DMA part has landed: loads retire in order, the MQ_RING*MQ_RT row loads issued
        // after the DMA may stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(MQ_RING * MQ_RT) : "memory");
        asm volatile("s_barrier" ::: "memory");
        const bool last_slice = j + 1 == S;
        const int pn = p + 1 < passes ? p + 1 : p;
        const int rot_n = last_slice ? rot_of(pn) : rot;
        int sl = j + rot;                 sl = sl >= S ? sl - S : sl;        // this unit's k-slice
        int sn = (last_slice ? 0 : j + 1) + rot_n; sn = sn >= S ? sn - S : sn;   // the next unit's
        // next slice into the other buffer (every wave is past its reads of slice u-1).  Issued unconditionally -- after the last unit it
        // is a harmless copy -- so that the vmcnt distances below are constants.
        dma(sn, (u + 1) & 1);
        asm volatile("" ::: "memory");                             // the DMA stays in front of the row loads below (vmcnt distances)
        uint64_t nxt[MQ_RT];                                      // the next unit's rows at its k-slice
        if (last_slice) {
            row_ptrs(pn, nxt);
#pragma unroll
            for (int i = 0; i < MQ_RT; ++i) nxt[i] += (uint64_t)sn * MQ_SLICE_STRIDE;
        } else {
#pragma unroll
            for (int i = 0; i < MQ_RT; ++i) nxt[i] = cur[i] + (int64_t)(sn - sl) * MQ_SLICE_STRIDE;
        }
        const uint32_t bx = bbase + ((u & 1) ? MQ_SLICE_BYTES : 0);
        mq_half8 b[8];
        mq_read_step(bx, b);
        mq_step<0>(bx, cur, nxt, a, b, acc);
        if (last_slice) {
            // pass done: keys[q][row] for this wave's tiles.  The queries are the A operand, the DB rows the B operand, so in the C layout
            // of the 16x16 tile a lane holds row l&15 of queries 4*(l>>4)+e: every store instruction writes four full 128-byte lines
            // (16 consecutive rows of one query each).
            // Stores share vmcnt with loads and retire out of order with them: drain the ring first so that the counted waits of the
            // next unit only ever have loads younger than the ones they wait for.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qq = 16 * t + 4 * c4 + e;
                    const float inv = inv_scale[qq];
                    const int64_t lim_q = lim.v[qq];
                    uint64_t* kq = keys + (int64_t)qq * key_stride;
#pragma unroll
                    for (int i = 0; i < MQ_RT; ++i) {
                        const int64_t row = (blk_of(p) * BT + i * MQ_WAVES + wave) * 16 + r16;
                        if (qq < nq && row < n_rows)
                            kq[row] = row < lim_q ? omni_make_key(acc[i][t][e] * inv, (uint32_t)row) : OMNI_KEY_EMPTY;
                    }
                    __builtin_amdgcn_sched_barrier(0);          // one (t, e) at a time: the ring and the accumulators leave few free registers
                }
            }
#pragma unroll
            for (int i = 0; i < MQ_RT; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[i][t] = mq_f32x4{0.f, 0.f, 0.f, 0.f};
            j = 0; ++p; rot = rot_n;
        } else {
            ++j;
        }
#pragma unroll
        for (int i = 0; i < MQ_RT; ++i) cur[i] = nxt[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // trailing ring refills (harmless re-reads) retire before the wave ends
}

__global__ void decode_topk_kernel(const uint64_t* __restrict__ keys, int nq, int k, int rank, int world,
                                   float* __restrict__ D, int64_t* __restrict__ I) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * k) return;
    uint64_t key = keys[i];
    uint32_t inv = (uint32_t)(key & 0xFFFFFFFFull);
    if (key == OMNI_KEY_EMPTY) { D[i] = -3.402823466e+38f; I[i] = -1; return; }
    uint32_t local = 0xFFFFFFFFu - inv;
    D[i] = omni_orderable_f32((uint32_t)(key >> 32));
    I[i] = (int64_t)local * world + rank;
}

// queries of a batched search picked out of a row buffer (e.g. MobileNetVLAD's output): dst[i] = src[idx.v[i]]
struct GatherIdx { int64_t v[MQ_NQ]; };
__global__ void gather_rows_kernel(const float* __restrict__ src, GatherIdx idx, int dim, float* __restrict__ dst) {
    const float4* s = reinterpret_cast<const float4*>(src + idx.v[blockIdx.x] * dim);
    float4* d = reinterpret_cast<float4*>(dst + (int64_t)blockIdx.x * dim);
    for (int i = threadIdx.x; i < dim / 4; i += blockDim.x) d[i] = s[i];
}

// decode_topk_kernel for a subset of the queries: result row i goes to query slot dst.v[i]
__global__ void decode_topk_scatter_kernel(const uint64_t* __restrict__ keys, int nq, int k, int rank, int world, GatherIdx dst, float* __restrict__ D,
                                           int64_t* __restrict__ I) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * k) return;
    const int64_t o = dst.v[i / k] * k + (i % k);
    const uint64_t key = keys[i];
    if (key == OMNI_KEY_EMPTY) { D[o] = -3.402823466e+38f; I[o] = -1; return; }
    D[o] = omni_orderable_f32((uint32_t)(key >> 32));
    I[o] = (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull)) * world + rank;
}

// half index of (row, 16-byte chunk) in the T16 layout
__device__ __forceinline__ int64_t t16_chunk(int64_t row, int chunk, int dim) {
    return (row >> 4) * 16 * (int64_t)dim + ((int64_t)chunk * 16 + (row & 15)) * 8;
}
// in: n x dim fp32 rows -> T16 rows row0 .. row0+n-1 (one thread per 8-element chunk)
__global__ void f32_to_t16_kernel(const float* __restrict__ in, __half* __restrict__ db, int64_t row0, int64_t n, int dim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cpr = dim >> 3;
    if (i >= n * cpr) return;
    const int64_t r = i / cpr;
    const int c = (int)(i - r * cpr);
    const float4 lo = *reinterpret_cast<const float4*>(in + r * dim + c * 8);
    const float4 hi = *reinterpret_cast<const float4*>(in + r * dim + c * 8 + 4);
    __half2 h[4] = {__floats2half2_rn(lo.x, lo.y), __floats2half2_rn(lo.z, lo.w), __floats2half2_rn(hi.x, hi.y), __floats2half2_rn(hi.z, hi.w)};
    *reinterpret_cast<uint4*>(db + t16_chunk(row0 + r, c, dim)) = *reinterpret_cast<uint4*>(h);
}
// snapshots hold plain row-major fp16 rows: T16 <-> rows (TO_ROWS: db -> rows, else rows -> db)
template <bool TO_ROWS>
__global__ void t16_rows_kernel(__half* __restrict__ db, __half* __restrict__ rows, int64_t row0, int64_t n, int dim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cpr = dim >> 3;
    if (i >= n * cpr) return;
    const int64_t r = i / cpr;
    const int c = (int)(i - r * cpr);
    uint4* a = reinterpret_cast<uint4*>(db + t16_chunk(row0 + r, c, dim));
    uint4* b = reinterpret_cast<uint4*>(rows + r * dim + c * 8);
    if (TO_ROWS) *b = *a; else *a = *b;
}

// fp16 shard, up to SCAN_MAX_QB queries on the VALU: one wave per 16-row block at a time, lane = (row r = l & 15, chunk c = l >> 4) --
// each wave instruction streams one contiguous KiB of the block; a row's dot product sits in its 4 lanes (r, r+16, r+32, r+48).
template <int QB>
__global__ void __launch_bounds__(SCAN_THREADS)
ip_scan_t16_kernel(const __half* __restrict__ db, int64_t n_rows, int dim, const float* __restrict__ queries,
                   uint64_t* __restrict__ keys, int64_t key_stride, ScanLimits<SCAN_MAX_QB> lim) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sq = reinterpret_cast<float*>(smem_raw);             // [QB][dim]
    for (int i = threadIdx.x * 4; i < QB * dim; i += SCAN_THREADS * 4)
        *reinterpret_cast<float4*>(sq + i) = *reinterpret_cast<const float4*>(queries + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, c4 = lane >> 4;
    const int64_t n_blocks = (n_rows + 15) >> 4;
    const int steps = dim >> 5;                                   // 32 k (4 chunks) per step
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    for (int64_t blk = (int64_t)blockIdx.x * SCAN_WAVES + wave; blk < n_blocks; blk += (int64_t)gridDim.x * SCAN_WAVES) {
        const u32x4_t* bp = reinterpret_cast<const u32x4_t*>(db + blk * 16 * dim) + lane;
        float acc[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) acc[q] = 0.f;
#pragma unroll 4
        for (int st = 0; st < steps; ++st) {
            const u32x4_t xx = NT_LOAD(bp + st * 64);
            const uint4 x = make_uint4(xx[0], xx[1], xx[2], xx[3]);
            const __half2* h = reinterpret_cast<const __half2*>(&x);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { float2 f = __half22float2(h[e]); v[2 * e] = f.x; v[2 * e + 1] = f.y; }
            const int k0 = (st * 4 + c4) * 8;
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const float4 q0 = *reinterpret_cast<const float4*>(sq + q * dim + k0);
                const float4 q1 = *reinterpret_cast<const float4*>(sq + q * dim + k0 + 4);
                acc[q] = fmaf(v[0], q0.x, acc[q]); acc[q] = fmaf(v[1], q0.y, acc[q]);
                acc[q] = fmaf(v[2], q0.z, acc[q]); acc[q] = fmaf(v[3], q0.w, acc[q]);
                acc[q] = fmaf(v[4], q1.x, acc[q]); acc[q] = fmaf(v[5], q1.y, acc[q]);
                acc[q] = fmaf(v[6], q1.z, acc[q]); acc[q] = fmaf(v[7], q1.w, acc[q]);
            }
        }
        const int64_t row = blk * 16 + r16;
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            float sum = acc[q];
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            if (c4 == 0 && row < n_rows) keys[(int64_t)q * key_stride + row] = row < lim.v[q] ? omni_make_key(sum, (uint32_t)row) : OMNI_KEY_EMPTY;
        }
    }
}

// largest squared norm of rows [0, n) of `rows` (row-major fp32), folded into *norm_max (bit pattern of a non-negative float: unsigned order)
__global__ void __launch_bounds__(256)
row_norm_max_kernel(const float* __restrict__ rows, int64_t n, int dim, uint32_t* __restrict__ norm_max) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* rp = rows + row * dim;
    float a = 0.f;
    for (int c = lane * 4; c < dim; c += 256) {
        const float4 x = *reinterpret_cast<const float4*>(rp + c);
        a = fmaf(x.x, x.x, a); a = fmaf(x.y, x.y, a); a = fmaf(x.z, x.z, a); a = fmaf(x.w, x.w, a);
    }
    a = wave_sum(a);
    if (lane == 0) atomicMax(norm_max, __float_as_uint(a == a ? a : __builtin_inff()));       // NaN rows poison the certificate (-> exact path)
}

// exact re-scoring of the mirror pass's candidates: one wave per (query, candidate), the lane-local fmaf chain and the wave reduction of
// ip_scan_kernel<float, 1> -- the score is bit-identical to the single-query scan's
__global__ void __launch_bounds__(256)
cert_refine_kernel(const float* __restrict__ db, int dim, const float* __restrict__ queries, const uint64_t* __restrict__ cand /*[nq][kp]*/, int nq, int kp,
                   uint64_t* __restrict__ out /*[nq][kp]*/) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (int64_t)nq * kp) return;
    const int q = (int)(w / kp);
    const uint64_t key = cand[w];
    if (key == OMNI_KEY_EMPTY) { if (lane == 0) out[w] = OMNI_KEY_EMPTY; return; }
    const uint32_t row = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
    const float* rp = db + (int64_t)row * dim;
    const float* qp = queries + (int64_t)q * dim;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    float acc = 0.f;
#pragma unroll 4
    for (int c = lane * 4; c < dim; c += 64 * 4) {
        const f32x4_t x = *reinterpret_cast<const f32x4_t*>(rp + c);
        const float4 qq = *reinterpret_cast<const float4*>(qp + c);
        acc = fmaf(x[0], qq.x, acc); acc = fmaf(x[1], qq.y, acc); acc = fmaf(x[2], qq.z, acc); acc = fmaf(x[3], qq.w, acc);
    }
    const float sc = wave_sum(acc);
    if (lane == 0) out[w] = omni_make_key(sc, row);
}

// per query (one wave): rank the kp <= 64 exact keys, write the best k (as decode_topk_kernel would) and the certificate:
//   every row NOT among the candidates has a mirror score <= m = the kp-th mirror score, hence an exact score <= m + eps with
//   eps = 2^-11 |q| max|row|  (fp16 rounding of the row, element-wise relative 2^-11, Cauchy-Schwarz)  +  slack for the mirror pass's own
//   fp32-class arithmetic and for fp16 subnormals;  if the k-th exact score is above that, no other row can enter the top k.
__global__ void __launch_bounds__(64)
cert_select_kernel(const uint64_t* __restrict__ exact /*[nq][kp]*/, const uint64_t* __restrict__ mirror /*[nq][kp]*/, int kp, int k, const float* __restrict__ queries,
                   int dim, const uint32_t* __restrict__ norm_max, int rank, int world, float* __restrict__ D, int64_t* __restrict__ I, int* __restrict__ flags) {
    const int q = blockIdx.x, lane = threadIdx.x;
    __shared__ uint64_t s[64];
    __shared__ uint64_t kth_key;
    if (lane == 0) kth_key = OMNI_KEY_EMPTY;
    const uint64_t mine = lane < kp ? exact[(int64_t)q * kp + lane] : OMNI_KEY_EMPTY;
    s[lane] = mine;
    float qn = 0.f, q1 = 0.f;
    for (int c = lane; c < dim; c += 64) { const float v = queries[(int64_t)q * dim + c]; qn = fmaf(v, v, qn); q1 += fabsf(v); }
    qn = wave_sum(qn); q1 = wave_sum(q1);
    __syncthreads();
    int r = 0;
    for (int j = 0; j < 64; ++j) r += (s[j] > mine);
    if (mine == OMNI_KEY_EMPTY) r = 64;                                   // all empties share rank >= number of valid keys
    if (r < k) {
        D[(int64_t)q * k + r] = omni_orderable_f32((uint32_t)(mine >> 32));
        I[(int64_t)q * k + r] = (int64_t)(0xFFFFFFFFu - (uint32_t)(mine & 0xFFFFFFFFull)) * world + rank;
    }
    int valid = 0;
    for (int j = 0; j < 64; ++j) valid += (s[j] != OMNI_KEY_EMPTY);
    for (int i = valid + lane; i < k; i += 64) { D[(int64_t)q * k + i] = -3.402823466e+38f; I[(int64_t)q * k + i] = -1; }
    // certificate
    if (r == k - 1) kth_key = mine;
    __syncthreads();
    if (lane == 0) {
        const uint64_t last = mirror[(int64_t)q * kp + kp - 1];           // sorted descending: the smallest retained mirror key
        int need = 0;
        if (last != OMNI_KEY_EMPTY) {                                     // otherwise every row of the prefix was a candidate: exact by construction
            if (valid < k) need = 1;
            else {
                const float kth = omni_orderable_f32((uint32_t)(kth_key >> 32));
                const float m = omni_orderable_f32((uint32_t)(last >> 32));
                const float rn = sqrtf(__uint_as_float(*norm_max)), qnorm = sqrtf(qn);
                const float eps = 4.8829e-4f * 1.0001f * qnorm * rn + 6.0e-8f * q1 + 4.0e-6f * qnorm * rn;
                need = !(kth > m + eps) || !(rn < 6.0e4f);                // NaN / inf / fp16-overflowing rows: not certifiable
            }
        }
        flags[q] = need;
    }
}

static int launch_scan(hipStream_t st, const omni_index* ix, int64_t n, int qb, const float* q_dev, uint64_t* keys, int64_t key_stride,
                       const int64_t* limits) {
    ScanLimits<SCAN_MAX_QB> lim;
    for (int q = 0; q < SCAN_MAX_QB; ++q) lim.v[q] = limits && q < qb ? limits[q] : INT64_MAX;
    const bool f16 = ix->storage == OMNI_STORE_F16;
    int cus = ix->ctx->prop.multiProcessorCount > 0 ? ix->ctx->prop.multiProcessorCount : 256;
    int64_t want = cdiv64(f16 ? cdiv64(n, 16) : n, SCAN_WAVES);      // a wave walks rows (fp32) or 16-row blocks (fp16)
    int grid = (int)(want < (int64_t)cus * 8 ? want : (int64_t)cus * 8);
    if (grid < 1) grid = 1;
    size_t smem = (size_t)qb * ix->dim * sizeof(float);
    static const int rows_min_qb = config_process()[CFG_SCAN_ROWS_MIN];
    if (!f16 && qb >= rows_min_qb && qb >= 4) {
        constexpr int R = 4;
        int64_t want_g = cdiv64(cdiv64(n, R), SCAN_WAVES);
        int grid_g = (int)(want_g < (int64_t)cus * 8 ? want_g : (int64_t)cus * 8);
        if (grid_g < 1) grid_g = 1;
#define OMNI_ROWS_CASE(QB)                                                                                              \
    case QB: {                                                                                                          \
        auto kf = ip_scan_rows_kernel<QB, R>;                                                                           \
        static DynSmemState attr;                                                                                       \
        OMNI_HIP_TRY(ensure_dyn_smem(attr, (const void*)kf, smem));                                                     \
        hipLaunchKernelGGL(kf, dim3(grid_g), dim3(SCAN_THREADS), smem, st, reinterpret_cast<const float*>(ix->db), n, ix->dim, q_dev, keys,  \
                           key_stride, lim);                                                                            \
        break; }
        switch (qb) {
            OMNI_ROWS_CASE(4) OMNI_ROWS_CASE(5) OMNI_ROWS_CASE(6) OMNI_ROWS_CASE(7) OMNI_ROWS_CASE(8)
            default: set_error("internal: bad query block %d", qb); return OMNI_ERR_INVALID;
        }
#undef OMNI_ROWS_CASE
        OMNI_LAUNCH_CHECK();
        return OMNI_OK;
    }
#define OMNI_SCAN_CASE(QB)                                                                                              \
    case QB:                                                                                                            \
        if (f16)                                                                                                        \
            hipLaunchKernelGGL((ip_scan_t16_kernel<QB>), dim3(grid), dim3(SCAN_THREADS), smem, st,                      \
                               reinterpret_cast<const __half*>(ix->db), n, ix->dim, q_dev, keys, key_stride, lim);      \
        else                                                                                                            \
            hipLaunchKernelGGL((ip_scan_kernel<float, QB>), dim3(grid), dim3(SCAN_THREADS), smem, st,                   \
                               reinterpret_cast<const float*>(ix->db), n, ix->dim, q_dev, keys, key_stride, lim);       \
        break;
    switch (qb) {
        OMNI_SCAN_CASE(1) OMNI_SCAN_CASE(2) OMNI_SCAN_CASE(3) OMNI_SCAN_CASE(4)
        OMNI_SCAN_CASE(5) OMNI_SCAN_CASE(6) OMNI_SCAN_CASE(7) OMNI_SCAN_CASE(8)
        default: set_error("internal: bad query block %d", qb); return OMNI_ERR_INVALID;
    }
#undef OMNI_SCAN_CASE
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// up to MQ_NQ queries in one pass over an fp16 shard (ip_scan_mq_kernel)
static int launch_scan_mq(hipStream_t st, omni_index* ix, int64_t n, int nq, const float* q_dev, uint64_t* keys, int64_t key_stride,
                          const int64_t* limits, const void* db_t16 = nullptr) {
    ScanLimits<MQ_NQ> lim;
    for (int q = 0; q < MQ_NQ; ++q) lim.v[q] = limits && q < nq ? limits[q] : INT64_MAX;
    auto kfn = ip_scan_mq_kernel;
    static DynSmemState attr;
    OMNI_HIP_TRY(ensure_dyn_smem(attr, (const void*)kfn, MQ_SMEM));
    const int slices = ix->dim / MQ_KS;
    int rc;
    if ((rc = ix->mq_q.ensure((size_t)slices * MQ_SLICE_BYTES))) return rc;
    if ((rc = ix->mq_inv.ensure(MQ_NQ * sizeof(float)))) return rc;
    hipLaunchKernelGGL(mq_prep_kernel, dim3(MQ_NQ), dim3(256), 0, st, q_dev, nq, ix->dim, ix->mq_q.as<uint4>(), ix->mq_inv.as<float>());
    OMNI_LAUNCH_CHECK();
    const int64_t tiles = cdiv64(n, 16);
    const int cus = ix->ctx->prop.multiProcessorCount > 0 ? ix->ctx->prop.multiProcessorCount : 256;
    // one workgroup per CU (128 KB of LDS each) walking 512-row blocks b, b + grid, ...
    const int64_t blocks = cdiv64(tiles, MQ_WAVES * MQ_RT);
    const int64_t grid = blocks < cus ? blocks : cus;
    static const int rotate = config_process()[CFG_MQ_ROT];
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(MQ_THREADS), MQ_SMEM, st, reinterpret_cast<const _Float16*>(db_t16 ? db_t16 : ix->db),
                       n, ix->dim, ix->mq_q.as<char>(), ix->mq_inv.as<float>(), nq, keys, key_stride, rotate, lim);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// number of queries from which an fp16 shard is searched on the matrix cores (OMNI_MQ_MIN overrides: 1 = always, 0 = never)
static int mq_min_queries() {
    Config c;                                         // (read per call: tests switch it between searches of one process)
    const int v = config_resolve(&c) == OMNI_OK ? c[CFG_MQ_MIN] : kCfgOptions[CFG_MQ_MIN].def;
    return v <= 0 ? (1 << 30) : v;
}

static int mirror_rows(omni_index* ix, int64_t row0, int64_t n);
static int ensure_capacity(omni_index* ix, int64_t rows) {
    if (rows <= ix->capacity) return OMNI_OK;
    int64_t cap = ix->capacity > 0 ? ix->capacity : 1024;
    while (cap < rows) cap *= 2;
    cap = (cap + 15) & ~(int64_t)15;                             // whole 16-row blocks (fp16 T16 layout)
    void* nd = nullptr;
    void* nm = nullptr;
    OMNI_HIP_TRY(hipMalloc(&nd, (size_t)cap * ix->dim * ix->elem()));
    // (every failure below frees what this call allocated: the index keeps its old buffers)
    auto fail = [&](int code, const char* what) { if (nd) (void)hipFree(nd); if (nm) (void)hipFree(nm); set_error("index growth to %lld rows: %s failed", (long long)cap, what); return code; };
    const size_t old_blocks = (size_t)((ix->ntotal + 15) & ~(int64_t)15);
    if (ix->db && ix->ntotal > 0 &&
        hipMemcpyAsync(nd, ix->db, old_blocks * ix->dim * ix->elem(), hipMemcpyDeviceToDevice, ix->ctx->stream) != hipSuccess) return fail(OMNI_ERR_HIP, "the copy of the rows");
    bool rebuild_mirror = false;
    if (ix->norm_max) {                                              // fp32 shard with an fp16 mirror
        if (hipMalloc(&nm, (size_t)cap * ix->dim * 2) != hipSuccess) { nm = nullptr; return fail(OMNI_ERR_NOMEM, "hipMalloc of the fp16 mirror"); }
        if (ix->db16 && ix->ntotal > 0) {
            if (hipMemcpyAsync(nm, ix->db16, old_blocks * ix->dim * 2, hipMemcpyDeviceToDevice, ix->ctx->stream) != hipSuccess) return fail(OMNI_ERR_HIP, "the copy of the mirror");
        } else if (ix->ntotal > 0) {
            rebuild_mirror = true;                                   // rows without a mirror (it could not be allocated when they were loaded): convert them now
        }
    }
    if (hipStreamSynchronize(ix->ctx->stream) != hipSuccess) return fail(OMNI_ERR_HIP, "the stream");
    if (ix->db) (void)hipFree(ix->db);
    if (ix->db16) (void)hipFree(ix->db16);
    ix->db = nd;
    ix->db16 = nm;
    ix->capacity = cap;
    if (rebuild_mirror) return mirror_rows(ix, 0, ix->ntotal);       // the certificate of the batched search assumes mirror == fp16(row) for EVERY row
    return OMNI_OK;
}

// rows [row0, row0 + n) of an fp32 shard were just written (stream order): bring the fp16 mirror and the norm bound up to date
static int mirror_rows(omni_index* ix, int64_t row0, int64_t n) {
    if (!ix->db16 || n <= 0) return OMNI_OK;
    hipStream_t st = ix->ctx->stream;
    const float* rows = (const float*)ix->db + row0 * ix->dim;
    hipLaunchKernelGGL(f32_to_t16_kernel, dim3((unsigned)cdiv64(n * ix->dim / 8, 256)), dim3(256), 0, st, rows, (__half*)ix->db16, row0, n, ix->dim);
    OMNI_LAUNCH_CHECK();
    hipLaunchKernelGGL(row_norm_max_kernel, dim3((unsigned)cdiv64(n, 4)), dim3(256), 0, st, rows, n, ix->dim, ix->norm_max);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

// x_dev: n x dim fp32 in HBM -> appended (converted when storage is fp16)
static int append_dev(omni_index* ix, int64_t n, const float* x_dev) {
    OMNI_REQUIRE(ix->ntotal + n < 0xFFFFFFFFll, OMNI_ERR_CAPACITY, "index full: row ids are packed into 32 bits (omni_make_key)");
    int rc = ensure_capacity(ix, ix->ntotal + n);
    if (rc) return rc;
    hipStream_t st = ix->ctx->stream;
    const int64_t cnt = n * ix->dim;
    if (ix->storage == OMNI_STORE_F32) {
        OMNI_HIP_TRY(hipMemcpyAsync((float*)ix->db + ix->ntotal * ix->dim, x_dev, (size_t)cnt * 4,
                                    hipMemcpyDeviceToDevice, st));
        if ((rc = mirror_rows(ix, ix->ntotal, n))) return rc;
    } else {
        hipLaunchKernelGGL(f32_to_t16_kernel, dim3((unsigned)cdiv64(cnt / 8, 256)), dim3(256), 0, st, x_dev, (__half*)ix->db, ix->ntotal, n,
                           ix->dim);
        OMNI_LAUNCH_CHECK();
    }
    ix->ntotal += n;
    return OMNI_OK;
}

// limits (host, [nq], optional): query q only sees rows [0, limits[q]) -- the scan covers [0, n) once for all of them
static int search_dev(omni_index* ix, int nq, const float* q_dev, int k, float* D_dev, int64_t* I_dev, int64_t n_limit = -1,
                      const int64_t* limits = nullptr) {
    hipStream_t st = ix->ctx->stream;
    const int64_t n = n_limit >= 0 && n_limit < ix->ntotal ? n_limit : ix->ntotal;      // rows [0, n) take part
    const int64_t chunks = cdiv64(n > 0 ? n : 1, TOPK_CHUNK);
    const int64_t per_q = (n > chunks * k ? n : chunks * k);
    int rc;
    if ((rc = ix->keys_a.ensure((size_t)nq * per_q * 8))) return rc;
    if ((rc = ix->keys_b.ensure((size_t)nq * per_q * 8))) return rc;
    uint64_t* ka = ix->keys_a.as<uint64_t>();
    uint64_t* kb = ix->keys_b.as<uint64_t>();
    // fp32 shard, a batch of queries: mirror pass + exact refinement + certificate (see omni_index::db16)
    const int kp = k + 24 > 2 * k ? k + 24 : 2 * k;                 // candidates per query
    // (small databases stay on the exact kernels: their scans are launch-bound, the mirror's extra launches and its host check would cost more
    // than the halved HBM traffic saves -- OMNI_INDEX_MIRROR_MIN_ROWS, default 32768)
    static const int64_t mirror_min_rows = config_process()[CFG_INDEX_MIRROR_MIN_ROWS];
    if (n >= mirror_min_rows && n > 0 && ix->storage == OMNI_STORE_F32 && ix->db16 && nq >= mq_min_queries() && nq <= MQ_NQ && kp <= TOPK_SEL_MAX_K) {
        if ((rc = ix->cert_keys.ensure((size_t)nq * kp * 8))) return rc;
        if ((rc = ix->cert_flags.ensure((size_t)MQ_NQ * 4))) return rc;
        if ((rc = ix->hflags.ensure((size_t)MQ_NQ * 4))) return rc;
        OMNI_HIP_TRY(hipEventRecord(ix->scan0, st));
        if ((rc = launch_scan_mq(st, ix, n, nq, q_dev, ka, n, limits, ix->db16))) return rc;
        OMNI_HIP_TRY(hipEventRecord(ix->scan1, st));
        ix->scan_timed = true;
        uint64_t* cand = nullptr;
        if ((rc = topk_keys(st, ka, kb, nq, n, kp, &cand))) return rc;
        hipLaunchKernelGGL(cert_refine_kernel, dim3((unsigned)cdiv64((int64_t)nq * kp, 4)), dim3(256), 0, st, reinterpret_cast<const float*>(ix->db), ix->dim, q_dev,
                           cand, nq, kp, ix->cert_keys.as<uint64_t>());
        OMNI_LAUNCH_CHECK();
        hipLaunchKernelGGL(cert_select_kernel, dim3(nq), dim3(64), 0, st, ix->cert_keys.as<uint64_t>(), cand, kp, k, q_dev, ix->dim, ix->norm_max, ix->rank, ix->world,
                           D_dev, I_dev, ix->cert_flags.as<int>());
        OMNI_LAUNCH_CHECK();
        // the certificate is checked on the host: one small copy + a wait for this search (its callers fetch the results right afterwards);
        // a query that is not certified is searched again with the exact scan -- results identical either way
        OMNI_HIP_TRY(hipMemcpyAsync(ix->hflags.p, ix->cert_flags.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
        OMNI_HIP_TRY(hipStreamSynchronize(st));
        ix->cert_searches += nq;
        const int* fl = ix->hflags.as<int>();
        static const bool force_fallback = config_process()[CFG_INDEX_CERT_FAIL] != 0;      // test hook
        // the uncertified queries, together, through the exact scan (blocks of <= 8 queries per pass over the fp32 rows: what every batch
        // cost before the mirror existed)
        GatherIdx gi;
        int64_t lim_f[MQ_NQ];
        int nfl = 0;
        for (int q = 0; q < nq; ++q)
            if (fl[q] || force_fallback) { gi.v[nfl] = q; lim_f[nfl] = limits ? limits[q] : INT64_MAX; ++nfl; }
        if (nfl == 0) return OMNI_OK;
        ix->cert_fallbacks += nfl;
        for (int i = nfl; i < MQ_NQ; ++i) gi.v[i] = 0;
        if ((rc = ix->stage.ensure((size_t)nfl * ix->dim * 4))) return rc;
        hipLaunchKernelGGL(gather_rows_kernel, dim3(nfl), dim3(256), 0, st, q_dev, gi, ix->dim, ix->stage.as<float>());
        OMNI_LAUNCH_CHECK();
        int max_qb = (int)(131072 / ((size_t)ix->dim * 4));
        max_qb = max_qb > SCAN_MAX_QB ? SCAN_MAX_QB : (max_qb < 1 ? 1 : max_qb);
        for (int q0 = 0; q0 < nfl; q0 += max_qb) {
            const int qb = nfl - q0 < max_qb ? nfl - q0 : max_qb;
            if ((rc = launch_scan(st, ix, n, qb, ix->stage.as<float>() + (int64_t)q0 * ix->dim, ka + (int64_t)q0 * n, n, lim_f + q0))) return rc;
        }
        uint64_t* r1 = nullptr;
        if ((rc = topk_keys(st, ka, kb, nfl, n, k, &r1))) return rc;
        hipLaunchKernelGGL(decode_topk_scatter_kernel, dim3(cdiv(nfl * k, 256)), dim3(256), 0, st, r1, nfl, k, ix->rank, ix->world, gi, D_dev, I_dev);
        OMNI_LAUNCH_CHECK();
        return OMNI_OK;
    }
    if (n > 0) {
        OMNI_HIP_TRY(hipEventRecord(ix->scan0, st));
        const bool mq = ix->storage == OMNI_STORE_F16 && nq >= mq_min_queries();
        for (int q0 = 0; mq && q0 < nq; q0 += MQ_NQ) {
            const int qb = nq - q0 < MQ_NQ ? nq - q0 : MQ_NQ;
            if ((rc = launch_scan_mq(st, ix, n, qb, q_dev + (int64_t)q0 * ix->dim, ka + (int64_t)q0 * n, n, limits ? limits + q0 : nullptr))) return rc;
        }
        int max_qb = (int)(131072 / ((size_t)ix->dim * 4));       // the query block lives in LDS: <= 128 KB of it
        max_qb = max_qb > SCAN_MAX_QB ? SCAN_MAX_QB : (max_qb < 1 ? 1 : max_qb);
        for (int q0 = 0; !mq && q0 < nq; q0 += max_qb) {
            int qb = nq - q0 < max_qb ? nq - q0 : max_qb;
            rc = launch_scan(st, ix, n, qb, q_dev + (int64_t)q0 * ix->dim, ka + (int64_t)q0 * n, n, limits ? limits + q0 : nullptr);
            if (rc) return rc;
        }
        OMNI_HIP_TRY(hipEventRecord(ix->scan1, st));
        ix->scan_timed = true;
    }
    uint64_t* res = nullptr;
    if ((rc = topk_keys(st, ka, kb, nq, n, k, &res))) return rc;
    hipLaunchKernelGGL(decode_topk_kernel, dim3(cdiv(nq * k, 256)), dim3(256), 0, st, res, nq, k, ix->rank, ix->world,
                       D_dev, I_dev);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

}  // namespace omni

extern "C" {

omni_index* omni_index_create(omni_ctx* ctx, int dim, int storage, int64_t initial_capacity_rows) {
    if (!ctx) { omni::set_error("null ctx"); return nullptr; }
    if (dim <= 0 || dim % 512 != 0 || dim > 8192) {
        omni::set_error("dim=%d unsupported: must be a positive multiple of 512, <= 8192 (reference: 4096)", dim);
        return nullptr;
    }
    if (storage != OMNI_STORE_F32 && storage != OMNI_STORE_F16) { omni::set_error("bad storage %d", storage); return nullptr; }
    (void)hipSetDevice(ctx->device);
    omni_index* ix = new omni_index();
    ix->ctx = ctx; ix->dim = dim; ix->storage = storage;
    if (hipEventCreate(&ix->scan0) != hipSuccess || hipEventCreate(&ix->scan1) != hipSuccess) {
        omni::set_error("hipEventCreate failed"); delete ix; return nullptr;
    }
    static const bool mirror_on = omni::config_process()[omni::CFG_INDEX_MIRROR] != 0;
    if (storage == OMNI_STORE_F32 && mirror_on) {
        if (hipMalloc((void**)&ix->norm_max, 4) != hipSuccess || hipMemsetAsync(ix->norm_max, 0, 4, ctx->stream) != hipSuccess) {
            omni::set_error("hipMalloc failed"); omni_index_destroy(ix); return nullptr;
        }
    }
    if (initial_capacity_rows > 0 && omni::ensure_capacity(ix, initial_capacity_rows) != OMNI_OK) { omni_index_destroy(ix); return nullptr; }
    return ix;
}

void omni_index_destroy(omni_index* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->ctx->device);
    (void)hipStreamSynchronize(ix->ctx->stream);
    if (ix->db) (void)hipFree(ix->db);
    if (ix->db16) (void)hipFree(ix->db16);
    if (ix->norm_max) (void)hipFree(ix->norm_max);
    ix->cert_keys.release(); ix->cert_flags.release(); ix->hflags.release();
    ix->qbuf.release(); ix->keys_a.release(); ix->keys_b.release(); ix->out_d.release(); ix->out_i.release();
    ix->stage.release(); ix->mq_q.release(); ix->mq_inv.release(); ix->hq.release(); ix->hout.release();
    if (ix->scan0) (void)hipEventDestroy(ix->scan0);
    if (ix->scan1) (void)hipEventDestroy(ix->scan1);
    delete ix;
}

int64_t omni_index_ntotal(const omni_index* ix) { return ix ? ix->ntotal : -1; }
int omni_index_dim(const omni_index* ix) { return ix ? ix->dim : -1; }

int omni_index_reset(omni_index* ix) {
    OMNI_REQUIRE(ix, OMNI_ERR_INVALID, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->ntotal = 0;
    return OMNI_OK;
}

int omni_index_truncate(omni_index* ix, int64_t n_rows) {
    OMNI_REQUIRE(ix && n_rows >= 0, OMNI_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    OMNI_REQUIRE(n_rows <= ix->ntotal, OMNI_ERR_INVALID, "cannot truncate %lld rows to %lld", (long long)ix->ntotal, (long long)n_rows);
    ix->ntotal = n_rows;            // stream order keeps earlier enqueued searches on their own prefix; later appends overwrite the tail
    return OMNI_OK;
}

int omni_index_set_shard(omni_index* ix, int rank, int world) {
    OMNI_REQUIRE(ix && world >= 1 && rank >= 0 && rank < world, OMNI_ERR_INVALID, "bad shard %d/%d", rank, world);
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->rank = rank; ix->world = world;
    // A shard's searches are ENQUEUED (omni_shard_step_enqueue: no host synchronisation between its collectives, the scan and the copy of the lists).  The
    // mirror path of the batched search checks its exactness certificate on the host -- a wait in the middle of that unit -- so a sharded index searches
    // its fp32 rows with the exact many-query scan (no certificate, nothing to check) and drops the mirror: +0 % HBM instead of +50 %.
    if (ix->db16) { (void)hipFree(ix->db16); ix->db16 = nullptr; }
    if (ix->norm_max) { (void)hipFree(ix->norm_max); ix->norm_max = nullptr; }
    return OMNI_OK;
}

int omni_index_add_dev(omni_index* ix, int64_t n, const float* x_dev) {
    omni::TraceRange trace_range("index add");
    OMNI_REQUIRE(ix && x_dev && n >= 0, OMNI_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    (void)hipSetDevice(ix->ctx->device);
    if (n == 0) return OMNI_OK;
    return omni::append_dev(ix, n, x_dev);
}

int omni_index_add(omni_index* ix, int64_t n, const float* x_host) {
    omni::TraceRange trace_range("index add (host rows)");
    OMNI_REQUIRE(ix && x_host && n >= 0, OMNI_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    (void)hipSetDevice(ix->ctx->device);
    if (n == 0) return OMNI_OK;
    // stream in slabs of <= 4096 rows through a device staging buffer (fp16 storage converts on device)
    const int64_t slab = 4096;
    int rc;
    for (int64_t s = 0; s < n; s += slab) {
        int64_t m = n - s < slab ? n - s : slab;
        size_t bytes = (size_t)m * ix->dim * 4;
        if (ix->storage == OMNI_STORE_F32) {
            OMNI_REQUIRE(ix->ntotal + m < 0xFFFFFFFFll, OMNI_ERR_CAPACITY, "index full: row ids are packed into 32 bits (omni_make_key)");
            if ((rc = omni::ensure_capacity(ix, ix->ntotal + m))) return rc;
            OMNI_HIP_TRY(hipMemcpyAsync((float*)ix->db + ix->ntotal * ix->dim, x_host + s * ix->dim, bytes,
                                        hipMemcpyHostToDevice, ix->ctx->stream));
            if ((rc = omni::mirror_rows(ix, ix->ntotal, m))) return rc;
            ix->ntotal += m;
        } else {
            if ((rc = ix->stage.ensure(bytes))) return rc;
            OMNI_HIP_TRY(hipMemcpyAsync(ix->stage.p, x_host + s * ix->dim, bytes, hipMemcpyHostToDevice, ix->ctx->stream));
            if ((rc = omni::append_dev(ix, m, ix->stage.as<float>()))) return rc;
        }
        OMNI_HIP_TRY(hipStreamSynchronize(ix->ctx->stream));   // x_host may be pageable / reused by the caller
    }
    return OMNI_OK;
}

int omni_index_search_dev(omni_index* ix, int nq, const float* q_dev, int k, float* D_dev, int64_t* I_dev) {
    omni::TraceRange trace_range("index search");
    OMNI_REQUIRE(ix && q_dev && D_dev && I_dev, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(nq >= 1 && nq <= 4096, OMNI_ERR_CAPACITY, "nq=%d outside [1,4096]", nq);
    OMNI_REQUIRE(k >= 1 && k <= TOPK_MAX_K, OMNI_ERR_CAPACITY, "k=%d outside [1,%d]", k, TOPK_MAX_K);
    std::lock_guard<std::mutex> lk(ix->mu);
    (void)hipSetDevice(ix->ctx->device);
    return omni::search_dev(ix, nq, q_dev, k, D_dev, I_dev);
}

int omni_index_search_prefix_dev(omni_index* ix, int nq, const float* q_dev, int k, int64_t n_limit, float* D_dev, int64_t* I_dev) {
    omni::TraceRange trace_range("index search (prefix)");
    OMNI_REQUIRE(ix && q_dev && D_dev && I_dev, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(nq >= 1 && nq <= 4096, OMNI_ERR_CAPACITY, "nq=%d outside [1,4096]", nq);
    OMNI_REQUIRE(k >= 1 && k <= TOPK_MAX_K, OMNI_ERR_CAPACITY, "k=%d outside [1,%d]", k, TOPK_MAX_K);
    OMNI_REQUIRE(n_limit >= 0, OMNI_ERR_INVALID, "n_limit=%lld < 0", (long long)n_limit);
    std::lock_guard<std::mutex> lk(ix->mu);
    (void)hipSetDevice(ix->ctx->device);
    return omni::search_dev(ix, nq, q_dev, k, D_dev, I_dev, n_limit);
}

int omni_index_search_batch_prefix_dev(omni_index* ix, int nq, const float* rows_dev, const int64_t* row_idx, int k,
                                       const int64_t* n_limits, float* D_dev, int64_t* I_dev) {
    omni::TraceRange trace_range("index search (batch, prefix)");
    OMNI_REQUIRE(ix && rows_dev && n_limits && D_dev && I_dev, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(nq >= 1 && nq <= MQ_NQ, OMNI_ERR_CAPACITY, "nq=%d outside [1,%d]", nq, MQ_NQ);
    OMNI_REQUIRE(k >= 1 && k <= TOPK_MAX_K, OMNI_ERR_CAPACITY, "k=%d outside [1,%d]", k, TOPK_MAX_K);
    std::lock_guard<std::mutex> lk(ix->mu);
    (void)hipSetDevice(ix->ctx->device);
    int64_t lim[MQ_NQ], nmax = 0;
    for (int q = 0; q < nq; ++q) {
        OMNI_REQUIRE(n_limits[q] >= 0, OMNI_ERR_INVALID, "n_limits[%d]=%lld < 0", q, (long long)n_limits[q]);
        lim[q] = n_limits[q] < ix->ntotal ? n_limits[q] : ix->ntotal;
        nmax = lim[q] > nmax ? lim[q] : nmax;
    }
    const float* q_dev = rows_dev;
    if (row_idx) {
        omni::GatherIdx gi;
        for (int q = 0; q < MQ_NQ; ++q) gi.v[q] = q < nq ? row_idx[q] : 0;
        int rc = ix->qbuf.ensure((size_t)MQ_NQ * ix->dim * 4);
        if (rc) return rc;
        hipLaunchKernelGGL(omni::gather_rows_kernel, dim3(nq), dim3(256), 0, ix->ctx->stream, rows_dev, gi, ix->dim, ix->qbuf.as<float>());
        OMNI_LAUNCH_CHECK();
        q_dev = ix->qbuf.as<float>();
    }
    return omni::search_dev(ix, nq, q_dev, k, D_dev, I_dev, nmax, lim);
}

int omni_index_search(omni_index* ix, int nq, const float* q_host, int k, float* D, int64_t* I) {
    omni::TraceRange trace_range("index search (host queries)");
    OMNI_REQUIRE(ix && q_host && D && I, OMNI_ERR_INVALID, "null argument");
    OMNI_REQUIRE(nq >= 1 && nq <= 4096, OMNI_ERR_CAPACITY, "nq=%d outside [1,4096]", nq);
    OMNI_REQUIRE(k >= 1 && k <= TOPK_MAX_K, OMNI_ERR_CAPACITY, "k=%d outside [1,%d]", k, TOPK_MAX_K);
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->ntotal == 0) {      // faiss pads an empty index's result with -1 labels; nothing to launch
        for (size_t i = 0; i < (size_t)nq * k; ++i) { D[i] = -3.402823466e+38f; I[i] = -1; }
        return OMNI_OK;
    }
    (void)hipSetDevice(ix->ctx->device);
    hipStream_t st = ix->ctx->stream;
    int rc;
    const size_t qbytes = (size_t)nq * ix->dim * 4;
    if ((rc = ix->qbuf.ensure(qbytes))) return rc;
    if ((rc = ix->hq.ensure(qbytes))) return rc;
    if ((rc = ix->out_d.ensure((size_t)nq * k * 4))) return rc;
    if ((rc = ix->out_i.ensure((size_t)nq * k * 8))) return rc;
    if ((rc = ix->hout.ensure((size_t)nq * k * 12))) return rc;
    memcpy(ix->hq.p, q_host, qbytes);
    OMNI_HIP_TRY(hipMemcpyAsync(ix->qbuf.p, ix->hq.p, qbytes, hipMemcpyHostToDevice, st));
    if ((rc = omni::search_dev(ix, nq, ix->qbuf.as<float>(), k, ix->out_d.as<float>(), ix->out_i.as<int64_t>()))) return rc;
    char* ho = ix->hout.as<char>();
    OMNI_HIP_TRY(hipMemcpyAsync(ho, ix->out_i.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
    OMNI_HIP_TRY(hipMemcpyAsync(ho + (size_t)nq * k * 8, ix->out_d.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
    OMNI_HIP_TRY(hipStreamSynchronize(st));
    memcpy(I, ho, (size_t)nq * k * 8);
    memcpy(D, ho + (size_t)nq * k * 8, (size_t)nq * k * 4);
    return OMNI_OK;
}

namespace {
struct OmnxHeader { char magic[8]; int32_t dim, storage; int64_t ntotal; };
}

int omni_index_save(omni_index* ix, const char* path) {
    OMNI_REQUIRE(ix && path, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    (void)hipSetDevice(ix->ctx->device);
    FILE* f = fopen(path, "wb");
    OMNI_REQUIRE(f, OMNI_ERR_INVALID, "cannot open %s for writing", path);
    OmnxHeader h{};
    memcpy(h.magic, "OMNX1\0\0", 8); h.dim = ix->dim; h.storage = ix->storage; h.ntotal = ix->ntotal;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    const size_t row = (size_t)ix->dim * ix->elem(), slab_rows = 4096;
    int rc = ix->hout.ensure(slab_rows * row);
    const bool f16 = ix->storage == OMNI_STORE_F16;              // snapshots hold plain row-major rows: de-block the T16 layout per slab
    if (!rc && f16) rc = ix->stage.ensure(slab_rows * row);
    for (int64_t s = 0; ok && rc == OMNI_OK && s < ix->ntotal; s += slab_rows) {
        const size_t m = (size_t)(ix->ntotal - s < (int64_t)slab_rows ? ix->ntotal - s : slab_rows);
        if (f16)
            hipLaunchKernelGGL((omni::t16_rows_kernel<true>), dim3((unsigned)omni::cdiv64((int64_t)m * (ix->dim / 8), 256)), dim3(256), 0,
                               ix->ctx->stream, (__half*)ix->db, ix->stage.as<__half>(), s, (int64_t)m, ix->dim);
        if (hipMemcpyAsync(ix->hout.p, f16 ? (const char*)ix->stage.p : (const char*)ix->db + (size_t)s * row, m * row, hipMemcpyDeviceToHost,
                           ix->ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ix->ctx->stream) != hipSuccess) { omni::set_error("device read failed while saving"); rc = OMNI_ERR_HIP; break; }
        ok = fwrite(ix->hout.p, row, m, f) == m;
    }
    ok = (fclose(f) == 0) && ok;
    if (rc) return rc;
    OMNI_REQUIRE(ok, OMNI_ERR_INVALID, "short write to %s", path);
    return OMNI_OK;
}

int omni_index_load(omni_index* ix, const char* path) {
    OMNI_REQUIRE(ix && path, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    (void)hipSetDevice(ix->ctx->device);
    FILE* f = fopen(path, "rb");
    OMNI_REQUIRE(f, OMNI_ERR_INVALID, "cannot open %s", path);
    OmnxHeader h{};
    int rc = OMNI_OK;
    const size_t row = (size_t)ix->dim * ix->elem(), slab_rows = 4096;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "OMNX1\0\0", 8) != 0) { omni::set_error("%s is not an OMNX1 snapshot", path); rc = OMNI_ERR_INVALID; }
    else if (h.dim != ix->dim || h.storage != ix->storage || h.ntotal < 0 || h.ntotal >= 0xFFFFFFFFll) {
        omni::set_error("snapshot %s is dim=%d storage=%d ntotal=%lld, the handle is dim=%d storage=%d", path, h.dim, h.storage, (long long)h.ntotal,
                        ix->dim, ix->storage);
        rc = OMNI_ERR_INVALID;
    } else {
        // the header is untrusted: the file must hold exactly ntotal rows before anything is allocated or overwritten
        struct stat sb;
        if (fstat(fileno(f), &sb) != 0 || (uint64_t)sb.st_size != sizeof(h) + (uint64_t)h.ntotal * row) {
            omni::set_error("%s is truncated or corrupt: header says %lld rows, file size does not match", path, (long long)h.ntotal);
            rc = OMNI_ERR_INVALID;
        }
    }
    // load into a FRESH buffer and swap on success: a failed load leaves the handle as it was
    void* nd = nullptr;
    int64_t cap = 0;
    if (!rc) {
        cap = ((h.ntotal > 0 ? h.ntotal : 1) + 15) & ~(int64_t)15;
        if (hipMalloc(&nd, (size_t)cap * row) != hipSuccess) { omni::set_error("hipMalloc(%zu) failed while loading", (size_t)cap * row); nd = nullptr; rc = OMNI_ERR_NOMEM; }
    }
    if (!rc) rc = ix->hout.ensure(slab_rows * row);
    const bool f16 = ix->storage == OMNI_STORE_F16;
    if (!rc && f16) rc = ix->stage.ensure(slab_rows * row);
    for (int64_t s = 0; !rc && s < h.ntotal; s += slab_rows) {
        const size_t m = (size_t)(h.ntotal - s < (int64_t)slab_rows ? h.ntotal - s : slab_rows);
        if (fread(ix->hout.p, row, m, f) != m) { omni::set_error("%s is truncated", path); rc = OMNI_ERR_INVALID; break; }
        bool okc = hipMemcpyAsync(f16 ? (char*)ix->stage.p : (char*)nd + (size_t)s * row, ix->hout.p, m * row, hipMemcpyHostToDevice,
                                  ix->ctx->stream) == hipSuccess;
        if (okc && f16)
            hipLaunchKernelGGL((omni::t16_rows_kernel<false>), dim3((unsigned)omni::cdiv64((int64_t)m * (ix->dim / 8), 256)), dim3(256), 0,
                               ix->ctx->stream, (__half*)nd, ix->stage.as<__half>(), s, (int64_t)m, ix->dim);
        if (!okc || hipStreamSynchronize(ix->ctx->stream) != hipSuccess) { omni::set_error("device write failed while loading"); rc = OMNI_ERR_HIP; }
    }
    fclose(f);
    if (rc) { if (nd) (void)hipFree(nd); return rc; }
    (void)hipStreamSynchronize(ix->ctx->stream);
    if (ix->db) (void)hipFree(ix->db);
    if (ix->db16) { (void)hipFree(ix->db16); ix->db16 = nullptr; }
    ix->db = nd; ix->capacity = cap; ix->ntotal = h.ntotal;
    if (ix->norm_max) {                                              // rebuild the fp16 mirror of the loaded rows
        if (hipMalloc(&ix->db16, (size_t)cap * ix->dim * 2) != hipSuccess) {
            // no room for the mirror: the rows are loaded, the index works -- every batch goes through the exact scan from now on.  (Leaving norm_max set with
            // no mirror made the next growth allocate an EMPTY mirror for the rows already held: the certificate would have trusted garbage.)
            ix->db16 = nullptr;
            (void)hipGetLastError();
            (void)hipFree(ix->norm_max); ix->norm_max = nullptr;
            return OMNI_OK;
        }
        OMNI_HIP_TRY(hipMemsetAsync(ix->norm_max, 0, 4, ix->ctx->stream));
        if ((rc = omni::mirror_rows(ix, 0, h.ntotal))) return rc;
        OMNI_HIP_TRY(hipStreamSynchronize(ix->ctx->stream));
    }
    return OMNI_OK;
}

int omni_index_cert_stats(omni_index* ix, int64_t* searches, int64_t* fallbacks) {
    OMNI_REQUIRE(ix, OMNI_ERR_INVALID, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (searches) *searches = ix->cert_searches;
    if (fallbacks) *fallbacks = ix->cert_fallbacks;
    return OMNI_OK;
}

int omni_index_last_scan_ms(omni_index* ix, float* ms) {
    OMNI_REQUIRE(ix && ms, OMNI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    OMNI_REQUIRE(ix->scan_timed, OMNI_ERR_INVALID, "no timed scan yet");
    OMNI_HIP_TRY(hipEventSynchronize(ix->scan1));
    OMNI_HIP_TRY(hipEventElapsedTime(ms, ix->scan0, ix->scan1));
    return OMNI_OK;
}

int omni_topk_merge(int n_lists, int nq, int k_each, const float* D_lists, const int64_t* I_lists, int k_out,
                    float* D, int64_t* I) {
    OMNI_REQUIRE(n_lists >= 1 && nq >= 1 && k_each >= 1 && k_out >= 1 && D_lists && I_lists && D && I, OMNI_ERR_INVALID,
                 "bad argument");
    // Host-side P*k-way merge (P <= 8 shards, k <= 1024): simple insertion into a k_out list per query.
    for (int q = 0; q < nq; ++q) {
        float* Dq = D + (size_t)q * k_out;
        int64_t* Iq = I + (size_t)q * k_out;
        for (int j = 0; j < k_out; ++j) { Dq[j] = -3.402823466e+38f; Iq[j] = -1; }
        for (int l = 0; l < n_lists; ++l) {
            const float* Dl = D_lists + ((size_t)l * nq + q) * k_each;
            const int64_t* Il = I_lists + ((size_t)l * nq + q) * k_each;
            for (int j = 0; j < k_each; ++j) {
                if (Il[j] < 0) continue;
                const float s = Dl[j];
                const int64_t id = Il[j];
                int pos = k_out;
                // position of first entry that (s,id) precedes: score desc, id asc
                while (pos > 0 && (Iq[pos - 1] < 0 || s > Dq[pos - 1] || (s == Dq[pos - 1] && id < Iq[pos - 1]))) --pos;
                if (pos >= k_out) continue;
                for (int m = k_out - 1; m > pos; --m) { Dq[m] = Dq[m - 1]; Iq[m] = Iq[m - 1]; }
                Dq[pos] = s; Iq[pos] = id;
            }
        }
    }
    return OMNI_OK;
}

}  // extern "C"
