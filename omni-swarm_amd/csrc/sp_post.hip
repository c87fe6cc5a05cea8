// SuperPoint post-processing kernels (see sp_post.h for the reference map).
//
// NMS2 is order dependent (a serial pass over candidates in row-major order, superpoint_tensorrt.cpp:265-283).
// Its outcome has the closed form (checked against a literal simulation in tests/):
//     alive(p)   <=> cand(p) and no EARLIER (row-major) alive q in the 9x9 window with conf(q) > conf(p)
//     survive(p) <=> alive(p) and no alive q anywhere in the window with conf(q) > conf(p)
// alive() only depends on strictly-higher-confidence earlier neighbours, so it is a DAG and a chaotic relaxation
// (unknown -> alive/dead as soon as all relevant neighbours are decided) converges to the unique serial answer.
// Split in two: (1) sp_cand_kernel, parallel over image tiles, thresholds the map and records for every candidate WHICH window
// positions hold a higher-confidence candidate (two 40-bit masks); (2) sp_nms_kernel, one 1024-thread workgroup per image,
// runs the relaxation over those masks with the 2-bit pixel state plane held in LDS (600x480 -> 72 KB of the CU's 160 KB,
// no heat-map reads inside the sweeps), then selects the top max_num survivors on 64-bit (confidence, row-major index) keys
// by radix-selecting the cut-off confidence and ranking the few keys above it: final order = confidence desc, index asc
// (the fixed spec; the reference's std::sort leaves ties unspecified).
#include "sp_post.h"
#include "conv.h"
#include "topk.h"

namespace omni {

#define NMS_THREADS 1024
#define NMS_SORT_CAP 8192          // keys held in LDS while sorting (64 KB)
#define ST_NONE 0u
#define ST_UNKNOWN 1u
#define ST_ALIVE 2u
#define ST_DEAD 3u

// ---- getKeyPoints: mask = prob > thres; findNonZero  (:167-173) + the static part of NMS2 --------------------------
// One workgroup per 64x16-pixel tile (4-pixel halo in LDS, non-candidates and out-of-image pixels stored as -inf).  Every
// candidate gets two 40-bit masks of the 9x9 window positions that hold a HIGHER-confidence candidate:
//     earlier mask: rows above + same row to the left  (what can kill it during the serial scan, :265-283)
//     later mask:   same row to the right + rows below (what can still beat an alive point, survive())
// bit (k+4)*9 + (j+4) for window row k in [-4,-1], column j in [-4,4]; bit 36 + (j+4) for k = 0, j in [-4,-1]; the later
// mask uses the same numbering for the NEGATED offset.  These masks are pure functions of the heat map, so this part is
// embarrassingly parallel over the whole chip; only the alive/dead recursion over them stays per image (sp_nms_kernel).
// The candidate list is unordered: NMS2's scan order is the row-major PIXEL order, which the masks encode.
#define CT_W 64
#define CT_H 16
#define NEG_SENTINEL (-3.0e38f)
__global__ void __launch_bounds__(256)
sp_cand_kernel(const float* __restrict__ semi, int W, int H, float thres, int* __restrict__ cand, uint64_t* __restrict__ masks,
               int* __restrict__ counters) {
    __shared__ float tile[CT_H + 8][CT_W + 8 + 1];
    __shared__ int s_wave_cnt[4];
    __shared__ int s_base;
    __shared__ unsigned short s_list[CT_W * CT_H];
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (W + CT_W - 1) / CT_W;
    const int ty0 = (blockIdx.x / tiles_x) * CT_H, tx0 = (blockIdx.x % tiles_x) * CT_W;
    const int hw = W * H;
    const float* sm = semi + (int64_t)b * hw;
    for (int i = tid; i < (CT_H + 8) * (CT_W + 8); i += 256) {
        const int iy = i / (CT_W + 8), ix = i - iy * (CT_W + 8);
        const int gy = ty0 - 4 + iy, gx = tx0 - 4 + ix;
        float v = NEG_SENTINEL;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) { const float t = sm[gy * W + gx]; v = (t > thres) ? t : NEG_SENTINEL; }
        tile[iy][ix] = v;
    }
    __syncthreads();
    // 1. local candidate list (tile coordinates), unordered: 4 consecutive pixels of one row per thread
    const int py = tid >> 4, px0 = (tid & 15) * 4;
    unsigned cmask = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) cmask |= (tile[py + 4][px0 + e + 4] > -1.0e38f) ? (1u << e) : 0u;
    const int mine = __popc(cmask);
    int incl = mine;                                                   // inclusive prefix over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    if (lane == 63) s_wave_cnt[wave] = incl;
    __syncthreads();
    const int total = s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
    if (total == 0) return;
    if (tid == 0) s_base = atomicAdd(&counters[b * 4 + 0], total);     // ONE global atomic per workgroup
    {
        int pos = incl - mine;
        for (int w = 0; w < wave; ++w) pos += s_wave_cnt[w];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cmask & (1u << e)) s_list[pos++] = (unsigned short)(((py + 4) << 8) | (px0 + e + 4));
    }
    __syncthreads();
    // 2. masks: one THREAD per candidate, the 40 + 40 window positions in registers' worth of independent LDS reads (round 4: the first version ran one
    // wave per candidate -- lane = window position, two ballots -- and its serial loop of dependent LDS round trips was the kernel's time: 84 us per 64 images)
    int* out = cand + (int64_t)b * hw;
    uint64_t* mo = masks + (int64_t)b * hw * 2;
    for (int ci = tid; ci < total; ci += 256) {
        const int cy = s_list[ci] >> 8, cx = s_list[ci] & 255;
        const float c0 = tile[cy][cx];
        uint32_t e_lo = 0, e_hi = 0, l_lo = 0, l_hi = 0;              // bits 0-31 / 32-39 of the earlier and the later mask
#pragma unroll
        for (int i = 0; i < 40; ++i) {
            const int k = (i < 36) ? (i / 9 - 4) : 0;
            const int j = (i < 36) ? (i - (k + 4) * 9 - 4) : (i - 40);
            const uint32_t be = tile[cy + k][cx + j] > c0 ? 1u : 0u, bl = tile[cy - k][cx - j] > c0 ? 1u : 0u;
            if (i < 32) { e_lo |= be << i; l_lo |= bl << i; } else { e_hi |= be << (i - 32); l_hi |= bl << (i - 32); }
        }
        const int pos = s_base + ci;
        out[pos] = (ty0 + cy - 4) * W + tx0 + cx - 4;
        mo[2 * (int64_t)pos] = ((uint64_t)e_hi << 32) | e_lo;
        mo[2 * (int64_t)pos + 1] = ((uint64_t)l_hi << 32) | l_lo;
    }
}

// ---- the same two steps when the detector head has thresholded its own output (conv.hip det_emit_candidates; OMNI_SP_FUSED_CAND) ----------------------
// cand_bits: word (cell, hh) of an image = the comparisons prob > thres of rows 0-7 x columns 4 hh .. 4 hh + 3 of the 8 x 8 cell, bit i = (row i >> 2,
// column i & 3).  sp_thresh_kernel makes that bitmap from a heat map the head did not produce (omni_sp_postprocess_dense); sp_mask_kernel turns the
// bitmap into the candidate lists (one prefix + ONE atomic per workgroup) and makes the two window masks of the candidates -- and of nothing else:
// sp_cand_kernel stages every pixel of the map through LDS tiles to find the 2 % that are candidates.
__global__ void __launch_bounds__(256)
sp_thresh_kernel(const float* __restrict__ semi, int W, int H, float thres, uint32_t* __restrict__ bits) {
    const int Wc = W >> 3, words = Wc * (H >> 3) * 2;
    const int b = blockIdx.y, w = blockIdx.x * 256 + threadIdx.x;
    if (w >= words) return;
    const int cell = w >> 1, hh = w & 1, cy = cell / Wc, cx = cell - cy * Wc;
    const float* p = semi + (int64_t)b * W * H + (int64_t)(cy * 8) * W + cx * 8 + 4 * hh;
    uint32_t cm = 0;
#pragma unroll
    for (int ry = 0; ry < 8; ++ry) {
        const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)ry * W);
        cm |= ((v.x > thres ? 1u : 0u) | (v.y > thres ? 2u : 0u) | (v.z > thres ? 4u : 0u) | (v.w > thres ? 8u : 0u)) << (4 * ry);
    }
    bits[(int64_t)b * words + w] = cm;
}

// One workgroup = 256 bitmap words (128 cells) of one image.  (1) popcount, prefix over the workgroup, one atomicAdd on the image's counter; (2) every set
// bit becomes an entry of the workgroup's candidate list in LDS (so that the mask work is spread evenly over the threads, whatever the cells hold);
// (3) one THREAD per candidate: the 9 rows of its 9 x 9 window as three ALIGNED 16-byte loads each (the window [x - 4, x + 4] lies inside the three
// float4 chunks from (x - 4) & ~3 on; chunks and rows outside the image are never loaded and count as "no candidate") -- 27 loads per candidate, all in
// flight together, instead of 80 scalar ones that each cost a cache-line transaction -- then the same two masks as sp_cand_kernel: a neighbour counts when
// it is a candidate of higher confidence, and the candidate's own confidence is above the threshold, so `inside the image and semi > c0` says exactly that.
#define MK_WORDS 256
#define MK_LIST_CAP (MK_WORDS * 32)
__global__ void __launch_bounds__(256)
sp_mask_kernel(const float* __restrict__ semi, int W, int H, const uint32_t* __restrict__ bits, int* __restrict__ cand, uint64_t* __restrict__ masks,
               int* __restrict__ counters) {
    __shared__ int s_list[MK_LIST_CAP];
    __shared__ int s_wave_cnt[4];
    __shared__ int s_base;
    const int Wc = W >> 3, words = Wc * (H >> 3) * 2, hw = W * H;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = blockIdx.x * MK_WORDS + tid;
    uint32_t cm = w < words ? bits[(int64_t)b * words + w] : 0u;
    const int mine = __popc(cm);
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
    if (lane == 63) s_wave_cnt[wave] = incl;
    __syncthreads();
    const int total = s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
    if (total == 0) return;
    if (tid == 0) s_base = atomicAdd(&counters[b * 4 + 0], total);
    {
        int pos = incl - mine;
        for (int q = 0; q < wave; ++q) pos += s_wave_cnt[q];
        const int cell = w >> 1, hh = w & 1, cy = cell / Wc, cx = cell - cy * Wc;
        const int pix0 = (cy * 8) * W + cx * 8 + 4 * hh;
        while (cm) {
            const int i = __ffs(cm) - 1;
            cm &= cm - 1;
            s_list[pos++] = pix0 + (i >> 2) * W + (i & 3);
        }
    }
    __syncthreads();
    const float* sm = semi + (int64_t)b * hw;
    int* out = cand + (int64_t)b * hw + s_base;
    uint64_t* mo = masks + ((int64_t)b * hw + s_base) * 2;
    for (int ci = tid; ci < total; ci += 256) {
        const int p = s_list[ci];
        const int y = p / W, x = p - y * W;
        const int xa = (x - 4) & ~3;                                 // first column of the three aligned chunks (may be -4: that chunk is outside the image)
        const int sh = (x - 4) - xa;                                 // window column j (-4..4) = chunk element sh + j + 4
        float win[9][12];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int yy = y + r - 4;
            const bool row_in = yy >= 0 && yy < H;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int xc = xa + 4 * c;
                const bool in = row_in && xc >= 0 && xc < W;      // (W is a multiple of 8: a chunk is inside or outside as a whole)
                const float4 v = *reinterpret_cast<const float4*>(sm + (in ? (int64_t)yy * W + xc : 0));
                win[r][4 * c + 0] = in ? v.x : NEG_SENTINEL; win[r][4 * c + 1] = in ? v.y : NEG_SENTINEL;
                win[r][4 * c + 2] = in ? v.z : NEG_SENTINEL; win[r][4 * c + 3] = in ? v.w : NEG_SENTINEL;
            }
        }
        // window element (k, j) = win[k + 4][sh + j + 4], sh in 0..3: a 4-way select per element (registers cannot be indexed by a variable)
        auto at = [&](int r, int e) -> float {                       // e = j + 4 in 0..8
            const float a0 = win[r][e], a1 = win[r][e + 1], a2 = win[r][e + 2], a3 = win[r][e + 3];
            return sh == 0 ? a0 : (sh == 1 ? a1 : (sh == 2 ? a2 : a3));
        };
        const float c0 = at(4, 4);
        uint32_t e_lo = 0, e_hi = 0, l_lo = 0, l_hi = 0;
#pragma unroll
        for (int i = 0; i < 40; ++i) {
            const int k = (i < 36) ? (i / 9 - 4) : 0;
            const int j = (i < 36) ? (i - (k + 4) * 9 - 4) : (i - 40);
            const uint32_t be = at(4 + k, 4 + j) > c0 ? 1u : 0u, bl = at(4 - k, 4 - j) > c0 ? 1u : 0u;
            if (i < 32) { e_lo |= be << i; l_lo |= bl << i; } else { e_hi |= be << (i - 32); l_hi |= bl << (i - 32); }
        }
        out[ci] = p;
        mo[2 * (int64_t)ci] = ((uint64_t)e_hi << 32) | e_lo;
        mo[2 * (int64_t)ci + 1] = ((uint64_t)l_hi << 32) | l_lo;
    }
}

__device__ __forceinline__ unsigned st_get(const unsigned* st, int p) { return (st[p >> 4] >> ((p & 15) * 2)) & 3u; }
// pixel offset of mask bit i (earlier-mask numbering)
__device__ __forceinline__ int mask_bit_offset(int i, int W) {
    const int k = (i < 36) ? (i / 9 - 4) : 0;
    const int j = (i < 36) ? (i - (k + 4) * 9 - 4) : (i - 40);
    return k * W + j;
}

// ---- NMS2 (:237-310): the alive/dead recursion over the masks + top max_num ------------------------------------------
#define NMS_RANK_CAP 1024          // keys ranked by counting (beyond that: bitonic fallback)
#define NMS_RC 8                   // candidates per thread held in registers during the relaxation
__global__ void __launch_bounds__(NMS_THREADS)
sp_nms_kernel(const float* __restrict__ semi, int W, int H, int max_num, const int* __restrict__ cand, const uint64_t* __restrict__ masks,
              int* __restrict__ counters, uint64_t* __restrict__ surv_keys, float* __restrict__ kps_xy,
              float* __restrict__ scores, int* __restrict__ n_kps, int state_words, int smem_main_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // all LDS is dynamic: keeps the base 16-B aligned
    unsigned* st = reinterpret_cast<unsigned*>(smem_raw);
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);     // reused after the relaxation
    int* sv = reinterpret_cast<int*>(smem_raw + smem_main_bytes);     // [0] n_surv  [1] gathered  [2] bucket  [3] need
    int* hist = sv + 4;                                               // [256]
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int hw = W * H;
    const float* sm = semi + (int64_t)b * hw;
    const int* cd = cand + (int64_t)b * hw;
    const uint64_t* mk = masks + (int64_t)b * hw * 2;
    uint64_t* sk = surv_keys + (int64_t)b * hw;
    const int n_cand = counters[b * 4 + 0];

    // a thread's first NMS_RC candidates (8192 per image) live in registers for the whole relaxation: their pixel index and earlier mask are loaded ONCE, all
    // loads in flight together (round 5: every sweep re-read both from global memory, one dependent L2 round trip per candidate per sweep -- most of
    // the kernel's 60 us); candidates beyond that (never at the reference's thresholds) keep the global path
    int pc[NMS_RC];
    uint64_t me[NMS_RC], ml[NMS_RC];                                  // pixel index, earlier mask, later mask
#pragma unroll
    for (int k = 0; k < NMS_RC; ++k) {
        const int ci = tid + k * NMS_THREADS;
        pc[k] = ci < n_cand ? cd[ci] : -1;
        me[k] = ci < n_cand ? mk[2 * (int64_t)ci] : 0ull;
        ml[k] = ci < n_cand ? mk[2 * (int64_t)ci + 1] : 0ull;
    }
    for (int i = tid; i < state_words; i += NMS_THREADS) st[i] = 0u;  // (the loads above fly meanwhile)
    if (tid < 4) sv[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NMS_RC; ++k)
        if (pc[k] >= 0) atomicOr(&st[pc[k] >> 4], (me[k] ? ST_UNKNOWN : ST_ALIVE) << ((pc[k] & 15) * 2));     // grid(vv,uu) = 1 (:259); nothing earlier can kill it -> alive
    for (int ci = tid + NMS_RC * NMS_THREADS; ci < n_cand; ci += NMS_THREADS) {
        const int p = cd[ci];
        atomicOr(&st[p >> 4], (mk[2 * (int64_t)ci] ? ST_UNKNOWN : ST_ALIVE) << ((p & 15) * 2));
    }
    __syncthreads();

    // chaotic relaxation of alive(): LDS state plane + the per-candidate mask only (no heat-map reads)
    int iters = 0;
    for (;;) {
        int changed = 0;
#pragma unroll
        for (int k = 0; k < NMS_RC; ++k) {
            const int p = pc[k];
            if (p < 0 || st_get(st, p) != ST_UNKNOWN) continue;
            uint64_t m = me[k];
            bool any_alive = false, any_unknown = false;
            while (m) {
                const int i = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                const unsigned sq = st_get(st, p + mask_bit_offset(i, W));
                any_alive |= (sq == ST_ALIVE);
                any_unknown |= (sq == ST_UNKNOWN);
            }
            if (any_alive) { atomicOr(&st[p >> 4], 2u << ((p & 15) * 2)); changed = 1; }            // 01 -> 11 dead
            else if (!any_unknown) { atomicXor(&st[p >> 4], 3u << ((p & 15) * 2)); changed = 1; }   // 01 -> 10 alive
        }
        for (int ci = tid + NMS_RC * NMS_THREADS; ci < n_cand; ci += NMS_THREADS) {
            const int p = cd[ci];
            if (st_get(st, p) != ST_UNKNOWN) continue;
            uint64_t m = mk[2 * (int64_t)ci];
            bool any_alive = false, any_unknown = false;
            while (m) {
                const int i = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                const unsigned sq = st_get(st, p + mask_bit_offset(i, W));
                any_alive |= (sq == ST_ALIVE);
                any_unknown |= (sq == ST_UNKNOWN);
            }
            if (any_alive) { atomicOr(&st[p >> 4], 2u << ((p & 15) * 2)); changed = 1; }            // 01 -> 11 dead
            else if (!any_unknown) { atomicXor(&st[p >> 4], 3u << ((p & 15) * 2)); changed = 1; }   // 01 -> 10 alive
        }
        ++iters;
        if (!__syncthreads_or(changed)) break;
        if (iters > (1 << 20)) break;   // bound every spin: a chain can never be longer than the candidate count
    }

    // survive(): alive and not beaten by any alive neighbour (an earlier one cannot exist; a later one can)
#pragma unroll
    for (int k = 0; k < NMS_RC; ++k) {
        const int p = pc[k];
        if (p < 0 || st_get(st, p) != ST_ALIVE) continue;
        uint64_t m = ml[k];
        bool beaten = false;
        while (m) {
            const int i = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            beaten |= (st_get(st, p - mask_bit_offset(i, W)) == ST_ALIVE);
        }
        if (!beaten) sk[atomicAdd(&sv[0], 1)] = omni_make_key(sm[p], (uint32_t)p);
    }
    for (int ci = tid + NMS_RC * NMS_THREADS; ci < n_cand; ci += NMS_THREADS) {
        const int p = cd[ci];
        if (st_get(st, p) != ST_ALIVE) continue;
        uint64_t m = mk[2 * (int64_t)ci + 1];
        bool beaten = false;
        while (m) {
            const int i = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            beaten |= (st_get(st, p - mask_bit_offset(i, W)) == ST_ALIVE);
        }
        if (!beaten) sk[atomicAdd(&sv[0], 1)] = omni_make_key(sm[p], (uint32_t)p);
    }
    __syncthreads();
    const int n_surv = sv[0];
    __syncthreads();   // everyone has read the state plane; LDS is reused for selection / sorting from here on

    // ---- top max_num by (conf desc, index asc) ----------------------------------------------------------------
    const int M = max_num;
    const int n_out = n_surv < M ? n_surv : M;
    uint64_t* res = keys;                                             // where the n_out sorted keys end up
    // 1. cut-off: the 32-bit confidence pattern c* of the M-th largest key (radix select, 4 passes of 8 bits)
    uint32_t cutoff = 0;
    if (n_surv > M) {
        int need = M;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n_surv; i += NMS_THREADS) {
                const uint32_t h = (uint32_t)(sk[i] >> 32);
                if (pass == 0 || (h >> (shift + 8)) == (cutoff >> (shift + 8))) atomicAdd(&hist[(h >> shift) & 255u], 1);
            }
            __syncthreads();
            if (tid < 64) {                                            // one wave: lane l owns the 4 buckets 255-4l .. 252-4l
                int c[4], tot = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) { c[e] = hist[255 - 4 * tid - e]; tot += c[e]; }
                int incl = tot;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (tid >= off) incl += t; }
                int above = incl - tot;                               // keys in strictly higher buckets than this lane's first
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (above < need && need <= above + c[e]) { sv[2] = 255 - 4 * tid - e; sv[3] = need - above; }
                    above += c[e];
                }
            }
            __syncthreads();
            cutoff |= (uint32_t)sv[2] << shift;
            need = sv[3];
            __syncthreads();
        }
    }
    // 2. gather every key with confidence >= c* (= the top M plus the ties at the cut-off) into LDS
    for (int i = tid; i < n_surv; i += NMS_THREADS) {
        const uint64_t k = sk[i];
        if ((uint32_t)(k >> 32) >= cutoff) {
            const int slot = atomicAdd(&sv[1], 1);
            if (slot < NMS_RANK_CAP) keys[slot] = k;
        }
    }
    __syncthreads();
    const int T = sv[1];
    if (T <= NMS_RANK_CAP) {
        // 3a. rank by counting (keys are distinct: they contain the pixel index) -- no barriers, LDS broadcast reads
        uint64_t* sorted = keys + NMS_RANK_CAP;
        for (int i = tid; i < T; i += NMS_THREADS) {
            const uint64_t k = keys[i];
            int rank = 0;
            for (int j = 0; j < T; ++j) rank += (keys[j] > k);
            if (rank < M) sorted[rank] = k;
        }
        res = sorted;
        __syncthreads();
    } else {
        // 3b. massive ties at the cut-off (e.g. saturated maps): running best in keys[0, M), batches appended behind it
        __syncthreads();
        for (int i = tid; i < M; i += NMS_THREADS) keys[i] = OMNI_KEY_EMPTY;
        const int batch_cap = NMS_SORT_CAP - M;
        int off = 0;
        do {
            const int cnt = (n_surv - off) < batch_cap ? (n_surv - off) : batch_cap;
            int n_pow2 = 2;
            while (n_pow2 < M + cnt) n_pow2 <<= 1;
            __syncthreads();
            for (int i = tid; i < n_pow2 - M; i += NMS_THREADS) keys[M + i] = (i < cnt) ? sk[off + i] : OMNI_KEY_EMPTY;
            __syncthreads();
            bitonic_sort_desc(keys, n_pow2, tid, NMS_THREADS);
            off += cnt;
        } while (off < n_surv);
        __syncthreads();
    }

    for (int i = tid; i < M; i += NMS_THREADS) {
        float x = 0.f, y = 0.f, c = 0.f;
        if (i < n_out) {
            const uint64_t key = res[i];
            const uint32_t p = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
            c = omni_orderable_f32((uint32_t)(key >> 32));
            y = (float)(p / (uint32_t)W);
            x = (float)(p % (uint32_t)W);
        }
        kps_xy[((int64_t)b * M + i) * 2 + 0] = x;
        kps_xy[((int64_t)b * M + i) * 2 + 1] = y;
        scores[(int64_t)b * M + i] = c;
    }
    if (tid == 0) { n_kps[b] = n_out; counters[b * 4 + 1] = n_surv; counters[b * 4 + 2] = iters; }
}

// ---- computeDescriptors (:192-230) --------------------------------------------------------------------------------
// Pass 1: torch::grid_sampler (bilinear, zeros, align_corners=false) of every key point; 8 key points per workgroup,
// thread = channel (NHWC: the 4 taps are coalesced 1 KiB rows).
#define SAMPLE_KPB 8
__global__ void __launch_bounds__(256)
sp_sample_kernel(const float* __restrict__ desc_nhwc, int W, int H, int max_num, const float* __restrict__ kps_xy,
                 const int* __restrict__ n_kps, float* __restrict__ raw_desc) {
    const int b = blockIdx.y;
    const int c = threadIdx.x;
    const int Wc = W >> 3, Hc = H >> 3;
    const int n = n_kps[b];
    const int i0 = blockIdx.x * SAMPLE_KPB;
    if (i0 >= n) return;
    const float* dm = desc_nhwc + (int64_t)b * Hc * Wc * 256;
    float* raw = raw_desc + (int64_t)b * max_num * 256;
    const float fW = (float)W, fH = (float)H, fWc = (float)Wc, fHc = (float)Hc;
#pragma unroll
    for (int kk = 0; kk < SAMPLE_KPB; ++kk) {
        const int i = i0 + kk;
        if (i >= n) break;
        const float kx = kps_xy[((int64_t)b * max_num + i) * 2 + 0];
        const float ky = kps_xy[((int64_t)b * max_num + i) * 2 + 1];
        // grid = 2*k/size - 1 (:204-205); unnormalise with align_corners=false: ((g+1)*size_c - 1)/2
        const float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, kx), fW), 1.0f);
        const float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, ky), fH), 1.0f);
        const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), fWc), 1.0f), 2.0f);
        const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), fHc), 1.0f), 2.0f);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
        float v = 0.f;
        if (y0 >= 0 && y0 < Hc) {
            if (x0 >= 0 && x0 < Wc) v += dm[((int64_t)y0 * Wc + x0) * 256 + c] * (wx0 * wy0);
            if (x1 >= 0 && x1 < Wc) v += dm[((int64_t)y0 * Wc + x1) * 256 + c] * (wx1 * wy0);
        }
        if (y1 >= 0 && y1 < Hc) {
            if (x0 >= 0 && x0 < Wc) v += dm[((int64_t)y1 * Wc + x0) * 256 + c] * (wx0 * wy1);
            if (x1 >= 0 && x1 < Wc) v += dm[((int64_t)y1 * Wc + x1) * 256 + c] * (wx1 * wy1);
        }
        raw[(int64_t)i * 256 + c] = v;
    }
}

// the coarse cell and weight of corner j (0: y0x0, 1: y0x1, 2: y1x0, 3: y1x1) of key point (kx, ky): sp_sample_kernel's arithmetic
__device__ __forceinline__ void sample_corner(float kx, float ky, int W, int H, int j, int& cx, int& cy, float& wgt) {
    const int Wc = W >> 3, Hc = H >> 3;
    const float fW = (float)W, fH = (float)H, fWc = (float)Wc, fHc = (float)Hc;
    const float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, kx), fW), 1.0f);
    const float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, ky), fH), 1.0f);
    const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), fWc), 1.0f), 2.0f);
    const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), fHc), 1.0f), 2.0f);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
    cx = (int)fx0 + (j & 1); cy = (int)fy0 + (j >> 1);
    wgt = ((j & 1) ? wx1 : wx0) * ((j >> 1) ? wy1 : wy0);
}
// fp32 sparse descriptor head, step 1: the cDa vectors of the four coarse cells around every key point, compactly: row ((b * max_num + i) * 4 + j);
// cells outside the map and key points beyond n_kps give zero rows (never read back).  grid = (max_num, batch), thread = channel
__global__ void __launch_bounds__(256)
sp_gather_cells_kernel(const float* __restrict__ cda, int cstride, int W, int H, int max_num, const float* __restrict__ kps_xy, const int* __restrict__ n_kps,
                       float* __restrict__ out) {
    const int b = blockIdx.y, i = blockIdx.x, c = threadIdx.x;
    const int Wc = W >> 3, Hc = H >> 3;
    float* o = out + ((int64_t)b * max_num + i) * 4 * 256 + c;
    if (i >= n_kps[b]) { o[0] = o[256] = o[512] = o[768] = 0.f; return; }
    const float kx = kps_xy[((int64_t)b * max_num + i) * 2 + 0], ky = kps_xy[((int64_t)b * max_num + i) * 2 + 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int cx, cy; float wgt;
        sample_corner(kx, ky, W, H, j, cx, cy, wgt);
        o[j * 256] = (cx >= 0 && cx < Wc && cy >= 0 && cy < Hc) ? cda[(((int64_t)b * Hc + cy) * Wc + cx) * cstride + c] : 0.f;
    }
}
// step 3: sp_sample_kernel on the compact rows (same taps, same order, same products: the same bits as sampling the dense map)
__global__ void __launch_bounds__(256)
sp_sample_compact_kernel(const float* __restrict__ rows, int W, int H, int max_num, const float* __restrict__ kps_xy, const int* __restrict__ n_kps,
                         float* __restrict__ raw_desc) {
    const int b = blockIdx.y, i = blockIdx.x, c = threadIdx.x;
    if (i >= n_kps[b]) return;
    const int Wc = W >> 3, Hc = H >> 3;
    const float kx = kps_xy[((int64_t)b * max_num + i) * 2 + 0], ky = kps_xy[((int64_t)b * max_num + i) * 2 + 1];
    const float* r = rows + ((int64_t)b * max_num + i) * 4 * 256 + c;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int cx, cy; float wgt;
        sample_corner(kx, ky, W, H, j, cx, cy, wgt);
        if (cx >= 0 && cx < Wc && cy >= 0 && cy < Hc) v += r[j * 256] * wgt;
    }
    raw_desc[((int64_t)b * max_num + i) * 256 + c] = v;
}

// Pass 1b: every CHANNEL is divided by its L2 norm ACROSS the image's key points -- torch::norm(desc, 2, /*dim=*/1) on the
// [256, n] tensor (:214): the reference normalises channels across key points, not descriptors across channels.
// Sum of squares in NORM_SEGS fixed segments of key points (grid = segments x images, thread = channel); the consumers add
// the segment sums in a fixed order (deterministic) and divide on the fly, so the normalised tensor is never materialised.
#define NORM_SEGS 8
__global__ void __launch_bounds__(256)
sp_chan_sumsq_kernel(int max_num, const int* __restrict__ n_kps, const float* __restrict__ raw_desc, float* __restrict__ partial) {
    const int b = blockIdx.y, seg = blockIdx.x, c = threadIdx.x;
    const int n = n_kps[b];
    const int per = (n + NORM_SEGS - 1) / NORM_SEGS;
    const int i0 = seg * per, i1 = (i0 + per < n) ? i0 + per : n;
    const float* raw = raw_desc + (int64_t)b * max_num * 256;
    float ss = 0.f;
#pragma unroll 8
    for (int i = i0; i < i1; ++i) { const float v = raw[(int64_t)i * 256 + c]; ss = fmaf(v, v, ss); }
    partial[((int64_t)b * NORM_SEGS + seg) * 256 + c] = ss;
}
__device__ __forceinline__ float chan_norm_of(const float* __restrict__ partial, int b, int c) {
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < NORM_SEGS; ++s) ss += partial[((int64_t)b * NORM_SEGS + s) * 256 + c];
    return sqrtf(ss);
}

// Pass 2: (d / norm - mean) * comp^T  (:214-221) -- 4 key points per workgroup, thread = (64-channel part, output dim)
__global__ void __launch_bounds__(256)
sp_pca_kernel(const float* __restrict__ raw_desc, const float* __restrict__ partial, int max_num, const int* __restrict__ n_kps, int pca_dim,
              const float* __restrict__ compT, const float* __restrict__ mean, float* __restrict__ out) {
    __shared__ float sx[4][256];
    __shared__ float sp[4][4][64];
    const int b = blockIdx.y;
    const int n = n_kps[b];
    const int i0 = blockIdx.x * 4;
    if (i0 >= n) return;
    const int tid = threadIdx.x;
    const float dn = chan_norm_of(partial, b, tid);
    for (int kp = 0; kp < 4; ++kp) {
        const int i = i0 + kp;
        sx[kp][tid] = (i < n) ? (raw_desc[((int64_t)b * max_num + i) * 256 + tid] / dn - mean[tid]) : 0.f;   // 0/0 -> NaN as in the reference
    }
    __syncthreads();
    const int part = tid >> 6;
    for (int j0 = 0; j0 < pca_dim; j0 += 64) {
        const int j = j0 + (tid & 63);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (j < pca_dim) {
            for (int cc = 0; cc < 64; ++cc) {
                const int c = part * 64 + cc;
                const float w = compT[(int64_t)c * pca_dim + j];
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) acc[kp] = fmaf(sx[kp][c], w, acc[kp]);
            }
        }
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) sp[kp][part][tid & 63] = acc[kp];
        __syncthreads();
        if (tid < 64 && j0 + tid < pca_dim) {
            for (int kp = 0; kp < 4; ++kp) {
                const int i = i0 + kp;
                if (i < n)
                    out[((int64_t)b * max_num + i) * pca_dim + j0 + tid] =
                        (sp[kp][0][tid] + sp[kp][1][tid]) + (sp[kp][2][tid] + sp[kp][3][tid]);
            }
        }
        __syncthreads();
    }
}

__global__ void copy_raw_desc_kernel(const float* __restrict__ raw, const float* __restrict__ partial, int max_num,
                                     const int* __restrict__ n_kps, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x;
    if (i >= n_kps[b]) return;
    out[((int64_t)b * max_num + i) * 256 + threadIdx.x] = raw[((int64_t)b * max_num + i) * 256 + threadIdx.x] / chan_norm_of(partial, b, threadIdx.x);
}

// ---- layout helpers -----------------------------------------------------------------------------------------------
__global__ void transpose_cl_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW, int to_nhwc) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 32 x 8
    const float* ib = in + (int64_t)b * C * HW;
    float* ob = out + (int64_t)b * C * HW;
    if (to_nhwc) {            // in [C][HW] -> out [HW][C]
        for (int r = ty; r < 32; r += 8) { int c = c0 + r, p = p0 + tx; tile[r][tx] = (c < C && p < HW) ? ib[(int64_t)c * HW + p] : 0.f; }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) { int p = p0 + r, c = c0 + tx; if (c < C && p < HW) ob[(int64_t)p * C + c] = tile[tx][r]; }
    } else {                  // in [HW][C] -> out [C][HW]
        for (int r = ty; r < 32; r += 8) { int p = p0 + r, c = c0 + tx; tile[r][tx] = (c < C && p < HW) ? ib[(int64_t)p * C + c] : 0.f; }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) { int c = c0 + r, p = p0 + tx; if (c < C && p < HW) ob[(int64_t)c * HW + p] = tile[tx][r]; }
    }
}

int nchw_to_nhwc(hipStream_t stream, const float* in, float* out, int batch, int C, int HW) {
    hipLaunchKernelGGL(transpose_cl_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), batch), dim3(256), 0, stream, in, out, C, HW, 1);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}
int nhwc_to_nchw(hipStream_t stream, const float* in, float* out, int batch, int C, int HW) {
    hipLaunchKernelGGL(transpose_cl_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), batch), dim3(256), 0, stream, in, out, C, HW, 0);
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

int sp_postprocess(hipStream_t stream, const SpPostParams& p, const SpPostBuffers& b, const float* semi,
                   const float* desc_nhwc, int batch, const SpSparseDesc& sparse) {
    const int hw = p.width * p.height;
    const int state_words = cdiv(hw, 16);
    size_t smem = (size_t)state_words * 4;
    if (smem < (size_t)NMS_SORT_CAP * 8) smem = (size_t)NMS_SORT_CAP * 8;
    const int smem_main = (int)smem;
    smem += 16 + 256 * 4;                                             // selection scalars + radix histogram
    OMNI_REQUIRE(smem <= 160 * 1024, OMNI_ERR_CAPACITY, "image %dx%d too large for the in-LDS NMS state plane", p.width, p.height);
    OMNI_REQUIRE(p.max_num >= 1 && p.max_num <= 1024, OMNI_ERR_CAPACITY, "max_num=%d outside [1,1024]", p.max_num);
    OMNI_REQUIRE(p.dist_thresh == 4, OMNI_ERR_INVALID, "NMS radius %d: the window masks are built for 4 (superpoint_tensorrt.cpp:183)", p.dist_thresh);
    if (sparse.cand_fused || sparse.cand_from_list) {
        const int words = (p.width / 8) * (p.height / 8) * 2;
        OMNI_HIP_TRY(hipMemsetAsync(b.counters, 0, (size_t)batch * 4 * sizeof(int), stream));
        if (!sparse.cand_fused) {
            hipLaunchKernelGGL(sp_thresh_kernel, dim3(cdiv(words, 256), batch), dim3(256), 0, stream, semi, p.width, p.height, p.thres, b.cand_bits);
            OMNI_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(sp_mask_kernel, dim3(cdiv(words, MK_WORDS), batch), dim3(256), 0, stream, semi, p.width, p.height, b.cand_bits, b.cand, b.cand_masks, b.counters);
    } else {
        OMNI_HIP_TRY(hipMemsetAsync(b.counters, 0, (size_t)batch * 4 * sizeof(int), stream));
        hipLaunchKernelGGL(sp_cand_kernel, dim3(cdiv(p.width, CT_W) * cdiv(p.height, CT_H), batch), dim3(256), 0, stream, semi, p.width, p.height,
                           p.thres, b.cand, b.cand_masks, b.counters);
    }
    OMNI_LAUNCH_CHECK();
    OMNI_HIP_TRY(hipFuncSetAttribute((const void*)sp_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(sp_nms_kernel, dim3(batch), dim3(NMS_THREADS), smem, stream, semi, p.width, p.height, p.max_num, b.cand,
                       b.cand_masks, b.counters, b.surv_keys, b.kps_xy, b.scores, b.n_kps, state_words, smem_main);
    OMNI_LAUNCH_CHECK();
    if (sparse.a4b) {
        int rc = conv_c128_sparse(stream, sparse.ctx, sparse.a4b, sparse.da_w, sparse.da_bias, p.height / 8, p.width / 8, sparse.da_g32_first, p.width, p.height,
                                  p.max_num, b.kps_xy, b.n_kps, sparse.da_compact, 256, batch);
        if (rc) return rc;
        rc = convdb_sparse_sample(stream, sparse.ctx, sparse.da_compact, 256, sparse.wfrag, sparse.bias, p.width, p.height, p.max_num, b.kps_xy, b.n_kps,
                                  b.raw_desc, batch, true);
        if (rc) return rc;
    } else if (sparse.in_f16) {
        int rc = convdb_sparse_sample(stream, sparse.ctx, sparse.in_f16, sparse.in_cstride, sparse.wfrag, sparse.bias, p.width, p.height, p.max_num,
                                      b.kps_xy, b.n_kps, b.raw_desc, batch, false);
        if (rc) return rc;
    } else if (sparse.cda_f32 || sparse.a4b_split) {
        // exact-f32 descriptor head only where the sampler reads: gather the cells -> the SAME 1x1 convolution kernel and per-cell norm the dense
        // map uses, on [8][N / 8] pixels instead of [batch][Hc][Wc] -> sample the compact rows
        const int64_t n_rows = (((int64_t)batch * p.max_num * 4) + 7) & ~(int64_t)7;
        if (sparse.a4b_split) {
            int rc = conv_split_c128_sparse(stream, sparse.ctx, sparse.a4b_split, sparse.da_w, sparse.da_bias, sparse.da_inv, p.height / 8, p.width / 8,
                                            sparse.da_g32_first, p.width, p.height, p.max_num, b.kps_xy, b.n_kps, sparse.cx, batch);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL(sp_gather_cells_kernel, dim3(p.max_num, batch), dim3(256), 0, stream, sparse.cda_f32, sparse.in_cstride, p.width, p.height, p.max_num,
                               b.kps_xy, b.n_kps, sparse.cx);
            OMNI_LAUNCH_CHECK();
        }
        int rc;
        if (sparse.a4b_split && sparse.wdb_hi) {
            if ((rc = convdb_l2norm_split(stream, sparse.ctx, sparse.cx, 256, sparse.wdb_hi, sparse.wdb_lo, sparse.bias, sparse.cy, n_rows))) return rc;
        } else {
            ConvArgs a;
            a.in = sparse.cx; a.out = sparse.cy; a.w_packed = sparse.wdb_f32; a.bias = sparse.bias; a.batch = 1; a.H = 8; a.W = (int)(n_rows / 8); a.cin = 256;
            a.cout = 256; a.ksize = 1; a.relu = false; a.pool = false; a.out_f32 = true; a.in_cstride = 256; a.n_cu = sparse.n_cu; a.zero_page = sparse.zero_page;
            if ((rc = conv_mfma(stream, OMNI_PREC_F32, a))) return rc;
            if ((rc = l2norm_channels(stream, sparse.cy, n_rows))) return rc;
        }
        hipLaunchKernelGGL(sp_sample_compact_kernel, dim3(p.max_num, batch), dim3(256), 0, stream, sparse.cy, p.width, p.height, p.max_num, b.kps_xy, b.n_kps,
                           b.raw_desc);
        OMNI_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(sp_sample_kernel, dim3(cdiv(p.max_num, SAMPLE_KPB), batch), dim3(256), 0, stream, desc_nhwc, p.width, p.height,
                           p.max_num, b.kps_xy, b.n_kps, b.raw_desc);
        OMNI_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sp_chan_sumsq_kernel, dim3(NORM_SEGS, batch), dim3(256), 0, stream, p.max_num, b.n_kps, b.raw_desc, b.norm_partial);
    OMNI_LAUNCH_CHECK();
    if (p.pca_dim > 0) {
        hipLaunchKernelGGL(sp_pca_kernel, dim3(cdiv(p.max_num, 4), batch), dim3(256), 0, stream, b.raw_desc, b.norm_partial, p.max_num, b.n_kps,
                           p.pca_dim, b.pca_compT, b.pca_mean, b.desc_out);
    } else {
        hipLaunchKernelGGL(copy_raw_desc_kernel, dim3(p.max_num, batch), dim3(256), 0, stream, b.raw_desc, b.norm_partial, p.max_num, b.n_kps,
                           b.desc_out);
    }
    OMNI_LAUNCH_CHECK();
    return OMNI_OK;
}

}  // namespace omni
