/*
 * CPU oracle (plain C) for the swarm_loop post-processing and matcher.
 * TEST INFRASTRUCTURE ONLY -- never linked into the product library.
 *
 * Each function restates one piece of the reference
 * (HKUST-Aerial-Robotics/Omni-swarm, paths relative to the reference root):
 *
 *   oracle_nms2_literal      swarm_loop/src/superpoint_tensorrt.cpp:164-189 (getKeyPoints)
 *                            + :237-310 (NMS2), literal serial simulation.
 *   oracle_bf_match          cv::BFMatcher(cv::NORM_L2, crossCheck=true).match(), call sites
 *                            swarm_loop/src/loop_cam.cpp:147-150, loop_detector.cpp:564-567.
 *                            OpenCV 3.4 is NOT in the reference tree: algorithm restated from
 *                            opencv/modules/core/src/batch_distance.cpp (batchDistance, K=1,
 *                            crosscheck) and features2d/src/matchers.cpp (knnMatchImpl).
 *                            PARITY UNPINNED.
 *   oracle_ip_search         faiss::IndexFlatIP::search, call site loop_detector.cpp:213.
 *                            faiss is NOT in the reference tree (unpinned system lib): exact
 *                            inner product, descending, ties -> lower row id (our spec).
 *                            PARITY UNPINNED.
 *
 * Build: see oracle/Makefile (gcc -O2 -shared -fPIC).  No -ffast-math: float op order is
 * part of the specification.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * getKeyPoints + NMS2, literal.
 *
 * Deviations from the literal text, both part of the build's FIXED SPEC:
 *   (1) neighbours outside the image are ignored (the reference indexes out of bounds:
 *       column overflow wraps to the adjacent row, row overflow is heap UB);
 *   (2) the final order is (confidence descending, row-major pixel index ascending); the
 *       reference uses std::sort (unstable) on confidence only, so its order among equal
 *       confidences is implementation-defined.
 *   (3) the candidate index plane is 32-bit (the reference's CV_16UC1 plane silently wraps
 *       at 65536 candidates, :246,260).
 * wrap_columns != 0 reproduces the reference's column wrap for in-range rows (used only to
 * characterise the quirk in tests; rows outside the image are still ignored).
 * Returns the number of key points written (<= max_num).
 * ---------------------------------------------------------------------------------------- */
typedef struct { float conf; int idx; } kp_t;

static int kp_cmp(const void* a, const void* b) {
    const kp_t* x = (const kp_t*)a; const kp_t* y = (const kp_t*)b;
    if (x->conf > y->conf) return -1;
    if (x->conf < y->conf) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

int oracle_nms2_literal(const float* prob, int W, int H, float thres, int dist_thresh,
                        int max_num, int wrap_columns,
                        int* out_xy /* max_num x 2 (x,y) */, float* out_conf /* max_num */,
                        int* out_num_candidates, int* out_num_survivors) {
    const long n_px = (long)W * H;
    unsigned char* grid = (unsigned char*)calloc(n_px, 1);
    float* confidence = (float*)calloc(n_px, sizeof(float));
    int* cand = (int*)malloc(n_px * sizeof(int));
    int n_cand = 0;
    /* getKeyPoints :167-180 -- mask = prob > threshold, findNonZero is row-major */
    for (long p = 0; p < n_px; ++p)
        if (prob[p] > thres) cand[n_cand++] = (int)p;
    /* NMS2 :254-263 -- scatter */
    for (int i = 0; i < n_cand; ++i) {
        grid[cand[i]] = 1;
        confidence[cand[i]] = prob[cand[i]];
    }
    /* NMS2 :265-283 -- serial pass in candidate (row-major) order */
    for (int i = 0; i < n_cand; ++i) {
        const int uu = cand[i] % W, vv = cand[i] / W;
        if (grid[cand[i]] != 1) continue;
        const float c0 = confidence[cand[i]];
        for (int k = -dist_thresh; k < dist_thresh + 1; ++k)
            for (int j = -dist_thresh; j < dist_thresh + 1; ++j) {
                if (j == 0 && k == 0) continue;
                int v = vv + k, u = uu + j;
                if (v < 0 || v >= H) continue;
                if (u < 0 || u >= W) {
                    if (!wrap_columns) continue;
                    long q = (long)v * W + u;          /* continuous cv::Mat: wraps rows */
                    if (q < 0 || q >= n_px) continue;
                    if (confidence[q] < c0) grid[q] = 0;
                    continue;
                }
                if (confidence[(long)v * W + u] < c0) grid[(long)v * W + u] = 0;
            }
        grid[cand[i]] = 2;
    }
    /* NMS2 :285-308 -- collect grid==2, sort by confidence, keep max_num */
    kp_t* surv = (kp_t*)malloc((n_cand > 0 ? n_cand : 1) * sizeof(kp_t));
    int n_surv = 0;
    for (long p = 0; p < n_px; ++p)
        if (grid[p] == 2) { surv[n_surv].conf = confidence[p]; surv[n_surv].idx = (int)p; ++n_surv; }
    qsort(surv, n_surv, sizeof(kp_t), kp_cmp);
    int n_out = n_surv < max_num ? n_surv : max_num;
    for (int i = 0; i < n_out; ++i) {
        out_xy[2 * i] = surv[i].idx % W;
        out_xy[2 * i + 1] = surv[i].idx / W;
        out_conf[i] = surv[i].conf;
    }
    if (out_num_candidates) *out_num_candidates = n_cand;
    if (out_num_survivors) *out_num_survivors = n_surv;
    free(surv); free(cand); free(confidence); free(grid);
    return n_out;
}

/* ------------------------------------------------------------------------------------------
 * cv::BFMatcher(NORM_L2, crossCheck=true).match(query, train)
 *
 * OpenCV 3.4 batchDistance(src1=query, src2=train, K=1, crosscheck=true):
 *   dist[:] = FLT_MAX, nidx[:] = -1
 *   (tdist, tidx) = for every TRAIN row j: nearest QUERY row (first minimum wins,
 *                   d = sqrtf(sum (a-b)^2), strict '<' update in ascending row order)
 *   for j ascending: idx = tidx[j]; if (tdist[j] < dist[idx]) { dist[idx]=tdist[j]; nidx[idx]=j; }
 * knnMatchImpl then emits DMatch(queryIdx=i, trainIdx=nidx[i], distance=dist[i]) for every i
 * with nidx[i] >= 0, in ascending i.
 *
 * NOTE this is what OpenCV computes; it is not the symmetric "mutual nearest neighbour"
 * (a query can be matched to a train row that is not the query's own nearest).  mode=1 gives
 * the strict mutual-NN variant of SURVEY.md 8c for comparison:
 *   keep (i, j*) iff j* = argmin_j d(i,j) (first min) and i = argmin_i' d(i', j*) (first min).
 * Returns the number of matches.
 * ---------------------------------------------------------------------------------------- */
static float l2_dist(const float* a, const float* b, int dim) {
    float s = 0.f;
    for (int k = 0; k < dim; ++k) { float d = a[k] - b[k]; s += d * d; }
    return sqrtf(s);
}

int oracle_bf_match(const float* q, int nq, const float* t, int nt, int dim, int mode,
                    int* q_idx, int* t_idx, float* dist_out) {
    if (nq <= 0 || nt <= 0) return 0;
    float* best_d = (float*)malloc(nq * sizeof(float));
    int* best_j = (int*)malloc(nq * sizeof(int));
    for (int i = 0; i < nq; ++i) { best_d[i] = 3.402823466e+38f; best_j[i] = -1; }
    if (mode == 0) {
        for (int j = 0; j < nt; ++j) {
            float bd = 3.402823466e+38f; int bi = -1;
            for (int i = 0; i < nq; ++i) {
                float d = l2_dist(t + (long)j * dim, q + (long)i * dim, dim);
                if (d < bd) { bd = d; bi = i; }
            }
            if (bi >= 0 && bd < best_d[bi]) { best_d[bi] = bd; best_j[bi] = j; }
        }
    } else {
        for (int i = 0; i < nq; ++i) {
            float bd = 3.402823466e+38f; int bj = -1;
            for (int j = 0; j < nt; ++j) {
                float d = l2_dist(q + (long)i * dim, t + (long)j * dim, dim);
                if (d < bd) { bd = d; bj = j; }
            }
            if (bj < 0) continue;
            float rd = 3.402823466e+38f; int ri = -1;
            for (int i2 = 0; i2 < nq; ++i2) {
                float d = l2_dist(t + (long)bj * dim, q + (long)i2 * dim, dim);
                if (d < rd) { rd = d; ri = i2; }
            }
            if (ri == i) { best_d[i] = bd; best_j[i] = bj; }
        }
    }
    int n = 0;
    for (int i = 0; i < nq; ++i)
        if (best_j[i] >= 0) { q_idx[n] = i; t_idx[n] = best_j[i]; dist_out[n] = best_d[i]; ++n; }
    free(best_d); free(best_j);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * faiss::IndexFlatIP::search(nq, q, k, D, I): exact inner product, k largest, descending.
 * Ties -> lower row id first.  Missing results (k > n) are padded with I=-1, D=-FLT_MAX
 * (faiss pads labels with -1; loop_detector.cpp:208-219 relies on that).
 * fp32 accumulation in row order (sequential), which is NOT faiss's blocked sgemm order;
 * scores agree to ~1e-6 relative, ids are what parity is judged on.
 * ---------------------------------------------------------------------------------------- */
void oracle_ip_search(const float* db, long n, int d, const float* q, int nq, int k,
                      float* D, int64_t* I) {
    for (int qi = 0; qi < nq; ++qi) {
        float* Dq = D + (long)qi * k; int64_t* Iq = I + (long)qi * k;
        for (int j = 0; j < k; ++j) { Dq[j] = -3.402823466e+38f; Iq[j] = -1; }
        const float* qv = q + (long)qi * d;
        for (long r = 0; r < n; ++r) {
            const float* row = db + r * d;
            /* 8 partial sums: matches a SIMD dot product's association closely enough and is
             * far more accurate than one serial chain at d=4096 */
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int c = 0;
            for (; c + 8 <= d; c += 8)
                for (int u = 0; u < 8; ++u) acc[u] += row[c + u] * qv[c + u];
            float s = ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
            for (; c < d; ++c) s += row[c] * qv[c];
            /* insert (s, r): strictly greater moves ahead; equal keeps earlier row first */
            if (s > Dq[k - 1] || (Iq[k - 1] < 0)) {
                int pos = k - 1;
                while (pos > 0 && (Iq[pos - 1] < 0 || s > Dq[pos - 1])) {
                    Dq[pos] = Dq[pos - 1]; Iq[pos] = Iq[pos - 1]; --pos;
                }
                Dq[pos] = s; Iq[pos] = r;
            }
        }
    }
}
