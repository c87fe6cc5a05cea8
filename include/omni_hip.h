/*
 * omni_hip.h -- C ABI of the MI355X-native (gfx950 / CDNA4) swarm_loop hot path.
 *
 * Drop-in boundary for HKUST-Aerial-Robotics/Omni-swarm's per-keyframe CNN frontend and
 * loop-closure matcher.  The reference has no FFI layer; the seam is four C++ call surfaces
 * used by LoopCam / LoopDetector.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference root):
 *
 *   omni_sp_*      SuperPointTensorRT            swarm_loop/include/swarm_loop/superpoint_tensorrt.h:12-29
 *                                                swarm_loop/src/superpoint_tensorrt.cpp:91-230,237-310
 *                  (+ the TensorRT engine = the graph of swarm_loop/superpoint.ipynb:135-205,
 *                   + TensorRTInferenceGeneric   swarm_loop/src/tensorrt_generic.cpp:14-120)
 *   omni_vlad_*    MobileNetVLADTensorRT         swarm_loop/include/swarm_loop/mobilenetvlad_tensorrt.h:6-22
 *                                                swarm_loop/src/mobilenetvlad_tensorrt.cpp:4-14
 *   omni_index_*   faiss::IndexFlatIP            used at swarm_loop/src/loop_detector.cpp:166,169,213,232,291,842
 *   omni_bf_*      cv::BFMatcher(NORM_L2,true)   used at swarm_loop/src/loop_cam.cpp:147-150,
 *                                                        swarm_loop/src/loop_detector.cpp:564-567
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch/OpenCV types cross this boundary.
 *   - every function that can fail returns int: 0 = OMNI_OK, otherwise an OMNI_ERR_* code;
 *     omni_last_error() gives a thread-local message.  The reference aborts on failure
 *     (live asserts, NV_CUDA_CHECK -- SURVEY.md F13); this library never aborts.
 *   - "_dev" variants take pointers to HBM (hipMalloc'd or omni_dev_alloc'd) and are asynchronous on the
 *     context's stream; the plain variants take host pointers, copy, run and synchronise (the reference's
 *     blocking batch-1 semantics, tensorrt_generic.cpp:58-75).
 *   - handles are thread-compatible: calls on one handle are serialised internally by a mutex
 *     (the reference's LoopDetector is entered from two threads with no lock).
 *   - there is NO CPU fallback: if the HIP device or a kernel launch fails the call returns OMNI_ERR_HIP.
 */
#ifndef OMNI_HIP_H
#define OMNI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMNI_ABI_VERSION 2      /* 2: omni_cam_result gained n_images; omni_cam_set_active / omni_cam_ready */

enum {
    OMNI_OK = 0,
    OMNI_ERR_INVALID = 1,   /* bad argument / shape mismatch (the reference asserts, superpoint_tensorrt.cpp:122) */
    OMNI_ERR_HIP = 2,       /* HIP runtime / launch failure */
    OMNI_ERR_NOMEM = 3,
    OMNI_ERR_CAPACITY = 4   /* batch / k / row count beyond what the handle was created for */
};

enum { OMNI_PREC_F32 = 0,   /* exact-f32 MFMA (v_mfma_f32_32x32x2_f32), fp32 activations: the parity mode */
       OMNI_PREC_F16 = 1,   /* fp16 storage + v_mfma_f32_32x32x16_f16, fp32 accumulate: the reference's engines
                               are fp16 TensorRT (launch/realsense.launch:10-11) */
       OMNI_PREC_SPLIT = 2 }; /* omni_sp only: fp32-class on the fp16 matrix cores -- every operand of the 3x3 convolutions is a
                               (hi, lo) pair of halfs, three MFMA terms per product, heads in exact f32: meets north_star's
                               tolerance (key points identical to the fp32 graph, descriptors <= 1e-3) at ~1/3 of the fp16 rate */

enum { OMNI_STORE_F32 = 0, OMNI_STORE_F16 = 1 };   /* global-descriptor DB storage */

enum { OMNI_BF_OPENCV = 0,  /* cv::batchDistance(K=1, crosscheck=true) semantics (what the reference runs) */
       OMNI_BF_MUTUAL = 1 };/* strict mutual nearest neighbour (SURVEY.md 8c restatement) */

typedef struct omni_ctx omni_ctx;
typedef struct omni_sp omni_sp;
typedef struct omni_vlad omni_vlad;
typedef struct omni_index omni_index;
typedef struct omni_cam omni_cam;
typedef struct omni_shard omni_shard;
typedef struct omni_flatten omni_flatten;

int         omni_abi_version(void);
const char* omni_last_error(void);
/* roctx ranges (OMNI_ROCTX=1; no-ops otherwise): the library brackets its own stages with them and the host loop its own through these two -- what the reference's
 * per-stage timers print (swarm_loop/src/superpoint_tensorrt.cpp:130-162, loop_cam.cpp:205-207) as a `rocprofv3 --marker-trace` timeline */
void        omni_trace_push(const char* name);
void        omni_trace_pop(void);

/* ---- configuration: EVERY switch of the library is an entry of one table (csrc/config.h): environment variable, default, valid range, class
 * (0 = variant: another kernel for the same results -- A/B measurements, bit-identity tests; 1 = tuning threshold; 2 = debug / timing ablation;
 * 3 = test fault injection; 4 = string), one line of documentation.  Handles resolve the table when they are created (a value outside its range
 * fails the creation); nothing else reads the environment.  The defaults ARE the production path (tests/test_config_cpu.py). */
int omni_config_count(void);
int omni_config_describe(int i, const char** env, int* def, int* lo, int* hi, int* cls, const char** doc);
/* what a handle created NOW would see for option `env` (defaults overridden by the current environment); OMNI_ERR_INVALID on an unknown name or
 * on any option holding a value outside its range */
int omni_config_value(const char* env, int* value);
/* 1: the option is read process-wide (frozen at first use: launch-site hooks, index thresholds), 0: snapshot per handle at creation */
int         omni_config_is_process_wide(int i);

/* ---- context: one per GPU/stream; owns a HIP stream, scratch and timers -------------------------------------
 * replaces TensorRTInferenceGeneric's cudaStreamCreate / cudaMalloc plumbing (tensorrt_generic.cpp:14-36,99-120) */
omni_ctx* omni_ctx_create(int device_id);
/* the same with the context's stream at the device's highest stream priority: for short, latency-bound work the host waits on (the detector's
 * searches) next to streams that keep every CU busy -- its kernels get the next free CUs instead of queueing behind whole launches */
omni_ctx* omni_ctx_create_priority(int device_id, int high_priority);
/* everything enqueued on `later`'s stream from now on runs after everything enqueued on `earlier`'s stream so far (an event on the device: the
 * host does not wait).  E.g. a network's stream behind the detector's, whose appends still read the network's output buffer. */
int       omni_ctx_order_after(omni_ctx* later, omni_ctx* earlier);
/* device -> PINNED host memory on the context's stream, without waiting: omni_ctx_sync() (or any later synchronising call) completes it */
int       omni_memcpy_d2h_async(omni_ctx* ctx, void* dst_pinned, const void* src, size_t bytes);
void      omni_ctx_destroy(omni_ctx* ctx);
int       omni_ctx_sync(omni_ctx* ctx);
void*     omni_ctx_stream(omni_ctx* ctx);                 /* hipStream_t, for callers that enqueue their own work */
int       omni_ctx_device_info(omni_ctx* ctx, char* name, int name_len, int* n_cu, int* clock_mhz, size_t* hbm_bytes);
/* Calibration (measurement support, no counterpart in the reference): v_mfma_f32_32x32x16_f16 back to back on every SIMD for about `ms` milliseconds;
 * returns what the board sustained (TFLOP/s) and the shader clock it held meanwhile -- bench.py quotes it next to the data-sheet peak. */
int       omni_ctx_mfma_ceiling(omni_ctx* ctx, float ms, float* tflops, float* sclk_ghz);

void* omni_dev_alloc(omni_ctx* ctx, size_t bytes);        /* HBM; NULL on failure */
int   omni_dev_free(omni_ctx* ctx, void* p);
void* omni_host_alloc(size_t bytes);                      /* pinned host memory (cudaMallocHost, tensorrt_generic.cpp:116-117); NULL on failure */
int   omni_host_free(void* p);
int   omni_memcpy_h2d(omni_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);  /* blocking */
int   omni_memcpy_d2h(omni_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);  /* blocking */
int   omni_timer_start(omni_ctx* ctx);                    /* hipEventRecord on the ctx stream */
int   omni_timer_stop(omni_ctx* ctx, float* ms);          /* records, synchronises, returns elapsed ms */

/* ---- SuperPoint -------------------------------------------------------------------------------------------------
 * weights: the 12 conv layers of superpoint.ipynb:143-160 in execution order
 *   conv1a conv1b conv2a conv2b conv3a conv3b conv4a conv4b convPa convPb convDa convDb,
 * each weight OIHW float32 and bias [O] float32 (the checkpoint's state_dict layout, superpoint.ipynb:270). */
#define OMNI_SP_NUM_LAYERS 12
typedef struct omni_sp_weights {
    const float* weight[OMNI_SP_NUM_LAYERS];
    const float* bias[OMNI_SP_NUM_LAYERS];
} omni_sp_weights;

/* SuperPointTensorRT::SuperPointTensorRT(engine, pca_comp, pca_mean, width, height, thres, max_num)
 * (superpoint_tensorrt.cpp:91-115).  pca_comp = sklearn components_ [pca_dim x 256] row-major, pca_mean [256]
 * (the two CSVs of :110-111); pca_comp == NULL disables PCA (#undef USE_PCA, :220-225) and desc is n x 256.
 * width, height must be multiples of 8.  max_batch = images processed per call (reference: 1). */
omni_sp* omni_sp_create(omni_ctx* ctx, const omni_sp_weights* w, const float* pca_comp, const float* pca_mean,
                        int pca_dim, int width, int height, float thres, int max_num, int precision, int max_batch);
void     omni_sp_destroy(omni_sp* sp);
int      omni_sp_desc_dim(const omni_sp* sp);              /* pca_dim, or 256 without PCA */
int      omni_sp_image_size(const omni_sp* sp, int* width, int* height);   /* the size the handle was created for (:122 asserts it per call) */

/* The rows the fisheye mask blanks in an image of `height` rows: [*row0, *row1) = cv::Rect(0, rows*3/4, cols, rows/4) of
 * LoopCam::extractor_img_desc_deepnet (loop_cam.cpp:536-539; integer divisions: a height that is not a multiple of 4 keeps its last rows).
 * fisheye_mask == 0: an empty range at the end of the image.  Every kernel that reads the gray image takes its mask rows from here. */
static inline void omni_fisheye_mask_rows(int height, int fisheye_mask, int* row0, int* row1) {
    *row0 = fisheye_mask ? height * 3 / 4 : height;
    *row1 = fisheye_mask ? height * 3 / 4 + height / 4 : height;
}

/* SuperPointTensorRT::inference(const cv::Mat&, std::vector<cv::Point2f>&, std::vector<float>&)
 * (superpoint_tensorrt.cpp:117-162) for `batch` images.
 *   gray      : batch images, u8, row stride `stride` bytes, image i at gray + i*stride*height
 *   fisheye_mask != 0 zeroes the rows omni_fisheye_mask_rows() names first (LoopCam::extractor_img_desc_deepnet, loop_cam.cpp:536-539)
 *   kps_xy    : [batch][max_num][2] float (x, y) integer-valued; order = (confidence desc, row-major index asc)
 *   n_kps     : [batch]
 *   desc      : [batch][max_num][desc_dim] float
 *   scores    : [batch][max_num] confidences, may be NULL */
int omni_sp_infer(omni_sp* sp, const uint8_t* gray_host, int stride, int batch, int fisheye_mask,
                  float* kps_xy, int* n_kps, float* desc, float* scores);
/* asynchronous, HBM-resident input; results stay in the handle's device buffers until fetched */
int omni_sp_enqueue_dev(omni_sp* sp, const uint8_t* gray_dev, int stride, int batch, int fisheye_mask);
int omni_sp_fetch(omni_sp* sp, int batch, float* kps_xy, int* n_kps, float* desc, float* scores);  /* D2H + sync */
/* device views of the last results: kps [max_batch][max_num][2] f32, n [max_batch] i32, desc [max_batch][max_num][dim] */
int omni_sp_dev_outputs(omni_sp* sp, const float** kps_xy_dev, const int** n_kps_dev, const float** desc_dev,
                        const float** scores_dev);
/* the engine's raw outputs in the reference's binding layout (tensorrt_generic.cpp:62-73):
 * semi [batch][H][W] f32 and desc [batch][256][H/8][W/8] f32 (NCHW).  For parity tests.  On the OMNI_PREC_F16 path a forward pass only
 * evaluates the descriptor head at the coarse cells around its key points (the key-point descriptors are bit-identical to sampling the
 * dense map); the dense `desc` is computed here, on demand, from the activations of the LAST forward pass (batch <= that pass's batch). */
int omni_sp_get_dense(omni_sp* sp, int batch, float* semi_host, float* desc_host);
/* run only the post-processing (getKeyPoints + NMS2 + computeDescriptors, superpoint_tensorrt.cpp:164-310) on
 * caller-supplied engine outputs in the layout above -- isolates the detector from conv rounding in tests */
int omni_sp_postprocess_dense(omni_sp* sp, const float* semi_host, const float* desc_host, int batch,
                              float* kps_xy, int* n_kps, float* desc, float* scores);
/* test hook: one intermediate activation of the last forward pass as NCHW float32 (name in {"conv1a","conv1b",...,
 * "conv4b","heads","desc"}; post-ReLU, post-pool where the layer pools).  out may be NULL to query the shape only. */
int omni_sp_debug_layer(omni_sp* sp, const char* name, int batch, float* out_nchw_host, int* C, int* H, int* W);
/* per-stage device time: runs the network `reps` times on HBM-resident input with HIP events between stages.
 * stage_ms [OMNI_SP_NUM_STAGES] MEDIAN ms per call over the repetitions; names via omni_sp_stage_name(). */
#define OMNI_SP_NUM_STAGES 16
int         omni_sp_profile(omni_sp* sp, const uint8_t* gray_dev, int stride, int batch, int reps, float* stage_ms);
/* enable_perf of the reference's runners (superpoint_tensorrt.cpp:130-162): with perf on every pass records its stage events; omni_sp_last_stage_ms waits for the
 * handle's stream and returns the LAST pass's device time per stage in milliseconds (stage_ms[OMNI_SP_NUM_STAGES], names: omni_sp_stage_name) */
int         omni_sp_set_perf(omni_sp* sp, int on);
int         omni_sp_last_stage_ms(omni_sp* sp, float* stage_ms);
const char* omni_sp_stage_name(int stage);
double      omni_sp_stage_flops(const omni_sp* sp, int stage);   /* algorithmic FLOP per image for that stage */
/* share of the stage's output tiles a fisheye-masked pass leaves out of the kernel's tile walk (the constant region of the mask, loop_cam.cpp:536-539:
 * written once per handle, bit-identical results); 0 when the stage computes every tile */
double      omni_sp_stage_tiles_left_out(const omni_sp* sp, int stage);
/* the plan itself (pure arithmetic on the image size and the kernels' tile shapes; no device needed): layer 0 = conv1a (OMNI_PREC_SPLIT only), 1..5 =
 * conv1b, conv2a, conv2b, conv3a, conv3b (both matrix-core precisions); rect = {tile row 0, tile row 1, tile column 0, tile column 1} of the layer's conv-output tile grid (empty = nothing
 * left out), frac = its share of the layer's tiles */
int         omni_sp_mask_skip_plan(int width, int height, int precision, int layer, int* rect, double* frac);

/* ---- MobileNetVLAD (ASSUMED architecture -- the reference ships only the I/O contract) --------- */
enum { OMNI_VLAD_CONV3X3_RELU6 = 0, OMNI_VLAD_PW_RELU6 = 1, OMNI_VLAD_DW3X3_RELU6 = 2,
       OMNI_VLAD_PW_LINEAR = 3, OMNI_VLAD_PW_LINEAR_RES = 4 };
typedef struct omni_vlad_layer {
    int kind, cin, cout, stride;
    const float* weight;   /* OIHW (depthwise: [C][1][3][3]) */
    const float* bias;
} omni_vlad_layer;
typedef struct omni_vlad_weights {
    int n_layers;
    const omni_vlad_layer* layers;
    int n_clusters, feat_dim, out_dim;
    const float* assign_w;   /* [K][D] */
    const float* assign_b;   /* [K] */
    const float* clusters;   /* [K][D] */
    const float* fc_w;       /* [out_dim][K*D] */
    const float* fc_b;       /* [out_dim] */
} omni_vlad_weights;

/* MobileNetVLADTensorRT(engine, width, height) (mobilenetvlad_tensorrt.h:10-19) */
omni_vlad* omni_vlad_create(omni_ctx* ctx, const omni_vlad_weights* w, int width, int height, int max_batch);
void       omni_vlad_destroy(omni_vlad* v);
/* OMNI_PREC_F32 (default): exact-f32 kernels, the parity mode.  OMNI_PREC_F16: the inverted-residual blocks run with fp16 matrix-core
 * operands and fp32 accumulation / residual stream -- the reference's engine is an fp16 TensorRT plan (launch/realsense.launch:10-11) */
int        omni_vlad_set_precision(omni_vlad* v, int precision);
/* test hook (host only, no device needed): the weight blob of one inverted-residual block as the split-fp16 matrix-core kernel reads it
 * (expand [hid][cin] + bias, depthwise [hid][9] + bias, projection [cout][hid], the layer table's OIHW order).  Returns the blob size in
 * bytes (out == NULL: size query), -1 when no kernel instantiation covers the shape, -2 on bad arguments.  Layout (vlad_s.hip): per 48-channel
 * chunk [10][48] fp32 depthwise taps + bias for all chunks, then per chunk the expand A fragments ([x_hi | x_lo | 1 1] pass with
 * We_hi, We_hi, be_hi, be_lo; x_hi pass with We_lo) and the projection A fragments (hi, lo per k-step), each fragment [64 lanes][8 halfs]. */
int64_t    omni_vlad_pack_block(int cin, int hid, int cout, int stride, const float* we, const float* be, const float* wd, const float* bd,
                                const float* wp, void* out, int64_t out_bytes);
/* Host-only test hooks (no GPU): the packed constants of two SuperPoint kernels, so that their algebra can be checked on the CPU.
 * which = 0: conv1a's matrix-core fragments for operands taken straight from the image bytes (csrc/conv.hip conv1a_pack_u8_weights): w [64][9],
 *   bias [64] -> out [2048] halfs, *scale = 1; which = 1: a cin = 64 layer's Winograd F(2x2,3x3) fragments (csrc/conv_wino.hip conv_pack_weights_wino):
 *   w OIHW [cout][64][3][3] -> out [64 * cout * 32] halfs, *scale = 2^-k of the packed values.  Returns the number of halfs written, -2 on bad arguments. */
int64_t omni_sp_pack_constants(int which, const float* w, const float* bias, int cout, uint16_t* out, int64_t out_halfs, float* scale);
/* std::vector<float> MobileNetVLADTensorRT::inference(const cv::Mat&) (mobilenetvlad_tensorrt.cpp:4-14):
 * u8 -> f32 with NO scaling feeds the net; out [batch][out_dim] */
int omni_vlad_infer(omni_vlad* v, const uint8_t* gray_host, int stride, int batch, int fisheye_mask, float* out);
int omni_vlad_enqueue_dev(omni_vlad* v, const uint8_t* gray_dev, int stride, int batch, int fisheye_mask);
int omni_vlad_fetch(omni_vlad* v, int batch, float* out);
int omni_vlad_dev_output(omni_vlad* v, const float** out_dev);
/* The layers (0 = stem + block 0, k = block k) whose tiles inside the constant region of the fisheye mask a masked pass leaves out (loop_cam.cpp:536-539
 * blanks the rows before netvlad_net.inference, :556-558): returns their number, frac[k] = the share of layer k's tiles left out (up to max_layers). */
int omni_vlad_mask_skip_layers(const omni_vlad* v, double* frac, int max_layers);

/* ---- global-descriptor index: faiss::IndexFlatIP(d) --------------------------------------------------------- */
omni_index* omni_index_create(omni_ctx* ctx, int dim, int storage, int64_t initial_capacity_rows);
void        omni_index_destroy(omni_index* idx);
int         omni_index_add(omni_index* idx, int64_t n, const float* x_host);       /* IndexFlatIP::add(n, x) */
int         omni_index_add_dev(omni_index* idx, int64_t n, const float* x_dev);
int64_t     omni_index_ntotal(const omni_index* idx);                              /* .ntotal */
int         omni_index_dim(const omni_index* idx);                                 /* .d */
int         omni_index_reset(omni_index* idx);
/* drop the rows appended last so that ntotal == n_rows again (undo of appends enqueued ahead by a batched caller that failed) */
int         omni_index_truncate(omni_index* idx, int64_t n_rows);
/* OMNI_STORE_F32 shards keep an fp16 mirror of their rows (+50 % HBM; OMNI_INDEX_MIRROR=0 disables it): a batch of >= 4 queries is scored
 * against the mirror in one matrix-core pass, the best k+24 candidates per query are re-scored exactly against the fp32 rows, and a per-query
 * certificate proves the result is the exact fp32 top-k (scores bit-identical to the single-query path); uncertified queries are re-run with
 * the exact scan.  Counters since creation: queries answered through the mirror / of those, the ones that needed the exact re-run. */
int         omni_index_cert_stats(omni_index* idx, int64_t* searches, int64_t* fallbacks);
/* IndexFlatIP::search(nq, q, k, D, I): exact inner product, k best descending, ties -> lower row id,
 * missing results padded with I = -1, D = -FLT_MAX.  k <= 1024 (the reference caps at 1000, loop_detector.cpp:200). */
int         omni_index_search(omni_index* idx, int nq, const float* q_host, int k, float* D, int64_t* I);
int         omni_index_search_dev(omni_index* idx, int nq, const float* q_dev, int k, float* D_dev, int64_t* I_dev);
/* search_dev restricted to the first n_limit local rows (the index as it was when it held n_limit rows).  Lets a caller enqueue
 * "add, query, add, query, ..." for several key frames -- LoopDetector::on_image_recv adds a frame's rows BEFORE querying
 * (loop_detector.cpp:89-98) -- on the stream without a host synchronisation in between: all adds first, then every query with the
 * ntotal it would have seen.  An empty prefix (n_limit == 0) fills the outputs with I = -1, D = -FLT_MAX. */
int         omni_index_search_prefix_dev(omni_index* idx, int nq, const float* q_dev, int k, int64_t n_limit, float* D_dev,
                                         int64_t* I_dev);
/* Several prefix searches in ONE pass over the shard (a micro-batch of key frames: every frame's query must see exactly the rows that
 * were in the index at its turn, loop_detector.cpp:89-98, but the rows are read from HBM once for all of them).  Query q is row
 * row_idx[q] of rows_dev (row_idx == NULL: rows 0..nq-1) and sees local rows [0, n_limits[q]) only; row_idx / n_limits are HOST arrays
 * read before the call returns.  nq <= 64.  Results as nq independent omni_index_search_prefix_dev calls would give them.
 * Ordering: rows_dev must be complete with respect to the index context's stream (e.g. the producer was waited for, or it ran on that
 * stream); the same holds for omni_index_add_dev / omni_index_search*_dev. */
int         omni_index_search_batch_prefix_dev(omni_index* idx, int nq, const float* rows_dev, const int64_t* row_idx, int k,
                                               const int64_t* n_limits, float* D_dev, int64_t* I_dev);
/* row sharding across GPUs: this handle holds rows g with g % world == rank at local slot g / world;
 * search then reports GLOBAL ids (local * world + rank).  Default rank 0, world 1. */
int         omni_index_set_shard(omni_index* idx, int rank, int world);   /* (an fp32 shard gives up its fp16 mirror: its batched searches must not wait on the host) */
/* host-side merge of per-shard top-k lists (after the all-gather): lists [n_lists][nq][k_each] -> [nq][k_out],
 * same ordering rule (score desc, id asc), entries with I < 0 ignored */
int         omni_topk_merge(int n_lists, int nq, int k_each, const float* D_lists, const int64_t* I_lists,
                            int k_out, float* D, int64_t* I);
/* Shard checkpoint (new -- the reference keeps its database in RAM only and loses it on restart):
 * "OMNX1" file = header {magic, dim, storage, ntotal} + the raw row matrix as stored (fp32 or fp16), streamed through a pinned
 * staging buffer.  load() replaces the handle's contents; dim and storage must match the handle. */
int         omni_index_save(omni_index* idx, const char* path);
int         omni_index_load(omni_index* idx, const char* path);
/* device time of the dominant scan kernel for the last search on this handle (HIP events on the ctx stream) */
int         omni_index_last_scan_ms(omni_index* idx, float* ms);

/* ---- the database row-sharded over the GPUs of one node: one process per GPU, RCCL over xGMI (new: the reference is single-GPU;
 * SURVEY.md 8e, BASELINE configs[3]/[4]).  Global row g lives on rank g % world at local slot g / world; every rank ends up with the
 * faiss::IndexFlatIP::search result of the UNSHARDED index (global ids, score desc, ties -> lower id).  RCCL is resolved at run time
 * (dlopen librccl.so.1); the launcher only has to carry the 128-byte unique id from rank 0 to the other ranks (any channel). */
#define OMNI_SHARD_ID_BYTES 128
/* the file the collective entry points (ncclAllGather ...) were resolved from: librccl of the process's ROCm stack, or what OMNI_RCCL_LIB names */
int         omni_shard_library_path(char* out, int cap);
int         omni_shard_unique_id(char* id_out /* [OMNI_SHARD_ID_BYTES], ncclGetUniqueId */);
/* collective: every rank calls it with the same id; `local` must be an empty index on ctx; its rows/ids are managed by the shard from here on */
omni_shard* omni_shard_create(omni_ctx* ctx, omni_index* local, int dim, int rank, int world, const char* unique_id);
void        omni_shard_destroy(omni_shard* s);
int64_t     omni_shard_ntotal(const omni_shard* s);                       /* GLOBAL row count */
int         omni_shard_preload_local(omni_shard* s, const float* rows_host, int64_t n_local, int64_t ntotal_global);
/* collective: F consecutive key-frame steps of every rank as ONE exchange unit.  rows_dev [F][m][dim] (HBM): this rank's m new rows of each
 * of its next F key frames.  Step f's rows of all ranks get the global ids ntotal + (f*world + r)*m + j; rank r's query of step f is its row
 * query_row of that step and sees exactly the rows up to and including step f (add before query, loop_detector.cpp:89-98).  Two
 * ncclAllGather (rows; per-shard top-k), one pass over the shard, one D2H; D_host / I_host [F][k] = this rank's merged results. */
int         omni_shard_step_batch_dev(omni_shard* s, int F, int m, const float* rows_dev, int query_row, int k, float* D_host, int64_t* I_host);
/* the same in two halves: _enqueue puts the whole unit -- both collectives, the scan, the copy of the lists to the host -- on the shard's stream
 * WITHOUT waiting (the caller goes on enqueuing CNN work); _rows_consumed blocks until rows_dev has been gathered and may be overwritten;
 * _wait blocks on the unit's completion event, merges this rank's results and moves the global row count.  One unit in flight at a time. */
int         omni_shard_step_enqueue(omni_shard* s, int F, int m, const float* rows_dev, int query_row, int k);
int         omni_shard_rows_consumed(omni_shard* s);
int         omni_shard_step_wait(omni_shard* s, float* D_host, int64_t* I_host);
/* device time (HIP events on the shard's stream, microseconds) of the two collectives of the exchange omni_shard_step_wait collected last: the all-gather
 * of every rank's new rows and the all-gather of the per-shard top-k lists (the one exchange step of the path) */
int         omni_shard_last_exchange_us(omni_shard* s, float* rows_gather_us, float* topk_gather_us);
/* collective: the same nq <= 64 queries on every rank -> the unsharded index's top-k on every rank */
int         omni_shard_search(omni_shard* s, int nq, const float* q_host, int k, float* D, int64_t* I);

/* ---- local-descriptor matcher: cv::BFMatcher(cv::NORM_L2, crossCheck=true).match(query, train, matches) -----
 * out arrays sized >= nq; matches ordered by query index; *n_matches = count.  dim <= 256. */
int omni_bf_match(omni_ctx* ctx, const float* q_host, int nq, const float* t_host, int nt, int dim, int mode,
                  int* q_idx, int* t_idx, float* dist, int* n_matches);
/* several pairs from host pointers in ONE upload / launch pair / download (the geometric verification matches up to four direction pairs per
 * loop candidate, loop_detector.cpp:431-537): outputs [n_pairs][max_n], n_matches [n_pairs]; nq[p], nt[p] <= max_n; n_pairs <= 64. */
int omni_bf_match_multi(omni_ctx* ctx, int n_pairs, const float* const* q_host, const int* nq, const float* const* t_host, const int* nt, int dim,
                        int mode, int max_n, int* q_idx, int* t_idx, float* dist, int* n_matches);
/* batched, HBM-resident: pair p uses q = q_dev + p*q_stride (floats), nq = nq_dev[p], likewise t.
 * outputs (device): q_idx/t_idx/dist [n_pairs][max_n], n_matches [n_pairs]. */
int omni_bf_match_batched_dev(omni_ctx* ctx, int n_pairs, int max_n, int dim, int mode,
                              const float* q_dev, int64_t q_stride, const int* nq_dev,
                              const float* t_dev, int64_t t_stride, const int* nt_dev,
                              int* q_idx_dev, int* t_idx_dev, float* dist_dev, int* n_matches_dev);

/* ---- fisheye flattening: FisheyeUndist::undist_all_cuda = cv::cuda::remap(INTER_LINEAR) per virtual pinhole view
 * (swarm_localization/test/fisheye_undist.hpp:57-90; VINS-Fisheye runs that class in front of swarm_loop).
 * map_xy[v] = the view's undistortion map [view_h][view_w][2] float (source x, y per output pixel: genOneUndistMap, :188-215; made on the host,
 * host/fisheye_flatten.hpp).  One launch remaps `batch` fisheye images into all views; image b's views are written back to back at
 * out_dev + b * omni_flatten_out_bytes(): view v at its running offset, rows packed -- ready for omni_sp_enqueue_dev / omni_cam_enqueue_dev.
 * Bilinear taps as cv::cuda's LinearFilter, border constant 0, round half to even. */
omni_flatten* omni_flatten_create(omni_ctx* ctx, int src_width, int src_height, int n_views, const int* view_w, const int* view_h,
                                  const float* const* map_xy);
void          omni_flatten_destroy(omni_flatten* f);
int64_t       omni_flatten_out_bytes(const omni_flatten* f);       /* bytes of all views of ONE source image */
int           omni_flatten_enqueue_dev(omni_flatten* f, const uint8_t* src_dev, int src_stride, int batch, uint8_t* out_dev);   /* asynchronous */

/* ---- key-frame frontend: the device work of LoopCam::on_flattened_images (swarm_loop/src/loop_cam.cpp:178-229;
 * generate_stereo_image_descriptor :341-523, extractor_img_desc_deepnet :525-585, match_HFNet_local_features :141-174)
 * as one asynchronous unit: SuperPoint on the 2*n_dirs images (up cameras first, then down), MobileNetVLAD on the n_dirs
 * up images, BFMatcher(L2, crossCheck) up <-> down per direction, and the D2H copy of every result into one pinned block.
 * The handles are borrowed (sp created with max_batch >= 2*n_dirs on sp_ctx, vlad with max_batch >= n_dirs on vlad_ctx,
 * both contexts on the same device).  Camera lifting / triangulation (:73-106, 405-454, 558-576) stay with the caller. */
typedef struct omni_cam_result {      /* pointers into the handle's pinned host block; valid until its next enqueue */
    int n_dirs, max_num, desc_dim, global_dim;
    const float* kps_xy;      /* [2*n_dirs][max_num][2] */
    const int*   n_kps;       /* [2*n_dirs] */
    const float* desc;        /* [2*n_dirs][max_num][desc_dim] */
    const float* scores;      /* [2*n_dirs][max_num] */
    const float* global_desc; /* [n_dirs][global_dim] */
    const int*   match_up;    /* [n_dirs][max_num]  key-point index in the up image   (cv::DMatch::queryIdx) */
    const int*   match_down;  /* [n_dirs][max_num]  key-point index in the down image (cv::DMatch::trainIdx) */
    const float* match_dist;  /* [n_dirs][max_num] */
    const int*   n_matches;   /* [n_dirs] */
    int n_images;             /* 2*n_dirs (omni_cam_create) or n_dirs (omni_cam_create_mono): the leading dimension of kps_xy / n_kps / desc / scores */
} omni_cam_result;
omni_cam* omni_cam_create(omni_ctx* sp_ctx, omni_sp* sp, omni_ctx* vlad_ctx, omni_vlad* vlad, int n_dirs, int max_num,
                          int global_dim, int bf_mode);
/* CameraConfig::PINHOLE_DEPTH (loop_cam.cpp:190-194; generate_gray_depth_image_descriptor :231-339; launch/realsense.launch): ONE camera per
 * image -- SuperPoint and MobileNetVLAD on each of the n_images images, no up/down match (n_matches = 0, the match arrays unused), the arrays
 * of omni_cam_result sized [n_images].  sp with max_batch >= n_images, vlad with max_batch >= n_images.  The landmarks come from the depth
 * image on the host (host/loop_geometry.hpp fill_depth_landmarks); pass fisheye_mask = 0 to the enqueue calls (:536 masks STEREO_FISHEYE only). */
omni_cam* omni_cam_create_mono(omni_ctx* sp_ctx, omni_sp* sp, omni_ctx* vlad_ctx, omni_vlad* vlad, int n_images, int max_num, int global_dim);
void      omni_cam_destroy(omni_cam* cam);
int       omni_cam_enqueue_dev(omni_cam* cam, const uint8_t* gray_dev, int stride, int fisheye_mask);   /* no host sync */
/* same with the images in HOST memory, image i at gray_host + i*stride*height (the reference hands every engine call a host cv::Mat
 * and copies it up synchronously, tensorrt_generic.cpp:58-75): ONE asynchronous upload of the 2*n_dirs images into a staging buffer
 * owned by the handle, then the work of omni_cam_enqueue_dev.  gray_host should be pinned (omni_host_alloc) and must stay untouched
 * until omni_cam_wait returns. */
int       omni_cam_enqueue_host(omni_cam* cam, const uint8_t* gray_host, int stride, int width, int height, int fisheye_mask);
/* the same from segments of (pinned) host memory, rows packed: the up cameras' images = the concatenation of n_up parts of up_images[i] images each, the down
 * cameras' likewise (n_down = 0 for a mono handle); the totals must be the unit's active size (omni_cam_set_active).  One asynchronous copy per part. */
int       omni_cam_enqueue_host_parts(omni_cam* cam, const uint8_t* const* up, const int* up_images, int n_up, const uint8_t* const* down, const int* down_images,
                                      int n_down, int width, int height, int fisheye_mask);
int       omni_cam_wait(omni_cam* cam, omni_cam_result* out);
/* A unit smaller than the handle was created for (a partly filled micro-batch of key frames that must not wait any longer): the next enqueues read
 * cams * n_dirs images -- the up cameras' first, the down cameras' right behind them -- and every array of omni_cam_result has that leading dimension.
 * 1 <= n_dirs <= the n_dirs of omni_cam_create; not while a unit is in flight. */
int       omni_cam_set_active(omni_cam* cam, int n_dirs);
/* non-blocking: *ready = 1 when omni_cam_wait would return at once (hipEventQuery on the unit's two events), or when nothing is pending */
int       omni_cam_ready(omni_cam* cam, int* ready);
/* Units in flight, oldest first: whatever is enqueued on `later` from now on starts behind the CONVOLUTION STACK of `earlier`'s last enqueue (an event
 * on the device; the host does not wait).  Without it the units' CU-filling kernels take turns, every unit finishes late and together; with it the
 * oldest unit finishes first while the next one's convolutions run under its small-grid tail (NMS, descriptor sampling, matcher).
 * streams: 1 = `later`'s SuperPoint stream waits, 2 = its MobileNetVLAD stream too, 0 = no-op. */
int       omni_cam_order_after(omni_cam* later, omni_cam* earlier, int streams);

#ifdef __cplusplus
}
#endif
#endif /* OMNI_HIP_H */
