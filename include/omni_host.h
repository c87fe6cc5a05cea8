/* omni_host.h -- C entry points of libomni_host.so (omni-swarm_amd/host/host_capi.cpp): the C++ host side of the path (host/keyframe_pipeline.hpp: LoopCam's
 * per-key-frame front end + LoopDetector + the geometry stage, over the kernels of libomni_hip.so) behind plain C, for bindings that cannot include C++
 * headers (pipeline.py uses exactly these).  The reference's counterpart is the node itself: SwarmLoop::Init / VIOKF_callback
 * (swarm_loop/src/swarm_loop.cpp:140-170,204-398), LoopCam (loop_cam.cpp:341-585) and LoopDetector::on_image_recv (loop_detector.cpp:11-137).
 * Every function returns 0 on success unless said otherwise; after a failure omni_pipeline_last_error() holds the message.  Plain pointers and sizes only. */
#ifndef OMNI_HOST_H
#define OMNI_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct omni_pipeline omni_pipeline;

/* message of the last call that failed (per calling thread) */
const char* omni_pipeline_last_error(void);

/* STEREO_FISHEYE (CameraConfig 1, swarm_loop.cpp:279-280: four directions, up + down cameras).  Weights: OMNW1 files (tools/export_weights.py); PCA:
 * the reference's two CSV files (superpoint_tensorrt.cpp:110-111).  The thresholds are LoopDetector's (swarm_loop.cpp:222-244).  pipelines <= 0: the
 * library's default for the precision.  NULL on failure */
omni_pipeline* omni_pipeline_create(int device, const char* sp_weights, const char* pca_comp_csv, const char* pca_mean_csv, const char* vlad_weights,
                                    int width, int height, float thres, int max_num, int precision, int microbatch, int pipelines, int storage,
                                    int self_id, double inner_product_thres, double init_mode_product_thres, int match_index_dist, int min_loop_num,
                                    int min_direction_loop, int geometry);

/* PINHOLE_DEPTH (CameraConfig 2, launch/realsense.launch, BASELINE.json configs[0]): one gray + one depth image per key frame, pinhole fx fy cx cy,
 * landmarks where depth_near < depth < depth_far (loop_cam.cpp:260-304) */
omni_pipeline* omni_pipeline_create_pinhole_depth(int device, const char* sp_weights, const char* pca_comp_csv, const char* pca_mean_csv,
                                                  const char* vlad_weights, int width, int height, float thres, int max_num, int precision,
                                                  int microbatch, int pipelines, int storage, int self_id, double inner_product_thres,
                                                  double init_mode_product_thres, int match_index_dist, int min_loop_num, int min_direction_loop,
                                                  int geometry, double fx, double fy, double cx, double cy, double depth_near, double depth_far,
                                                  int accept_min_3d_pts);

/* configured by one of the reference's launch files (the four files of swarm_loop/launch/, read as roslaunch + swarm_loop.cpp:205-270 would): launch_xml = the file's text, args =
 * "name:=value ..." overrides; what a launch file cannot know stays an argument */
omni_pipeline* omni_pipeline_create_from_launch(int device, const char* launch_xml, const char* node_name, const char* args, const char* sp_weights,
                                                const char* vlad_weights, const char* pca_comp_csv, const char* pca_mean_csv, int precision,
                                                int microbatch, int pipelines, int storage, int geometry, const double* intrinsics4);

/* the launch file's detector / geometry thresholds onto an existing pipeline */
int omni_pipeline_apply_launch(omni_pipeline* h, const char* launch_xml, const char* node_name);

void omni_pipeline_destroy(omni_pipeline* h);

/* PINHOLE_DEPTH: depth images (u16 millimetres, [n][height][width]) of key frames first_msg_id .. first_msg_id + n - 1; BORROWED until the run that
 * uses them returns */
int omni_pipeline_set_depth(omni_pipeline* h, int64_t first_msg_id, int64_t n, const uint16_t* depth);

/* odometry poses (xyz + quaternion wxyz) of key frames first_msg_id ..: what VIOKF_callback receives with the frame (swarm_loop.cpp:140-170); used by
 * the geometry stage */
int omni_pipeline_set_poses(omni_pipeline* h, int64_t first_msg_id, int64_t n, const double* poses7);

/* rows already in the key-frame database (n x 4096 fp32): a map built earlier */
int omni_pipeline_preload(omni_pipeline* h, const float* rows, int64_t n);

int64_t omni_pipeline_db_rows(omni_pipeline* h);

/* collective over all ranks (one process per GPU): the database becomes this rank's part of a row-sharded one; unique_id =
 * omni_shard_unique_id() of rank 0 */
int omni_pipeline_attach_shard(omni_pipeline* h, int rank, int world, const char* unique_id);

/* allocates what a later omni_pipeline_run(h, n_keyframes, ...) would allocate on its first call */
int omni_pipeline_prepare(omni_pipeline* h, int n_keyframes);

/* a batch of key frames through the whole path (LoopCam front end -> LoopDetector::on_image_recv, loop_detector.cpp:11-137); pool[s] = the packed
 * images of micro-batch s; *hits = loop candidates found */
int omni_pipeline_run(omni_pipeline* h, int n_keyframes, int64_t first_msg_id, const uint8_t* const* pool, int n_pool, int first_slot,
                      const uint8_t* tail, int from_host, int* hits);

/* the streaming entry = SwarmLoop::VIOKF_callback's hand-over (swarm_loop.cpp:140-170): one key frame, images[] host pointers (up cameras, then down
 * cameras), pose7 = xyz + quaternion wxyz */
int omni_pipeline_push_keyframe(omni_pipeline* h, const uint8_t* const* images, int stride, int64_t msg_id, double stamp, const double* pose7,
                                int prevent_adding_db, const uint16_t* depth, int* hits);

/* the latency bound of the streaming intake: call from a timer; never waits for a CNN unit */
int omni_pipeline_poll(omni_pipeline* h, int* hits);

/* everything pushed so far through the detector (and the geometry stage) */
int omni_pipeline_flush(omni_pipeline* h, int* hits);

/* max_wait_ms (< 0: never) and dispatch_when_idle of the streaming intake */
int omni_pipeline_set_latency(omni_pipeline* h, double max_wait_ms, int dispatch_when_idle);

/* units (micro-batches) in flight; *units_oldest_first = what the last run() used */
int omni_pipeline_units(omni_pipeline* h, int* units_oldest_first);

int omni_pipeline_sync(omni_pipeline* h);

/* loop candidates so far, [i][4] = {new key frame, old key frame, direction_new, direction_old} (loop_detector.cpp:150-287) */
int omni_pipeline_get_candidates(omni_pipeline* h, int64_t* out, int max);

/* accepted loop edges so far (swarm_msgs::LoopEdge as compute_loop fills it, loop_detector.cpp:627-836) */
int omni_pipeline_get_edges(omni_pipeline* h, double* out, int max);

/* compute_loop calls and accepted edges so far */
int omni_pipeline_geometry_stats(omni_pipeline* h, int* compute_loop_calls, int* edges);

/* per-micro-batch latencies (ms): upload start -> detector (+ geometry) step done */
int omni_pipeline_get_latencies(omni_pipeline* h, double* out, int max, int reset);

/* sharded mode: device microseconds of the two all-gathers of every exchange unit, [i][2] */
int omni_pipeline_get_exchange_us(omni_pipeline* h, float* out, int max, int reset);

/* the host thread's milliseconds per unit: enqueue, waiting for the GPU, messages, detector, geometry */
int omni_pipeline_host_times(omni_pipeline* h, double* out5, int reset);

/* FisheyeUndist's undistortion maps (swarm_localization/test/fisheye_undist.hpp:57-90) for omni_flatten_create */
int omni_fisheye_maps(const double* mei, int img_width, double fov_deg, int cam_id, int* n_views, int* view_w, int* view_h, float* const* maps);

/* the parameter server a launch file gives the node, as "name<TAB>type<TAB>value" lines (swarm_loop.cpp:205-270 read through roscpp's param<T>
 * conversions) */
int omni_swarm_params_from_launch(const char* launch_xml, const char* node_name, const char* args, char* out, int cap);

/* the node's parameter table: "name<TAB>I|B|D|S<TAB>default" per line */
int omni_swarm_params_table(char* out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* OMNI_HOST_H */
