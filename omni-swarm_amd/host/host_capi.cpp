// host_capi.cpp -> lib/libomni_host.so: C entry points over omni::KeyframePipeline (keyframe_pipeline.hpp) so that a launcher written
// in any language (bench.py via ctypes; a ROS nodelet via plain linking) can run the C++ key-frame host loop in-process.
// Plain g++ -- no HIP headers: everything below the adapters goes through the C ABI of libomni_hip.so (include/omni_hip.h).
#include <cstring>
#include <string>

#include "keyframe_pipeline.hpp"

namespace {
thread_local std::string g_err;
}

extern "C" {

struct omni_pipeline { omni::KeyframePipeline* p; };

const char* omni_pipeline_last_error(void) { return g_err.c_str(); }

// weights: OMNW1 files (tools/export_weights.py), PCA: the reference's two CSV files (superpoint_tensorrt.cpp:110-111)
omni_pipeline* omni_pipeline_create(int device, const char* sp_weights, const char* pca_comp_csv, const char* pca_mean_csv, const char* vlad_weights,
                                    int width, int height, float thres, int max_num, int precision, int microbatch, int pipelines, int storage,
                                    int self_id, double inner_product_thres, double init_mode_product_thres, int match_index_dist, int min_loop_num,
                                    int min_direction_loop, int geometry) {
    try {
        omni::KeyframePipeline::Config c;
        c.device = device; c.sp_weights = sp_weights; c.pca_comp = pca_comp_csv ? pca_comp_csv : ""; c.pca_mean = pca_mean_csv ? pca_mean_csv : "";
        c.vlad_weights = vlad_weights; c.width = width; c.height = height; c.thres = thres; c.max_num = max_num; c.precision = precision;
        c.microbatch = microbatch; c.pipelines = pipelines; c.storage = storage; c.self_id = self_id;
        c.inner_product_thres = inner_product_thres; c.init_mode_product_thres = init_mode_product_thres; c.match_index_dist = match_index_dist;
        c.min_loop_num = min_loop_num; c.min_direction_loop = min_direction_loop; c.geometry = geometry != 0;
        c.cx = width / 2.0; c.cy = height / 2.0; c.fx = c.fy = width / 2.0;      // 90 degree horizontal field of view
        return new omni_pipeline{new omni::KeyframePipeline(c)};
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

void omni_pipeline_destroy(omni_pipeline* h) {
    if (!h) return;
    delete h->p;
    delete h;
}

int omni_pipeline_preload(omni_pipeline* h, const float* rows, int64_t n) {
    try { h->p->preload(rows, n); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

int64_t omni_pipeline_db_rows(omni_pipeline* h) { return h->p->db_rows(); }

// collective over all ranks (one process per GPU): switch the database to the row-sharded index (omni_shard_*, RCCL inside libomni_hip.so);
// unique_id = omni_shard_unique_id() of rank 0.  Call before preload / run; preload then takes THIS rank's rows.
int omni_pipeline_attach_shard(omni_pipeline* h, int rank, int world, const char* unique_id) {
    try { h->p->attach_shard(rank, world, unique_id); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// see omni::KeyframePipeline::run; *hits = loop candidates found.  Returns 0 on success.
int omni_pipeline_run(omni_pipeline* h, int n_keyframes, int64_t first_msg_id, const uint8_t* const* pool, int n_pool, int first_slot,
                      const uint8_t* tail, int from_host, int* hits) {
    try {
        const int n = h->p->run(n_keyframes, first_msg_id, pool, n_pool, first_slot, tail, from_host != 0);
        if (hits) *hits = n;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// allocates whatever a later omni_pipeline_run(h, n_keyframes, ...) would allocate on first use (the unit for a partial micro-batch)
int omni_pipeline_prepare(omni_pipeline* h, int n_keyframes) {
    try { h->p->prepare(n_keyframes); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// must be called before the first run: switches the geometric verification stage on (host/loop_geometry.hpp) with a pinhole model for the
// flattened views.  omni_pipeline_geometry_stats: candidates handed to compute_loop / loop edges accepted so far.
int omni_pipeline_geometry_stats(omni_pipeline* h, int* compute_loop_calls, int* edges) {
    if (compute_loop_calls) *compute_loop_calls = h->p->geometry_calls();
    if (edges) *edges = (int)h->p->edges().size();
    return 0;
}

int omni_pipeline_sync(omni_pipeline* h) {
    try { h->p->sync(); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

}  // extern "C"
