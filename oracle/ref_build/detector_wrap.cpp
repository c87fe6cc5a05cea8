// C entry points around the reference's LoopDetector decision functions compiled from their own text (oracle/Makefile target
// _ref/libref_detector.so): swarm_loop/src/loop_detector.cpp:11-137 (on_image_recv) and :150-292 (add_to_database x2, query_from_database x2,
// query_fisheyeframe_from_database, database_size), extracted at build time into _ref/ (git-ignored).  Test infrastructure only.
#include "detector_shim.h"
using namespace std::chrono;       // loop_detector.cpp:6

#include REF_DET_SNIPPET_A         // void LoopDetector::on_image_recv(...)                      :11-137
// the identifiers (not the text) of two functions are renamed so that on_image_recv above reaches them through the observers of detector_shim.h
#define add_to_database add_to_database_REF
#define query_fisheyeframe_from_database query_fisheyeframe_from_database_REF
#include REF_DET_SNIPPET_B         // add_to_database ... database_size                           :150-292
#undef add_to_database
#undef query_fisheyeframe_from_database

namespace {
struct Handle {
    LoopDetector det; LoopCam cam;
    int min_loop_num = 15, min_direction_loop = 3, match_index_dist = 10, inter_init = 50;
    double inner = 0.6, init = 0.3;
    void publish() const {           // the reference keeps these as process globals
        MIN_LOOP_NUM = min_loop_num; MIN_DIRECTION_LOOP = min_direction_loop; MATCH_INDEX_DIST = match_index_dist; inter_drone_init_frames = inter_init;
        INNER_PRODUCT_THRES = inner; INIT_MODE_PRODUCT_THRES = init;
    }
};
}  // namespace

extern "C" {

typedef int (*ref_loop_verdict_fn)(int64_t new_msg_id, int64_t old_msg_id);

void* ref_det_create(int self_id, double inner_product_thres, double init_mode_product_thres, int match_index_dist, int min_loop_num,
                     int min_direction_loop, int inter_drone_init_frames_, int camera_configuration, ref_loop_verdict_fn verdict) {
    Handle* h = new Handle();
    h->det.self_id = self_id; h->det.loop_cam = &h->cam; h->cam.cfg = (CameraConfig)camera_configuration;
    h->inner = inner_product_thres; h->init = init_mode_product_thres; h->match_index_dist = match_index_dist; h->min_loop_num = min_loop_num;
    h->min_direction_loop = min_direction_loop; h->inter_init = inter_drone_init_frames_;
    if (verdict) h->det.loop_verdict = [verdict](int64_t a, int64_t b) { return verdict(a, b); };
    return h;
}
void ref_det_destroy(void* hv) { delete static_cast<Handle*>(hv); }

// one frame through the reference's on_image_recv.  descs: [n_images][4096].  out[8] = {added, queried, image_id, old_msg_id, dir_old, loop,
// database_size after the call, compute_loop calls} (the function returns void: read off the observers of detector_shim.h).
int ref_det_on_image_recv(void* hv, int64_t msg_id, int drone_id, int landmark_num, int prevent_adding_db, int n_images, const int* img_landmark_num,
                          const int* img_drone_id, const float* descs, int64_t* out) {
    Handle* h = static_cast<Handle*>(hv);
    h->publish();
    FisheyeFrameDescriptor_t f;
    f.msg_id = msg_id; f.drone_id = drone_id; f.landmark_num = landmark_num; f.prevent_adding_db = prevent_adding_db != 0;
    f.images.resize(n_images);
    for (int i = 0; i < n_images; ++i) {
        f.images[i].drone_id = img_drone_id[i]; f.images[i].landmark_num = img_landmark_num[i];
        f.images[i].image_desc.assign(descs + (size_t)i * 4096, descs + (size_t)(i + 1) * 4096);
    }
    LoopDetector& d = h->det;
    d.tr = LoopDetector::Trace();
    d.on_image_recv(f);
    int64_t image_id = -1;                 // best_image_id is local to query_fisheyeframe_from_database: the row whose maps point at (old frame, direction_old)
    if (d.tr.dir_old >= 0)
        for (auto& kv : d.imgid2fisheye)
            if (kv.second == d.tr.old_msg_id && d.imgid2dir.count(kv.first) && d.imgid2dir[kv.first] == d.tr.dir_old) { image_id = kv.first; break; }
    out[0] = d.tr.added; out[1] = d.tr.queried; out[2] = image_id; out[3] = d.tr.old_msg_id; out[4] = d.tr.dir_old;
    out[5] = d.tr.loop; out[6] = d.database_size(); out[7] = d.tr.n_compute_loop;
    return 0;
}

// the candidate a frame WOULD get from the current database (query_fisheyeframe_from_database, :245-287), without inserting it:
// out[4] = {found, image id via imgid maps (-1 when not found), old frame's msg_id, direction_old}; *distance as the reference leaves it
int ref_det_query(void* hv, int drone_id, int prevent_adding_db, int init_mode, int n_images, const int* img_landmark_num, const int* img_drone_id,
                  const float* descs, int64_t* out) {
    Handle* h = static_cast<Handle*>(hv);
    h->publish();
    FisheyeFrameDescriptor_t f;
    f.drone_id = drone_id; f.prevent_adding_db = prevent_adding_db != 0;
    f.images.resize(n_images);
    for (int i = 0; i < n_images; ++i) {
        f.images[i].drone_id = img_drone_id[i]; f.images[i].landmark_num = img_landmark_num[i];
        f.images[i].image_desc.assign(descs + (size_t)i * 4096, descs + (size_t)(i + 1) * 4096);
    }
    int direction_new = -1, direction_old = -1;
    FisheyeFrameDescriptor_t& old = h->det.query_fisheyeframe_from_database(f, init_mode != 0, prevent_adding_db != 0, direction_new, direction_old);
    out[0] = direction_old >= 0; out[1] = direction_new; out[2] = direction_old >= 0 ? old.msg_id : -1; out[3] = direction_old;    // (the not-found return value is a dangling reference in the reference: not touched)
    return 0;
}

// the 6-argument query_from_database (:199-242) against one of the two indexes: returns its return value, *distance in/out
int ref_det_query_index(void* hv, int remote_db, const float* desc, double thres, int max_index, double* distance) {
    Handle* h = static_cast<Handle*>(hv);
    h->publish();
    ImageDescriptor_t img;
    img.image_desc.assign(desc, desc + 4096);
    return h->det.query_from_database(img, remote_db ? h->det.remote_index : h->det.local_index, remote_db != 0, thres, max_index, *distance);
}
// the 4-argument query_from_database (:176-197)
int ref_det_query_image(void* hv, int drone_id, const float* desc, int init_mode, int nonkeyframe, double* distance) {
    Handle* h = static_cast<Handle*>(hv);
    h->publish();
    ImageDescriptor_t img;
    img.drone_id = drone_id; img.image_desc.assign(desc, desc + 4096);
    return h->det.query_from_database(img, init_mode != 0, nonkeyframe != 0, *distance);
}
int ref_det_database_size(void* hv) { return static_cast<Handle*>(hv)->det.database_size(); }
int ref_det_inter_drone_loop_count(void* hv, int a, int b) { return static_cast<Handle*>(hv)->det.inter_drone_loop_count[a][b]; }

}  // extern "C"
