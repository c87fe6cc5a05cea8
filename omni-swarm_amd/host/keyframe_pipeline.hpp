// keyframe_pipeline.hpp -- the per-key-frame host flow of SwarmLoop::VIOKF_callback (swarm_loop/src/swarm_loop.cpp:140-170:
//     ret = loop_cam->on_flattened_images(stereoframe, imgs);  ...  loop_detector->on_image_recv(ret, imgs);)
// as a C++17 driver over the adapters of omni_swarm.hpp, for a STREAM of key frames: `pipelines` micro-batches of `microbatch` key
// frames are in flight on the GPU (each an omni_cam unit: upload + SuperPoint + MobileNetVLAD + up/down BF + one D2H), while the host
// hands the finished ones to LoopDetectorCore::on_images_recv_batch in arrival order.  The reference does the same work strictly
// serially, one blocking engine call at a time.  This is the host loop bench.py times (through host_capi.cpp); Python only
// prepares the synthetic inputs and brackets the run with the barrier.
#pragma once
#include <chrono>
#include <condition_variable>
#include <deque>
#include <future>
#include <memory>
#include <thread>

#include "loop_geometry.hpp"
#include "omni_swarm.hpp"

namespace omni {

// fixed pool of host threads for the geometry stage (one candidate per task)
class TaskPool {
public:
    explicit TaskPool(int n) {
        for (int i = 0; i < n; ++i)
            workers_.emplace_back([this] {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
                        if (stop_ && jobs_.empty()) return;
                        job = std::move(jobs_.front());
                        jobs_.pop_front();
                    }
                    job();
                }
            });
    }
    ~TaskPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    template <typename F> auto submit(F&& f) -> std::future<decltype(f())> {
        auto task = std::make_shared<std::packaged_task<decltype(f())()>>(std::forward<F>(f));
        auto fut = task->get_future();
        { std::lock_guard<std::mutex> lk(mu_); jobs_.emplace_back([task] { (*task)(); }); }
        cv_.notify_one();
        return fut;
    }
    int size() const { return (int)workers_.size(); }
private:
    std::vector<std::thread> workers_;
    std::deque<std::function<void()>> jobs_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
};

class KeyframePipeline {
public:
    struct Config {
        int device = 0, width = 600, height = 480, max_num = 200, precision = OMNI_PREC_F16;
        float thres = 0.02f;
        int microbatch = 8, pipelines = 0 /* units in flight; <= 0: default_pipelines(precision) */, storage = OMNI_STORE_F32, self_id = 1;
        double inner_product_thres = 0.3, init_mode_product_thres = 0.2;
        int match_index_dist = 5, min_loop_num = 30, min_direction_loop = 3;
        std::string sp_weights, pca_comp, pca_mean, vlad_weights;
        // geometric verification (f64, host): lift the key points of the flattened pinhole views, triangulate up/down matches
        // (loop_cam.cpp:397-444, 558-569) and hand every candidate to LoopGeometry::compute_loop (loop_detector.cpp:627-836)
        bool geometry = false;
        double fx = 300, fy = 300, cx = 300, cy = 240, triangle_thres = 0.006, stereo_baseline = 0.10;
        int accept_min_3d_pts = 50;
        // CameraConfig (loop_defines.h:111-116): STEREO_FISHEYE = 1 -- a key frame is 4 directions x (up, down) flattened views, the bottom quarter of
        // every view blanked; PINHOLE_DEPTH = 2 (launch/realsense.launch, BASELINE.json configs[0]: 640 x 480) -- a key frame is ONE gray image, not
        // blanked, plus its 16-bit depth image in millimetres (set_depth), MAX_DIRS = 1 (swarm_loop.cpp:279-280), the query image is direction 0
        // (loop_detector.cpp:252-258) and the landmarks are read from the depth image (loop_cam.cpp:260-304)
        int camera_configuration = 1;
        double depth_near = 0.3, depth_far = 10.0;      // DEPTH_NEAR_THRES / DEPTH_FAR_THRES: the reference's own defaults (swarm_loop.cpp:251-252; launch/realsense.launch sets 0.3 / 100)
        // The streaming intake's latency bound (push_keyframe / poll).  The reference reports a key frame's loop synchronously (swarm_loop.cpp:140-170); a
        // micro-batch that only left for the GPU when `microbatch` key frames had arrived would, at the reference's 1 Hz key-frame rate (max_freq), hold a key
        // frame for seconds.  So: (1) dispatch_when_idle: a key frame that arrives while NO unit is in flight goes to the GPU at once, as a unit of one --
        // batches only grow while the GPU is busy anyway (adaptive batching: full throughput under load, one frame's compute time when idle);
        // (2) max_wait_ms: a partly filled micro-batch older than this is sent as it is by the next push_keyframe() or poll() (< 0: never -- only flush()).
        bool dispatch_when_idle = true;
        double max_wait_ms = 50.0;
        bool mono() const { return camera_configuration == 2; }
        int dirs() const { return mono() ? 1 : 4; }
    };

    // units in flight when the caller does not say: fp16's small-grid tails (NMS, sampling, matcher, MobileNetVLAD's last blocks) are filled by the next
    // units' kernels, so four pay; the fp32-class modes are CU-filling convolutions end to end and gain nothing beyond two
    static int default_pipelines(int precision) { return precision == OMNI_PREC_F16 ? 4 : 2; }
    static Config resolved(Config c) { if (c.pipelines <= 0) c.pipelines = default_pipelines(c.precision); return c; }
    int pipelines() const { return cfg_.pipelines; }

    explicit KeyframePipeline(const Config& c0) : KeyframePipeline(resolved(c0), 0) {}
private:
    KeyframePipeline(const Config& c, int) : cfg_(c), index_ctx_(c.device, true), det_(index_ctx_, c.self_id, c.storage) {
        det_.INNER_PRODUCT_THRES = c.inner_product_thres; det_.INIT_MODE_PRODUCT_THRES = c.init_mode_product_thres;
        det_.MATCH_INDEX_DIST = c.match_index_dist; det_.MIN_LOOP_NUM = c.min_loop_num; det_.MIN_DIRECTION_LOOP = c.min_direction_loop;
        if (c.camera_configuration != 1 && c.camera_configuration != 2) throw std::runtime_error("KeyframePipeline: camera_configuration must be 1 (STEREO_FISHEYE) or 2 (PINHOLE_DEPTH)");
        det_.stereo_fisheye = !c.mono();
        geo_.MAX_DIRS = c.dirs();
        for (int p = 0; p < c.pipelines; ++p) lanes_.push_back(std::make_unique<Lane>(c, c.microbatch));
        if (c.geometry) {
            geo_.self_id = c.self_id; geo_.MIN_LOOP_NUM = c.min_loop_num; geo_.MIN_DIRECTION_LOOP = c.min_direction_loop;
            // Per candidate: (1) on this thread, the up to four direction pairs of compute_correspond_features (loop_detector.cpp:431-537) are
            // matched in ONE GPU round trip (the matcher is a pure function of its two descriptor sets); (2) the rest of compute_loop -- flag
            // filter, homography-RANSAC mask, PnP-RANSAC + refit, verification: f64, milliseconds -- runs as a task on a pool of host threads,
            // its per-pair match() calls served from (1).  Tasks of one micro-batch run side by side; finish() collects them IN CANDIDATE ORDER,
            // so edges, their ids and the on_loop order are those of the serial flow.  The detector itself only needs compute_loop's verdict for
            // the init-mode counters of REMOTE drones (inter_drone_loop_count, loop_detector.cpp:66-72,826-827): candidates between two frames
            // of the self drone are deferred, anything else is verified on the spot.
            {
                int nt = -1;                                            // (the library's one table of switches: csrc/config.h)
                check(omni_config_value("OMNI_GEOMETRY_THREADS", &nt), "omni_config_value");
                if (nt < 0) nt = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 2));
                if (nt > 0) pool_ = std::make_unique<TaskPool>(nt);
            }
            det_.compute_loop = [this](const FisheyeFrameDescriptor& a, const FisheyeFrameDescriptor& b, int da, int db, bool im) {
                ++geometry_calls_;
                if (a.drone_id == cfg_.self_id && b.drone_id == cfg_.self_id) {      // verdict not needed by the detector: deferred to collect_geometry()
                    // CONTRACT: for a candidate between two frames of the self drone the callback returns false HERE, whatever the verdict will be:
                    // LoopCandidate::loop of the detector's record stays false and inter_drone_loop_count[{self, self}] is not counted by the detector (the
                    // reference counts it, loop_detector.cpp:826-827, and never reads it: init mode only exists for OTHER drones, :66-72).  The verdict is
                    // the edge: edges() / geometry_stats() hold it once the micro-batch's tasks are drained, and drain_geometry() adds the count then.
                    deferred_.push_back({&a, &b, da, db, im});
                    return false;
                }
                collect_geometry();                                     // keep the order: everything deferred earlier comes first
                deferred_.push_back({&a, &b, da, db, im});
                const size_t before = edges_.size();
                collect_geometry();
                return edges_.size() > before;
            };
        }
    }
public:
    // The geometry of every deferred candidate: (1) ONE GPU round trip matches the direction pairs of all of them (compute_correspond_features
    // pairs up to four directions per candidate, loop_detector.cpp:431-537; the matcher is a pure function of its two descriptor sets);
    // (2) one task per candidate on the pool: flag filter, homography-RANSAC masks, PnP-RANSAC + refit, verification (f64), its match() calls
    // served from (1); (3) the accepted edges are numbered and appended in candidate order.
    // start_geometry() does (1) and submits (2); drain_geometry() does (3).  finish() leaves a micro-batch's tasks running while the host
    // waits for the NEXT micro-batch's CNN unit and drains them before it starts that one's geometry: batches are drained in the order they
    // were started and candidates in the order the detector returned them, so edges, ids and latencies are those of the serial flow.
    void collect_geometry() { drain_geometry(); start_geometry(nullptr); drain_geometry(); }
    void drain_geometry() {
        while (!geo_inflight_.empty()) {
            GeoBatch& gb = geo_inflight_.front();
            for (size_t fi = 0; fi < gb.futs.size(); ++fi) {
                std::pair<bool, LoopEdge> r = gb.futs[fi].get();
                if (r.first) {
                    geo_.number_edge(r.second); edges_.push_back(r.second);
                    if (fi < gb.self_self.size() && gb.self_self[fi]) det_.inter_drone_loop_count[{cfg_.self_id, cfg_.self_id}] += 2;      // what the detector would have counted (:826-827, both orders)
                }
            }
            if (gb.timed) latencies_ms_.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - gb.t_enqueue).count());
            geo_inflight_.pop_front();
        }
    }
    // held: frames of the micro-batch that were not moved into the database (the tasks point into them); t_enqueue: its latency is recorded
    // when its last task has been collected.  Returns true when tasks were left in flight.
    bool start_geometry(std::vector<FisheyeFrameDescriptor>* held, const std::chrono::steady_clock::time_point* t_enqueue = nullptr) {
        if (deferred_.empty()) return false;
        struct Prepared { size_t first = 0, count = 0; };
        std::vector<BFMatcherL2X::Pair> pairs;
        std::vector<int> pair_dim;                                   // descriptor length of every pair (one match_multi call per distinct length)
        std::vector<Prepared> prep(deferred_.size());
        const int nd = geo_.MAX_DIRS;
        for (size_t ci = 0; ci < deferred_.size(); ++ci) {
            const FisheyeFrameDescriptor &a = *deferred_[ci].a, &b = *deferred_[ci].b;
            const int da = deferred_[ci].da, db = deferred_[ci].db;
            prep[ci].first = pairs.size();
            for (int d = da; d < da + nd; ++d) {                       // the pairing rule of compute_correspond_features (frame pair)
                const int dn = d % nd, dold = ((db - da + nd) % nd + d) % nd;
                if (dn < (int)a.images.size() && dold < (int)b.images.size() && b.images[dold].landmark_num > 0 && a.images[dn].landmark_num > 0) {
                    const ImageDescriptor &x = a.images[dn], &y = b.images[dold];
                    const int nx = (int)x.landmarks_2d.size(), ny = (int)y.landmarks_2d.size();
                    if (nx > 0 && ny > 0 && x.feature_descriptor.size() % nx == 0) {
                        const int dim = (int)(x.feature_descriptor.size() / nx);
                        if (dim > 0 && (int)y.feature_descriptor.size() == ny * dim) { pairs.push_back({x.feature_descriptor.data(), nx, y.feature_descriptor.data(), ny}); pair_dim.push_back(dim); }
                    }
                }
            }
            prep[ci].count = pairs.size() - prep[ci].first;
        }
        std::vector<std::vector<DMatch>> outs(pairs.size());
        {
            std::vector<char> done(pairs.size(), 0);
            for (size_t p0 = 0; p0 < pairs.size(); ++p0) {
                if (done[p0]) continue;
                std::vector<BFMatcherL2X::Pair> sub;
                std::vector<size_t> where;
                for (size_t p = p0; p < pairs.size(); ++p)
                    if (!done[p] && pair_dim[p] == pair_dim[p0]) { sub.push_back(pairs[p]); where.push_back(p); done[p] = 1; }
                std::vector<std::vector<DMatch>> so;
                bf_.match_multi(sub, pair_dim[p0], so);
                for (size_t j = 0; j < where.size(); ++j) outs[where[j]] = std::move(so[j]);
            }
        }
        using Result = std::pair<bool, LoopEdge>;
        geo_inflight_.emplace_back();
        GeoBatch& gb = geo_inflight_.back();
        if (held) gb.held = std::move(*held);                      // (a moved vector keeps its elements where they are)
        if (t_enqueue) { gb.timed = true; gb.t_enqueue = *t_enqueue; }
        std::vector<std::future<Result>>& futs = gb.futs;
        for (size_t ci = 0; ci < deferred_.size(); ++ci) {
            auto mine_p = std::make_shared<std::vector<BFMatcherL2X::Pair>>(pairs.begin() + prep[ci].first, pairs.begin() + prep[ci].first + prep[ci].count);
            auto mine_o = std::make_shared<std::vector<std::vector<DMatch>>>();
            auto mine_d = std::make_shared<std::vector<int>>(pair_dim.begin() + prep[ci].first, pair_dim.begin() + prep[ci].first + prep[ci].count);
            for (size_t j = 0; j < prep[ci].count; ++j) mine_o->push_back(std::move(outs[prep[ci].first + j]));
            const Deferred c = deferred_[ci];
            auto work = [g0 = geo_, mine_p, mine_o, mine_d, c]() -> Result {    // g0: the parameters, copied on this thread
                LoopGeometry g = g0;
                g.match = [&](const float* q, int nq, const float* t, int nt, int dim, std::vector<DMatch>& out) {
                    for (size_t p = 0; p < mine_p->size(); ++p)
                        if ((*mine_p)[p].query == q && (*mine_p)[p].train == t && (*mine_p)[p].nq == nq && (*mine_p)[p].nt == nt && (*mine_d)[p] == dim) { out = (*mine_o)[p]; return; }
                    // every pair compute_correspond_features can ask for was listed above (the same rule, whatever its descriptor length); a pair that
                    // was not has malformed descriptors (a length that is not a multiple of its key-point count): no matches
                    out.clear();
                };
                Result r;
                r.first = g.compute_loop_core(*c.a, *c.b, c.da, c.db, r.second, c.im);
                return r;
            };
            gb.self_self.push_back(c.a->drone_id == cfg_.self_id && c.b->drone_id == cfg_.self_id);
            if (pool_) futs.push_back(pool_->submit(work));
            else { std::promise<Result> pr; pr.set_value(work()); futs.push_back(pr.get_future()); }
        }
        deferred_.clear();
        return true;
    }
    // device time of the two all-gathers of every exchange unit collected so far (microseconds: new rows, per-shard top-k lists); sharded mode only
    const std::vector<std::pair<float, float>>& exchange_us() const { return exchange_us_; }
    void clear_exchange_us() { exchange_us_.clear(); }
    int geometry_calls() const { return geometry_calls_; }
    int last_run_fifo() const { return last_fifo_; }
    // where the host thread's time goes, per unit (micro-batch), in milliseconds since the last reset: [0] enqueue (upload + launches), [1] waiting for
    // the unit's results, [2] building the frame messages (+ stereo landmarks), [3] the detector step (index appends / searches, one GPU round trip),
    // [4] geometry hand-over; returns the number of units.  The loop runs at the GPU's pace while [1] > 0: the host then waits for the GPU, not the GPU for it.
    int host_times(double out[5], bool reset) {
        for (int i = 0; i < 5; ++i) out[i] = host_units_ ? host_ms_[i] / host_units_ : 0.0;
        const int n = host_units_;
        if (reset) { for (double& v : host_ms_) v = 0; host_units_ = 0; }
        return n;
    }
    const std::vector<LoopEdge>& edges() const { return edges_; }
    // every loop candidate the detector returned (query_fisheyeframe_from_database found an old frame), in key-frame order
    struct Candidate { int64_t new_msg_id, old_msg_id; int dir_new, dir_old; };
    const std::vector<Candidate>& candidates() const { return candidates_; }
    // odometry poses of the key frames (VIO's pose_drone, swarm_loop.cpp:140-170 takes it from the key-frame message): key frame msg_id gets
    // poses7[msg_id - first_msg_id] = position xyz + quaternion wxyz; key frames outside the range keep the identity
    void set_poses(int64_t first_msg_id, const double* poses7, int64_t n) {
        pose_base_ = first_msg_id;
        poses_.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            PoseMsg m;
            for (int k = 0; k < 3; ++k) m.position[k] = poses7[i * 7 + k];
            for (int k = 0; k < 4; ++k) m.quat_wxyz[k] = poses7[i * 7 + 3 + k];
            poses_[(size_t)i] = m;
        }
    }
    // extrinsics of the virtual pinhole views of the stacked fisheye pair: direction d looks along the body x axis rotated by 90 deg * d,
    // the up / down cameras sit +- baseline/2 along body z (camera axes: x right, y down, z forward)
    geom::Pose view_extrinsic(int direction, bool up) const {
        if (cfg_.mono()) {                                  // the one forward-looking camera, at the body origin
            geom::Mat3 R;
            R.m[0][0] = 0;  R.m[0][1] = 0;  R.m[0][2] = 1;
            R.m[1][0] = -1; R.m[1][1] = 0;  R.m[1][2] = 0;
            R.m[2][0] = 0;  R.m[2][1] = -1; R.m[2][2] = 0;
            return {{0, 0, 0}, geom::quat_from_R(R)};
        }
        const double yaw = M_PI / 2 * direction, c = std::cos(yaw), s = std::sin(yaw);
        geom::Mat3 R;            // Rz(yaw) * [[0,0,1],[-1,0,0],[0,-1,0]]
        R.m[0][0] = s;  R.m[0][1] = 0;  R.m[0][2] = c;
        R.m[1][0] = -c; R.m[1][1] = 0;  R.m[1][2] = s;
        R.m[2][0] = 0;  R.m[2][1] = -1; R.m[2][2] = 0;
        return {{0, 0, (up ? 0.5 : -0.5) * cfg_.stereo_baseline}, geom::quat_from_R(R)};
    }

    LoopDetectorCore& detector() { return det_; }
    LoopGeometry& geometry() { return geo_; }
    // the reference's launch parameters (host/swarm_loop_params.hpp) that reach the detector and the geometry stage AFTER construction (the constructor took
    // what Config carries: SwarmLoopParams::to_pipeline_config)
    template <class Params> void apply_params(const Params& p) { p.to_detector(det_); p.to_geometry(geo_); }

    // PINHOLE_DEPTH: the depth images (u16 millimetres, height x width, rows packed) of key frames msg_id = first_msg_id .. first_msg_id + n - 1,
    // image i at depth + i * width * height.  Borrowed: they must stay valid until the run() that consumes them returns.  A key frame without
    // one gets no landmarks.
    void set_depth(int64_t first_msg_id, const uint16_t* depth, int64_t n) { depth_base_ = first_msg_id; depth_ = depth; depth_n_ = n; }

    // N > 1 GPUs, one process per GPU: the database becomes one row-sharded index over all ranks (omni_shard_*, RCCL inside libomni_hip.so).
    // Collective: every rank attaches with rank 0's unique id.  Each micro-batch is then one exchange unit (two ncclAllGather) and a key
    // frame's candidate is decided on GLOBAL ids with the reference's rule (loop_detector.cpp:232: recency + threshold); the id maps of
    // LoopDetectorCore describe one drone's own database and are not used in this mode.
    void attach_shard(int rank, int world, const char* unique_id) {
        shard_index_ = std::make_unique<IndexFlatIP>(index_ctx_, 4096, cfg_.storage);
        shard_ = omni_shard_create(index_ctx_.get(), shard_index_->handle(), 4096, rank, world, unique_id);
        if (!shard_) throw std::runtime_error(std::string("omni_shard_create: ") + omni_last_error());
        world_ = world;
    }
    ~KeyframePipeline() { if (shard_) omni_shard_destroy(shard_); }
    int64_t db_rows() const { return shard_ ? omni_shard_ntotal(shard_) : det_.local_index.ntotal + det_.remote_index.ntotal; }

    // bulk pre-load of the key-frame database: rows [n][4096], dirs() consecutive rows = the directions of one earlier key frame
    void preload(const float* rows, int64_t n) {
        if (shard_) { check(omni_shard_preload_local(shard_, rows, n, n * world_), "omni_shard_preload_local"); return; }    // this rank's rows
        const int64_t base = det_.local_index.ntotal;
        for (int64_t s = 0; s < n; s += 4096) det_.local_index.add(std::min<int64_t>(4096, n - s), rows + s * 4096);
        const int nd = cfg_.dirs();
        for (int64_t i = 0; i < n; ++i) { det_.imgid2fisheye[(int)(base + i)] = -((base + i) / nd) - 1; det_.imgid2dir[(int)(base + i)] = (int)((base + i) % nd); }
    }

    // n_keyframes key frames through the whole hot path.  pool[e] = one micro-batch of images in (pinned) host memory,
    // [up cameras of the MB frames (4 each) | down cameras of the MB frames] (PINHOLE_DEPTH: the MB gray images), u8, rows packed; micro-batch s uses pool[(first_slot + s) %
    // n_pool].  When n_keyframes is not a multiple of the micro-batch the last rem frames run as their own, smaller unit from `tail`
    // (same layout for rem frames): EXACTLY n_keyframes key frames are processed.  from_host: upload inside the loop
    // (omni_cam_enqueue_host); otherwise the pool entries are device pointers.  Returns the number of loop candidates found.
    int run(int n_keyframes, int64_t first_msg_id, const uint8_t* const* pool, int n_pool, int first_slot, const uint8_t* tail, bool from_host) {
        if (open_ || !stream_pending_.empty()) throw std::runtime_error("run: key frames pushed through push_keyframe are still open -- flush() first");
        const int MB = cfg_.microbatch, nd = cfg_.dirs();
        const int full = n_keyframes / MB, rem = n_keyframes % MB;
        // The units of this call.  Host blocks can be cut anywhere (a unit's upload is a list of segments: omni_cam_enqueue_host_parts), so a run that is not a
        // whole number of micro-batches is cut into units of (nearly) EQUAL size -- 20 key frames = 7 + 7 + 6, not 8 + 8 + 4: a unit far below the size the
        // kernels' grids were sized for runs the same launches at a fraction of the work, and at the reference's key-frame rates (swarm_loop.cpp:140-170: 0.3-1 Hz
        // per drone) every run() is a short one.  Device-resident blocks and the sharded database keep the blocks' own cut (a unit there is one exchange step).
        std::vector<int> sizes;
        const bool recut = from_host && !shard_ && unit_plan_ != 0 && rem != 0 && full >= 1;
        if (recut) sizes = plan_units(n_keyframes, MB, unit_plan_);
        else { sizes.assign(full, MB); if (rem) sizes.push_back(rem); }
        Lane* tail_lane = (!recut && rem) ? prepare(n_keyframes) : nullptr;
        std::deque<std::pair<Lane*, int64_t>> pending;
        int hits = 0;
        // units oldest first (omni_cam_order_after)?  OMNI_PIPELINE_FIFO >= 0 decides; the default (-1) does by what this call is given: the fp32-class
        // modes always (CU-filling convolutions end to end: +2-4 %); fp16 when the whole run is in flight at once (no more units than lanes: the first
        // unit then finishes early and the host's work on it overlaps the rest, +13 % at 3 units, +1.5 % at 4), not in a run the lanes' in-flight limit
        // staggers anyway (the units' kernels taking turns fill each other's tails: +6 % at 5 units, +4 % at 7, +5 % at 8;
        // profiles/r06k_fifo_by_units.log -- until round 6 the rule chained every run of fewer than 8 units)
        const int n_units = (int)sizes.size();
        const int fifo = fifo_cfg_ >= 0 ? fifo_cfg_ : ((cfg_.precision != OMNI_PREC_F16 || n_units <= (int)lanes_.size()) ? 1 : 0);
        last_fifo_ = fifo;
        const size_t img = (size_t)cfg_.width * cfg_.height;
        // key frame f of the call: its block (pool entry or the tail), the block's frame count, its index inside
        auto block_of = [&](int f, const uint8_t*& base, int& cnt, int& idx) {
            const int e = f / MB;
            if (e < full) { base = pool[(first_slot + e) % n_pool]; cnt = MB; idx = f - e * MB; }
            else { base = tail; cnt = rem; idx = f - full * MB; }
        };
        std::vector<const uint8_t*> up, down;
        std::vector<int> upn, downn;
        int f0 = 0;
        for (int s = 0; s < n_units; ++s) {
            const int m = sizes[s];
            Lane* lane = (recut || s < full) ? lanes_[s % lanes_.size()].get() : tail_lane;
            if (pend_lane_ == lane) check(omni_shard_rows_consumed(shard_), "omni_shard_rows_consumed");      // its row buffer is the exchange's input
            lane->t_enqueue = std::chrono::steady_clock::now();
            omni_trace_push("host: unit enqueue (upload + launches)");
            lane->meta.clear();
            const int want = recut ? m : lane->mb;
            if (lane->cur != want) { lane->cam.set_active(nd * want); lane->cur = want; }     // (the streaming intake or a recut run may have left another size)
            chain(lane, fifo);
            if (recut) {
                up.clear(); down.clear(); upn.clear(); downn.clear();
                for (int f = f0; f < f0 + m;) {                                  // maximal runs of frames inside one block
                    const uint8_t* base; int cnt, idx;
                    block_of(f, base, cnt, idx);
                    const int take = std::min(cnt - idx, f0 + m - f);
                    up.push_back(base + (size_t)idx * nd * img); upn.push_back(take * nd);
                    if (!cfg_.mono()) { down.push_back(base + ((size_t)cnt + idx) * nd * img); downn.push_back(take * nd); }
                    f += take;
                }
                lane->cam.enqueue_host_parts(up, upn, down, downn, !cfg_.mono());
            } else {
                const uint8_t* src = s < full ? pool[(first_slot + s) % n_pool] : tail;
                if (from_host) lane->cam.enqueue_host(src, cfg_.width, !cfg_.mono());       // loop_cam.cpp:536: only STEREO_FISHEYE blanks rows
                else lane->cam.enqueue_dev(src, cfg_.width, !cfg_.mono());
            }
            host_ms_[0] += since(lane->t_enqueue);
            omni_trace_pop();
            pending.emplace_back(lane, first_msg_id + f0);
            f0 += m;
            if (pending.size() >= lanes_.size()) { hits += finish_timed(*pending.front().first, pending.front().second); pending.pop_front(); }
        }
        while (!pending.empty()) { hits += finish_timed(*pending.front().first, pending.front().second); pending.pop_front(); }
        hits += collect_exchange();
        hits += collect_detector();
        drain_geometry();
        return hits;
    }
    // the sizes of the units a run of n key frames is cut into (each <= MB, sum = n); plan 1: ceil(n / MB) units of equal size (+- 1, the larger ones first);
    // plan 2: the same with half a unit in front, so that the first kernels start behind half an upload
    static std::vector<int> plan_units(int n, int MB, int plan) {
        std::vector<int> out;
        if (n <= 0) return out;
        int head = 0;
        if (plan == 2 && n > MB) { head = std::max(1, MB / 2); out.push_back(head); n -= head; }
        const int k = (n + MB - 1) / MB;
        for (int i = 0; i < k; ++i) out.push_back(n / k + (i < n % k ? 1 : 0));
        return out;
    }

    // latency of every micro-batch processed so far: from the start of its upload to the end of its detector / geometry step (milliseconds;
    // every key frame of a micro-batch shares it).  The reference is batch-1 and serial (tensorrt_generic.cpp:58-75): a key frame there waits for
    // nothing but its own 12 engine calls; here it waits for its micro-batch and for the micro-batches in flight before it.
    const std::vector<double>& latencies_ms() const { return latencies_ms_; }
    void clear_latencies() { latencies_ms_.clear(); }

    void sync() { for (auto& l : lanes_) l->sync(); for (auto& t : tail_lanes_) t.second->sync(); check(omni_ctx_sync(index_ctx_.get()), "sync"); }

    // one key frame as SwarmLoop::VIOKF_callback hands it on (swarm_loop.cpp:140-170): the flattened views -- images[0..dirs) = up cameras, images[dirs..2*dirs)
    // = down cameras (PINHOLE_DEPTH: the one gray image), each height rows of `stride` bytes -- the key-frame id and stamp (StereoFrame::keyframe_id,
    // ::stamp), VIO's pose, and prevent_adding_db (:156: a non-key frame that moved less than min_movement_keyframe is matched but not added);
    // depth: PINHOLE_DEPTH's 16-bit depth image (rows packed), borrowed until the key frame's unit is finished
    struct KeyframeIn {
        const uint8_t* const* images = nullptr; int stride = 0;
        int64_t msg_id = 0; double stamp = 0; PoseMsg pose_drone; bool prevent_adding_db = false;
        const uint16_t* depth = nullptr;
    };
private:
    struct SlotMeta { int64_t msg_id = 0; double stamp = 0; PoseMsg pose; bool prevent_adding_db = false; const uint16_t* depth = nullptr; };
    struct Lane;
public:
    // The streaming intake (what a ROS callback calls per key frame; KeyframeIntake::extract in keyframe_intake.hpp): the images are packed into the
    // open micro-batch's pinned block and the call returns; the `microbatch`-th key frame sends the unit to the GPU (one upload, SuperPoint,
    // MobileNetVLAD, up/down match, one download) and, when `pipelines` units are in flight, finishes the oldest (detector, geometry).  Returns the
    // loop candidates found by the units this call finished.  The reference handles one key frame at a time, synchronously (tensorrt_generic.cpp:58-75).
    int push_keyframe(const KeyframeIn& k) {
        std::lock_guard<std::mutex> lk(intake_mu_);
        if (shard_) throw std::runtime_error("push_keyframe: the sharded database is driven through run()");
        if (!k.images || k.stride < cfg_.width) throw std::invalid_argument("push_keyframe: no images / a row stride below the image width");
        const int MB = cfg_.microbatch, nd = cfg_.dirs(), cams = cfg_.mono() ? 1 : 2;
        for (int i = 0; i < cams * nd; ++i) if (!k.images[i]) throw std::invalid_argument("push_keyframe: a null image pointer");
        const size_t img = (size_t)cfg_.width * cfg_.height;
        int hits = carried_hits_; carried_hits_ = 0;
        // whatever throws below: the candidates counted so far are not lost -- they go back into carried_hits_ and the next call returns them
        struct Carry { int& hits; int& carried; bool ok = false; ~Carry() { if (!ok) carried += hits; } } carry{hits, carried_hits_};
        hits += reap_ready();                                   // units the GPU has finished meanwhile (non-blocking): "idle" below is then the truth
        if (!open_) {
            // round robin over the lanes: with fewer than `pipelines` units in flight the next lane is free (a poll() may have sent a partial unit without
            // finishing the oldest -- it never waits --, so that is made sure of here)
            while (stream_pending_.size() >= lanes_.size()) { Lane* l = stream_pending_.front(); stream_pending_.pop_front(); hits += finish_timed(*l, 0); }
            Lane* l = lanes_[next_lane_++ % lanes_.size()].get();
            l->meta.clear();
            if (!l->stage) {
                l->stage_bytes = (size_t)cams * nd * MB * img;
                l->stage = static_cast<uint8_t*>(omni_host_alloc(l->stage_bytes));
                if (!l->stage) throw std::runtime_error(std::string("omni_host_alloc: ") + omni_last_error());
            }
            l->t_enqueue = std::chrono::steady_clock::now();
            open_ = l;
        }
        const int m = (int)open_->meta.size();
        if (m >= MB) throw std::logic_error("push_keyframe: the open micro-batch is already full");      // (cannot happen: a full unit leaves below, whatever throws)
        for (int c = 0; c < cams; ++c)
            for (int d = 0; d < nd; ++d) {
                uint8_t* dst = open_->stage + ((size_t)c * nd * MB + (size_t)nd * m + d) * img;
                const uint8_t* src = k.images[c * nd + d];
                for (int y = 0; y < cfg_.height; ++y) std::memcpy(dst + (size_t)y * cfg_.width, src + (size_t)y * k.stride, (size_t)cfg_.width);
            }
        SlotMeta sm; sm.msg_id = k.msg_id; sm.stamp = k.stamp; sm.pose = k.pose_drone; sm.prevent_adding_db = k.prevent_adding_db; sm.depth = k.depth;
        open_->meta.push_back(sm);
        const bool full = (int)open_->meta.size() == MB;
        const bool idle = cfg_.dispatch_when_idle && stream_pending_.empty();
        const bool aged = cfg_.max_wait_ms >= 0 && since(open_->t_enqueue) >= cfg_.max_wait_ms;
        if (full || idle || aged) {
            dispatch_open();
            while (stream_pending_.size() >= lanes_.size()) { Lane* l = stream_pending_.front(); stream_pending_.pop_front(); hits += finish_timed(*l, 0); }
        }
        carry.ok = true;
        return hits;
    }
    // The latency bound of the streaming intake, to be called from a timer (or after every push): never waits for a CNN unit.  (1) a partly filled
    // micro-batch older than Config::max_wait_ms leaves for the GPU as it is; (2) every unit the GPU has finished goes through the detector; (3) when no
    // unit is in flight any more, the detector step enqueued last (reported one unit later otherwise: OMNI_DETECTOR_ASYNC) and its geometry are
    // collected -- a short wait for searches that run on an otherwise idle GPU.  Returns the loop candidates found.
    int poll() {
        std::lock_guard<std::mutex> lk(intake_mu_);
        if (shard_) return 0;
        int hits = carried_hits_; carried_hits_ = 0;
        hits += reap_ready();
        if (open_ && !open_->meta.empty() && cfg_.max_wait_ms >= 0 && since(open_->t_enqueue) >= cfg_.max_wait_ms) dispatch_open();
        return hits;
    }
    // everything pushed so far through the detector (and the geometry stage): a partial micro-batch runs as its own, smaller unit
    int flush() {
        std::lock_guard<std::mutex> lk(intake_mu_);
        int hits = carried_hits_; carried_hits_ = 0;
        std::exception_ptr first;
        try { if (open_ && !open_->meta.empty()) dispatch_open(); } catch (...) { first = std::current_exception(); }
        open_ = nullptr;
        while (!stream_pending_.empty()) {
            Lane* l = stream_pending_.front(); stream_pending_.pop_front();
            try { hits += finish_timed(*l, 0); } catch (...) { if (!first) first = std::current_exception(); }      // the other units are still finished
        }
        try { hits += collect_detector(); drain_geometry(); } catch (...) { if (!first) first = std::current_exception(); }
        if (first) std::rethrow_exception(first);
        return hits;
    }
    // the streaming intake's settings after construction (what the C entry point omni_pipeline_set_latency sets)
    void set_latency(double max_wait_ms, bool dispatch_when_idle) { std::lock_guard<std::mutex> lk(intake_mu_); cfg_.max_wait_ms = max_wait_ms; cfg_.dispatch_when_idle = dispatch_when_idle; }

    // creates (once) the smaller unit a run of n_keyframes needs for its trailing n_keyframes % microbatch key frames, so that the first
    // timed run does not pay for it
    Lane* prepare(int n_keyframes) {
        const int rem = n_keyframes % cfg_.microbatch;
        if (!rem) return nullptr;
        auto it = tail_lanes_.find(rem);
        if (it == tail_lanes_.end()) it = tail_lanes_.emplace(rem, std::make_unique<Lane>(cfg_, rem)).first;
        return it->second.get();
    }

private:
    struct Lane {                                      // one micro-batch in flight: its own streams, networks and result block
        Lane(const Config& c, int mb_)
            : mb(mb_), sp_ctx(c.device), vlad_ctx(c.device),
              sp(sp_ctx, c.sp_weights, c.pca_comp, c.pca_mean, c.width, c.height, c.thres, c.max_num, false, c.precision, (c.mono() ? 1 : 8) * mb_),
              one_stream(cfg_int("OMNI_PIPELINE_ONE_STREAM") != 0),
              vlad(one_stream ? sp_ctx : vlad_ctx, c.vlad_weights, c.width, c.height, false, c.dirs() * mb_),
              cam(sp_ctx, sp, one_stream ? sp_ctx : vlad_ctx, vlad, c.dirs() * mb_, c.max_num, c.width, c.height, c.mono()) {
            check(omni_vlad_dev_output(vlad.handle(), &rows_dev), "omni_vlad_dev_output");
            cur = mb_;
        }
        void sync() { check(omni_ctx_sync(sp_ctx.get()), "sync"); check(omni_ctx_sync(vlad_ctx.get()), "sync"); }
        std::chrono::steady_clock::time_point t_enqueue;
        int mb;                                        // key frames the unit was created for
        int cur = 0;                                   // ... and of the unit in flight (a partly filled micro-batch of the streaming intake: omni_cam_set_active)
        std::vector<SlotMeta> meta;                    // streaming intake (push_keyframe): what each key frame of the unit came with; empty = run()'s numbering
        uint8_t* stage = nullptr;                      // pinned block the streaming intake packs the unit's images into
        size_t stage_bytes = 0;
        ~Lane() { if (stage) omni_host_free(stage); }
        Context sp_ctx, vlad_ctx;
        Swarm::SuperPointHIP sp;
        bool one_stream;                               // MobileNetVLAD behind SuperPoint on ONE stream (OMNI_PIPELINE_ONE_STREAM) instead of next to it on its own
        Context& vlad_stream_ctx() { return one_stream ? sp_ctx : vlad_ctx; }
        Swarm::MobileNetVLADHIP vlad;
        LoopCamHIP cam;
        const float* rows_dev = nullptr;
    };

    // The open unit of the streaming intake leaves for the GPU, full or not (a partial one: the down cameras' block moves up behind the up cameras' and the
    // unit runs with fewer directions, omni_cam_set_active -- no second set of networks).  Exception-safe: open_ is cleared FIRST, so whatever throws
    // below the next push starts a fresh unit (a failed unit's key frames are dropped with it: the caller got the exception).
    void dispatch_open() {
        Lane* u = open_;
        open_ = nullptr;
        const int MB = cfg_.microbatch, nd = cfg_.dirs(), rem = (int)u->meta.size();
        const size_t img = (size_t)cfg_.width * cfg_.height;
        try {
            if (rem < MB && !cfg_.mono()) std::memmove(u->stage + (size_t)nd * rem * img, u->stage + (size_t)nd * MB * img, (size_t)nd * rem * img);
            if (u->cur != rem) { u->cam.set_active(nd * rem); u->cur = rem; }
            chain(u, fifo_cfg_ >= 0 ? fifo_cfg_ : (cfg_.precision == OMNI_PREC_F16 ? 0 : 1));
            u->cam.enqueue_host(u->stage, cfg_.width, !cfg_.mono());
        } catch (...) { u->meta.clear(); throw; }
        stream_pending_.push_back(u);
    }
    // finishes, oldest first, the units the GPU is done with (never waits for a CNN unit); with nothing left in flight also the detector step enqueued
    // last and its geometry
    int reap_ready() {
        int hits = 0;
        while (!stream_pending_.empty() && stream_pending_.front()->cam.ready()) { Lane* l = stream_pending_.front(); stream_pending_.pop_front(); hits += finish_timed(*l, 0); }
        if (stream_pending_.empty() && det_pending_.active) { hits += collect_detector(); drain_geometry(); }
        return hits;
    }

    // waits for the exchange in flight (if any) and applies the reference's rule (loop_detector.cpp:232: recency + threshold) on GLOBAL row ids
    int collect_exchange() {
        if (!pend_lane_) return 0;
        const int k = LoopDetectorCore::SEARCH_NEAREST_NUM + cfg_.match_index_dist, mb = pend_mb_;
        D_.resize((size_t)mb * k); I_.resize((size_t)mb * k);
        pend_lane_ = nullptr;
        check(omni_shard_step_wait(shard_, D_.data(), I_.data()), "omni_shard_step_wait");
        { float a = 0, b = 0; if (omni_shard_last_exchange_us(shard_, &a, &b) == OMNI_OK) exchange_us_.push_back({a, b}); }
        int hits = 0;
        for (int m = 0; m < mb; ++m) {
            const int64_t nt = pend_base_ + (int64_t)(m + 1) * world_ * cfg_.dirs();          // ntotal as of this key frame's step
            for (int j = 0; j < k; ++j) {
                const int64_t id = I_[(size_t)m * k + j];
                if (id >= 0 && id <= nt - cfg_.match_index_dist && D_[(size_t)m * k + j] > cfg_.inner_product_thres) { ++hits; break; }
            }
        }
        return hits;
    }
    int finish_timed(Lane& lane, int64_t first_id) {
        const int hits = finish(lane, first_id);
        if (shard_) latencies_ms_.push_back(since(lane.t_enqueue));       // (the local path records a unit's latency when its detector step is collected)
        return hits;
    }
    // The detector step of the unit whose batch was begun last: LoopDetectorCore::end_images_batch waits for the copy of its result lists -- enqueued one
    // unit ago, so it ran behind the CNN kernels of the unit that is in flight and the host does not wait here -- and replays the rules; then the unit's
    // geometry.  (The host used to wait for the searches right where it enqueued them: on a GPU kept busy by the next unit's kernels that round trip took
    // 2.9 ms of a 3.8 ms cycle, the next unit was enqueued late and the GPU idled 15 % of the time -- host_times(), DESIGN.md.)
    int collect_detector() {
        if (!det_pending_.active) return 0;
        auto t_a = std::chrono::steady_clock::now();
        omni_trace_push("host: detector collect (result lists, rules) + geometry hand-over");
        struct Pop { ~Pop() { omni_trace_pop(); } } pop_at_exit;
        int hits = 0, fi = 0;
        det_pending_.active = false;                     // first: if end_images_batch() throws (its rollback has dropped the batch), the next unit starts clean
        for (auto& c : det_.end_images_batch()) {
            if (c.found) { ++hits; candidates_.push_back({det_pending_.ids[(size_t)fi], c.old_msg_id, c.direction_new, c.direction_old}); }
            ++fi;
        }
        host_ms_[3] += since(t_a);
        t_a = std::chrono::steady_clock::now();
        // the previous micro-batch's tasks ran meanwhile; this one's run until the next is collected.  The tasks reference this micro-batch's frames:
        // those moved into the database live there (std::map: stable), the others are handed over
        drain_geometry();
        const bool left = start_geometry(&det_.held_frames(), &det_pending_.t_enqueue);
        if (!async_geometry_) drain_geometry();
        det_.held_frames().clear();
        if (!left) latencies_ms_.push_back(since(det_pending_.t_enqueue));        // (otherwise drain_geometry() records it, when the micro-batch's last edge is in)
        host_ms_[4] += since(t_a);
        return hits;
    }
    // the micro-batch's key frames reach the detector in order, as one batch; rows and queries are taken from MobileNetVLAD's output
    // buffer in HBM ([4*mb][4096], key-frame major) -- wait() has synchronised with the MobileNetVLAD stream
    int finish(Lane& lane, int64_t first_id) {
        auto t_a = std::chrono::steady_clock::now();
        omni_trace_push("host: wait for the unit's GPU work");
        const omni_cam_result r = lane.cam.wait();
        omni_trace_pop();
        host_ms_[1] += since(t_a); ++host_units_;
        t_a = std::chrono::steady_clock::now();
        struct Range { explicit Range(const char* n) { omni_trace_push(n); } ~Range() { omni_trace_pop(); } } range_rest("host: messages + detector step of the unit");
        if (shard_) {
            // the exchange of THIS micro-batch is enqueued (two collectives, the scan, the copy of the lists: no host wait) and its results are
            // collected when the NEXT micro-batch gets here (or at the end of run()): meanwhile the host enqueues the next CNN unit
            int hits = collect_exchange();
            const int k = LoopDetectorCore::SEARCH_NEAREST_NUM + cfg_.match_index_dist;
            check(omni_shard_step_enqueue(shard_, lane.cur, cfg_.dirs(), lane.rows_dev, cfg_.mono() ? 0 : 1, k), "omni_shard_step_enqueue");
            pend_lane_ = &lane; pend_mb_ = lane.cur; pend_base_ = omni_shard_ntotal(shard_);
            return hits;
        }
        const int n = r.n_dirs, M = r.max_num, D = r.desc_dim, nd = cfg_.dirs();
        frames_.resize(lane.cur);
        const int G = r.global_dim;
        // Without the geometry stage the detector's plan (which rows are appended, which searches run: LoopDetectorCore::begin_batch) reads counts and ids only:
        // the messages are first built LIGHT (landmark_num, stamps), the detector step is enqueued, and the heavy part -- 75 KB of key points, descriptors and
        // lifted points per image -- is copied into the frames the detector now holds while its searches run on the GPU.  With the geometry stage the landmarks
        // can change landmark_num (generate_gray_depth_image_descriptor drops an image below ACCEPT_MIN_3D_PTS): everything is built first, as before.
        const bool defer_heavy = !cfg_.geometry;
        auto heavy = [&](ImageDescriptor& im, int i) {
            fill_image_descriptor(im, r.kps_xy + (size_t)i * M * 2, r.n_kps[i], r.desc + (size_t)i * M * D, D, r.global_desc + (size_t)i * G, G, lift64_);
        };
        for (int m = 0; m < lane.cur; ++m) {
            FisheyeFrameDescriptor& f = frames_[m];
            const bool streamed = !lane.meta.empty();
            const int64_t kf_id = streamed ? lane.meta[m].msg_id : first_id + m;
            const double stamp = streamed ? lane.meta[m].stamp : (double)kf_id;
            const PoseMsg pose = streamed ? lane.meta[m].pose
                                          : (kf_id >= pose_base_ && kf_id < pose_base_ + (int64_t)poses_.size()) ? poses_[(size_t)(kf_id - pose_base_)] : PoseMsg{};
            f.prevent_adding_db = streamed && lane.meta[m].prevent_adding_db;
            f.images.resize(nd);
            for (int d = 0; d < nd; ++d) {
                const int i = nd * m + d;                                           // image i of the up cameras
                ImageDescriptor& im = f.images[d];
                // extractor_img_desc_deepnet (loop_cam.cpp:525-585) + the stamps of generate_stereo_image_descriptor (:362-374)
                if (defer_heavy) im.landmark_num = r.n_kps[i]; else heavy(im, i);
                stamp_image_descriptor(im, stamp, cfg_.self_id, to_msg(view_extrinsic(d, true)), pose, kf_id);
                if (cfg_.geometry && cfg_.mono()) {
                    // generate_gray_depth_image_descriptor's landmarks (loop_cam.cpp:260-304): read from the depth image under each key point
                    const bool have = streamed ? lane.meta[m].depth != nullptr : (depth_ && kf_id >= depth_base_ && kf_id < depth_base_ + depth_n_);
                    if (have) fill_depth_landmarks(im, streamed ? lane.meta[m].depth : depth_ + (size_t)(kf_id - depth_base_) * cfg_.width * cfg_.height, cfg_.width, cfg_.width, cfg_.height, cfg_.depth_near,
                                                   cfg_.depth_far, cfg_.accept_min_3d_pts, lift64_);
                } else if (cfg_.geometry) {
                    // the stereo half of generate_stereo_image_descriptor (loop_cam.cpp:341-454): the down image of this direction, triangulation
                    // (one task per direction on the geometry pool: ~170 SVD triangulations each; joined before the frames reach the detector)
                    if (downs_.size() < (size_t)4 * lane.cur) downs_.resize((size_t)4 * lane.cur);
                    ImageDescriptor& down = downs_[(size_t)i];
                    down = ImageDescriptor{};
                    const int j = n + i;
                    fill_image_descriptor(down, r.kps_xy + (size_t)j * M * 2, r.n_kps[j], nullptr, D, nullptr, 0, lift64_);     // (its descriptors were matched on the device)
                    stamp_image_descriptor(down, stamp, cfg_.self_id, to_msg(view_extrinsic(d, false)), pose, kf_id);
                    auto tri = [this, &im, &down, r, i, M] {
                        // the triangulation lifts the pixels again, in double (loop_cam.cpp:403-407); the message keeps the float points
                        fill_stereo_landmarks(im, down, r.match_up + (size_t)i * M, r.match_down + (size_t)i * M, r.n_matches[i], cfg_.triangle_thres, cfg_.accept_min_3d_pts,
                                              &lift64_);
                    };
                    if (pool_) stereo_tasks_.push_back(pool_->submit(tri)); else tri();
                }
            }
            finish_frame_descriptor(f, stamp, kf_id, pose, cfg_.self_id);             // on_flattened_images (loop_cam.cpp:178-217)
        }
        {   // join the triangulation tasks; if one throws, the others are still joined (they reference this micro-batch's frames) and the list is
            // emptied before the exception leaves: the next finish() must not meet futures that were already consumed
            std::exception_ptr first;
            for (auto& t : stereo_tasks_) {
                try { t.get(); } catch (...) { if (!first) first = std::current_exception(); }
            }
            stereo_tasks_.clear();
            if (first) std::rethrow_exception(first);
        }
        host_ms_[2] += since(t_a);
        t_a = std::chrono::steady_clock::now();
        int hits = collect_detector();                                          // the unit before this one
        t_a = std::chrono::steady_clock::now();
        det_pending_.ids.resize((size_t)lane.cur);
        for (int m = 0; m < lane.cur; ++m) det_pending_.ids[(size_t)m] = lane.meta.empty() ? first_id + m : lane.meta[(size_t)m].msg_id;
        det_pending_.t_enqueue = lane.t_enqueue;
        det_.begin_images_batch(std::move(frames_), lane.rows_dev);             // appends + searches + the copy of the lists: enqueued, not waited for
        det_pending_.active = true;
        frames_.clear();
        // the appends and the query gather read MobileNetVLAD's output buffer: the lane's next unit must not overwrite it before they have
        check(omni_ctx_order_after(lane.vlad_stream_ctx().get(), index_ctx_.get()), "omni_ctx_order_after");
        host_ms_[3] += since(t_a);
        if (defer_heavy) {                                                      // the messages' contents, into the frames the detector holds, under its GPU work
            t_a = std::chrono::steady_clock::now();
            std::vector<FisheyeFrameDescriptor>& held = det_.held_frames();
            // key frames in stripes over the helper threads and this one (OMNI_MESSAGE_THREADS): each message is written by one thread, the result block is
            // only read; what the stripes share is the allocator.  Matters where this is exposed: behind the LAST unit of a run() call
            const int n_kf = std::min(lane.cur, (int)held.size()), parts = msg_pool_ ? std::min(n_kf, msg_pool_->size() + 1) : 1;
            auto stripe = [&, n_kf, parts](int p) {
                for (int m = p; m < n_kf; m += parts)
                    for (int d = 0; d < nd && d < (int)held[(size_t)m].images.size(); ++d) heavy(held[(size_t)m].images[(size_t)d], nd * m + d);
            };
            std::vector<std::future<void>> helpers;
            for (int p = 1; p < parts; ++p) helpers.push_back(msg_pool_->submit([&stripe, p] { stripe(p); }));
            std::exception_ptr first;
            try { stripe(0); } catch (...) { first = std::current_exception(); }
            for (auto& h : helpers) { try { h.get(); } catch (...) { if (!first) first = std::current_exception(); } }      // (every helper is joined: they reference this frame's locals)
            if (first) std::rethrow_exception(first);
            host_ms_[2] += since(t_a);
        }
        if (!async_detector_) hits += collect_detector();                       // OMNI_DETECTOR_ASYNC=0: wait for it here, as before (A/B)
        return hits;
    }

    Config cfg_;
    // cam->liftProjective of the flattened (pinhole) views, then the division by z: in double, as the reference's camera model
    const std::function<geom::Vec2(const Point2f&)> lift64_ = [this](const Point2f& p) { return geom::Vec2{((double)p.x - cfg_.cx) / cfg_.fx, ((double)p.y - cfg_.cy) / cfg_.fy}; };
    Context index_ctx_;
    LoopDetectorCore det_;
    BFMatcherL2X bf_{index_ctx_};
    LoopGeometry geo_;
    std::vector<double> latencies_ms_;
    struct GeoBatch {                           // the geometry tasks of one micro-batch, in candidate order
        std::vector<std::future<std::pair<bool, LoopEdge>>> futs;
        std::vector<char> self_self;            // per task: a candidate between two frames of the self drone (the detector did not get its verdict)
        std::vector<FisheyeFrameDescriptor> held;
        std::chrono::steady_clock::time_point t_enqueue;
        bool timed = false;
    };
    std::deque<GeoBatch> geo_inflight_;         // (declared before the pool: outlives its threads)
    struct DetPending { bool active = false; std::vector<int64_t> ids; std::chrono::steady_clock::time_point t_enqueue; } det_pending_;
    bool async_detector_ = [] { int v = 1; check(omni_config_value("OMNI_DETECTOR_ASYNC", &v), "omni_config_value"); return v != 0; }();
    bool async_geometry_ = [] { int v = 1; check(omni_config_value("OMNI_GEOMETRY_ASYNC", &v), "omni_config_value"); return v != 0; }();
    std::unique_ptr<TaskPool> pool_;
    std::unique_ptr<TaskPool> msg_pool_ = [] { const int n = cfg_int("OMNI_MESSAGE_THREADS"); return n > 0 ? std::make_unique<TaskPool>(n) : nullptr; }();
    std::vector<ImageDescriptor> downs_;        // the down-camera halves of the micro-batch being finished
    std::vector<std::future<void>> stereo_tasks_;
    struct Deferred { const FisheyeFrameDescriptor *a, *b; int da, db; bool im; };
    std::vector<Deferred> deferred_;            // candidates waiting for collect_geometry(): references into the database / the current micro-batch
    std::vector<LoopEdge> edges_;
    std::vector<Candidate> candidates_;
    std::vector<PoseMsg> poses_;
    int64_t pose_base_ = 0;
    const uint16_t* depth_ = nullptr;           // PINHOLE_DEPTH: borrowed depth images (set_depth)
    int64_t depth_base_ = 0, depth_n_ = 0;
    int geometry_calls_ = 0;
    std::vector<std::unique_ptr<Lane>> lanes_;
    std::map<int, std::unique_ptr<Lane>> tail_lanes_;
    double host_ms_[5] = {0, 0, 0, 0, 0};
    int host_units_ = 0;
    // units in flight run oldest first (omni_cam_order_after): the unit about to be enqueued starts behind the convolution stack of the one enqueued last
    Lane* last_enqueued_ = nullptr;
    int unit_plan_ = cfg_int("OMNI_PIPELINE_UNIT_PLAN");    // how run() cuts a run that is not a whole number of micro-batches (plan_units; 0: the blocks' own cut)
    int fifo_cfg_ = cfg_int("OMNI_PIPELINE_FIFO");          // >= 0: as asked; -1: run() and the streaming intake decide (see run())
    int last_fifo_ = 0;                                      // what the last run() used (reported by the bench line)
    void chain(Lane* lane, int fifo_streams) {
        if (fifo_streams > 0 && last_enqueued_ && last_enqueued_ != lane) lane->cam.order_after(last_enqueued_->cam, fifo_streams);
        last_enqueued_ = lane;
    }
    std::mutex intake_mu_;                                   // push_keyframe / poll / flush may come from different threads (a callback and a timer)
    static int cfg_int(const char* name) { int v = 0; check(omni_config_value(name, &v), "omni_config_value"); return v; }
    static double since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); }
    Lane* open_ = nullptr;                      // streaming intake: the micro-batch being filled
    std::deque<Lane*> stream_pending_;          // ... and the units in flight, oldest first
    size_t next_lane_ = 0;
    int carried_hits_ = 0;
    std::vector<FisheyeFrameDescriptor> frames_;
    std::unique_ptr<IndexFlatIP> shard_index_;
    omni_shard* shard_ = nullptr;
    int world_ = 1;
    Lane* pend_lane_ = nullptr;                 // the micro-batch whose exchange is in flight
    int pend_mb_ = 0;
    int64_t pend_base_ = 0;
    std::vector<float> D_;
    std::vector<int64_t> I_;
    std::vector<std::pair<float, float>> exchange_us_;
};

}  // namespace omni
