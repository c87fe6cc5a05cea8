#!/usr/bin/env python
"""bench.py -- keyframes/sec of the swarm_loop hot path on MI355X + p50 loop-match latency.

A "step" is ONE fisheye key frame through the whole hot path (BASELINE.json configs[1]):
    upload of its 8 images (pinned host -> HBM, inside the timed region; the reference does one H2D per engine call, tensorrt_generic.cpp:58-75)
    8 SuperPoint images (4 directions x up/down, 600x480 u8, fisheye-masked) + 4 MobileNetVLAD images
    + 4 up<->down descriptor cross-check matches                    (LoopCam::on_flattened_images, loop_cam.cpp:178-229)
    + <=4 row inserts into the 4096-d global index and the top-k inner-product query with the recency/threshold rule
                                                                    (LoopDetector::on_image_recv, loop_detector.cpp:11-137)
    + the D2H copy of every result (key points, descriptors, global descriptors, match lists, query result).
The timed host loop is C++ (omni-swarm_amd/host/keyframe_pipeline.hpp, the flow of SwarmLoop::VIOKF_callback swarm_loop.cpp:140-170);
Python prepares the synthetic inputs, starts the run and brackets it with barrier + synchronize.  EXACTLY --steps key frames are
processed per timed region (a trailing partial micro-batch runs as its own smaller unit); when one region is shorter than --min-time
it is repeated and the MEDIAN region is reported (`repeats`, `ms_per_step_minmax` say so).

One process per GPU.  N > 1 (launched by torch.distributed.run): key frames are data parallel, the global index is row-sharded across
ranks and every micro-batch has one exchange (all_gather of the new rows, all_gather of per-shard top-k).

Prints ONE JSON line on rank 0 (see the task contract) with the extra objects `roofline` (dominant kernel), `roofline_knn`,
`roofline_knn_batched`, `loop_match`, `db100k` (the same loop against a 100k-key-frame = 400k-row database, fp16 and fp32 rows),
`value_f32`, `python_host`, `parity` and `cpu_baseline`.
"""
import argparse
import json
import math
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the key-frame pipeline's five streams must not share a hardware queue (csrc/ctx.hip: libomni_hip.so asks for 8 when it is loaded; here as well, because
# torch may initialise the HIP runtime before the library is loaded -- the runtime reads the variable once)
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("OMNI_HW_QUEUES", "8") if os.environ.get("OMNI_HW_QUEUES", "8") != "0" else "4")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # multi-process GPU work on this driver: dmabuf IPC only (RCCL fails without it)
os.environ.setdefault("OMNI_SP_PROFILE_MASK", "1")      # omni_sp_profile: stage times with the fisheye mask on, as the key-frame pipeline runs the network

PEAK_F16_TFLOPS = 2500.0      # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF figure is 2:1 sparse)
PEAK_F32_TFLOPS = 157.3       # f32-input MFMA = vector peak
PEAK_HBM_GBS = 8000.0         # HBM3E spec peak (6.29 TB/s measured achievable)
SP_FLOP_PER_IMAGE = 48.85e9   # SURVEY.md 2.3 @600x480: the dense network
KF_IMAGES = 8                 # reference-faithful key frame: 8 SuperPoint + 4 MobileNetVLAD
# fp16 path: convDa (128 -> 256, 3x3) and convDb (256 -> 256, 1x1) run only at the <= 4 * max_num coarse cells around the key points instead of all
# 75 x 60 = 4500 (conv3x3_c128_sparse_kernel, convdb_sparse_kernel): FLOP actually executed per image
SP_DA_FLOP_PER_CELL, SP_DB_FLOP_PER_CELL, SP_CELLS = 2.0 * 9 * 128 * 256, 2.0 * 256 * 256, 4500


def sp_flop_executed(precision, max_num, conv_stages_only=False, left_out_flop=0.0):
    """FLOP per image actually executed (conv_stages_only: by the stages named conv*, i.e. without the sparse descriptor kernels, which run inside the
    post-processing stage; left_out_flop: the FLOP of the tiles a fisheye-masked pass leaves out of the tile walk -- the constant region of the mask,
    csrc/superpoint.hip sp_plan_mask_skip -- summed from the library's own per-stage figures, omni_sp_stage_flops x omni_sp_stage_tiles_left_out)."""
    sparse_db = precision == "f16" and os.environ.get("OMNI_SP_SPARSE_DESC", "1") != "0"
    # convDa at the key points only: the fp16 path (conv_c128_sparse) and, since round 4, OMNI_PREC_SPLIT (conv_split_c128_sparse); the exact-f32 path keeps
    # the dense convDa.  (The fp32 / split convDb is sparse too, but not part of SP_FLOP_PER_IMAGE's matrix-core figure in those modes.)
    sparse_da = precision in ("f16", "split") and os.environ.get("OMNI_SP_SPARSE_DESC", "1") != "0" and os.environ.get("OMNI_SP_SPARSE_DA", "1") != "0"
    cells = 0 if conv_stages_only else min(4 * max_num, SP_CELLS)
    f = SP_FLOP_PER_IMAGE - left_out_flop
    if sparse_db:
        f -= SP_DB_FLOP_PER_CELL * (SP_CELLS - cells)
    if sparse_da:
        f -= SP_DA_FLOP_PER_CELL * (SP_CELLS - cells)
    return f


def rocprof_trace(pattern):
    """Launch duration of a kernel from the newest committed rocprofv3 kernel trace (profiles/*_kernel_times.json, written by tools/rocprof_summary.py
    from the same bench command under `rocprofv3 --kernel-trace --stats`): the number the HIP-event time of this run must agree with."""
    import glob
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_times.json")))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        for name, v in t.get("kernels", {}).items():
            if pattern in name:
                return {"kernel": name[:100], "median_us": v["median_us"], "avg_us": v["avg_us"], "calls": v["calls"], "source": os.path.relpath(f, ROOT),
                        "command": t.get("label", "")}
    return None


def mfma_terms_of(precision):
    return 3 if precision == "split" else 1


def traffic_fields(key, enabled=True, images_per_launch=None):
    """{"traffic": HBM bytes per launch (a number, or None), "traffic_detail": where the number comes from}.  The committed PMC pass
    only applies to the launch shape it was collected at."""
    t = pmc_traffic(key) if enabled else None
    if t and images_per_launch is not None and t.get("images_per_launch") not in (None, images_per_launch):
        t = None
    return {"traffic": t["bytes_per_launch"] if t else None, "traffic_detail": t}


def pmc_traffic(key):
    """HBM bytes per launch of a kernel from the newest committed PMC pass (profiles/*_traffic.json, written from
    `tools/profile_gpu.sh` output: bench.py cannot collect PMC counters on itself).  None when no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    for f in reversed(files):                      # the newest pass that profiled this kernel (passes differ in precision / legs)
        try:
            t = json.load(open(f))
            if key in t:
                return {"bytes_per_launch": t[key]["bytes_per_launch"], "images_per_launch": t[key].get("images_per_launch"),
                        "source": os.path.relpath(f, ROOT)}
        except Exception:
            continue
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", choices=["f16", "f32", "split"], default="f16")
    ap.add_argument("--host", choices=["cpp", "python"], default="cpp", help="language of the timed host loop")
    ap.add_argument("--min-time", type=float, default=1.0, help="repeat the timed region of --steps key frames until this many seconds were timed; the median region is reported")
    ap.add_argument("--db-keyframes", type=int, default=1000, help="key frames pre-loaded in the index (x4 rows) for the throughput loop")
    ap.add_argument("--match-db-rows", type=int, default=100_000, help="index rows for the p50 loop-match measurement (node total)")
    ap.add_argument("--pipelines", type=int, default=0,
                    help="micro-batches (units) in flight per GPU, each with its own HIP streams; 0 = by measurement (round 4, profiles/r04*): 4 for f16 -- the small-grid "
                         "kernels of a unit's tail (NMS, descriptor sampling, the matcher, MobileNetVLAD's last blocks) leave CUs idle that the next units' kernels fill: "
                         "2215 -> 2476 kf/s -- and 2 for the fp32-class modes, whose time is all in CU-filling convolutions (no gain from depth, only latency)")
    ap.add_argument("--microbatch", type=int, default=8,
                    help="consecutive key frames enqueued together (one SuperPoint launch over 8*MB images, one MobileNetVLAD launch over 4*MB): "
                         "the low-resolution layers of both nets are launch/latency-bound at one key frame")
    ap.add_argument("--batched-rows", type=int, default=125_000,
                    help="fp16 rows per GPU for the 64-concurrent-query search measurement (configs[4]: 1 M key frames over 8 GPUs); 0 = skip")
    ap.add_argument("--big-db-keyframes", type=int, default=100_000, help="key frames (x4 rows) of the big-database throughput legs; 0 = skip")
    ap.add_argument("--big-db-steps", type=int, default=64, help="key frames per timed region of the big-database legs")
    ap.add_argument("--f32-steps", type=int, default=16, help="key frames of the f32-precision leg (value_f32); 0 = skip")
    ap.add_argument("--c5-rows", type=int, default=500_000, help="rows of the fp16 shard of the configs[4] leg (c5_shard); 0 = skip")
    ap.add_argument("--long-region-steps", type=int, default=200, help="when --steps is shorter than 8 micro-batches: also time regions of this many key frames (value_long_regions); 0 = skip")
    ap.add_argument("--parity-steps", type=int, default=64, help="key frames per region of the OMNI_PREC_SPLIT leg (value_parity: the mode that meets north_star's tolerance); 0 = skip")
    ap.add_argument("--geometry-steps", type=int, default=64, help="key frames of the leg with the geometric verification stage on (with_geometry); 0 = skip")
    ap.add_argument("--python-steps", type=int, default=64, help="key frames of the Python-host leg (python_host); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-keyframes", type=int, default=20, help="key frames timed on the host cores for cpu_baseline (after 3 warm-ups; the median is reported: SURVEY 8d asks for >= 20)")
    return ap.parse_args()


class RowFactory:
    """Unit-norm pseudo-random 4096-d rows, fast enough for 400k-row databases: one seeded Gaussian block, re-used with a different
    cyclic column shift and sign pattern per block (rows stay unit norm, distinct and mutually near-orthogonal)."""

    def __init__(self, seed, block=8192):
        rng = np.random.default_rng(seed)
        x = rng.standard_normal((block, 4096), dtype=np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        self.base, self.block, self.n = x, block, 0
        self.signs = (rng.integers(0, 2, (64, 4096)) * 2 - 1).astype(np.float32)

    def rows(self, n):
        out = []
        while n > 0:
            b = self.n // self.block
            o = self.n % self.block
            m = min(n, self.block - o)
            blk = self.base[o:o + m]
            if b:
                blk = np.roll(blk, 37 * b, axis=1) * self.signs[b % 64]
            out.append(blk)
            self.n += m
            n -= m
        return out[0] if len(out) == 1 else np.concatenate(out)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    import torch
    # OMNI_BENCH_ONE_GPU=1 (bring-up only): run the N > 1 code path with every rank on GPU 0 and gloo instead of RCCL, to
    # exercise the sharded step on a 1-GPU box; the result is NOT a scaling number
    one_gpu = os.environ.get("OMNI_BENCH_ONE_GPU", "0") == "1"
    if one_gpu:
        local_rank = 0
    coll_dev = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # torch.distributed is plumbing here (the 128-byte unique id, barriers, the MAX over ranks): a gloo group.  The data path's collectives are RCCL
        # inside libomni_hip.so (csrc/shard.hip: ncclAllGather over xGMI on the shard's own stream) -- ONE RCCL user per process, no second communicator
        # next to it.  OMNI_BENCH_TORCH_NCCL=1 puts torch's side on its nccl backend instead (A/B; both then share one librccl: the
        # line reports every librccl mapped into the process).
        torch_nccl = os.environ.get("OMNI_BENCH_TORCH_NCCL", "0") == "1" and not one_gpu
        if torch_nccl:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        coll_dev = torch.device("cuda", local_rank) if torch_nccl else torch.device("cpu")     # where torch's all_reduce / all_gather payloads live
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import omni_loader
    omni_loader.load()
    from omni_swarm_amd import capi, detector, frontend, pipeline, shard, synth, weights
    PREC = {"f16": capi.PREC_F16, "f32": capi.PREC_F32, "split": capi.PREC_SPLIT}
    prec = PREC[args.precision]
    W, H, MAXN = 600, 480, 200
    THRES = 0.02                      # superpoint_thres of the fisheye launch files (nodelet-sfisheye.launch)
    MATCH_INDEX_DIST, QUERY_THRES, INIT_THRES = 5, 0.3, 0.2   # launch values
    K_SEARCH = 5 + MATCH_INDEX_DIST
    MB = max(1, args.microbatch)
    # N > 1: C++ loop + RCCL inside the library.  The one-GPU bring-up (gloo for the torch side) stays on the Python loop unless OMNI_RCCL_LIB
    # points the library at the single-GPU stand-in for librccl (tests/stub_rccl): then the REAL N > 1 flow of this file -- the C++ loop on
    # omni_shard_*, the unique id broadcast, barriers, MAX over ranks -- runs with every rank on GPU 0: a rehearsal, not a scaling number
    cpp_host = args.host == "cpp" and (world == 1 or not one_gpu or bool(os.environ.get("OMNI_RCCL_LIB")))

    sp_w = weights.superpoint_synth_weights(0)
    comp, mean = synth.pca()
    vl_w = weights.mobilenetvlad_synth_weights()
    vl_specs = weights.mobilenetvlad_layer_specs()
    vl_shape = (weights.VLAD_N_CLUSTERS, weights.VLAD_FEAT_DIM, weights.VLAD_OUT_DIM)
    tmp = tempfile.TemporaryDirectory(prefix=f"omni_bench_r{rank}_")
    files = weights.write_pipeline_files(tmp.name, sp_w, comp, mean, vl_w, vl_specs, capi.VLAD_KINDS)
    ictx = capi.Context(local_rank)
    info = ictx.device_info()

    # ---- synthetic key-frame pool in PINNED HOST memory: every micro-batch is uploaded inside the timed region -------------------
    def micro_batch_images(first_seed, mb):
        """[up cameras of the mb frames (4 each) | down cameras of the mb frames] -- LoopCam's image order"""
        kf = [[synth.image_u8(first_seed + 8 * m + i, H, W) for i in range(KF_IMAGES)] for m in range(mb)]
        return np.stack([kf[m][i] for m in range(mb) for i in range(4)] + [kf[m][4 + i] for m in range(mb) for i in range(4)])

    POOL = 4
    img_cache = {}

    def pinned_batch(first_seed, mb):
        key = (first_seed, mb)
        if key not in img_cache:
            a = ictx.host_alloc((KF_IMAGES * mb, H, W), np.uint8)
            a[:] = micro_batch_images(first_seed, mb)
            img_cache[key] = a
        return img_cache[key]

    pool = [pinned_batch(1000 * rank + 8 * MB * p, MB) for p in range(POOL)]
    pool_ptrs = [a.ctypes.data for a in pool]

    def tail_for(steps):
        rem = steps % MB
        return (pinned_batch(1000 * rank + 8 * MB * POOL, rem), rem) if rem else (None, 0)

    rowgen = RowFactory(7 + rank)

    def barrier(*syncs):
        for s in syncs:
            s()
        ictx.sync()
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        if dist is not None:
            dist.barrier()

    def shard_uid():
        """ncclGetUniqueId of rank 0 for one omni_shard group, carried to the other ranks by torch.distributed (plumbing only)."""
        box = [capi.shard_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_all(x):
        """x of every rank, in rank order (a list of floats)"""
        if dist is None:
            return [float(x)]
        out = [torch.zeros(1, device=coll_dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(out, torch.tensor([x], device=coll_dev, dtype=torch.float64))
        return [float(o.item()) for o in out]

    own = []                                                   # this rank's own seconds per region of the last timed_regions() call

    def timed_regions(run, sync, steps, warmup, min_time, max_repeats=200):
        """run(n_steps) processes exactly n_steps key frames.  Warm-up, then regions of `steps` key frames, each bracketed by barrier +
        synchronize on both sides and reduced with MAX over ranks; repeated until min_time seconds are covered; returns the list."""
        if warmup > 0:
            run(warmup)
        dts = []
        own.clear()
        while True:
            barrier(sync)
            t0 = time.perf_counter()
            run(steps)
            sync()                                             # this rank's own work done: its own rate (per_rank below), before the others are waited for
            own.append(time.perf_counter() - t0)
            barrier(sync)
            dts.append(reduce_max(time.perf_counter() - t0))
            if sum(dts) >= min_time or len(dts) >= max_repeats:
                return dts

    def summarize(dts, steps):
        med = float(np.median(dts))
        return {"value": round(steps * world / med, 2), "ms_per_step": round(med / steps * 1e3, 4), "repeats": len(dts),
                "ms_per_step_minmax": [round(min(dts) / steps * 1e3, 4), round(max(dts) / steps * 1e3, 4)]}

    # ---- the C++ host loop (N = 1) ------------------------------------------------------------------------------------------------
    def room_pool():
        """pool of RENDERED key frames (synth.room_keyframe: a stereo rig in textured rooms, the down view = the up view 16 rows further): cycling
        it revisits every place again and again -- the geometry stage then has real loops to close (stereo landmarks, PnP, LoopEdges)"""
        out = []
        for p in range(POOL):
            key = ("room", p)
            if key not in img_cache:
                a = ictx.host_alloc((KF_IMAGES * MB, H, W), np.uint8)
                kf = [synth.room_keyframe(100 * rank + MB * p + m, H, W) for m in range(MB)]
                a[:] = np.stack([kf[m][i] for m in range(MB) for i in range(4)] + [kf[m][4 + i] for m in range(MB) for i in range(4)])
                img_cache[key] = a
            out.append(img_cache[key])
        return out

    def cpp_leg(precision, storage, db_rows, steps, warmup, min_time, geometry=False, ptrs=None, mb=None, pipelines=None):
        mb = mb or MB
        # ONE configuration: the library's defaults, whatever --steps is.  Units in flight: --pipelines, else 0 = the library's default
        # for the precision (KeyframePipeline::default_pipelines: 4 for fp16, 2 for the fp32-class modes); whether the units run oldest-first is the
        # library's own rule (KeyframePipeline::run: by precision and by how many units the call holds), not an environment variable set here
        pipelines = pipelines or args.pipelines or 0
        ptrs = ptrs or (pool_ptrs if mb == MB else [pinned_batch(1000 * rank + 50_000 + 8 * mb * p, mb).ctypes.data for p in range(2)])
        pl = pipeline.KeyframePipeline(local_rank, files["sp"], files["comp"], files["mean"], files["vlad"], W, H, THRES, MAXN, precision, mb,
                                       pipelines, storage, 1, QUERY_THRES, INIT_THRES, MATCH_INDEX_DIST, 30, 3, geometry=geometry)
        gen = RowFactory(7 + rank)
        if world > 1:
            pl.attach_shard(rank, world, shard_uid())
            pl.preload(gen.rows(db_rows // world))            # this rank's part of a db_rows-row database
        else:
            for s in range(0, db_rows, 32768):
                pl.preload(gen.rows(min(32768, db_rows - s)))
        assert mb == MB or (steps % mb == 0 and warmup % mb == 0)
        tail, _ = tail_for(steps) if mb == MB else (None, 0)
        wtail, _ = tail_for(warmup) if mb == MB else (None, 0)
        pl.prepare(steps)
        pl.prepare(warmup)
        state = {"id": 0, "slot": 0, "hits": 0, "calls": 0}

        def run(n):
            if state["calls"] == (1 if warmup > 0 else 0):
                pl.latencies_ms(reset=True)                    # the warm-up's micro-batches do not count
                pl.host_times(reset=True)
            state["calls"] += 1
            t = tail if n == steps else wtail
            state["hits"] += pl.run(n, state["id"], ptrs, state["slot"], None if t is None else t.ctypes.data, True)
            state["id"] += n
            state["slot"] += n // mb
        dts = timed_regions(run, pl.sync, steps, warmup, min_time)
        out = summarize(dts, steps)
        lat = pl.latencies_ms()
        out.update(db_rows_start=db_rows, db_rows_end=int(pl.db_rows), loop_candidates_found=state["hits"])
        out["host_ms_per_microbatch"] = pl.host_times()
        units, fifo = pl.units()
        out["pipelines"] = units
        out["units_oldest_first"] = f"{fifo} (" + ("OMNI_PIPELINE_FIFO" if os.environ.get("OMNI_PIPELINE_FIFO") not in (None, "-1") else
                                                    "library default: by precision and by the units a run() call holds") + ")"
        if world > 1:
            # per rank: its own rate over its own time per region (before the barrier), the database rows it holds, and the device time of the exchange's two
            # all-gathers (HIP events on the shard's stream inside libomni_hip.so)
            ex = pl.exchange_us()
            mine = steps / float(np.median(own)) if own else 0.0
            out["per_rank_keyframes_per_s"] = [round(v, 2) for v in gather_all(mine)]
            out["db_rows_per_gpu"] = int(pl.db_rows) // world            # global row g lives on rank g % world: every shard holds ntotal / world rows
            p50 = lambda a: float(np.median(a)) if len(a) else 0.0
            out["all_gather_us_p50"] = {"new_rows": [round(v, 1) for v in gather_all(p50(ex[:, 0]))], "topk_lists": [round(v, 1) for v in gather_all(p50(ex[:, 1]))],
                                        "exchange_units": int(len(ex)),
                                        "definition": "device time between the events that bracket each ncclAllGather on the shard's stream: the collective INCLUDING the wait for "
                                                      "the slowest rank to arrive (per rank; the minimum over the ranks is the closest to pure transfer time)",
                                        "bytes_per_rank": {"new_rows": mb * 4 * 4096 * 4, "topk_lists": mb * world * K_SEARCH * 12}}
        if len(lat):
            out["keyframe_latency_ms"] = {"p50": round(float(np.percentile(lat, 50)), 3), "p99": round(float(np.percentile(lat, 99)), 3),
                                          "micro_batches": int(len(lat)), "keyframes_in_flight": mb * pl.units()[0],
                                          "definition": "start of a micro-batch's upload -> its key frames' detector (+ geometry) step done; the reference is "
                                                        "batch-1 and serial (tensorrt_generic.cpp:58-75)"}
        if geometry:
            calls, edges = pl.geometry_stats()
            out.update(compute_loop_calls=calls, loop_edges=edges)
        pl.close()
        return out

    # ---- the Python host loop (N > 1, --host python, and the `python_host` comparison leg) -----------------------------------------
    class PythonLoop:
        def __init__(self, precision):
            self.ctxs = [capi.Context(local_rank) for _ in range(args.pipelines or 2)]
            self.prec = precision
            self.cams = [frontend.LoopCam(c, sp_w, comp, mean, vl_w, vl_specs, vl_shape, W, H, THRES, MAXN, precision, n_dirs=4 * MB) for c in self.ctxs]
            self.tail_cam = None
            self.hits, self.step = 0, 0
            if world == 1:
                self.det = detector.LoopDetector(ictx, self_id=1, inner_product_thres=QUERY_THRES, init_mode_product_thres=INIT_THRES,
                                                 match_index_dist=MATCH_INDEX_DIST, min_loop_num=30, min_direction_loop=3)
                base_rows = 4 * args.db_keyframes
                gen = RowFactory(7 + rank)
                for s in range(0, base_rows, 8192):
                    self.det.local_index.add(gen.rows(min(8192, base_rows - s)))
                for i in range(base_rows):      # bookkeeping for pre-loaded rows: frame ids / directions
                    self.det.imgid2fisheye[i] = -(i // 4) - 1
                    self.det.imgid2dir[i] = i % 4
                    self.det.fisheyeframe_database.setdefault(-(i // 4) - 1, detector.FisheyeFrameDescriptor(msg_id=-(i // 4) - 1))
                self.swarm = None
            else:
                self.det = None
                self.native = not one_gpu and args.host == "python"
                if self.native:   # the exchange inside libomni_hip.so: ncclAllGather on device buffers (csrc/shard.hip)
                    self.swarm = shard.NativeSwarmIndex(ictx, capi.IndexFlatIP(ictx, 4096), rank, world, shard_uid())
                else:             # torch.distributed collectives over host copies (gloo bring-up on one GPU, or the fallback)
                    self.swarm = shard.SwarmIndex(capi.IndexFlatIP(ictx, 4096), rank, world, dist, coll_dev)
                per_rank = 4 * args.db_keyframes // world
                self.swarm.preload_local(RowFactory(7 + rank).rows(per_rank), per_rank * world)

        def finish(self, cam, step, mb):
            out = cam.fetch()
            if os.environ.get("OMNI_BENCH_SKIP_DETECTOR") == "1":      # diagnostic only (host/index share of a step); never a reported number
                return
            if world == 1:
                # the mb key frames of the micro-batch reach the detector in order, as one batch: rows and queries are taken from
                # MobileNetVLAD's output buffer in HBM ([4*mb][4096], key-frame major), one host synchronisation for all of them
                frames = []
                for m in range(mb):
                    ims = out["images"][4 * m:4 * m + 4]
                    frames.append(detector.FisheyeFrameDescriptor(
                        msg_id=step + m, drone_id=1, landmark_num=int(sum(i["landmark_num"] for i in ims)), prevent_adding_db=False,
                        images=[detector.ImageDescriptor(drone_id=1, landmark_num=i["landmark_num"], image_desc=i["image_desc"],
                                                         feature_descriptor=i["feature_descriptor"], landmarks_2d=i["landmarks_2d"])
                                for i in ims]))
                if os.environ.get("OMNI_BENCH_DETECTOR_PER_FRAME") == "1":      # A/B: the reference's call pattern, ~6 host syncs per key frame
                    recs = [self.det.on_image_recv(fr) for fr in frames]
                else:
                    recs = self.det.on_images_recv_batch(frames, rows_dev=cam.vlad.dev_output())
                self.hits += sum(int(r["old_msg_id"] != -1) for r in recs)
            else:
                # the micro-batch's mb steps (each: add world*4 rows, query direction 1) in two collectives + one index sync
                base = self.swarm.ntotal
                if self.native:   # rows and queries straight from MobileNetVLAD's output buffer in HBM ([mb][4][4096])
                    results = self.swarm.step_batch_dev(mb, 4, cam.vlad.dev_output(), 1, K_SEARCH)
                else:
                    rows = np.stack([np.stack([i["image_desc"] for i in out["images"][4 * m:4 * m + 4]]) for m in range(mb)])
                    results = self.swarm.step_batch(rows, query_row=1, k=K_SEARCH)
                for m, (D, I) in enumerate(results):
                    nt = base + (m + 1) * world * 4                                 # ntotal as of this key frame's step
                    ok = (I[0] >= 0) & (I[0] <= nt - MATCH_INDEX_DIST) & (D[0] > QUERY_THRES)
                    self.hits += int(ok.any())

        def run(self, n_steps):
            """`pipelines` micro-batches in flight: micro-batch s is enqueued (upload + kernels + D2H copies, no host sync) before the host
            waits for micro-batch s - pipelines + 1.  Exactly n_steps key frames: a trailing partial micro-batch is its own smaller unit."""
            from collections import deque
            pending = deque()
            full, rem = n_steps // MB, n_steps % MB
            slot0 = self.step // MB
            for s in range(full):
                cam = self.cams[s % len(self.cams)]
                cam.enqueue_host(pool[(slot0 + s) % POOL])
                pending.append((cam, self.step + s * MB, MB))
                if len(pending) >= len(self.cams):
                    self.finish(*pending.popleft())
            if rem:
                if self.tail_cam is None or self.tail_cam.n_dirs != 4 * rem:
                    self.tail_ctx = capi.Context(local_rank)
                    self.tail_cam = frontend.LoopCam(self.tail_ctx, sp_w, comp, mean, vl_w, vl_specs, vl_shape, W, H, THRES, MAXN, self.prec, n_dirs=4 * rem)
                self.tail_cam.enqueue_host(tail_for(n_steps)[0])
                pending.append((self.tail_cam, self.step + full * MB, rem))
            while pending:
                self.finish(*pending.popleft())
            self.step += n_steps

        def sync(self):
            for c in self.ctxs:
                c.sync()

    hits = 0
    pyloop = None
    if cpp_host and world > 1:
        # the sharded C++ loop needs RCCL inside libomni_hip.so (dlopen) and one GPU per rank: probe it collectively, and if ANY rank cannot
        # set it up every rank takes the torch.distributed path instead (still GPU compute; the line's host_loop field says which ran)
        ok = 1.0
        try:
            capi.shard_unique_id()
        except Exception as e:                                # noqa: BLE001
            print(f"[bench] rank {rank}: omni_shard unavailable ({e}); falling back to the torch.distributed exchange", file=sys.stderr)
            ok = 0.0
        t = torch.tensor([ok], device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        cpp_host = bool(t.item() > 0.5)
        if cpp_host:
            # ... and the communicator itself (ncclCommInitRank over every rank's GPU), once, on a throw-away shard: a node on which RCCL loads but
            # cannot build its rings gives a line from the fallback path instead of no line
            uid = shard_uid()
            try:
                probe = capi.Shard(ictx, capi.IndexFlatIP(ictx, 4096), rank, world, uid)
                probe.close()
            except Exception as e:                            # noqa: BLE001
                print(f"[bench] rank {rank}: omni_shard_create failed ({e}); falling back to the torch.distributed exchange", file=sys.stderr)
                ok = 0.0
            t = torch.tensor([ok], device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            cpp_host = bool(t.item() > 0.5)
    if cpp_host:
        main_leg = cpp_leg(prec, capi.STORE_F32, 4 * args.db_keyframes, args.steps, args.warmup, args.min_time)
        hits = main_leg["loop_candidates_found"]
    else:
        pyloop = PythonLoop(prec)
        dts = timed_regions(pyloop.run, pyloop.sync, args.steps, args.warmup, args.min_time)
        main_leg = summarize(dts, args.steps)
        hits = pyloop.hits
    kfps = main_leg["value"]

    # ---- comparison legs (N = 1 only; bounded) ----------------------------------------------------------------------------------------
    python_host = value_f32 = value_parity = db100k = with_geometry = c5_shard = long_regions = None
    if world == 1 and cpp_host and args.steps < 8 * MB and args.long_region_steps > 0:
        # a timed region shorter than the pipeline is deep (the driver's 20 key frames = 2.5 micro-batches) measures ramp-up and drain, not the rate: the
        # same loop over regions of 200 key frames, for the record (`value` above stays what was asked for)
        n = max(8 * MB, args.long_region_steps // MB * MB)
        long_regions = cpp_leg(prec, capi.STORE_F32, 4 * args.db_keyframes, n, 4 * MB, min(args.min_time, 0.5))
        long_regions.update(steps=n, note=f"the headline loop timed over regions of {n} key frames instead of {args.steps}: {args.steps} key frames are "
                                          f"{args.steps / MB:.1f} micro-batches -- the first upload and the last unit's host work are exposed in every region; a "
                                          "running system never drains")
    if world == 1:
        if cpp_host and args.geometry_steps > 0:
            n = max(MB, args.geometry_steps // MB * MB)
            with_geometry = cpp_leg(prec, capi.STORE_F32, 4 * args.db_keyframes, n, MB * (args.pipelines or 4), min(args.min_time, 0.5), geometry=True,
                                    ptrs=[a.ctypes.data for a in room_pool()])
            with_geometry.update(steps=n, note="same loop + the host geometry stage per candidate (lifting, up/down triangulation, BF + homography-RANSAC "
                                               "mask, PnP-RANSAC; f64, host) on key frames of a RENDERED scene (synth.room_keyframe: a stereo rig in textured "
                                               f"rooms, {POOL * MB} places revisited cyclically): every revisit is a real loop -- stereo landmarks, PnP, an accepted LoopEdge")
        if cpp_host and args.python_steps > 0:
            pyloop = PythonLoop(prec)
            n = max(MB, args.python_steps // MB * MB)
            python_host = summarize(timed_regions(pyloop.run, pyloop.sync, n, MB * (args.pipelines or 4), min(args.min_time, 0.5)), n)
            python_host["steps"] = n
        if args.f32_steps > 0 and args.precision == "f16":
            n = max(MB, args.f32_steps // MB * MB)
            if cpp_host:
                value_f32 = cpp_leg(capi.PREC_F32, capi.STORE_F32, 4 * args.db_keyframes, n, MB * (args.pipelines or 4), 0.0)
            else:
                l32 = PythonLoop(capi.PREC_F32)
                value_f32 = summarize(timed_regions(l32.run, l32.sync, n, MB * (args.pipelines or 4), 0.0), n)
            value_f32.update(steps=n, dtype="f32", note="OMNI_PREC_F32: exact-f32 MFMA network (key points identical to the fp32 oracle), same workload")
        if args.parity_steps > 0 and args.precision == "f16" and cpp_host:
            n = max(MB, args.parity_steps // MB * MB)
            value_parity = cpp_leg(capi.PREC_SPLIT, capi.STORE_F32, 4 * args.db_keyframes, n, MB * (args.pipelines or 4), min(args.min_time, 0.5))
            value_parity.update(steps=n, dtype="f16 x3 (split hi+lo operands, fp32-class)",
                                note="OMNI_PREC_SPLIT: the SAME workload and host loop with SuperPoint's 3x3 convolutions on the fp16 matrix cores at fp32-class "
                                     "accuracy (every operand a (hi, lo) pair of halfs, three MFMA terms per product; heads in exact f32; MobileNetVLAD is "
                                     "fp32-class in every mode): key points identical to the fp32 oracle, descriptors ~1e-6 -- `parity_split` below; "
                                     "gates: tests/test_gpu_bench_shape.py::test_superpoint_64_images_split_precision_meets_the_north_star_bar")
        if args.c5_rows > 0 and cpp_host:
            # BASELINE configs[4] as one GPU of eight sees it: 64 key frames in flight (micro-batches of 16 x 4 pipelines: 128 SuperPoint + 64
            # MobileNetVLAD images per launch sequence), every micro-batch's 16 queries in ONE matrix-core pass over an fp16 shard
            c5_mb, c5_pipes = 16, 4
            n = c5_mb * c5_pipes * 2
            c5_shard = cpp_leg(prec, capi.STORE_F16, args.c5_rows, n, c5_mb * c5_pipes, min(args.min_time, 0.5), mb=c5_mb, pipelines=c5_pipes)
            if args.precision == "f16" and args.parity_steps > 0:
                c5_split = cpp_leg(capi.PREC_SPLIT, capi.STORE_F16, args.c5_rows, n, c5_mb * c5_pipes, min(args.min_time, 0.5), mb=c5_mb, pipelines=c5_pipes)
                c5_shard["split"] = {k: c5_split[k] for k in ("value", "ms_per_step", "repeats", "keyframe_latency_ms", "db_rows_end", "loop_candidates_found")}
                c5_shard["split"]["dtype"] = "f16 x3 (split operands, fp32-class): the same leg in the mode that meets north_star's tolerance"
            c5_shard.update(steps=n, shard_rows=args.c5_rows, keyframes_per_microbatch=c5_mb, pipelines=c5_pipes,
                            note="configs[4] end to end on ONE GPU's share: 64 concurrent key frames against a 500k-row fp16 shard (1/8 of the 4M rows of a "
                                 "1M-key-frame database), the batched search inside the timed loop; the 8-GPU exchange itself is the driver's run")
        if args.big_db_keyframes > 0 and cpp_host:
            n = max(MB, args.big_db_steps // MB * MB)
            db100k = {"db_keyframes": args.big_db_keyframes, "db_rows": 4 * args.big_db_keyframes, "steps": n,
                      "note": "same key-frame loop, every key frame's query scans the whole database (one pass per micro-batch, per-query row limits)"}
            for name, st in (("f16_rows", capi.STORE_F16), ("f32_rows", capi.STORE_F32)):
                db100k[name] = cpp_leg(prec, st, 4 * args.big_db_keyframes, n, MB * (args.pipelines or 4), min(args.min_time, 0.5))
            if args.precision == "f16" and args.parity_steps > 0:
                # north_star's "against a 100k-keyframe DB" in the mode that meets its tolerance: OMNI_PREC_SPLIT, exact fp32 rows
                db100k["split_f32_rows"] = cpp_leg(capi.PREC_SPLIT, capi.STORE_F32, 4 * args.big_db_keyframes, n, MB * (args.pipelines or 2), min(args.min_time, 0.5))
                db100k["split_f32_rows"]["dtype"] = "f16 x3 (split operands, fp32-class): key points identical to the fp32 oracle (parity_split)"

    # ---- roofline of the dominant kernel (conv1b + pool, 43 % of the FLOPs): HIP events on the kernel's own stream --
    # same launch shape as in the timed loop: one micro-batch = 8 * MB images per launch (HIP events between the stages, on the
    # kernels' own stream); stage times are then quoted per key frame (8 images)
    n_img = KF_IMAGES * MB
    if pyloop is not None:
        prof_sp = pyloop.cams[0].sp
    else:
        prof_sp = capi.SuperPoint(ictx, sp_w, comp, mean, W, H, THRES, MAXN, prec, n_img)
    pool_dev = ictx.to_device(pool[0])
    prof = prof_sp.profile(pool_dev, W, n_img, reps=21)
    conv_ms = sum(p["ms"] for p in prof if p["stage"].startswith("conv")) / MB
    sp_ms = sum(p["ms"] for p in prof) / MB
    peak = PEAK_F32_TFLOPS if args.precision == "f32" else PEAK_F16_TFLOPS
    left_out_flop = sum(p["flops_per_image"] * p["tiles_left_out"] for p in prof)       # per image, by the library's own plan

    def conv1b_roofline(stages, precision):
        """`frac` = FLOP the launch EXECUTED / its HIP-event time / peak: the tiles that ran (a fisheye-masked pass leaves the constant region of the mask
        out of the tile walk) x 3 MFMA terms per product in OMNI_PREC_SPLIT; the layer's algorithmic FLOP / time is kept as frac_algorithmic."""
        c1b = next(p for p in stages if p["stage"].startswith("conv1b"))
        alg = c1b["flops_per_image"] * n_img
        split = precision == "split"
        wino = split and (capi.config_value("OMNI_SPLIT_WINO") & 1) != 0      # conv1b as the Winograd F(2x2,3x3) kernel (csrc/conv_wino.hip): 16 products per 2 x 2 outputs, not 36
        # OMNI_PREC_SPLIT executes three fp16 MFMA terms per product; the Winograd form has 16 / 36 of the direct form's products: 4/3 of the algorithmic FLOP
        terms = (3.0 * 16.0 / 36.0 if wino else 3.0) if split else 1.0
        executed = terms * alg * (1.0 - c1b["tiles_left_out"])
        t = c1b["ms"] * 1e-3
        pk = PEAK_F32_TFLOPS if precision == "f32" else PEAK_F16_TFLOPS
        trace = rocprof_trace(("conv3x3_wino_kernelILb1ELb0ELb1" if wino else "conv3x3_split_kernelILb0ELb1ELb0ELb0ELb1") if split else "conv3x3_c64_pp_kernelILb1ELi0ELb1")
        return {"bound": "mfma",
                "kernel": ("conv3x3_wino_kernel<POOL, FUSE1A> = conv1a (built tile by tile from the u8 image, on the matrix cores) + conv1b 3x3 64->64 as Winograd F(2x2,3x3) + ReLU + maxpool2 "
                           "with split (hi, lo) fp16 operands: 16 x 3 MFMA terms per 2 x 2 outputs and input channel (the direct form: 36 x 3) -- `frac` counts the MFMA FLOP the "
                           "kernel executes (4/3 of the layer's algorithmic FLOP), `frac_algorithmic` the layer's own FLOP: the fraction that says how fast the LAYER runs; "
                           "FLOP counted for conv1b only" if wino else
                           "conv3x3_split_kernel<cin 64, POOL, FUSE1A> = conv1a (built tile by tile from the u8 image, on the matrix cores) + conv1b 3x3 64->64 + ReLU + maxpool2 with split "
                           "(hi, lo) fp16 operands: three MFMA terms per product; FLOP counted for conv1b only" if split else
                           "conv3x3_c64_pp_kernel<POOL, FUSE1A> = conv1a (u8 -> 64 ch, on the matrix cores) + conv1b 3x3 64->64 + ReLU + maxpool2 in one launch; "
                           "FLOP counted for conv1b only"),
                "achieved": round(executed / t / 1e12, 1), "peak": pk, "unit": "TFLOP/s", "frac": round(executed / t / 1e12 / pk, 4),
                "definition": "achieved = MFMA FLOP of the tiles the launch ran / HIP-event time between the stage markers on the kernel's own stream",
                "flop_executed_per_launch": executed, "launch_ms": round(c1b["ms"], 4), "images_per_launch": n_img, "tiles_left_out": round(c1b["tiles_left_out"], 4),
                "flop_algorithmic_per_launch": alg, "achieved_algorithmic": round(alg / t / 1e12, 1), "frac_algorithmic": round(alg / t / 1e12 / pk, 4),
                "rocprof_trace": trace,
                "rocprof_trace_note": "median launch duration of the same kernel under rocprofv3 --kernel-trace (committed summary); frac recomputed from it = "
                                      + (str(round(executed / (trace["median_us"] * 1e-6) / 1e12 / pk, 4)) if trace else "n/a: no trace committed yet"),
                **(traffic_fields("conv3x3_wino_kernel<POOL,FUSE1A> (conv1b)" if wino else "conv3x3_split_kernel<cin64,POOL,FUSE1A> (conv1b)", True, n_img) if split else
                   traffic_fields("conv3x3_c64_pp_kernel<POOL,FUSE1A>", precision == "f16", n_img))}

    roofline = conv1b_roofline(prof, args.precision)
    if args.precision != "f32" and rank == 0:
        # what THIS board's fp16 matrix cores sustain (omni_ctx_mfma_ceiling: back-to-back MFMAs on every SIMD, ~0.1 s): `peak` is the guide's figure at the
        # 2.4 GHz engine clock; saturated matrix cores run at the clock the power budget leaves (profiles/r05u_mfma_ceiling_probe.log: ~1.95 GHz, 2.0 PFLOP/s)
        try:
            cal = ictx.mfma_ceiling(100.0)
            roofline["measured_ceiling"] = {"tflops": round(cal["tflops"], 1), "sclk_ghz": round(cal["sclk_ghz"], 3), "frac_of_peak": round(cal["tflops"] / PEAK_F16_TFLOPS, 4),
                                            "frac_of_measured_ceiling": round(roofline["achieved"] / cal["tflops"], 4) if cal["tflops"] > 0 else None,
                                            "definition": "v_mfma_f32_32x32x16_f16 back to back on every SIMD of this GPU for ~0.1 s, no memory traffic: TFLOP/s by HIP events, "
                                                          "shader clock by s_memtime against s_memrealtime; extra information -- `frac` stays against `peak`"}
        except Exception as e:                                    # noqa: BLE001
            roofline["measured_ceiling"] = {"error": str(e)}
    try:
        _vl = capi.MobileNetVLAD(ictx, vl_w, vl_specs, *vl_shape, W, H, 1)
        vlad_left_out = [round(f, 4) for f in _vl.mask_skip_layers()]
        _vl.close()
    except Exception as e:                                        # noqa: BLE001
        vlad_left_out = str(e)
    roofline.update({
        "mask_skip": {"note": "stage times with the fisheye mask on, as the key-frame pipeline runs the network (loop_cam.cpp:536-539): the tiles whose whole "
                              "receptive field lies in the blanked rows hold one constant vector per layer, written once, and are left out of the tile walk "
                              "(bit-identical: tests/test_gpu_mask_skip.py; OMNI_SP_MASK_SKIP=0 / OMNI_SP_MASK_SKIP_SPLIT=0 = the dense pass)",
                      "tiles_left_out": {p["stage"]: round(p["tiles_left_out"], 4) for p in prof if p["tiles_left_out"] > 0},
                      "mobilenetvlad_tiles_left_out": vlad_left_out,
                      "mobilenetvlad_note": "the same for MobileNetVLAD (the frame is blanked before both networks, loop_cam.cpp:536-539, 556-558): stem + block 0, blocks 1, 2, ... "
                                            "(csrc/vlad.hip, omni_vlad::MaskSkip; OMNI_VLAD_MASK_SKIP=0 = the dense pass; tests/test_gpu_vlad_detector.py)"},
        "conv_stack_tflops": round(mfma_terms_of(args.precision) * sp_flop_executed(args.precision, MAXN, True, left_out_flop) * KF_IMAGES / (conv_ms * 1e-3) / 1e12, 1),
        "conv_stack_note": "MFMA FLOP executed by the stages named conv* (fp16 path: without convDa / convDb, which run only at the cells around the key points "
                           "inside the post-processing stage; split: x 3 terms of the direct form for every layer -- the layers OMNI_SPLIT_WINO runs as Winograd kernels execute 4/3 -- so this is a rate of useful work) / their time",
        "stages_ms_per_keyframe": {p["stage"]: round(p["ms"] / MB, 4) for p in prof}, "superpoint_ms_per_keyframe": round(sp_ms, 3)})
    # the same kernel figure for the precision that meets north_star's tolerance (value_parity's conv1b), in the default run
    roofline_parity = None
    if args.precision == "f16" and args.parity_steps > 0 and world == 1:
        sp_split = capi.SuperPoint(ictx, sp_w, comp, mean, W, H, THRES, MAXN, capi.PREC_SPLIT, n_img)
        prof_s = sp_split.profile(pool_dev, W, n_img, reps=11)
        sp_split.close()
        roofline_parity = conv1b_roofline(prof_s, "split")
        roofline_parity.update({"stages_ms_per_keyframe": {p["stage"]: round(p["ms"] / MB, 4) for p in prof_s},
                                "superpoint_ms_per_keyframe": round(sum(p["ms"] for p in prof_s) / MB, 3),
                                "tiles_left_out_by_stage": {p["stage"]: round(p["tiles_left_out"], 4) for p in prof_s if p["tiles_left_out"] > 0}})
        mc = roofline.get("measured_ceiling") or {}
        if mc.get("tflops"):
            roofline_parity["measured_ceiling"] = {"tflops": mc["tflops"], "sclk_ghz": mc["sclk_ghz"], "frac_of_measured_ceiling": round(roofline_parity["achieved"] / mc["tflops"], 4),
                                                   "definition": "see roofline.measured_ceiling"}
    ictx.free(pool_dev)

    # ---- p50 loop-match latency on big DBs (node total rows, sharded when N > 1): 100k rows and 100k key frames = 400k rows ----------
    def p50_leg(total_rows):
        rows_here = total_rows // world
        midx = capi.IndexFlatIP(ictx, 4096, capi.STORE_F32, rows_here)
        gen = RowFactory(11 + rank)
        if world == 1:
            for s in range(0, rows_here, 32768):
                midx.add(gen.rows(min(32768, rows_here - s)))
        mq = RowFactory(99).rows(1)
        native_p50 = world > 1 and cpp_host
        if world > 1 and not native_p50:
            big = shard.ShardedIndex(midx, rank, world, dist, coll_dev)
            for s in range(0, rows_here, 32768):
                midx.add(gen.rows(min(32768, rows_here - s)))
            search = lambda: big.search(mq, K_SEARCH)
        elif world > 1:
            big = capi.Shard(ictx, midx, rank, world, shard_uid())
            for s in range(0, rows_here, 32768):
                big.local.add(gen.rows(min(32768, rows_here - s)))
            search = lambda: big.search(mq, K_SEARCH)
        else:
            search = lambda: midx.search(mq, K_SEARCH)
        lat, scan = [], []
        for i in range(60):
            if dist is not None:
                dist.barrier()
            t = time.perf_counter()
            search()
            lat.append((time.perf_counter() - t) * 1e3)
            scan.append(midx.last_scan_ms())
        p50 = reduce_max(float(np.median(lat[10:])))
        scan_ms = float(np.median(scan[10:]))
        if world > 1 and native_p50:
            big.close()
        midx.close()
        return p50, scan_ms, rows_here

    p50, scan_ms, rows_here = p50_leg(args.match_db_rows)
    gbs = rows_here * 4096 * 4 / (scan_ms * 1e-3) / 1e9
    roofline_knn = {"bound": "hbm", "kernel": "ip_scan_kernel<float,1>", "achieved": round(gbs, 0), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4),
                    **traffic_fields(f"ip_scan_kernel<float,1>@{rows_here}"), "bytes_per_launch": rows_here * 16384,
                    "launch_ms": round(scan_ms, 4), "rows_per_gpu": rows_here}
    loop_match = {"p50_ms": round(p50, 4), "db_rows_node": args.match_db_rows, "db_rows_per_gpu": rows_here, "k": K_SEARCH,
                  "includes": "H2D query, scan, top-k, D2H result" + (", all_gather + merge" if world > 1 else "")}
    if args.match_db_rows != 400_000:
        p50b, scan_b, rows_b = p50_leg(400_000)
        loop_match["db_100k_keyframes"] = {"p50_ms": round(p50b, 4), "db_rows_node": 400_000, "db_rows_per_gpu": rows_b, "scan_ms": round(scan_b, 4),
                                           "scan_gbs": round(rows_b * 16384 / (scan_b * 1e-3) / 1e9, 0)}

    # ---- BASELINE config 5 per-GPU shard: 64 concurrent queries against 1 M key frames / 8 GPUs = 125 000 fp16 rows --------------
    batched = None
    if args.batched_rows > 0:
        bidx = capi.IndexFlatIP(ictx, 4096, capi.STORE_F16, args.batched_rows)
        gen = RowFactory(13 + rank)
        for s in range(0, args.batched_rows, 32768):
            bidx.add(gen.rows(min(32768, args.batched_rows - s)))
        bq = RowFactory(98).rows(64)
        blat, bscan = [], []
        for i in range(30):
            t = time.perf_counter()
            bidx.search(bq, K_SEARCH)
            blat.append((time.perf_counter() - t) * 1e3)
            bscan.append(bidx.last_scan_ms())
        b_ms, b_p50 = float(np.median(bscan[8:])), float(np.median(blat[8:]))
        b_bytes = args.batched_rows * (4096 * 2 + 64 * 8)            # every fp16 row read once + one 64-bit key per (query, row)
        batched = {"bound": "hbm", "kernel": "ip_scan_mq_kernel (64 queries, fp16 rows, v_mfma_f32_16x16x32_f16)",
                   "achieved": round(b_bytes / (b_ms * 1e-3) / 1e9, 0), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                   "frac": round(b_bytes / (b_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), **traffic_fields("ip_scan_mq_kernel", args.batched_rows == 125_000),
                   "bytes_per_launch": b_bytes, "launch_ms": round(b_ms, 4), "rows_per_gpu": args.batched_rows, "queries": 64,
                   "search_p50_ms": round(b_p50, 4), "queries_per_s": round(64 / (b_p50 * 1e-3), 0)}
        bidx.close()

    # ---- CPU baseline: the reference's PyTorch-CPU SuperPoint path + oracle post-processing, same host; and, with the oracle's outputs
    # of that sample at hand, how far the GPU path at the benchmarked precision is from them (`parity`) -------------------------------
    cpu = parity = parity_split = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu, parity, parity_split = cpu_baseline(args.cpu_keyframes, W, H, THRES, MAXN, comp, mean, sp_w, vl_w, vl_specs, vl_shape, capi, ictx, prec)

    librccl = None
    rccl_mapped = None
    if world > 1 and cpp_host:
        try:
            librccl = capi.shard_library_path()
        except Exception as e:                                    # noqa: BLE001
            librccl = f"unavailable: {e}"
        try:                                                      # every librccl mapped into this process: one path = one RCCL, whoever else loaded it
            rccl_mapped = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln or "libstub_rccl" in ln})
        except OSError:
            rccl_mapped = None

    def gate(par, label):
        """north_star's tolerance, as a boolean next to the number it qualifies: key-point sets identical on every image of the sample (the fixed NMS
        ordering makes the index lists bit-exact then), descriptors and the MobileNetVLAD vector within 1e-3 relative"""
        if par is None:
            return None
        a, b = par["images_with_identical_keypoint_set"].split("/")
        ok = a == b and par["desc64_rel_err_p99"] <= 1e-3 and par["vlad_rel_err_max"] <= 1e-3
        return {"precision": label, "within_north_star_tolerance": bool(ok), "images_with_identical_keypoint_set": par["images_with_identical_keypoint_set"],
                "desc64_rel_err_p99": par["desc64_rel_err_p99"], "vlad_rel_err_max": par["vlad_rel_err_max"]}
    headline_gate = gate(parity, args.precision)
    split_gate = gate(parity_split, "split") if parity_split is not None else None
    if rank == 0:
        line = {
            "metric": "keyframes/sec (4x fisheye 600x480) + p50 loop-match ms @ 100k-frame DB",
            "value": kfps, "unit": "keyframes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # the honesty gate: is `value` a number INSIDE north_star's tolerance (key-point indices bit-exact, descriptors / VLAD vectors
            # within 1e-3)?  fp16 -- the reference's own engine precision (launch/realsense.launch:10-11) -- is not; the mode that is runs at
            # `value_within_north_star_tolerance` (= value_parity.value, OMNI_PREC_SPLIT, same workload and host loop, same run)
            "within_north_star_tolerance": None if headline_gate is None else headline_gate["within_north_star_tolerance"],
            "parity_gate": {"headline": headline_gate, "value_parity": split_gate},
            "value_within_north_star_tolerance": (kfps if (headline_gate or {}).get("within_north_star_tolerance") else
                                                  (value_parity or {}).get("value") if (split_gate or {}).get("within_north_star_tolerance") else None),
            "ms_per_step": main_leg["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f16": "f16", "f32": "f32", "split": "f16 x3 (split operands)"}[args.precision], "data": "synthetic",
            "repeats": main_leg["repeats"], "ms_per_step_minmax": main_leg["ms_per_step_minmax"],
            "keyframe_latency_ms": main_leg.get("keyframe_latency_ms"),
            "host_ms_per_microbatch": main_leg.get("host_ms_per_microbatch"),
            "value_long_regions": long_regions,
            "config": {"workload": "configs[1]: reference-faithful fisheye key frame = upload of 8 images + 8 SuperPoint + 4 MobileNetVLAD(assumed arch) "
                                   "images 600x480 + 4 up/down BF matches + <=4 index inserts + top-k query + results to host; "
                                   f"{args.db_keyframes}-keyframe DB ({4 * args.db_keyframes} rows); seeded synthetic weights",
                       "images_per_keyframe": KF_IMAGES, "superpoint_thres": THRES, "max_num": MAXN, "pipelines_per_gpu": main_leg.get("pipelines", args.pipelines), "units_oldest_first": main_leg.get("units_oldest_first"),
                       "keyframes_per_microbatch": MB, "host_loop": ("c++ (host/keyframe_pipeline.hpp)" + (" + omni_shard (RCCL inside libomni_hip.so)" if world > 1 else "")) if cpp_host else
                                    "python" + (" + torch.distributed exchange" if world > 1 else ""),
                       "image_upload": "inside the timed region (pinned host -> HBM, one async copy per micro-batch)",
                       "parallelism": f"dp{world} keyframes + {world}-way row-sharded index" if world > 1 else "single GPU",
                       "device": info["name"], "n_cu": info["n_cu"]},
            "loop_candidates_found": hits,
            "gflop_per_keyframe_superpoint": round(sp_flop_executed(args.precision, MAXN, False, left_out_flop) * KF_IMAGES / 1e9, 1),
            "gflop_per_keyframe_superpoint_dense": round(SP_FLOP_PER_IMAGE * KF_IMAGES / 1e9, 1),
            "achieved_tflops_end_to_end": round(kfps * sp_flop_executed(args.precision, MAXN, False, left_out_flop) * KF_IMAGES / 1e12 / world, 1),
            "rccl_ranks": world if (world > 1 and cpp_host) else (1 if world == 1 else 0), "librccl": librccl, "librccl_mapped_in_process": rccl_mapped,
            "torch_distributed_backend": (dist.get_backend() if dist is not None else None),
            "per_rank_keyframes_per_s": main_leg.get("per_rank_keyframes_per_s"), "db_rows_per_gpu": main_leg.get("db_rows_per_gpu"),
            "all_gather_us_p50": main_leg.get("all_gather_us_p50"),
            "roofline": roofline, "roofline_parity": roofline_parity, "roofline_knn": roofline_knn, "roofline_knn_batched": batched, "loop_match": loop_match,
            "db100k": db100k, "with_geometry": with_geometry, "value_f32": value_f32, "value_parity": value_parity, "c5_shard": c5_shard,
            "python_host": python_host, "parity": parity, "parity_split": parity_split, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _cpu_topology():
    """model name, the CPUs of NUMA node 0 ordered one hardware thread per physical core first, the node's physical core count (Linux sysfs / procfs; anything
    missing: every CPU the process may run on, in numeric order)"""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass

    def parse_list(txt):
        out = []
        for part in txt.strip().split(","):
            if "-" in part:
                a, b = part.split("-")
                out += list(range(int(a), int(b) + 1))
            elif part:
                out.append(int(part))
        return out

    node = allowed
    try:
        node = [c for c in parse_list(open("/sys/devices/system/node/node0/cpulist").read()) if c in set(allowed)] or allowed
    except OSError:
        pass
    by_core = {}
    for c in node:
        try:
            key = (open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id").read().strip(), open(f"/sys/devices/system/cpu/cpu{c}/topology/core_id").read().strip())
        except OSError:
            key = ("?", str(c))
        by_core.setdefault(key, []).append(c)
    firsts = [v[0] for v in by_core.values()]
    rest = [c for v in by_core.values() for c in v[1:]]
    return {"model": model, "order": firsts + rest, "physical_cores_node0": len(by_core)}


def cpu_baseline(n_kf, W, H, thres, max_num, comp, mean, sp_w, vl_w, vl_specs, vl_shape, capi, ictx, prec):
    """The reference's CPU path restated (oracle): PyTorch SuperPointNet fp32 on all host cores + C post-processing +
    MobileNetVLAD(assumed) + BF match + flat IP search, timed on a bounded sample of the same workload.  Second result: the GPU path at
    the benchmarked precision on the same 8 images, compared with what the oracle just computed (the oracle is the checker here)."""
    import torch
    from oracle import match_ref, mobilenetvlad_ref, postproc_ref, superpoint_ref
    from omni_swarm_amd import synth
    avail = os.cpu_count() or 1
    db = synth.global_db(4000, seed=3)
    imgs = np.stack([synth.image_u8(i, H, W) for i in range(8)])
    last = {}
    # The host cores this baseline runs on: ONE NUMA node, one hardware thread per physical core first (os.sched_setaffinity; restored at the end), and the
    # thread count swept -- oneDNN on all 256 hardware threads of the GPU box thrashes (96 s per key frame), 32 un-pinned threads measured 175 ms per image in
    # round 5 against the survey container's 118 ms on 8: an un-swept, un-pinned count is not "the GPU box's own host cores".
    topo = _cpu_topology()
    old_aff = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    x_probe = superpoint_ref.preprocess_u8(imgs, fisheye_mask=True)

    def pin(n):
        cpus = topo["order"][:max(1, n)]
        if old_aff is not None:
            try:
                os.sched_setaffinity(0, set(cpus) & set(old_aff) or old_aff)
            except OSError:
                pass
        torch.set_num_threads(max(1, n))

    sweep = []
    counts = sorted({c for c in (8, 16, 32, 64, topo["physical_cores_node0"]) if 1 <= c <= max(1, len(topo["order"]))} or {min(avail, 8)})
    for n in counts:                            # SuperPoint alone (98 % of the key frame's CPU time): 1 warm-up + 3 timed batches of 8 images per setting
        pin(n)
        superpoint_ref.forward(sp_w, x_probe)
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            superpoint_ref.forward(sp_w, x_probe)
            ts.append(time.perf_counter() - t)
        sweep.append({"threads": n, "ms_per_image": round(float(np.median(ts)) * 1e3 / 8, 1)})
    cores = min(sweep, key=lambda e: e["ms_per_image"])["threads"]
    pin(cores)

    def keyframe():
        x = superpoint_ref.preprocess_u8(imgs, fisheye_mask=True)
        semi, desc = superpoint_ref.forward(sp_w, x)            # the 8 images of the key frame as one batch
        feats, kps = [], []
        for b in range(8):
            xy, conf, _, _ = postproc_ref.get_keypoints(semi[b], thres, max_num)
            d64, _ = postproc_ref.compute_descriptors(desc[b], xy, W, H, comp, mean)
            feats.append(d64)
            kps.append(xy)
        masked = imgs[:4].copy()
        masked[:, H * 3 // 4:] = 0
        g = mobilenetvlad_ref.forward(vl_w, masked)
        for d in range(4):
            match_ref.bf_match(feats[d], feats[4 + d], 0)
        match_ref.ip_search_numpy(db, g[1], 10)
        last.update(kps=kps, feats=feats, g=g)

    for _ in range(3):                          # warm-ups (median of >= 20 after 3)
        keyframe()
    dts = []
    for _ in range(n_kf):
        t = time.perf_counter()
        keyframe()
        dts.append(time.perf_counter() - t)
    dt = float(np.median(dts))
    t_sp = []
    for _ in range(3):                          # the SuperPoint share at the chosen setting, per image (next to the survey container's probe)
        t = time.perf_counter()
        superpoint_ref.forward(sp_w, x_probe)
        t_sp.append(time.perf_counter() - t)
    if old_aff is not None:
        try:
            os.sched_setaffinity(0, old_aff)
        except OSError:
            pass
    cpu = {"value": round(1.0 / dt, 4), "unit": "keyframes/s", "cores": cores, "kind": "port",
           "sample": f"median of {n_kf} key frames (8 SuperPoint + 4 MobileNetVLAD 600x480, post-processing, 4 BF matches, 1 search over 4000 rows) "
                     f"after 3 warm-ups (SURVEY 8d); torch {torch.__version__} fp32, {cores} threads pinned to one hardware thread each of NUMA node 0 "
                     f"({topo['physical_cores_node0']} physical cores; {avail} host CPUs in all); the thread count is the best of `sweep`",
           "ms_per_keyframe": round(dt * 1e3, 1), "ms_per_keyframe_minmax": [round(min(dts) * 1e3, 1), round(max(dts) * 1e3, 1)],
           "cpu_model": topo["model"], "superpoint_ms_per_image": round(float(np.median(t_sp)) * 1e3 / 8, 1), "sweep": sweep,
           "survey_probe": "118 ms per image on 8 threads of the survey container (BASELINE.md section 4)"}
    # GPU path on the same key frame vs the oracle's outputs
    vl = capi.MobileNetVLAD(ictx, vl_w, vl_specs, *vl_shape, W, H, 4)
    g = vl.inference(imgs[:4], fisheye_mask=True)
    vl.close()
    vrel = np.linalg.norm(g - last["g"], axis=1) / np.linalg.norm(last["g"], axis=1)

    def parity_of(precision, label):
        sp = capi.SuperPoint(ictx, sp_w, comp, mean, W, H, thres, max_num, precision, 8)
        res = sp.inference(imgs, fisheye_mask=True)
        _, dense = sp.get_dense(8)                 # the network's own dense descriptor map [8][256][Hc][Wc]
        sp.close()
        overlap, derr, derr_same, identical = [], [], [], 0
        for b in range(8):
            ref = {tuple(p): i for i, p in enumerate(last["kps"][b].tolist())}
            got = res[b][0].astype(np.int32).tolist()
            common = [(i, ref[tuple(p)]) for i, p in enumerate(got) if tuple(p) in ref]
            overlap.append(len(common) / max(1, len(ref)))
            identical += int(len(common) == len(ref) == len(got))
            if common:
                gi, ri = zip(*common)
                a, r = res[b][1][list(gi)], last["feats"][b][list(ri)]
                derr.append(np.linalg.norm(a - r, axis=1) / np.maximum(np.linalg.norm(r, axis=1), 1e-12))
            # the same descriptor arithmetic (superpoint_tensorrt.cpp:192-230) on the GPU network's dense map but at the ORACLE's key points: the
            # network's rounding alone, without the effect a flipped key point has on every descriptor of its image through the reference's
            # per-channel normalisation across key points (:211-215)
            same, _ = postproc_ref.compute_descriptors(dense[b], last["kps"][b], W, H, comp, mean)
            r = last["feats"][b]
            if len(r):
                derr_same.append(np.linalg.norm(same - r, axis=1) / np.maximum(np.linalg.norm(r, axis=1), 1e-12))
        derr = np.concatenate(derr) if derr else np.zeros(1)
        derr_same = np.concatenate(derr_same) if derr_same else np.zeros(1)
        return {"vs": "CPU oracle (torch fp32 notebook graph + literal post-processing) on the cpu_baseline key frame, GPU path at " + label,
                "keypoint_overlap_min": round(float(min(overlap)), 4), "keypoint_overlap_mean": round(float(np.mean(overlap)), 4),
                "images_with_identical_keypoint_set": f"{identical}/8",
                "desc64_rel_err_p50": float(np.round(np.percentile(derr, 50), 7)), "desc64_rel_err_p99": float(np.round(np.percentile(derr, 99), 7)),
                "desc64_at_oracle_keypoints_rel_err_p50": float(np.round(np.percentile(derr_same, 50), 7)),
                "desc64_at_oracle_keypoints_rel_err_p99": float(np.round(np.percentile(derr_same, 99), 7)),
                "desc64_note": "desc64_rel_err_*: the pipeline's descriptors at the key points it shares with the oracle; desc64_at_oracle_keypoints_*: the "
                               "network's dense map sampled, normalised and projected at the oracle's own key-point set -- the rounding of the network alone; the "
                               "difference is the reference's per-channel normalisation ACROSS key points spreading every flipped key point over its image",
                "vlad_rel_err_max": float(np.round(vrel.max(), 7))}

    parity = parity_of(prec, "the benchmarked precision")
    parity_split = parity_of(capi.PREC_SPLIT, "OMNI_PREC_SPLIT (value_parity)") if prec != capi.PREC_SPLIT else None
    return cpu, parity, parity_split


if __name__ == "__main__":
    main()
