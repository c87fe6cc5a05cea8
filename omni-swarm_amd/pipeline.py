"""ctypes launcher of the C++ key-frame host loop (host/keyframe_pipeline.hpp -> lib/libomni_host.so).

The loop itself -- SwarmLoop::VIOKF_callback's hot part, swarm_loop/src/swarm_loop.cpp:140-170: LoopCam::on_flattened_images followed
by LoopDetector::on_image_recv, for a stream of key frames -- runs in C++ (omni::KeyframePipeline over the adapters of
host/omni_swarm.hpp and the C ABI of libomni_hip.so).  Python only writes the weight files, owns the pinned input pool and starts the run.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libomni_host.so")
SYMBOLS = ["omni_pipeline_last_error", "omni_pipeline_create", "omni_pipeline_destroy", "omni_pipeline_preload", "omni_pipeline_db_rows",
           "omni_pipeline_run", "omni_pipeline_attach_shard", "omni_pipeline_prepare", "omni_pipeline_geometry_stats", "omni_pipeline_sync",
           "omni_pipeline_set_poses", "omni_pipeline_create_pinhole_depth", "omni_pipeline_set_depth", "omni_pipeline_push_keyframe", "omni_pipeline_flush", "omni_pipeline_host_times", "omni_pipeline_get_candidates", "omni_pipeline_get_edges", "omni_pipeline_get_latencies",
           "omni_pipeline_poll", "omni_pipeline_set_latency", "omni_pipeline_units", "omni_pipeline_get_exchange_us", "omni_swarm_params_from_launch", "omni_swarm_params_table", "omni_pipeline_apply_launch", "omni_pipeline_create_from_launch"]
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is missing: run `make -C omni-swarm_amd`")
        capi.lib()                                     # libomni_hip.so first (libomni_host.so links against it)
        L = C.CDLL(LIB_PATH)
        L.omni_pipeline_last_error.restype = C.c_char_p
        L.omni_pipeline_create.restype = C.c_void_p
        L.omni_pipeline_create.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]
        L.omni_pipeline_create_pinhole_depth.restype = C.c_void_p
        L.omni_pipeline_create_pinhole_depth.argtypes = L.omni_pipeline_create.argtypes + [C.c_double] * 6 + [C.c_int]
        L.omni_pipeline_set_depth.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.omni_pipeline_push_keyframe.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_double, C.POINTER(C.c_double), C.c_int, C.c_void_p,
                                                  C.POINTER(C.c_int)]
        L.omni_pipeline_flush.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.omni_pipeline_poll.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.omni_pipeline_set_latency.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.omni_pipeline_units.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.omni_pipeline_get_exchange_us.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]
        L.omni_swarm_params_from_launch.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.omni_pipeline_apply_launch.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.omni_swarm_params_table.argtypes = [C.c_char_p, C.c_int]
        L.omni_pipeline_create_from_launch.restype = C.c_void_p
        L.omni_pipeline_create_from_launch.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                                      C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.omni_pipeline_host_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        L.omni_pipeline_geometry_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.omni_pipeline_destroy.argtypes = [C.c_void_p]
        L.omni_pipeline_destroy.restype = None
        L.omni_pipeline_preload.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int64]
        L.omni_pipeline_db_rows.argtypes = [C.c_void_p]
        L.omni_pipeline_db_rows.restype = C.c_int64
        L.omni_pipeline_run.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_int)]
        L.omni_pipeline_sync.argtypes = [C.c_void_p]
        L.omni_pipeline_attach_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        L.omni_pipeline_prepare.argtypes = [C.c_void_p, C.c_int]
        L.omni_pipeline_set_poses.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_double)]
        L.omni_pipeline_get_candidates.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int]
        L.omni_pipeline_get_edges.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        L.omni_pipeline_get_latencies.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]
        _lib = L
    return _lib


def _err(what):
    return capi.OmniError(f"{what}: {lib().omni_pipeline_last_error().decode()}")


def swarm_params_from_launch(launch_xml: str, node_name: str = "swarm_loop", args: dict | None = None):
    """The node's parameters as one of the reference's launch files sets them (host/swarm_loop_params.hpp; no ROS): -> (dict name -> value text,
    names whose stored type nh.param<T> refuses (defaults kept), names the node never reads)."""
    buf = C.create_string_buffer(1 << 16)
    a = "\n".join(f"{k}:={v}" for k, v in (args or {}).items()).encode() or None
    if lib().omni_swarm_params_from_launch(launch_xml.encode(), node_name.encode(), a, buf, len(buf)):
        raise _err("omni_swarm_params_from_launch")
    vals, mism, unk = {}, [], []
    for line in buf.value.decode().splitlines():
        if line.startswith("#mismatch "):
            mism.append(line[10:])
        elif line.startswith("#unknown "):
            unk.append(line[9:])
        else:
            k, _, v = line.partition("=")
            vals[k] = v
    return vals, mism, unk


def swarm_params_table():
    """[(name, 'I' | 'B' | 'D' | 'S', default text)]: every parameter SwarmLoop::Init reads, in its order (host/swarm_loop_params.hpp)"""
    buf = C.create_string_buffer(1 << 16)
    if lib().omni_swarm_params_table(buf, len(buf)):
        raise _err("omni_swarm_params_table")
    return [tuple(l.split("\t")) if l.count("\t") == 2 else tuple(l.split("\t")) + ("",) for l in buf.value.decode().split("\n") if l]


class KeyframePipeline:
    def __init__(self, device: int, sp_weights_path: str, pca_comp_csv: str, pca_mean_csv: str, vlad_weights_path: str, width=600, height=480,
                 thres=0.02, max_num=200, precision=capi.PREC_F16, microbatch=8, pipelines=0, storage=capi.STORE_F32, self_id=1,
                 inner_product_thres=0.3, init_mode_product_thres=0.2, match_index_dist=5, min_loop_num=30, min_direction_loop=3, geometry=False, pinhole_depth=None):
        """pinhole_depth: None = CameraConfig::STEREO_FISHEYE (4 directions x up/down views per key frame); a dict(fx, fy, cx, cy, depth_near, depth_far,
        accept_min_3d_pts) = CameraConfig::PINHOLE_DEPTH (launch/realsense.launch): one gray image + one depth image (set_depth) per key frame.
        pipelines <= 0: the library's default number of units in flight for the precision (4 for fp16, 2 otherwise)"""
        self.microbatch = microbatch
        common = (device, sp_weights_path.encode(), pca_comp_csv.encode(), pca_mean_csv.encode(), vlad_weights_path.encode(), width, height, thres, max_num,
                  precision, microbatch, pipelines, storage, self_id, inner_product_thres, init_mode_product_thres, match_index_dist, min_loop_num,
                  min_direction_loop, int(geometry))
        if pinhole_depth is None:
            self.h = lib().omni_pipeline_create(*common)
        else:
            d = pinhole_depth
            self.h = lib().omni_pipeline_create_pinhole_depth(*common, d["fx"], d["fy"], d["cx"], d["cy"], d.get("depth_near", 0.3), d.get("depth_far", 10.0),
                                                              d.get("accept_min_3d_pts", 50))
        self._depth = None
        if not self.h:
            raise _err("omni_pipeline_create")

    @classmethod
    def from_launch(cls, device: int, launch_xml: str, sp_weights_path: str, vlad_weights_path: str, pca_comp_csv: str | None = None, pca_mean_csv: str | None = None,
                    precision=capi.PREC_F16, microbatch=8, pipelines=0, storage=capi.STORE_F32, geometry=False, node_name="swarm_loop", args: dict | None = None,
                    intrinsics=None):
        """A pipeline configured by one of the reference's launch files (omni_pipeline_create_from_launch): image size, thresholds, camera configuration and
        self_id come from the file; weights, precision and batching are this build's."""
        self = cls.__new__(cls)
        a = "\n".join(f"{k}:={v}" for k, v in (args or {}).items()).encode() or None
        k4 = (C.c_double * 4)(*intrinsics) if intrinsics is not None else None
        self.microbatch = microbatch
        self._depth = None
        self.h = lib().omni_pipeline_create_from_launch(device, launch_xml.encode(), node_name.encode(), a, sp_weights_path.encode(), vlad_weights_path.encode(),
                                                       pca_comp_csv.encode() if pca_comp_csv else None, pca_mean_csv.encode() if pca_mean_csv else None, precision,
                                                       microbatch, pipelines, storage, int(geometry), k4)
        if not self.h:
            raise _err("omni_pipeline_create_from_launch")
        return self

    def set_depth(self, first_msg_id: int, depth: np.ndarray):
        """depth [n][H][W] u16 millimetres of key frames first_msg_id .. first_msg_id + n - 1 (kept alive by this object)"""
        self._depth = np.ascontiguousarray(depth, np.uint16)
        if lib().omni_pipeline_set_depth(self.h, first_msg_id, self._depth.shape[0], self._depth.ctypes.data):
            raise _err("omni_pipeline_set_depth")

    def close(self):
        if getattr(self, "h", None):
            lib().omni_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def attach_shard(self, rank: int, world: int, unique_id: bytes):
        """Collective: the database becomes the row-sharded index over `world` ranks (RCCL inside libomni_hip.so)."""
        if lib().omni_pipeline_attach_shard(self.h, rank, world, unique_id):
            raise _err("omni_pipeline_attach_shard")

    def preload(self, rows: np.ndarray):
        rows = np.ascontiguousarray(rows, np.float32)
        if lib().omni_pipeline_preload(self.h, rows.ctypes.data_as(C.POINTER(C.c_float)), rows.shape[0]):
            raise _err("omni_pipeline_preload")

    @property
    def db_rows(self) -> int:
        return lib().omni_pipeline_db_rows(self.h)

    def run(self, n_keyframes: int, first_msg_id: int, pool_ptrs, first_slot: int = 0, tail_ptr=None, from_host: bool = True) -> int:
        """pool_ptrs: addresses (ints) of the micro-batch image blocks -- pinned host memory (from_host) or HBM."""
        arr = (C.c_void_p * len(pool_ptrs))(*pool_ptrs)
        hits = C.c_int(0)
        if lib().omni_pipeline_run(self.h, n_keyframes, first_msg_id, arr, len(pool_ptrs), first_slot, tail_ptr, int(from_host), C.byref(hits)):
            raise _err("omni_pipeline_run")
        return hits.value

    def push_keyframe(self, images, msg_id: int, stamp: float, pose7=None, prevent_adding_db: bool = False, depth: np.ndarray = None) -> int:
        """The streaming intake: one key frame (images: list of u8 arrays [H][W], up cameras then down cameras; PINHOLE_DEPTH: one) with its id, stamp,
        odometry pose and prevent_adding_db flag (swarm_loop.cpp:140-170); returns the loop candidates found by the units this call finished.  The
        images are copied before the call returns; a depth image must stay alive until flush()."""
        imgs = [np.ascontiguousarray(i, np.uint8) for i in images]
        arr = (C.c_void_p * len(imgs))(*[i.ctypes.data for i in imgs])
        p7 = None
        if pose7 is not None:
            p7v = np.ascontiguousarray(pose7, np.float64).reshape(7)
            p7 = p7v.ctypes.data_as(C.POINTER(C.c_double))
        if depth is not None:
            depth = np.ascontiguousarray(depth, np.uint16)
            self._depths = getattr(self, "_depths", []) + [depth]
        hits = C.c_int(0)
        if lib().omni_pipeline_push_keyframe(self.h, arr, imgs[0].shape[1], msg_id, float(stamp), p7, int(prevent_adding_db),
                                             depth.ctypes.data if depth is not None else None, C.byref(hits)):
            raise _err("omni_pipeline_push_keyframe")
        return hits.value

    def poll(self) -> int:
        """The streaming intake's latency bound (call from a timer or after every push; never waits for a CNN unit): sends a partly filled micro-batch
        older than max_wait_ms, finishes the units the GPU is done with, and -- with nothing left in flight -- collects the last detector step."""
        hits = C.c_int(0)
        if lib().omni_pipeline_poll(self.h, C.byref(hits)):
            raise _err("omni_pipeline_poll")
        return hits.value

    def set_latency(self, max_wait_ms: float = 50.0, dispatch_when_idle: bool = True):
        if lib().omni_pipeline_set_latency(self.h, float(max_wait_ms), int(dispatch_when_idle)):
            raise _err("omni_pipeline_set_latency")

    def units(self):
        """(units in flight, how the last run() ordered them: 0 = kernels take turns, 1 / 2 = oldest first)"""
        f = C.c_int(0)
        n = lib().omni_pipeline_units(self.h, C.byref(f))
        return n, f.value

    def flush(self) -> int:
        hits = C.c_int(0)
        if lib().omni_pipeline_flush(self.h, C.byref(hits)):
            raise _err("omni_pipeline_flush")
        self._depths = []
        return hits.value

    def host_times(self, reset: bool = True) -> dict:
        """the host thread's milliseconds per unit (micro-batch): enqueue, wait_gpu, messages, detector, geometry; `units` = how many were averaged"""
        out = (C.c_double * 5)()
        n = lib().omni_pipeline_host_times(self.h, out, int(reset))
        return dict(zip(("enqueue", "wait_gpu", "messages", "detector", "geometry"), [round(v, 4) for v in out]), units=n)

    def prepare(self, n_keyframes: int):
        if lib().omni_pipeline_prepare(self.h, n_keyframes):
            raise _err("omni_pipeline_prepare")

    def geometry_stats(self):
        """(candidates handed to compute_loop, loop edges accepted)"""
        a, b = C.c_int(0), C.c_int(0)
        lib().omni_pipeline_geometry_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def set_poses(self, first_msg_id: int, poses7: np.ndarray):
        """odometry poses [n][7] = position xyz + quaternion wxyz of key frames first_msg_id .. first_msg_id + n - 1"""
        p = np.ascontiguousarray(poses7, np.float64).reshape(-1, 7)
        if lib().omni_pipeline_set_poses(self.h, first_msg_id, len(p), p.ctypes.data_as(C.POINTER(C.c_double))):
            raise _err("omni_pipeline_set_poses")

    def candidates(self) -> np.ndarray:
        """[n][4] = (new key frame, old key frame, direction_new, direction_old) of every loop candidate so far"""
        n = lib().omni_pipeline_get_candidates(self.h, None, 0)
        out = np.zeros((max(n, 1), 4), np.int64)
        lib().omni_pipeline_get_candidates(self.h, out.ctypes.data_as(C.POINTER(C.c_int64)), n)
        return out[:n]

    def edges(self) -> np.ndarray:
        """[n][12] = (keyframe_id_a, keyframe_id_b, drone_id_a, drone_id_b, pnp inliers, relative position xyz, relative quaternion wxyz)"""
        n = lib().omni_pipeline_get_edges(self.h, None, 0)
        out = np.zeros((max(n, 1), 12), np.float64)
        lib().omni_pipeline_get_edges(self.h, out.ctypes.data_as(C.POINTER(C.c_double)), n)
        return out[:n]

    def latencies_ms(self, reset: bool = True) -> np.ndarray:
        """latency of every micro-batch since the last reset: upload start -> detector (+ geometry) done, milliseconds"""
        n = lib().omni_pipeline_get_latencies(self.h, None, 0, 0)
        out = np.zeros(max(n, 1), np.float64)
        lib().omni_pipeline_get_latencies(self.h, out.ctypes.data_as(C.POINTER(C.c_double)), n, int(reset))
        return out[:n]

    def apply_launch(self, launch_xml: str, node_name: str = "swarm_loop"):
        """the detector's and the geometry stage's thresholds from one of the reference's launch files (omni_pipeline_apply_launch)"""
        if lib().omni_pipeline_apply_launch(self.h, launch_xml.encode(), node_name.encode()):
            raise _err("omni_pipeline_apply_launch")

    def exchange_us(self, reset: bool = True) -> np.ndarray:
        """sharded mode: [n][2] device microseconds of the two all-gathers (new rows, per-shard top-k lists) of every exchange unit since the last reset"""
        n = lib().omni_pipeline_get_exchange_us(self.h, None, 0, 0)
        out = np.zeros((max(n, 1), 2), np.float32)
        lib().omni_pipeline_get_exchange_us(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), n, int(reset))
        return out[:n]

    def sync(self):
        if lib().omni_pipeline_sync(self.h):
            raise _err("omni_pipeline_sync")
