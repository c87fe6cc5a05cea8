"""ctypes binding of include/omni_hip.h plus thin numpy-friendly wrappers that mirror the reference classes.

Fails loudly: a missing ``lib/libomni_hip.so`` raises ImportError-like OSError from ``lib()``, a missing GPU raises
``OmniError`` from ``Context()``.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OMNI_LIB") or os.path.join(_HERE, "lib", "libomni_hip.so")   # OMNI_LIB: A/B a second build of the same ABI

OK, ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_CAPACITY = 0, 1, 2, 3, 4
PREC_F32, PREC_F16, PREC_SPLIT = 0, 1, 2
STORE_F32, STORE_F16 = 0, 1
BF_OPENCV, BF_MUTUAL = 0, 1
ABI_VERSION = 2               # include/omni_hip.h OMNI_ABI_VERSION: checked when the library is loaded
SP_NUM_LAYERS = 12
SP_NUM_STAGES = 16
SP_LAYER_NAMES = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b",
                  "convPa", "convPb", "convDa", "convDb"]
VLAD_KINDS = {"conv3x3": 0, "pw_relu6": 1, "dw3x3_relu6": 2, "pw_linear": 3, "pw_linear_res": 4}

# every symbol include/omni_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "omni_abi_version", "omni_last_error", "omni_ctx_create", "omni_ctx_create_priority", "omni_ctx_order_after", "omni_memcpy_d2h_async", "omni_ctx_destroy", "omni_ctx_sync", "omni_ctx_stream",
    "omni_ctx_device_info", "omni_ctx_mfma_ceiling", "omni_dev_alloc", "omni_dev_free", "omni_host_alloc", "omni_host_free", "omni_memcpy_h2d", "omni_memcpy_d2h", "omni_timer_start",
    "omni_timer_stop", "omni_sp_create", "omni_sp_destroy", "omni_sp_desc_dim", "omni_sp_image_size", "omni_sp_infer", "omni_sp_enqueue_dev",
    "omni_sp_fetch", "omni_sp_dev_outputs", "omni_sp_get_dense", "omni_sp_postprocess_dense", "omni_sp_debug_layer",
    "omni_sp_profile", "omni_sp_stage_name", "omni_sp_stage_flops", "omni_sp_stage_tiles_left_out", "omni_sp_mask_skip_plan", "omni_vlad_create", "omni_vlad_destroy", "omni_vlad_set_precision", "omni_vlad_pack_block", "omni_sp_pack_constants",
    "omni_vlad_infer", "omni_vlad_enqueue_dev", "omni_vlad_fetch", "omni_vlad_dev_output", "omni_vlad_mask_skip_layers", "omni_index_create",
    "omni_index_destroy", "omni_index_add", "omni_index_add_dev", "omni_index_ntotal", "omni_index_dim", "omni_index_reset", "omni_index_truncate", "omni_index_cert_stats",
    "omni_index_search", "omni_index_search_dev", "omni_index_search_prefix_dev", "omni_index_search_batch_prefix_dev", "omni_index_set_shard", "omni_topk_merge", "omni_index_last_scan_ms",
    "omni_index_save", "omni_index_load",
    "omni_trace_push", "omni_trace_pop", "omni_sp_set_perf", "omni_sp_last_stage_ms", "omni_bf_match", "omni_bf_match_multi", "omni_bf_match_batched_dev", "omni_config_count", "omni_config_describe", "omni_config_value", "omni_config_is_process_wide", "omni_cam_create", "omni_cam_create_mono", "omni_cam_destroy", "omni_cam_enqueue_dev", "omni_cam_enqueue_host", "omni_cam_enqueue_host_parts", "omni_cam_wait", "omni_cam_order_after", "omni_cam_set_active", "omni_cam_ready",
    "omni_shard_unique_id", "omni_shard_library_path", "omni_shard_create", "omni_shard_destroy", "omni_shard_ntotal", "omni_shard_preload_local", "omni_shard_step_batch_dev", "omni_shard_step_enqueue", "omni_shard_rows_consumed", "omni_shard_step_wait", "omni_shard_last_exchange_us",
    "omni_shard_search", "omni_flatten_create", "omni_flatten_destroy", "omni_flatten_out_bytes", "omni_flatten_enqueue_dev",
]


class OmniError(RuntimeError):
    pass


class _SpWeights(C.Structure):
    _fields_ = [("weight", C.POINTER(C.c_float) * SP_NUM_LAYERS), ("bias", C.POINTER(C.c_float) * SP_NUM_LAYERS)]


class _VladLayer(C.Structure):
    _fields_ = [("kind", C.c_int), ("cin", C.c_int), ("cout", C.c_int), ("stride", C.c_int),
                ("weight", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_float))]


class _CamResult(C.Structure):
    _fields_ = [("n_dirs", C.c_int), ("max_num", C.c_int), ("desc_dim", C.c_int), ("global_dim", C.c_int),
                ("kps_xy", C.POINTER(C.c_float)), ("n_kps", C.POINTER(C.c_int)), ("desc", C.POINTER(C.c_float)),
                ("scores", C.POINTER(C.c_float)), ("global_desc", C.POINTER(C.c_float)), ("match_up", C.POINTER(C.c_int)),
                ("match_down", C.POINTER(C.c_int)), ("match_dist", C.POINTER(C.c_float)), ("n_matches", C.POINTER(C.c_int)), ("n_images", C.c_int)]


class _VladWeights(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("layers", C.POINTER(_VladLayer)), ("n_clusters", C.c_int), ("feat_dim", C.c_int),
                ("out_dim", C.c_int), ("assign_w", C.POINTER(C.c_float)), ("assign_b", C.POINTER(C.c_float)),
                ("clusters", C.POINTER(C.c_float)), ("fc_w", C.POINTER(C.c_float)), ("fc_b", C.POINTER(C.c_float))]


_lib = None
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)
_vp = C.c_void_p


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"(or `make -C omni-swarm_amd`); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)

    def sig(name, res, args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args

    sig("omni_abi_version", C.c_int, [])
    sig("omni_last_error", C.c_char_p, [])
    sig("omni_ctx_create", _vp, [C.c_int])
    sig("omni_ctx_create_priority", _vp, [C.c_int, C.c_int])
    sig("omni_ctx_order_after", C.c_int, [_vp, _vp])
    sig("omni_memcpy_d2h_async", C.c_int, [_vp, _vp, _vp, C.c_size_t])
    sig("omni_ctx_destroy", None, [_vp])
    sig("omni_ctx_sync", C.c_int, [_vp])
    sig("omni_ctx_stream", _vp, [_vp])
    sig("omni_ctx_device_info", C.c_int, [_vp, C.c_char_p, C.c_int, _ip, _ip, C.POINTER(C.c_size_t)])
    sig("omni_ctx_mfma_ceiling", C.c_int, [_vp, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)])
    sig("omni_dev_alloc", _vp, [_vp, C.c_size_t])
    sig("omni_dev_free", C.c_int, [_vp, _vp])
    sig("omni_host_alloc", _vp, [C.c_size_t])
    sig("omni_host_free", C.c_int, [_vp])
    sig("omni_memcpy_h2d", C.c_int, [_vp, _vp, _vp, C.c_size_t])
    sig("omni_memcpy_d2h", C.c_int, [_vp, _vp, _vp, C.c_size_t])
    sig("omni_timer_start", C.c_int, [_vp])
    sig("omni_timer_stop", C.c_int, [_vp, _fp])
    sig("omni_sp_create", _vp, [_vp, C.POINTER(_SpWeights), _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                C.c_int, C.c_int])
    sig("omni_sp_destroy", None, [_vp])
    sig("omni_sp_desc_dim", C.c_int, [_vp])
    sig("omni_sp_infer", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _fp, _ip, _fp, _fp])
    sig("omni_sp_enqueue_dev", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int])
    sig("omni_sp_fetch", C.c_int, [_vp, C.c_int, _fp, _ip, _fp, _fp])
    sig("omni_sp_dev_outputs", C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)])
    sig("omni_sp_get_dense", C.c_int, [_vp, C.c_int, _fp, _fp])
    sig("omni_sp_postprocess_dense", C.c_int, [_vp, _fp, _fp, C.c_int, _fp, _ip, _fp, _fp])
    sig("omni_sp_debug_layer", C.c_int, [_vp, C.c_char_p, C.c_int, _fp, _ip, _ip, _ip])
    sig("omni_sp_profile", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _fp])
    sig("omni_sp_stage_name", C.c_char_p, [C.c_int])
    sig("omni_sp_stage_flops", C.c_double, [_vp, C.c_int])
    sig("omni_sp_stage_tiles_left_out", C.c_double, [_vp, C.c_int])
    sig("omni_sp_mask_skip_plan", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)])
    sig("omni_vlad_create", _vp, [_vp, C.POINTER(_VladWeights), C.c_int, C.c_int, C.c_int])
    sig("omni_vlad_destroy", None, [_vp])
    sig("omni_vlad_set_precision", C.c_int, [_vp, C.c_int])
    sig("omni_sp_pack_constants", C.c_int64, [C.c_int, _fp, _fp, C.c_int, _vp, C.c_int64, C.POINTER(C.c_float)])
    sig("omni_vlad_pack_block", C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _vp, C.c_int64])
    sig("omni_vlad_infer", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _fp])
    sig("omni_vlad_enqueue_dev", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int])
    sig("omni_vlad_fetch", C.c_int, [_vp, C.c_int, _fp])
    sig("omni_vlad_dev_output", C.c_int, [_vp, C.POINTER(_vp)])
    sig("omni_vlad_mask_skip_layers", C.c_int, [_vp, C.POINTER(C.c_double), C.c_int])
    sig("omni_index_create", _vp, [_vp, C.c_int, C.c_int, C.c_int64])
    sig("omni_index_destroy", None, [_vp])
    sig("omni_index_add", C.c_int, [_vp, C.c_int64, _fp])
    sig("omni_index_add_dev", C.c_int, [_vp, C.c_int64, _vp])
    sig("omni_index_ntotal", C.c_int64, [_vp])
    sig("omni_index_reset", C.c_int, [_vp])
    sig("omni_index_truncate", C.c_int, [_vp, C.c_int64])
    sig("omni_index_cert_stats", C.c_int, [_vp, _i64p, _i64p])
    sig("omni_index_dim", C.c_int, [_vp])
    sig("omni_sp_image_size", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)])
    sig("omni_index_search", C.c_int, [_vp, C.c_int, _fp, C.c_int, _fp, _i64p])
    sig("omni_index_search_dev", C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, _vp])
    sig("omni_index_search_prefix_dev", C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int64, _vp, _vp])
    sig("omni_index_search_batch_prefix_dev", C.c_int, [_vp, C.c_int, _vp, _i64p, C.c_int, _i64p, _vp, _vp])
    sig("omni_index_set_shard", C.c_int, [_vp, C.c_int, C.c_int])
    sig("omni_topk_merge", C.c_int, [C.c_int, C.c_int, C.c_int, _fp, _i64p, C.c_int, _fp, _i64p])
    sig("omni_index_last_scan_ms", C.c_int, [_vp, _fp])
    sig("omni_index_save", C.c_int, [_vp, C.c_char_p])
    sig("omni_index_load", C.c_int, [_vp, C.c_char_p])
    sig("omni_bf_match", C.c_int, [_vp, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, _ip, _ip, _fp, _ip])
    sig("omni_bf_match_multi", C.c_int, [_vp, C.c_int, C.POINTER(_fp), _ip, C.POINTER(_fp), _ip, C.c_int, C.c_int, C.c_int, _ip, _ip, _fp, _ip])
    sig("omni_bf_match_batched_dev", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int64, _vp, _vp, C.c_int64,
                                               _vp, _vp, _vp, _vp, _vp])
    sig("omni_cam_create", _vp, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int])
    sig("omni_cam_create_mono", _vp, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int])
    sig("omni_cam_destroy", None, [_vp])
    sig("omni_cam_enqueue_dev", C.c_int, [_vp, _vp, C.c_int, C.c_int])
    sig("omni_trace_push", None, [C.c_char_p])
    sig("omni_config_is_process_wide", C.c_int, [C.c_int])
    sig("omni_sp_set_perf", C.c_int, [_vp, C.c_int])
    sig("omni_sp_last_stage_ms", C.c_int, [_vp, _vp])
    sig("omni_trace_pop", None, [])
    sig("omni_cam_enqueue_host", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int])
    sig("omni_cam_enqueue_host_parts", C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int])
    sig("omni_cam_wait", C.c_int, [_vp, C.POINTER(_CamResult)])
    sig("omni_cam_order_after", C.c_int, [_vp, _vp, C.c_int])
    sig("omni_cam_set_active", C.c_int, [_vp, C.c_int])
    sig("omni_cam_ready", C.c_int, [_vp, C.POINTER(C.c_int)])
    sig("omni_flatten_create", _vp, [_vp, C.c_int, C.c_int, C.c_int, _ip, _ip, C.POINTER(_fp)])
    sig("omni_flatten_destroy", None, [_vp])
    sig("omni_flatten_out_bytes", C.c_int64, [_vp])
    sig("omni_flatten_enqueue_dev", C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp])
    sig("omni_config_count", C.c_int, [])
    sig("omni_config_describe", C.c_int, [C.c_int, C.POINTER(C.c_char_p), _ip, _ip, _ip, _ip, C.POINTER(C.c_char_p)])
    sig("omni_config_value", C.c_int, [C.c_char_p, _ip])
    sig("omni_shard_unique_id", C.c_int, [C.c_char_p])
    sig("omni_shard_library_path", C.c_int, [C.c_char_p, C.c_int])
    sig("omni_shard_create", _vp, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_char_p])
    sig("omni_shard_destroy", None, [_vp])
    sig("omni_shard_ntotal", C.c_int64, [_vp])
    sig("omni_shard_preload_local", C.c_int, [_vp, _fp, C.c_int64, C.c_int64])
    sig("omni_shard_step_batch_dev", C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _fp, _i64p])
    sig("omni_shard_step_enqueue", C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int])
    sig("omni_shard_rows_consumed", C.c_int, [_vp])
    sig("omni_shard_step_wait", C.c_int, [_vp, _fp, _i64p])
    sig("omni_shard_last_exchange_us", C.c_int, [_vp, _fp, _fp])
    sig("omni_shard_search", C.c_int, [_vp, C.c_int, _fp, C.c_int, _fp, _i64p])
    if L.omni_abi_version() != ABI_VERSION:
        raise OmniError("libomni_hip.so ABI version mismatch")
    _lib = L
    return L


def _check(rc):
    if rc != OK:
        raise OmniError(f"omni error {rc}: {lib().omni_last_error().decode()}")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _pf(a):
    return a.ctypes.data_as(_fp)


class Context:
    """One HIP stream + scratch on one GPU (omni_ctx)."""

    def __init__(self, device_id: int = 0, high_priority: bool = False):
        import weakref
        self.device_id = device_id
        self._children = weakref.WeakSet()      # handles created on this context: closed before the context is (their destroy uses its stream)
        self.h = lib().omni_ctx_create_priority(device_id, 1) if high_priority else lib().omni_ctx_create(device_id)
        if not self.h:
            raise OmniError(f"omni_ctx_create failed: {lib().omni_last_error().decode()}")

    def _adopt(self, child):
        self._children.add(child)

    def close(self):
        if self.h:
            for kind in ("Cam", "Shard", None):     # omni_cam / omni_shard borrow other handles: they go first
                for ch in list(self._children):
                    if kind is None or type(ch).__name__ == kind:   # noqa: E721
                        try:
                            ch.close()
                        except Exception:
                            pass
            lib().omni_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(lib().omni_ctx_sync(self.h))

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu, mhz, mem = C.c_int(), C.c_int(), C.c_size_t()
        _check(lib().omni_ctx_device_info(self.h, name, 256, C.byref(ncu), C.byref(mhz), C.byref(mem)))
        return {"name": name.value.decode(), "n_cu": ncu.value, "clock_mhz": mhz.value, "hbm_bytes": mem.value}

    def mfma_ceiling(self, ms: float = 100.0):
        """what the fp16 matrix cores of this board sustain (calibration: back-to-back MFMAs on every SIMD for ~ms milliseconds)"""
        tf, ghz = C.c_float(), C.c_float()
        _check(lib().omni_ctx_mfma_ceiling(self.h, C.c_float(ms), C.byref(tf), C.byref(ghz)))
        return {"tflops": tf.value, "sclk_ghz": ghz.value}

    def alloc(self, nbytes: int) -> int:
        p = lib().omni_dev_alloc(self.h, nbytes)
        if not p:
            raise OmniError(f"omni_dev_alloc({nbytes}) failed: {lib().omni_last_error().decode()}")
        return p

    def free(self, p):
        _check(lib().omni_dev_free(self.h, p))

    def to_device(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        p = self.alloc(max(arr.nbytes, 1))
        _check(lib().omni_memcpy_h2d(self.h, p, arr.ctypes.data_as(_vp), arr.nbytes))
        return p

    def from_device(self, p: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype)
        _check(lib().omni_memcpy_d2h(self.h, out.ctypes.data_as(_vp), p, out.nbytes))
        return out

    def host_alloc(self, shape, dtype=np.uint8) -> np.ndarray:
        """Pinned host array (hipHostMalloc): the source of asynchronous uploads (omni_cam_enqueue_host).  The array keeps no
        reference to this context; free with host_free()."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = lib().omni_host_alloc(n)
        if not p:
            raise OmniError(f"omni_host_alloc({n}) failed: {lib().omni_last_error().decode()}")
        buf = (C.c_uint8 * n).from_address(p)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr: np.ndarray):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p:
            _check(lib().omni_host_free(p))

    def timer_start(self):
        _check(lib().omni_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        _check(lib().omni_timer_stop(self.h, C.byref(ms)))
        return ms.value


class SuperPoint:
    """SuperPointTensorRT(engine, pca_comp, pca_mean, width, height, thres, max_num) -- weights replace the engine file."""

    def __init__(self, ctx: Context, weights: dict, pca_comp, pca_mean, width: int, height: int, thres: float = 0.015,
                 max_num: int = 200, precision: int = PREC_F32, max_batch: int = 1):
        self.ctx, self.W, self.H, self.max_num, self.max_batch, self.precision = ctx, width, height, max_num, max_batch, precision
        self._keep = []
        w = _SpWeights()
        for i, n in enumerate(SP_LAYER_NAMES):
            wt, bs = _f32(weights[n + ".weight"]), _f32(weights[n + ".bias"])
            self._keep += [wt, bs]
            w.weight[i], w.bias[i] = _pf(wt), _pf(bs)
        pc = pm = None
        pca_dim = 0
        if pca_comp is not None:
            pc, pm = _f32(pca_comp), _f32(pca_mean)
            pca_dim = pc.shape[0]
        self.h = lib().omni_sp_create(ctx.h, C.byref(w), _pf(pc) if pc is not None else None,
                                      _pf(pm) if pm is not None else None, pca_dim, width, height, thres, max_num,
                                      precision, max_batch)
        if not self.h:
            raise OmniError(f"omni_sp_create failed: {lib().omni_last_error().decode()}")
        ctx._adopt(self)
        self.desc_dim = lib().omni_sp_desc_dim(self.h)

    def close(self):
        if getattr(self, "h", None):
            lib().omni_sp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _outs(self, batch):
        return (np.zeros((batch, self.max_num, 2), np.float32), np.zeros(batch, np.int32),
                np.zeros((batch, self.max_num, self.desc_dim), np.float32), np.zeros((batch, self.max_num), np.float32))

    @staticmethod
    def _split(kps, n, desc, sc):
        return [(kps[b, :n[b]].copy(), desc[b, :n[b]].copy(), sc[b, :n[b]].copy()) for b in range(len(n))]

    def inference(self, gray_u8: np.ndarray, fisheye_mask: bool = False):
        """[H,W] or [B,H,W] uint8 -> list of (kps [n,2] (x,y), desc [n,D], scores [n]) per image."""
        g = np.ascontiguousarray(gray_u8, np.uint8)
        if g.ndim == 2:
            g = g[None]
        b = g.shape[0]
        kps, n, desc, sc = self._outs(b)
        _check(lib().omni_sp_infer(self.h, g.ctypes.data_as(_vp), g.shape[2], b, int(fisheye_mask), _pf(kps),
                                   n.ctypes.data_as(_ip), _pf(desc), _pf(sc)))
        return self._split(kps, n, desc, sc)

    def enqueue_dev(self, gray_dev: int, stride: int, batch: int, fisheye_mask: bool = False):
        _check(lib().omni_sp_enqueue_dev(self.h, gray_dev, stride, batch, int(fisheye_mask)))

    def fetch(self, batch: int):
        kps, n, desc, sc = self._outs(batch)
        _check(lib().omni_sp_fetch(self.h, batch, _pf(kps), n.ctypes.data_as(_ip), _pf(desc), _pf(sc)))
        return self._split(kps, n, desc, sc)

    def dev_outputs(self):
        k, n, d, s = _vp(), _vp(), _vp(), _vp()
        _check(lib().omni_sp_dev_outputs(self.h, C.byref(k), C.byref(n), C.byref(d), C.byref(s)))
        return k.value, n.value, d.value, s.value

    def get_dense(self, batch: int):
        semi = np.empty((batch, self.H, self.W), np.float32)
        desc = np.empty((batch, 256, self.H // 8, self.W // 8), np.float32)
        _check(lib().omni_sp_get_dense(self.h, batch, _pf(semi), _pf(desc)))
        return semi, desc

    def postprocess_dense(self, semi: np.ndarray, desc: np.ndarray):
        semi, desc = _f32(semi), _f32(desc)
        if semi.ndim == 2:
            semi, desc = semi[None], desc[None]
        b = semi.shape[0]
        kps, n, d, sc = self._outs(b)
        _check(lib().omni_sp_postprocess_dense(self.h, _pf(semi), _pf(desc), b, _pf(kps), n.ctypes.data_as(_ip), _pf(d), _pf(sc)))
        return self._split(kps, n, d, sc)

    def debug_layer(self, name: str, batch: int = 1) -> np.ndarray:
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        _check(lib().omni_sp_debug_layer(self.h, name.encode(), batch, None, C.byref(c), C.byref(h), C.byref(w)))
        out = np.empty((batch, c.value, h.value, w.value), np.float32)
        _check(lib().omni_sp_debug_layer(self.h, name.encode(), batch, _pf(out), C.byref(c), C.byref(h), C.byref(w)))
        return out

    def profile(self, gray_dev: int, stride: int, batch: int, reps: int = 5):
        ms = np.zeros(SP_NUM_STAGES, np.float32)
        _check(lib().omni_sp_profile(self.h, gray_dev, stride, batch, reps, _pf(ms)))
        out = []
        for i in range(SP_NUM_STAGES):
            name = lib().omni_sp_stage_name(i).decode()
            if name:
                out.append({"stage": name, "ms": float(ms[i]), "flops_per_image": lib().omni_sp_stage_flops(self.h, i),
                            "tiles_left_out": lib().omni_sp_stage_tiles_left_out(self.h, i)})
        return out


class MobileNetVLAD:
    """MobileNetVLADTensorRT(engine, width, height); weights (ASSUMED architecture) replace the engine file."""

    def __init__(self, ctx: Context, weights: dict, layer_specs, n_clusters: int, feat_dim: int, out_dim: int,
                 width: int, height: int, max_batch: int = 1):
        self.ctx, self.W, self.H, self.out_dim, self.max_batch = ctx, width, height, out_dim, max_batch
        self._keep = []
        layers = (_VladLayer * len(layer_specs))()
        for i, (name, kind, cin, cout, stride) in enumerate(layer_specs):
            wt, bs = _f32(weights[name + ".weight"]), _f32(weights[name + ".bias"])
            self._keep += [wt, bs]
            layers[i] = _VladLayer(VLAD_KINDS[kind], cin, cout, stride, _pf(wt), _pf(bs))
        aw, ab = _f32(weights["vlad.assign.weight"]).reshape(n_clusters, feat_dim), _f32(weights["vlad.assign.bias"])
        cl, fw, fb = _f32(weights["vlad.clusters"]), _f32(weights["fc.weight"]), _f32(weights["fc.bias"])
        self._keep += [aw, ab, cl, fw, fb, layers]
        vw = _VladWeights(len(layer_specs), layers, n_clusters, feat_dim, out_dim, _pf(aw), _pf(ab), _pf(cl), _pf(fw), _pf(fb))
        self.h = lib().omni_vlad_create(ctx.h, C.byref(vw), width, height, max_batch)
        if not self.h:
            raise OmniError(f"omni_vlad_create failed: {lib().omni_last_error().decode()}")
        ctx._adopt(self)

    def close(self):
        if getattr(self, "h", None):
            lib().omni_vlad_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_precision(self, precision: int):
        """PREC_F32 (default, parity mode) or PREC_F16 (fp16 matrix-core operands in the inverted-residual blocks)."""
        _check(lib().omni_vlad_set_precision(self.h, int(precision)))

    def inference(self, gray_u8: np.ndarray, fisheye_mask: bool = False) -> np.ndarray:
        g = np.ascontiguousarray(gray_u8, np.uint8)
        if g.ndim == 2:
            g = g[None]
        out = np.empty((g.shape[0], self.out_dim), np.float32)
        _check(lib().omni_vlad_infer(self.h, g.ctypes.data_as(_vp), g.shape[2], g.shape[0], int(fisheye_mask), _pf(out)))
        return out

    def enqueue_dev(self, gray_dev: int, stride: int, batch: int, fisheye_mask: bool = False):
        _check(lib().omni_vlad_enqueue_dev(self.h, gray_dev, stride, batch, int(fisheye_mask)))

    def fetch(self, batch: int) -> np.ndarray:
        out = np.empty((batch, self.out_dim), np.float32)
        _check(lib().omni_vlad_fetch(self.h, batch, _pf(out)))
        return out

    def dev_output(self) -> int:
        p = _vp()
        _check(lib().omni_vlad_dev_output(self.h, C.byref(p)))
        return p.value

    def mask_skip_layers(self) -> list:
        """Share of the tiles of the stem (+ block 0) and of blocks 1, 2, ... that a fisheye-masked pass leaves out (the mask's constant region)."""
        f = (C.c_double * 32)()
        n = lib().omni_vlad_mask_skip_layers(self.h, f, 32)
        return [f[i] for i in range(min(n, 32))]


def sp_pack_constants(which, w, bias=None, cout=64):
    """Host-only test hook: (halfs as uint16, scale) of conv1a's byte-operand fragments (which = 0) or a cin = 64 layer's Winograd fragments (which = 1)."""
    w = _f32(w)
    b = _f32(bias) if bias is not None else None
    out = np.zeros(2048 if which == 0 else 64 * cout * 32, np.uint16)
    sc = C.c_float(0)
    n = lib().omni_sp_pack_constants(which, _pf(w), _pf(b) if b is not None else None, cout, out.ctypes.data_as(_vp), out.size, C.byref(sc))
    if n != out.size:
        raise OmniError(f"omni_sp_pack_constants failed: {lib().omni_last_error().decode()}")
    return out, float(sc.value)


def vlad_pack_block(cin, hid, cout, stride, we, be, wd, bd, wp):
    """Host-only test hook: the split-fp16 weight blob of one inverted-residual block (bytes), or None when no kernel covers the shape."""
    we, be, wd, bd, wp = (_f32(x) for x in (we, be, wd, bd, wp))
    need = lib().omni_vlad_pack_block(cin, hid, cout, stride, None, None, None, None, None, None, 0)
    if need < 0:
        return None
    out = np.zeros(need, np.uint8)
    got = lib().omni_vlad_pack_block(cin, hid, cout, stride, _pf(we), _pf(be), _pf(wd), _pf(bd), _pf(wp), out.ctypes.data_as(_vp), need)
    if got != need:
        raise OmniError(f"omni_vlad_pack_block failed: {lib().omni_last_error().decode()}")
    return out


class IndexFlatIP:
    """faiss::IndexFlatIP(d): add / search / ntotal (+ row sharding)."""

    def __init__(self, ctx: Context, d: int = 4096, storage: int = STORE_F32, capacity: int = 0):
        self.ctx, self.d = ctx, d
        self.h = lib().omni_index_create(ctx.h, d, storage, capacity)
        if not self.h:
            raise OmniError(f"omni_index_create failed: {lib().omni_last_error().decode()}")
        ctx._adopt(self)

    def close(self):
        if getattr(self, "h", None):
            lib().omni_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ntotal(self) -> int:
        return lib().omni_index_ntotal(self.h)

    def add(self, x: np.ndarray):
        x = _f32(np.atleast_2d(x))
        assert x.shape[1] == self.d
        _check(lib().omni_index_add(self.h, x.shape[0], _pf(x)))

    def add_dev(self, n: int, x_dev: int):
        _check(lib().omni_index_add_dev(self.h, n, x_dev))

    def reset(self):
        _check(lib().omni_index_reset(self.h))

    def cert_stats(self):
        """(queries answered through the fp16 mirror + exact refinement, of those the ones re-run with the exact scan)"""
        a, b = C.c_int64(0), C.c_int64(0)
        _check(lib().omni_index_cert_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def truncate(self, n_rows: int):
        _check(lib().omni_index_truncate(self.h, n_rows))

    def save(self, path: str):
        """Shard checkpoint (OMNX1): header + the raw row matrix as stored."""
        _check(lib().omni_index_save(self.h, path.encode()))

    def load(self, path: str):
        _check(lib().omni_index_load(self.h, path.encode()))

    def set_shard(self, rank: int, world: int):
        _check(lib().omni_index_set_shard(self.h, rank, world))

    def search(self, q: np.ndarray, k: int):
        q = _f32(np.atleast_2d(q))
        D = np.empty((q.shape[0], k), np.float32)
        I = np.empty((q.shape[0], k), np.int64)
        _check(lib().omni_index_search(self.h, q.shape[0], _pf(q), k, _pf(D), I.ctypes.data_as(_i64p)))
        return D, I

    def search_dev(self, nq: int, q_dev: int, k: int, D_dev: int, I_dev: int):
        _check(lib().omni_index_search_dev(self.h, nq, q_dev, k, D_dev, I_dev))

    def search_prefix_dev(self, nq: int, q_dev: int, k: int, n_limit: int, D_dev: int, I_dev: int):
        """search_dev over the first n_limit rows only (asynchronous; see include/omni_hip.h)."""
        _check(lib().omni_index_search_prefix_dev(self.h, nq, q_dev, k, n_limit, D_dev, I_dev))

    def search_batch_prefix_dev(self, rows_dev: int, row_idx, k: int, limits, D_dev: int, I_dev: int):
        """len(limits) <= 64 prefix searches in ONE pass over the shard: query j = row row_idx[j] of rows_dev (None: rows 0..), it sees
        local rows [0, limits[j]).  Asynchronous (see include/omni_hip.h)."""
        lim = np.ascontiguousarray(limits, np.int64)
        ri = None if row_idx is None else np.ascontiguousarray(row_idx, np.int64)
        _check(lib().omni_index_search_batch_prefix_dev(self.h, len(lim), rows_dev, None if ri is None else ri.ctypes.data_as(_i64p), k,
                                                        lim.ctypes.data_as(_i64p), D_dev, I_dev))

    def search_prefix_many(self, q: np.ndarray, k: int, limits) -> tuple:
        """q [F][nq][d], limits [F]: F searches, search f over this shard's first limits[f] rows only; all enqueued back to back,
        ONE host synchronisation.  Returns D [F][nq][k], I [F][nq][k] (global ids under set_shard)."""
        q = _f32(q)
        F, nq = q.shape[0], q.shape[1]
        D = np.full((F, nq, k), -3.4028235e38, np.float32)
        I = np.full((F, nq, k), -1, np.int64)
        live = [f for f in range(F) if int(limits[f]) > 0]
        if not live:
            return D, I
        qd = self.ctx.to_device(q)
        buf = self.ctx.alloc(F * nq * k * 12)
        try:
            for f in live:
                self.search_prefix_dev(nq, qd + f * nq * self.d * 4, k, int(limits[f]), buf + F * nq * k * 8 + f * nq * k * 4, buf + f * nq * k * 8)
            raw = self.ctx.from_device(buf, (F * nq * k * 12,), np.uint8)
        finally:
            self.ctx.free(qd)
            self.ctx.free(buf)
        Ia = raw[:F * nq * k * 8].view(np.int64).reshape(F, nq, k)
        Da = raw[F * nq * k * 8:].view(np.float32).reshape(F, nq, k)
        for f in live:
            D[f], I[f] = Da[f], Ia[f]
        return D, I

    def last_scan_ms(self) -> float:
        ms = C.c_float()
        _check(lib().omni_index_last_scan_ms(self.h, C.byref(ms)))
        return ms.value


class Flatten:
    """omni_flatten: FisheyeUndist::undist_all_cuda -- one launch remaps fisheye images into all virtual pinhole views (maps from
    omni_swarm_amd.flatten.generate_undist_maps or host/fisheye_flatten.hpp)."""

    def __init__(self, ctx: Context, src_width: int, src_height: int, maps):
        self.ctx = ctx
        self.maps = [np.ascontiguousarray(m, np.float32) for m in maps]          # [h][w][2]
        self.shapes = [(m.shape[0], m.shape[1]) for m in self.maps]
        n = len(self.maps)
        vw = (C.c_int * n)(*[s[1] for s in self.shapes])
        vh = (C.c_int * n)(*[s[0] for s in self.shapes])
        ptrs = (_fp * n)(*[_pf(m) for m in self.maps])
        self.h = lib().omni_flatten_create(ctx.h, src_width, src_height, n, vw, vh, ptrs)
        if not self.h:
            raise OmniError(f"omni_flatten_create failed: {lib().omni_last_error().decode()}")
        ctx._adopt(self)
        self.out_bytes = lib().omni_flatten_out_bytes(self.h)

    def close(self):
        if getattr(self, "h", None):
            lib().omni_flatten_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enqueue_dev(self, src_dev: int, src_stride: int, batch: int, out_dev: int):
        _check(lib().omni_flatten_enqueue_dev(self.h, src_dev, src_stride, batch, out_dev))

    def __call__(self, fisheye_u8: np.ndarray):
        """[H,W] or [B,H,W] uint8 -> list over images of lists of views (host convenience: upload, remap, download)."""
        g = np.ascontiguousarray(fisheye_u8, np.uint8)
        if g.ndim == 2:
            g = g[None]
        src = self.ctx.to_device(g)
        out = self.ctx.alloc(self.out_bytes * g.shape[0])
        try:
            self.enqueue_dev(src, g.shape[2], g.shape[0], out)
            raw = self.ctx.from_device(out, (g.shape[0], self.out_bytes), np.uint8)
        finally:
            self.ctx.free(src)
            self.ctx.free(out)
        res = []
        for b in range(g.shape[0]):
            views, o = [], 0
            for (h, w) in self.shapes:
                views.append(raw[b, o:o + h * w].reshape(h, w).copy())
                o += h * w
            res.append(views)
        return res


SHARD_ID_BYTES = 128


def shard_unique_id() -> bytes:
    """ncclGetUniqueId through the library: call on rank 0, carry the 128 bytes to the other ranks by any channel."""
    buf = C.create_string_buffer(SHARD_ID_BYTES)
    _check(lib().omni_shard_unique_id(buf))
    return buf.raw


def sp_mask_skip_plan(width: int, height: int, precision: int, layer: int):
    """((tile row 0, tile row 1, tile column 0, tile column 1), share of the layer's tiles) a fisheye-masked pass leaves out of layer `layer`
    (0 = conv1a [split only], 1..5 = conv1b, conv2a, conv2b, conv3a, conv3b): csrc/superpoint.hip, pure arithmetic (no device)."""
    rect = (C.c_int * 4)()
    frac = C.c_double()
    _check(lib().omni_sp_mask_skip_plan(width, height, precision, layer, rect, C.byref(frac)))
    return tuple(rect), frac.value


def config_table() -> list:
    """csrc/config.h's table: [{env, default, lo, hi, cls, doc}] (cls: 0 variant, 1 tuning, 2 debug, 3 test, 4 string)"""
    out = []
    for i in range(lib().omni_config_count()):
        env, doc = C.c_char_p(), C.c_char_p()
        d, lo, hi, cls = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _check(lib().omni_config_describe(i, C.byref(env), C.byref(d), C.byref(lo), C.byref(hi), C.byref(cls), C.byref(doc)))
        out.append({"env": env.value.decode(), "default": d.value, "lo": lo.value, "hi": hi.value, "cls": cls.value, "doc": doc.value.decode()})
    return out


def config_value(env: str) -> int:
    """what a handle created now would see for this option"""
    v = C.c_int()
    _check(lib().omni_config_value(env.encode(), C.byref(v)))
    return v.value


def shard_library_path() -> str:
    """The file ncclAllGather & co. were resolved from (the process's librccl, or what OMNI_RCCL_LIB names)."""
    buf = C.create_string_buffer(1024)
    _check(lib().omni_shard_library_path(buf, 1024))
    return buf.value.decode()


class Shard:
    """omni_shard: this rank's part of the row-sharded key-frame database; collectives are RCCL inside libomni_hip.so (csrc/shard.hip)."""

    def __init__(self, ctx: Context, local: IndexFlatIP, rank: int, world: int, unique_id: bytes):
        assert len(unique_id) == SHARD_ID_BYTES
        self.ctx, self.local, self.rank, self.world = ctx, local, rank, world
        self.h = lib().omni_shard_create(ctx.h, local.h, local.d, rank, world, unique_id)
        if not self.h:
            raise OmniError(f"omni_shard_create failed: {lib().omni_last_error().decode()}")
        ctx._adopt(self)

    def close(self):
        if getattr(self, "h", None):
            lib().omni_shard_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ntotal(self) -> int:
        return lib().omni_shard_ntotal(self.h)

    def preload_local(self, rows_local: np.ndarray, ntotal_global: int):
        rows_local = _f32(rows_local)
        _check(lib().omni_shard_preload_local(self.h, _pf(rows_local), rows_local.shape[0], ntotal_global))

    def step_batch_dev(self, F: int, m: int, rows_dev: int, query_row: int, k: int):
        D = np.empty((F, k), np.float32)
        I = np.empty((F, k), np.int64)
        _check(lib().omni_shard_step_batch_dev(self.h, F, m, rows_dev, query_row, k, _pf(D), I.ctypes.data_as(_i64p)))
        return D, I

    def step_enqueue(self, F: int, m: int, rows_dev: int, query_row: int, k: int):
        """asynchronous half of step_batch_dev: the whole exchange unit is put on the shard's stream, nothing is waited for"""
        self._pend = (F, k)
        _check(lib().omni_shard_step_enqueue(self.h, F, m, rows_dev, query_row, k))

    def rows_consumed(self):
        _check(lib().omni_shard_rows_consumed(self.h))

    def step_wait(self):
        F, k = self._pend
        D = np.empty((F, k), np.float32)
        I = np.empty((F, k), np.int64)
        _check(lib().omni_shard_step_wait(self.h, _pf(D), I.ctypes.data_as(_i64p)))
        return D, I

    def search(self, q: np.ndarray, k: int):
        q = _f32(np.atleast_2d(q))
        D = np.empty((q.shape[0], k), np.float32)
        I = np.empty((q.shape[0], k), np.int64)
        _check(lib().omni_shard_search(self.h, q.shape[0], _pf(q), k, _pf(D), I.ctypes.data_as(_i64p)))
        return D, I


def topk_merge(D_lists: np.ndarray, I_lists: np.ndarray, k_out: int):
    """[n_lists, nq, k_each] per-shard lists -> merged [nq, k_out] (score desc, id asc).  Host-side, no GPU needed."""
    D_lists, I_lists = _f32(D_lists), np.ascontiguousarray(I_lists, np.int64)
    n_lists, nq, k_each = D_lists.shape
    D = np.empty((nq, k_out), np.float32)
    I = np.empty((nq, k_out), np.int64)
    _check(lib().omni_topk_merge(n_lists, nq, k_each, _pf(D_lists), I_lists.ctypes.data_as(_i64p), k_out, _pf(D),
                                 I.ctypes.data_as(_i64p)))
    return D, I


def bf_match(ctx: Context, q: np.ndarray, t: np.ndarray, mode: int = BF_OPENCV):
    """cv::BFMatcher(NORM_L2, crossCheck=true).match(q, t) -> (query_idx, train_idx, distance)."""
    q, t = _f32(q), _f32(t)
    nq, nt = q.shape[0], t.shape[0]
    dim = q.shape[1] if q.ndim == 2 and nq else (t.shape[1] if t.ndim == 2 else 64)
    qi, ti, dd = np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.float32)
    n = C.c_int(0)
    _check(lib().omni_bf_match(ctx.h, _pf(q), nq, _pf(t), nt, dim, mode, qi.ctypes.data_as(_ip), ti.ctypes.data_as(_ip),
                               _pf(dd), C.byref(n)))
    return qi[:n.value].copy(), ti[:n.value].copy(), dd[:n.value].copy()


def bf_match_multi(ctx: Context, pairs, mode: int = BF_OPENCV):
    """[(q, t), ...] -> [(query_idx, train_idx, distance), ...]: bf_match of every pair in one GPU round trip."""
    P = len(pairs)
    qs, ts = [_f32(q) for q, _ in pairs], [_f32(t) for _, t in pairs]
    dim = next((a.shape[1] for a in qs + ts if a.ndim == 2 and a.shape[0]), 64)
    nq = np.array([q.shape[0] for q in qs], np.int32)
    nt = np.array([t.shape[0] for t in ts], np.int32)
    max_n = int(max(1, nq.max(), nt.max()))
    qp = (_fp * P)(*[_pf(q) if q.shape[0] else None for q in qs])
    tp = (_fp * P)(*[_pf(t) if t.shape[0] else None for t in ts])
    qi, ti, dd, n = np.zeros((P, max_n), np.int32), np.zeros((P, max_n), np.int32), np.zeros((P, max_n), np.float32), np.zeros(P, np.int32)
    _check(lib().omni_bf_match_multi(ctx.h, P, qp, nq.ctypes.data_as(_ip), tp, nt.ctypes.data_as(_ip), dim, mode, max_n, qi.ctypes.data_as(_ip),
                                     ti.ctypes.data_as(_ip), _pf(dd), n.ctypes.data_as(_ip)))
    return [(qi[p, :n[p]].copy(), ti[p, :n[p]].copy(), dd[p, :n[p]].copy()) for p in range(P)]


def bf_match_batched_dev(ctx: Context, n_pairs, max_n, dim, mode, q_dev, q_stride, nq_dev, t_dev, t_stride, nt_dev,
                         qidx_dev, tidx_dev, dist_dev, n_dev):
    _check(lib().omni_bf_match_batched_dev(ctx.h, n_pairs, max_n, dim, mode, q_dev, q_stride, nq_dev, t_dev, t_stride,
                                           nt_dev, qidx_dev, tidx_dev, dist_dev, n_dev))


class Cam:
    """omni_cam: one key frame's CNN + matching work as a single asynchronous unit (LoopCam::on_flattened_images,
    loop_cam.cpp:178-229).  wait() returns numpy VIEWS of the handle's pinned host block (valid until the next enqueue)."""

    def __init__(self, sp: SuperPoint, vlad, n_dirs: int, global_dim: int, bf_mode: int = BF_OPENCV, mono: bool = False):
        """mono: CameraConfig::PINHOLE_DEPTH -- one camera per image (omni_cam_create_mono): n_dirs images, no up/down match"""
        self.sp, self.vlad, self.n, self.cams = sp, vlad, n_dirs, (1 if mono else 2)
        self.n_active = n_dirs
        if mono:
            self.h = lib().omni_cam_create_mono(sp.ctx.h, sp.h, vlad.ctx.h, vlad.h, n_dirs, sp.max_num, global_dim)
        else:
            self.h = lib().omni_cam_create(sp.ctx.h, sp.h, vlad.ctx.h, vlad.h, n_dirs, sp.max_num, global_dim, bf_mode)
        if not self.h:
            raise OmniError(f"omni_cam_create failed: {lib().omni_last_error().decode()}")
        sp.ctx._adopt(self)
        vlad.ctx._adopt(self)
        self._res = _CamResult()

    def close(self):
        if getattr(self, "h", None):
            lib().omni_cam_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enqueue_dev(self, gray_dev: int, stride: int, fisheye_mask: bool = True):
        _check(lib().omni_cam_enqueue_dev(self.h, gray_dev, stride, int(fisheye_mask)))

    def enqueue_host(self, gray_host: np.ndarray, fisheye_mask: bool = True):
        """gray_host [2*n_dirs][H][W] u8 ([n_dirs] for a mono handle), ideally pinned (Context.host_alloc); must stay untouched until wait() returns."""
        assert gray_host.dtype == np.uint8 and gray_host.ndim == 3 and gray_host.shape[0] == self.cams * self.n_active and gray_host.flags.c_contiguous
        _check(lib().omni_cam_enqueue_host(self.h, gray_host.ctypes.data_as(_vp), gray_host.shape[2], gray_host.shape[2], gray_host.shape[1],
                                           int(fisheye_mask)))

    def set_active(self, n_dirs: int):
        """a unit of fewer directions than the handle was created for: the next enqueues read cams * n_dirs images (up cameras first, down right behind)"""
        _check(lib().omni_cam_set_active(self.h, n_dirs))
        self.n_active = n_dirs

    def ready(self) -> bool:
        r = C.c_int(0)
        _check(lib().omni_cam_ready(self.h, C.byref(r)))
        return bool(r.value)

    def wait(self) -> dict:
        r = self._res
        _check(lib().omni_cam_wait(self.h, C.byref(r)))
        n, m, d, g, ni = r.n_dirs, r.max_num, r.desc_dim, r.global_dim, r.n_images
        A = np.ctypeslib.as_array
        return {"kps_xy": A(r.kps_xy, (ni, m, 2)), "n_kps": A(r.n_kps, (ni,)), "desc": A(r.desc, (ni, m, d)),
                "scores": A(r.scores, (ni, m)), "global_desc": A(r.global_desc, (n, g)), "match_up": A(r.match_up, (n, m)),
                "match_down": A(r.match_down, (n, m)), "match_dist": A(r.match_dist, (n, m)), "n_matches": A(r.n_matches, (n,))}
