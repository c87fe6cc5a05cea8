"""SuperPoint post-processing oracle (CPU) -- test infrastructure only.

Restates /root/reference/swarm_loop/src/superpoint_tensorrt.cpp:
  get_keypoints        :164-189 getKeyPoints  +  :237-310 NMS2
  compute_descriptors  :192-230 computeDescriptors (torch grid_sampler, L2 norm, PCA)
The reference has no golden vectors for these; the restatement is
checked three ways in tests/: C literal simulation (oracle/csrc/oracle.c) == python literal
simulation == the closed "alive / survive" characterisation below.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    """ctypes handle of oracle/liboracle.so (built by oracle/Makefile / __graft_entry__.build)."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
        L = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        L.oracle_nms2_literal.restype = ctypes.c_int
        L.oracle_nms2_literal.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ip, fp, ip, ip]
        L.oracle_bf_match.restype = ctypes.c_int
        L.oracle_bf_match.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, ip, fp]
        L.oracle_ip_search.restype = None
        L.oracle_ip_search.argtypes = [fp, ctypes.c_long, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                       fp, ctypes.POINTER(ctypes.c_int64)]
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def get_keypoints(prob: np.ndarray, thres: float, max_num: int = 200, dist_thresh: int = 4,
                  wrap_columns: bool = False):
    """C literal simulation. prob [H,W] f32 -> (xy int32 [n,2] (x,y), conf f32 [n], n_cand, n_surv).

    Order: confidence descending, then row-major pixel index ascending (fixed spec)."""
    prob = np.ascontiguousarray(prob, dtype=np.float32)
    H, W = prob.shape
    xy = np.zeros((max_num, 2), np.int32)
    conf = np.zeros(max_num, np.float32)
    nc = ctypes.c_int(0)
    ns = ctypes.c_int(0)
    n = lib().oracle_nms2_literal(_fp(prob), W, H, np.float32(thres), dist_thresh, max_num,
                                  int(wrap_columns), _ip(xy), _fp(conf), ctypes.byref(nc), ctypes.byref(ns))
    return xy[:n].copy(), conf[:n].copy(), nc.value, ns.value


def get_keypoints_py_literal(prob: np.ndarray, thres: float, max_num: int = 200, dist_thresh: int = 4):
    """Pure-python literal simulation of superpoint_tensorrt.cpp:164-189,237-310 (small inputs only)."""
    prob = np.asarray(prob, dtype=np.float32)
    H, W = prob.shape
    ys, xs = np.nonzero(prob > np.float32(thres))          # row-major (findNonZero)
    grid = np.zeros((H, W), np.uint8)
    confidence = np.zeros((H, W), np.float32)
    grid[ys, xs] = 1
    confidence[ys, xs] = prob[ys, xs]
    for vv, uu in zip(ys.tolist(), xs.tolist()):
        if grid[vv, uu] != 1:
            continue
        c0 = confidence[vv, uu]
        for k in range(-dist_thresh, dist_thresh + 1):
            for j in range(-dist_thresh, dist_thresh + 1):
                if j == 0 and k == 0:
                    continue
                v, u = vv + k, uu + j
                if v < 0 or v >= H or u < 0 or u >= W:     # fixed spec: ignore out-of-image
                    continue
                if confidence[v, u] < c0:
                    grid[v, u] = 0
        grid[vv, uu] = 2
    sy, sx = np.nonzero(grid == 2)
    c = confidence[sy, sx]
    idx = sy.astype(np.int64) * W + sx
    order = np.lexsort((idx, -c.astype(np.float64)))       # conf desc, then idx asc
    order = order[:max_num]
    return np.stack([sx[order], sy[order]], 1).astype(np.int32), c[order]


def get_keypoints_characterised(prob: np.ndarray, thres: float, max_num: int = 200, dist_thresh: int = 4):
    """Closed form of NMS2, evaluated by fixed-point iteration -- this is the
    formulation the HIP kernel implements:

      alive(p)   <=> cand(p) and no EARLIER (row-major) alive q in the (2r+1)^2 window with conf(q) > conf(p)
      survive(p) <=> alive(p) and no alive q (anywhere in the window) with conf(q) > conf(p)
    """
    prob = np.asarray(prob, dtype=np.float32)
    H, W = prob.shape
    r = dist_thresh
    cand = prob > np.float32(thres)
    conf = np.where(cand, prob, np.float32(0))
    UNKNOWN, ALIVE, DEAD = 1, 2, 3
    state = np.where(cand, UNKNOWN, 0).astype(np.uint8)
    pad_c = np.pad(conf, r, constant_values=0)
    offs_earlier = [(k, j) for k in range(-r, r + 1) for j in range(-r, r + 1)
                    if (k < 0) or (k == 0 and j < 0)]
    offs_all = [(k, j) for k in range(-r, r + 1) for j in range(-r, r + 1) if not (k == 0 and j == 0)]
    while True:
        pad_s = np.pad(state, r, constant_values=0)
        any_alive = np.zeros((H, W), bool)
        any_unknown = np.zeros((H, W), bool)
        for k, j in offs_earlier:
            cq = pad_c[r + k:r + k + H, r + j:r + j + W]
            sq = pad_s[r + k:r + k + H, r + j:r + j + W]
            higher = cq > conf
            any_alive |= higher & (sq == ALIVE)
            any_unknown |= higher & (sq == UNKNOWN)
        new = state.copy()
        unk = state == UNKNOWN
        new[unk & any_alive] = DEAD
        new[unk & ~any_alive & ~any_unknown] = ALIVE
        if np.array_equal(new, state):
            break
        state = new
    assert not (state == UNKNOWN).any()
    pad_s = np.pad(state, r, constant_values=0)
    beaten = np.zeros((H, W), bool)
    for k, j in offs_all:
        cq = pad_c[r + k:r + k + H, r + j:r + j + W]
        sq = pad_s[r + k:r + k + H, r + j:r + j + W]
        beaten |= (cq > conf) & (sq == ALIVE)
    surv = (state == ALIVE) & ~beaten
    sy, sx = np.nonzero(surv)
    c = conf[sy, sx]
    idx = sy.astype(np.int64) * W + sx
    order = np.lexsort((idx, -c.astype(np.float64)))[:max_num]
    return np.stack([sx[order], sy[order]], 1).astype(np.int32), c[order]


def compute_descriptors(desc: np.ndarray, kps_xy: np.ndarray, width: int, height: int,
                        pca_comp: np.ndarray | None = None, pca_mean: np.ndarray | None = None):
    """superpoint_tensorrt.cpp:192-230.

    desc [256, H/8, W/8] f32 (one image), kps_xy [n,2] (x,y).  Returns (n x 64 if PCA else n x 256, raw n x 256).
      grid x = 2*x/width - 1, y = 2*y/height - 1                 (:203-205, float32 arithmetic)
      torch::grid_sampler(desc, grid, bilinear=0, zeros=0, align_corners=false)   (:209)
      d = d.squeeze(0).squeeze(1) -> [256, n];  dn = torch::norm(d, 2, /*dim=*/1)  (:211-215)
        !! dim 1 of a [256, n] tensor is the KEY-POINT axis: the reference divides every
        !! channel by its L2 norm ACROSS the image's key points (not every descriptor by its
        !! own norm, which is what the original SuperPoint demo does and what SURVEY.md 8a-6
        !! assumed).  Restated literally; a channel that is zero at every key point gives
        !! 0/0 = NaN here exactly as in the reference.
      (d - mean) * comp^T, no re-normalisation                   (:220-222, USE_PCA)
    """
    n = len(kps_xy)
    if n == 0:
        d_out = 64 if pca_comp is not None else SP_RAW
        return np.zeros((0, d_out), np.float32), np.zeros((0, SP_RAW), np.float32)
    fk = torch.from_numpy(np.asarray(kps_xy, dtype=np.float32))
    grid = torch.zeros(1, 1, n, 2)
    grid[0, 0, :, 0] = 2.0 * fk[:, 0] / width - 1
    grid[0, 0, :, 1] = 2.0 * fk[:, 1] / height - 1
    md = torch.from_numpy(np.ascontiguousarray(desc, dtype=np.float32))[None]
    d = F.grid_sample(md, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    d = d.squeeze(0).squeeze(1)                       # [256, n]
    dn = torch.norm(d, 2, 1)                          # [256]: across key points (sic)
    d = d.div(torch.unsqueeze(dn, 1)).transpose(0, 1).contiguous().numpy()   # [n, 256]
    if pca_comp is None:
        return d, d
    out = (d - np.asarray(pca_mean, np.float32)[None, :]) @ np.asarray(pca_comp, np.float32).T
    return out.astype(np.float32), d


SP_RAW = 256
