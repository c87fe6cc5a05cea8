"""Fisheye-flattening oracle (CPU, numpy) -- test infrastructure only.  PARITY UNPINNED (OpenCV and camodocal are un-vendored).

Restates /root/reference/swarm_localization/test/fisheye_undist.hpp:
    generate_all_undist_maps   generateAllUndistMap :118-186 + genOneUndistMap :188-215 (camodocal MEI / CataCamera spaceToPlane)
    remap_linear               cv::cuda::remap(INTER_LINEAR, BORDER_CONSTANT 0) as undist_all_cuda :67-90 uses it: floor, four float taps,
                               products and sums rounded one by one, saturate_cast<uchar> = round half to even
"""
from __future__ import annotations

import math

import numpy as np


def space_to_plane(mei, P):
    xi, k1, k2, p1, p2, g1, g2, u0, v0 = mei
    P = np.asarray(P, np.float64)
    n = np.linalg.norm(P, axis=-1)
    z = P[..., 2] / n + xi
    mx, my = P[..., 0] / n / z, P[..., 1] / n / z
    rho2 = mx * mx + my * my
    rad = k1 * rho2 + k2 * rho2 * rho2
    dx = mx * rad + 2 * p1 * mx * my + p2 * (rho2 + 2 * mx * mx)
    dy = my * rad + 2 * p2 * mx * my + p1 * (rho2 + 2 * my * my)
    return np.stack([g1 * (mx + dx) + u0, g2 * (my + dy) + v0], -1)


def _rot(axis, angle):
    c, s = math.cos(angle), math.sin(angle)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def gen_one_undist_map(mei, R, w, h, f):
    x, y = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    obj = np.stack([x - w / 2, y - h / 2, np.full_like(x, f)], -1) @ R.T
    return space_to_plane(mei, obj).astype(np.float32)


def generate_all_undist_maps(mei, img_width, fov_deg, cam_id=0):
    side = max((fov_deg - 180) * math.pi / 180, 0.0)
    center = fov_deg * math.pi / 180 - 2 * side
    f_center, f_side = img_width / 2 / math.tan(center / 2), img_width / 2
    side_h = int(2 * f_side * math.tan(side / 2))
    t = np.eye(3)
    maps = [gen_one_undist_map(mei, t, img_width, img_width, f_center)]
    if cam_id == 1:
        t = _rot("x", math.pi)
    if side_h > 0:
        t = t @ _rot("x", -math.pi / 2)
        maps.append(gen_one_undist_map(mei, t, img_width, side_h, f_side))
        for _ in range(3):
            t = t @ _rot("y", math.pi / 2)
            maps.append(gen_one_undist_map(mei, t, img_width, side_h, f_side))
    return maps


def remap_linear(src_u8, map_xy):
    src = np.asarray(src_u8, np.uint8)
    H, W = src.shape
    mx, my = map_xy[..., 0].astype(np.float32), map_xy[..., 1].astype(np.float32)
    x1, y1 = np.floor(mx).astype(np.int64), np.floor(my).astype(np.int64)
    x2, y2 = x1 + 1, y1 + 1

    def at(y, x):
        ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
        return np.where(ok, src[np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)], 0).astype(np.float32)

    f = np.float32
    ax2, ax1 = x2.astype(f) - mx, mx - x1.astype(f)
    ay2, ay1 = y2.astype(f) - my, my - y1.astype(f)
    acc = at(y1, x1) * (ax2 * ay2)
    acc = acc + at(y1, x2) * (ax1 * ay2)
    acc = acc + at(y2, x1) * (ax2 * ay1)
    acc = acc + at(y2, x2) * (ax1 * ay1)
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)
