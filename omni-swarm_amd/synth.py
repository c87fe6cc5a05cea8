"""Seeded synthetic inputs for bench.py and the parity tests.  No reference algorithm lives here.

No dataset, checkpoint or bag exists offline: every generator here is deterministic in its seed so the
GPU box, the build container and the committed golden fixtures all see the same bytes.
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import gaussian_filter

BASE_SEED = 20260925


def image_u8(index: int, height: int = 480, width: int = 600, n_shapes: int = 200,
             fisheye_mask: bool = False) -> np.ndarray:
    """Image-like u8 frame: random rectangles/ellipses, blur sigma=1.5, N(0,4) noise (8d)."""
    rng = np.random.default_rng(BASE_SEED + index)
    img = np.full((height, width), 96.0, np.float32)
    for _ in range(n_shapes):
        cx, cy = rng.uniform(0, width), rng.uniform(0, height)
        rx, ry = rng.uniform(4, width / 8), rng.uniform(4, height / 8)
        val = rng.uniform(-90, 110)
        # both shapes live inside [c - r, c + r]: the masks are evaluated on that bounding box only (same arithmetic, same bytes as
        # the full-frame masks the golden fixtures were generated with)
        x0, x1 = max(0, int(np.floor(cx - rx))), min(width, int(np.ceil(cx + rx)) + 1)
        y0, y1 = max(0, int(np.floor(cy - ry))), min(height, int(np.ceil(cy + ry)) + 1)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        if rng.random() < 0.5:
            m = (np.abs(xx - cx) < rx) & (np.abs(yy - cy) < ry)
        else:
            m = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 < 1.0
        img[y0:y1, x0:x1][m] += val
    img = gaussian_filter(img, 1.5)
    img += rng.normal(0, 4.0, img.shape).astype(np.float32)
    out = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if fisheye_mask:                       # loop_cam.cpp:536-539
        out[height * 3 // 4: height * 3 // 4 + height // 4] = 0
    return out


def pca(seed: int = 1, d_out: int = 64, d_in: int = 256):
    """Orthonormal components (QR of a Gaussian) + small mean, shaped like sklearn PCA (pca.ipynb:28-31)."""
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.normal(size=(d_in, d_out)))
    comp = np.ascontiguousarray(q.T.astype(np.float32))          # [64, 256]
    mean = (np.random.default_rng(seed + 1).normal(size=d_in) * 0.01).astype(np.float32)
    return comp, mean


def global_db(n_rows: int, dim: int = 4096, seed: int = 3, dup_frac: float = 0.01, dup_cos: float = 0.8,
              chunk: int = 8192) -> np.ndarray:
    """L2-normalised Gaussian rows; ``dup_frac`` of rows are near-duplicates (cos ~ dup_cos) of an earlier row."""
    rng = np.random.default_rng(seed)
    db = np.empty((n_rows, dim), np.float32)
    for s in range(0, n_rows, chunk):
        e = min(n_rows, s + chunk)
        x = rng.standard_normal((e - s, dim), dtype=np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        db[s:e] = x
    n_dup = int(n_rows * dup_frac)
    if n_dup and n_rows > 16:
        rng2 = np.random.default_rng(seed + 100)
        dst = rng2.choice(np.arange(n_rows // 2, n_rows), size=min(n_dup, n_rows - n_rows // 2), replace=False)
        src = rng2.integers(0, n_rows // 2, size=len(dst))
        noise = rng2.standard_normal((len(dst), dim), dtype=np.float32)
        noise /= np.linalg.norm(noise, axis=1, keepdims=True)
        v = dup_cos * db[src] + np.sqrt(1 - dup_cos ** 2) * noise
        db[dst] = v / np.linalg.norm(v, axis=1, keepdims=True)
    return db


def queries_from_db(db: np.ndarray, n_q: int, seed: int = 4, sigma: float = 0.02):
    """Rows perturbed with N(0, sigma) per element, renormalised (8d)."""
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, db.shape[0], size=n_q)
    q = db[rows] + rng.normal(0, sigma, size=(n_q, db.shape[1])).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32), rows


def local_descriptors(n: int, dim: int = 64, seed: int = 5, pair_noise: float | None = None):
    """n x dim float descriptors (unit vectors); with pair_noise returns a second, permuted+noised set."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((n, dim)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    if pair_noise is None:
        return a
    perm = rng.permutation(n)
    b = a[perm] + rng.normal(0, pair_noise, (n, dim)).astype(np.float32)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    return a, b.astype(np.float32), perm


# ---- key frames of a rendered scene: a stereo fisheye rig (up / down cameras, 4 flattened 90-degree pinhole views each) in a room --------
ROOM_DISPARITY = 16          # pixels between the up and the down view of a wall point (a multiple of the network's 8-pixel cell)


def room_wall_depth(width: int = 600, baseline: float = 0.10) -> float:
    """Distance of the four walls from the rig: focal length (width / 2: 90-degree horizontal field of view) * baseline / ROOM_DISPARITY."""
    return (width / 2.0) * baseline / ROOM_DISPARITY


def room_keyframe(place: int, height: int = 480, width: int = 600, revisit: int = 0, noise_sigma: float = 0.0) -> np.ndarray:
    """The 8 images [up d0..d3, down d0..d3] a stereo rig at the centre of a square room sees: direction d looks straight at wall d
    (fronto-parallel, at room_wall_depth()), walls carry seeded textures at one texel per pixel, and the up / down cameras sit
    +- baseline / 2 along the vertical axis -- so the down view is the up view moved by exactly ROOM_DISPARITY rows and every wall point
    triangulates to the wall's depth.  The room is tall enough that the views never see floor or ceiling (vertical field of view 77 degrees).
    Rendering a fronto-parallel plane through a pinhole at one texel per pixel is a crop: up = texture rows [0, H), down = rows
    [ROOM_DISPARITY, ROOM_DISPARITY + H).  `revisit` > 0 with `noise_sigma` > 0 adds per-camera sensor noise seeded by (place, revisit):
    the same place seen again."""
    out = np.empty((8, height, width), np.uint8)
    for d in range(4):
        tex = image_u8(50_000 + 8 * place + d, height + ROOM_DISPARITY, width, n_shapes=260)
        out[d] = tex[:height]
        out[4 + d] = tex[ROOM_DISPARITY:ROOM_DISPARITY + height]
    if revisit > 0 and noise_sigma > 0:
        rng = np.random.default_rng(BASE_SEED + 7_000_000 + 1000 * place + revisit)
        out = np.clip(np.rint(out.astype(np.float32) + rng.normal(0, noise_sigma, out.shape)), 0, 255).astype(np.uint8)
    return out


# ---- key frames of a PINHOLE_DEPTH camera (launch/realsense.launch: one 640 x 480 gray image + one depth image) -------------------------------
DEPTH_STEP_M = (2.0, 3.5)    # the wall in front of the camera has a step: its left half is 2.0 m away, its right half 3.5 m


def depth_keyframe(place: int, height: int = 480, width: int = 640, revisit: int = 0, noise_sigma: float = 0.0):
    """(gray u8 [H][W], depth u16 millimetres [H][W]) of a forward-looking camera in front of a textured, fronto-parallel wall with a step
    (DEPTH_STEP_M): the landmarks are not coplanar.  One pixel in 16 of the depth image has no return (0), as a real depth camera's does.
    `revisit` > 0 with `noise_sigma` > 0: the same place seen again, with sensor noise on the gray image and +- 2 mm on the depth."""
    gray = image_u8(90_000 + place, height, width, n_shapes=260)
    depth = np.empty((height, width), np.uint16)
    depth[:, :width // 2] = int(DEPTH_STEP_M[0] * 1000)
    depth[:, width // 2:] = int(DEPTH_STEP_M[1] * 1000)
    rng = np.random.default_rng(BASE_SEED + 9_000_000 + 1000 * place + revisit)
    if revisit > 0 and noise_sigma > 0:
        gray = np.clip(np.rint(gray.astype(np.float32) + rng.normal(0, noise_sigma, gray.shape)), 0, 255).astype(np.uint8)
        depth = (depth.astype(np.int32) + rng.integers(-2, 3, depth.shape)).astype(np.uint16)
    depth[rng.integers(0, 16, depth.shape) == 0] = 0
    return gray, depth
