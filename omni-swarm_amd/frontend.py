"""LoopCam: the per-keyframe CNN frontend (host-side mirror over the C ABI).

Mirrors /root/reference/swarm_loop/src/loop_cam.cpp:
    on_flattened_images                :178-229   loop over the camera directions of one fisheye key frame
    generate_stereo_image_descriptor   :341-523   SuperPoint(up), SuperPoint(down), MobileNetVLAD(up), BF up<->down
    extractor_img_desc_deepnet         :525-585   fisheye mask (rows [3H/4,H) zeroed), the two CNN calls
    match_HFNet_local_features         :141-174   cv::BFMatcher(NORM_L2, crossCheck) on the 64-d descriptors
The reference runs these 8 + 4 engine calls and 4 matches strictly one after another, each with its own H2D/D2H and
stream sync; here one key frame is three batched enqueues on one HIP stream (8 SuperPoint images, 4
MobileNetVLAD images on a second stream, 4 descriptor-set pairs) with every intermediate resident in HBM and one small
D2H at the end.
Camera geometry (camodocal liftProjective, SVD triangulation, loop_cam.cpp:73-106,405-454,558-576) is host-side f64
work outside the kernel scope and is left to the caller: the 2-D key points, descriptors, global
descriptors and the up/down match lists are everything those steps consume.
"""
from __future__ import annotations

import numpy as np

from . import capi


class LoopCam:
    def __init__(self, ctx: capi.Context, sp_weights: dict, pca_comp, pca_mean, vlad_weights: dict, vlad_specs,
                 vlad_shape=(32, 112, 4096), width: int = 600, height: int = 480, thres: float = 0.015,
                 max_num: int = 200, precision: int = capi.PREC_F16, n_dirs: int = 4, fisheye: bool = True,
                 accept_min_3d_pts: int = 0):
        self.ctx, self.W, self.H, self.n_dirs, self.max_num, self.fisheye = ctx, width, height, n_dirs, max_num, fisheye
        self.accept_min_3d_pts = accept_min_3d_pts
        self.sp = capi.SuperPoint(ctx, sp_weights, pca_comp, pca_mean, width, height, thres, max_num, precision, 2 * n_dirs)
        k, d, o = vlad_shape
        # MobileNetVLAD is ~50 small launches: on its own HIP stream they overlap SuperPoint's large convolutions
        import os
        self._own_vctx = os.environ.get("OMNI_VLAD_SAME_STREAM", "0") != "1"
        self.vctx = capi.Context(ctx.device_id) if self._own_vctx else ctx
        self.vlad = capi.MobileNetVLAD(self.vctx, vlad_weights, vlad_specs, k, d, o, width, height, n_dirs)
        self.out_dim = o
        self.dim = self.sp.desc_dim
        # the whole key frame as one asynchronous unit with a single pinned result block (csrc/cam.hip)
        self.cam = capi.Cam(self.sp, self.vlad, n_dirs, o, capi.BF_OPENCV)

    def close(self):
        self.cam.close()
        self.sp.close()
        self.vlad.close()
        if self._own_vctx:
            self.vctx.close()

    def enqueue_dev(self, gray_dev: int, stride: int):
        """gray_dev: [2*n_dirs][H][W] u8 in HBM -- images 0..n_dirs-1 are the 'up' (main) camera of each direction,
        n_dirs..2*n_dirs-1 the 'down' camera.  Asynchronous: SuperPoint + BF matching on the context's stream,
        MobileNetVLAD on a second stream, all D2H copies included (loop_cam.cpp:350-351, 553-556, 147-150)."""
        self.cam.enqueue_dev(gray_dev, stride, self.fisheye)

    def enqueue_host(self, gray_host: np.ndarray):
        """Same with the images in (pinned) host memory, [2*n_dirs][H][W] u8: one asynchronous upload in front of the kernels
        (omni_cam_enqueue_host).  The array must stay untouched until fetch() returns."""
        self.cam.enqueue_host(gray_host, self.fisheye)

    def fetch(self) -> dict:
        """Waits for the key frame and returns one FisheyeFrameDescriptor_t's worth of CNN outputs (copies: the pinned
        block is reused by the next enqueue)."""
        r = self.cam.wait()
        n = r["global_desc"].shape[0]          # the unit's directions: n_dirs, or fewer after Cam.set_active
        nk = r["n_kps"]
        images = []
        for d in range(n):
            nu, nd = int(nk[d]), int(nk[n + d])
            k = int(r["n_matches"][d]) if nu > self.accept_min_3d_pts else 0    # :388 `if (pts_up.size() > ACCEPT_MIN_3D_PTS)`
            images.append({"landmarks_2d": r["kps_xy"][d, :nu].copy(), "feature_descriptor": r["desc"][d, :nu].copy(),
                           "scores": r["scores"][d, :nu].copy(), "landmark_num": nu, "image_desc": r["global_desc"][d].copy(),
                           "landmarks_2d_down": r["kps_xy"][n + d, :nd].copy(), "feature_descriptor_down": r["desc"][n + d, :nd].copy(),
                           "ids_up": r["match_up"][d, :k].copy(), "ids_down": r["match_down"][d, :k].copy(), "direction": d})
        return {"images": images, "landmark_num": int(sum(i["landmark_num"] for i in images))}

    def on_flattened_images(self, up: np.ndarray, down: np.ndarray) -> dict:
        """Host-pointer convenience (the reference's blocking call): up/down [n_dirs][H][W] uint8."""
        g = np.ascontiguousarray(np.concatenate([up, down]), np.uint8)
        p = self.ctx.to_device(g)
        try:
            self.enqueue_dev(p, self.W)
            return self.fetch()
        finally:
            self.ctx.free(p)
