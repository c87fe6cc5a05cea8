#!/usr/bin/env python
"""Where does a bench.py step spend HOST time?  (run on the GPU box)  One pipeline, serial: enqueue | sync | fetch | detector."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omni_loader
omni = omni_loader.load()
from omni_swarm_amd import capi, detector, frontend, synth, weights

W, H = 600, 480
ctx = capi.Context(0); ictx = capi.Context(0)
cam = frontend.LoopCam(ctx, weights.superpoint_synth_weights(0), *synth.pca(), weights.mobilenetvlad_synth_weights(),
                       weights.mobilenetvlad_layer_specs(), (32, 112, 4096), W, H, 0.02, 200, capi.PREC_F16)
imgs = np.stack([synth.image_u8(i, H, W) for i in range(8)])
dev = ictx.to_device(imgs)
det = detector.LoopDetector(ictx, self_id=1, inner_product_thres=0.3, init_mode_product_thres=0.2, match_index_dist=5, min_loop_num=30, min_direction_loop=3)
rng = np.random.default_rng(0)
x = rng.standard_normal((4000, 4096), dtype=np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
det.local_index.add(x)
for i in range(4000):
    det.imgid2fisheye[i] = -(i // 4) - 1; det.imgid2dir[i] = i % 4
T = {"enqueue": 0.0, "gpu_wait": 0.0, "fetch": 0.0, "detector": 0.0}
N = 60
for it in range(N + 5):
    if it == 5: T = {k: 0.0 for k in T}
    t0 = time.perf_counter(); cam.enqueue_dev(dev, W)
    t1 = time.perf_counter(); ctx.sync(); cam.vctx.sync()
    t2 = time.perf_counter(); out = cam.fetch()
    t3 = time.perf_counter()
    fr = detector.FisheyeFrameDescriptor(msg_id=it, drone_id=1, landmark_num=out["landmark_num"], prevent_adding_db=False,
        images=[detector.ImageDescriptor(drone_id=1, landmark_num=i["landmark_num"], image_desc=i["image_desc"],
                feature_descriptor=i["feature_descriptor"], landmarks_2d=i["landmarks_2d"]) for i in out["images"]])
    det.on_image_recv(fr)
    t4 = time.perf_counter()
    T["enqueue"] += t1 - t0; T["gpu_wait"] += t2 - t1; T["fetch"] += t3 - t2; T["detector"] += t4 - t3
print({k: round(v / N * 1e3, 3) for k, v in T.items()}, "ms per key frame; fused vlad:", os.environ.get("OMNI_VLAD_UNFUSED", "0") != "1")
