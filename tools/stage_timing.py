#!/usr/bin/env python
"""SuperPoint stage times and stand-alone MobileNetVLAD time at a given batch (run on the GPU box): BATCH=32 python tools/stage_timing.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omni_loader
omni = omni_loader.load()
from omni_swarm_amd import capi, synth, weights
B = int(os.environ.get("BATCH", 32))
ctx = capi.Context(0)
comp, mean = synth.pca()
PREC = {"f16": capi.PREC_F16, "split": capi.PREC_SPLIT, "f32": capi.PREC_F32}[os.environ.get("PREC", "f16")]
sp = capi.SuperPoint(ctx, weights.superpoint_synth_weights(0), comp, mean, 600, 480, 0.02, 200, PREC, B)
imgs = np.stack([synth.image_u8(i, 480, 600) for i in range(B)])
dev = ctx.to_device(imgs)
for _ in range(3):
    sp.profile(dev, 600, B, reps=20)      # warm the clocks up
prof = sp.profile(dev, 600, B, reps=20)
tot = sum(p["ms"] for p in prof)
print(f"SuperPoint batch {B}: {tot:.3f} ms = {tot / B * 8:.3f} ms per 8 images;", {p["stage"]: round(p["ms"] / B * 8, 4) for p in prof})
if os.environ.get("NO_VLAD") == "1":
    sys.exit(0)
nb = max(1, B // 2)
net = capi.MobileNetVLAD(ctx, weights.mobilenetvlad_synth_weights(), weights.mobilenetvlad_layer_specs(), 32, 112, 4096, 600, 480, nb)
for _ in range(3):
    net.enqueue_dev(dev, 600, nb, True)
ctx.sync()
t = time.perf_counter()
for _ in range(30):
    net.enqueue_dev(dev, 600, nb, True)
ctx.sync()
ms = (time.perf_counter() - t) / 30 * 1e3
print(f"MobileNetVLAD batch {nb}: {ms:.3f} ms = {ms / nb * 4:.3f} ms per 4 images")
