#!/usr/bin/env python
"""MobileNetVLAD alone at the bench's launch shape (32 images): for a rocprofv3 --kernel-trace --stats run on the GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omni_loader
omni = omni_loader.load()
from omni_swarm_amd import capi, synth, weights
B = int(os.environ.get("BATCH", 32))
ctx = capi.Context(0)
net = capi.MobileNetVLAD(ctx, weights.mobilenetvlad_synth_weights(), weights.mobilenetvlad_layer_specs(), 32, 112, 4096, 600, 480, B)
if os.environ.get("PREC", "f32") == "f16":
    net.set_precision(capi.PREC_F16)
dev = ctx.to_device(np.stack([synth.image_u8(i, 480, 600) for i in range(B)]))
for _ in range(3):
    net.enqueue_dev(dev, 600, B, True)
ctx.sync()
t = time.perf_counter()
N = 20
for _ in range(N):
    net.enqueue_dev(dev, 600, B, True)
ctx.sync()
print(f"vlad batch-{B} ms:", round((time.perf_counter() - t) / N * 1e3, 4), "stem fuse", os.environ.get("OMNI_VLAD_STEM_FUSE", "1"))
