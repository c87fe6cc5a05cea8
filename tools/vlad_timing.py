#!/usr/bin/env python
"""Stand-alone MobileNetVLAD time per 4-image batch (run on the GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omni_loader
omni = omni_loader.load()
from omni_swarm_amd import capi, synth, weights
ctx = capi.Context(0)
net = capi.MobileNetVLAD(ctx, weights.mobilenetvlad_synth_weights(), weights.mobilenetvlad_layer_specs(), 32, 112, 4096, 600, 480, 4)
imgs = np.stack([synth.image_u8(i, 480, 600) for i in range(4)])
dev = ctx.to_device(imgs)
for _ in range(5):
    net.enqueue_dev(dev, 600, 4, True)
ctx.sync()
t = time.perf_counter()
N = 50
for _ in range(N):
    net.enqueue_dev(dev, 600, 4, True)
ctx.sync()
print("vlad batch-4 ms:", round((time.perf_counter() - t) / N * 1e3, 4), {k: os.environ.get(k) for k in ("OMNI_VLAD_MFMA", "OMNI_VLAD_UNFUSED")})
