#!/usr/bin/env python
"""MobileNetVLAD at OMNI_PREC_F16 vs OMNI_PREC_F32 vs the oracle: descriptor error and time per launch (run on the GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omni_loader
omni = omni_loader.load()
from omni_swarm_amd import capi, synth, weights
B = int(os.environ.get("BATCH", 32))
ctx = capi.Context(0)
w = weights.mobilenetvlad_synth_weights()
net = capi.MobileNetVLAD(ctx, w, weights.mobilenetvlad_layer_specs(), 32, 112, 4096, 600, 480, B)
imgs = np.stack([synth.image_u8(i, 480, 600) for i in range(B)])
dev = ctx.to_device(imgs)
out = {}
for name, prec in (("f32", capi.PREC_F32), ("f16", capi.PREC_F16)):
    net.set_precision(prec)
    for _ in range(3):
        net.enqueue_dev(dev, 600, B, True)
    ctx.sync()
    t = time.perf_counter()
    N = 20
    for _ in range(N):
        net.enqueue_dev(dev, 600, B, True)
    ctx.sync()
    ms = (time.perf_counter() - t) / N * 1e3
    out[name] = net.fetch(B)
    print(f"{name}: {ms:.4f} ms per {B} images")
e = np.linalg.norm(out["f16"] - out["f32"], axis=1) / np.linalg.norm(out["f32"], axis=1)
print("f16 vs f32 relative L2 per image: max %.3e mean %.3e; min cosine %.6f" % (e.max(), e.mean(), (out["f16"] * out["f32"]).sum(1).min()))
one = net.inference(imgs[:1], True)
print("batch-1 == batch-32 image 0 (f16):", np.array_equal(one[0], out["f16"][0]))
if os.environ.get("ORACLE", "1") == "1":
    from oracle import mobilenetvlad_ref
    m = imgs[0].copy(); m[480 * 3 // 4:] = 0
    r = mobilenetvlad_ref.forward(w, m)
    for k in out:
        print(k, "vs oracle image 0:", np.linalg.norm(out[k][0] - r) / np.linalg.norm(r))
