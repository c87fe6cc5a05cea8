"""Diagnostic (GPU): OMNI_PREC_SPLIT dense pass vs mask-skip pass, layer by layer, repeated (is a difference deterministic? which layer first?)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omni_loader
omni = omni_loader.load()
from oracle import superpoint_ref as S
from omni_swarm_amd import synth
c = omni.capi
ctx = c.Context(0)
h, w, batch = 480, 600, 3
weights = S.synth_weights(0)
comp, mean = synth.pca()
imgs = np.stack([synth.image_u8(900 + i, h, w, n_shapes=200) for i in range(batch)])
sps = []
for flag in ("0", "1"):
    os.environ["OMNI_SP_MASK_SKIP_SPLIT"] = flag
    sps.append(c.SuperPoint(ctx, weights, comp, mean, w, h, 0.015, 200, c.PREC_SPLIT, batch + 1))
dense, skip = sps
LAYERS = ["conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "heads"]
for rep in range(4):
    ra, rb = dense.inference(imgs, True), skip.inference(imgs, True)
    for n in LAYERS:
        a, b = dense.debug_layer(n, batch), skip.debug_layer(n, batch)
        if not np.array_equal(a, b):
            idx = np.argwhere(a != b)
            print(f"rep {rep} layer {n}: {len(idx)} values differ; first {idx[:5].tolist()} max |d| {np.abs(a - b).max():.3e}; rows {sorted(set(idx[:, 2].tolist()))[:12]} cols {sorted(set(idx[:, 3].tolist()))[:12]}")
            break
    else:
        same = all(np.array_equal(x[1], y[1]) for x, y in zip(ra, rb))
        print(f"rep {rep}: all layers identical; descriptors identical: {same}")
    # the same handle twice: deterministic?
    a1 = dense.debug_layer("conv1b", batch); dense.inference(imgs, True); a2 = dense.debug_layer("conv1b", batch)
    print("   dense conv1b run-to-run identical:", np.array_equal(a1, a2))
