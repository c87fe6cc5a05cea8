#!/usr/bin/env python
"""OMNI_PREC_SPLIT bring-up (run on the GPU box): every layer of the split-fp16 SuperPoint path against the torch-fp32 oracle with error
localisation (never raises), then stage times at BATCH images per launch:  BATCH=64 python tools/split_check.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omni_loader  # noqa: E402
from oracle import postproc_ref as P, superpoint_ref as S  # noqa: E402

omni = omni_loader.load()
from omni_swarm_amd import synth  # noqa: E402
c = omni.capi
ctx = c.Context(0)
print("device:", ctx.device_info(), flush=True)
LAYERS = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b"]
PREC = getattr(c, os.environ.get("PREC", "PREC_SPLIT"))


def oracle_layers(w, x):
    import torch
    import torch.nn.functional as F
    semi, desc, inter = S.forward(w, x, return_intermediates=True)
    out = {}
    for n in LAYERS:
        a = torch.from_numpy(inter[n])
        out[n] = (F.max_pool2d(a, 2, 2) if n in ("conv1b", "conv2b", "conv3b") else a).numpy()
    out["heads"] = np.concatenate([inter["convPa"], inter["convDa"]], 1)
    return semi, desc, out


weights = S.synth_weights(0)
comp, mean = synth.pca()
for (h, w) in [(64, 96), (72, 104), (480, 600)]:
    imgs = np.stack([synth.image_u8(400 + i, h, w, n_shapes=60 if h < 100 else 200) for i in range(2)])
    try:
        sp = c.SuperPoint(ctx, weights, comp, mean, w, h, 0.015, 200, PREC, 2)
        res = sp.inference(imgs)
        semi_r, desc_r, layers_r = oracle_layers(weights, S.preprocess_u8(imgs))
        for n in LAYERS + ["heads"]:
            got, ref = sp.debug_layer(n, 2), layers_r[n]
            d = np.abs(got - ref)
            err, mag = d.max(), max(1.0, np.abs(ref).max())
            tag = "PASS" if err < 2e-5 * mag else "FAIL"
            where = np.unravel_index(np.argmax(d), d.shape)
            # per-channel / per-row / per-column error profile of the worst image localises indexing bugs
            msg = ""
            if tag == "FAIL":
                dc = d.max(axis=(0, 2, 3)); dy = d.max(axis=(0, 1, 3)); dx = d.max(axis=(0, 1, 2))
                msg = (f"\n      bad channels {np.nonzero(dc > 2e-5 * mag)[0][:40].tolist()} ({(dc > 2e-5 * mag).sum()}/{len(dc)})"
                       f"\n      bad rows {np.nonzero(dy > 2e-5 * mag)[0][:40].tolist()} ({(dy > 2e-5 * mag).sum()}/{len(dy)})"
                       f"\n      bad cols {np.nonzero(dx > 2e-5 * mag)[0][:40].tolist()} ({(dx > 2e-5 * mag).sum()}/{len(dx)})"
                       f"\n      got {got[where]:.6f} ref {ref[where]:.6f}; mean|got| {np.abs(got).mean():.4f} mean|ref| {np.abs(ref).mean():.4f}")
            print(f"[{tag}] {h}x{w} {n}: max err {err:.3e} (magnitude {mag:.2f}) at {where}{msg}", flush=True)
        semi, desc = sp.get_dense(2)
        print(f"       semi max err {np.abs(semi - semi_r).max():.3e}   desc max err {np.abs(desc - desc_r).max():.3e}", flush=True)
        for b in range(2):
            xy, conf, _, _ = P.get_keypoints(semi_r[b], 0.015, 200)
            kp = res[b][0].astype(np.int32)
            same = {tuple(p) for p in kp.tolist()} == {tuple(p) for p in xy.tolist()}
            d_r, _ = P.compute_descriptors(desc_r[b], xy, w, h, comp, mean)
            derr = np.abs(res[b][1] - d_r).max() if same and np.array_equal(kp, xy) else float("nan")
            print(f"       image {b}: {len(kp)} key points, same set as the oracle: {same}, same order: {np.array_equal(kp, xy)}, 64-d max err {derr:.3e}", flush=True)
        sp.close()
    except Exception as e:  # noqa: BLE001
        import traceback
        print(f"[FAIL] {h}x{w}: {e!r}", flush=True)
        traceback.print_exc()

B = int(os.environ.get("BATCH", 64))
sp = c.SuperPoint(ctx, weights, comp, mean, 600, 480, 0.02, 200, PREC, B)
imgs = np.stack([synth.image_u8(i, 480, 600) for i in range(min(B, 8))] * (B // min(B, 8)))
dev = ctx.to_device(imgs)
for _ in range(2):
    sp.profile(dev, 600, B, reps=5)
prof = sp.profile(dev, 600, B, reps=10)
tot = sum(p["ms"] for p in prof)
print(f"SuperPoint batch {B}: {tot:.3f} ms = {tot / B * 8:.4f} ms per key frame (8 images);", {p["stage"]: round(p["ms"] / B * 8, 4) for p in prof}, flush=True)
for p in prof:
    if p.get("flops_per_image") and p["ms"] > 0:
        print(f"   {p['stage']:>20s}: {p['ms']:.3f} ms  {p['flops_per_image'] * B / (p['ms'] * 1e-3) / 1e12:.0f} TFLOP/s (algorithmic)")
