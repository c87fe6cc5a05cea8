#!/usr/bin/env python
"""GPU bring-up diagnostics (run on the MI355X box via gpurun): every stage of the hot path against the oracle with
verbose error localisation, plus first timings.  Never raises: prints PASS/FAIL per check so one run shows everything."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omni_loader  # noqa: E402
from oracle import match_ref, mobilenetvlad_ref, postproc_ref, superpoint_ref  # noqa: E402

omni = omni_loader.load()
from omni_swarm_amd import synth  # noqa: E402
c = omni.capi
RESULTS = []


def check(name):
    def deco(fn):
        t = time.time()
        try:
            msg = fn()
            RESULTS.append((name, "PASS", msg or ""))
            print(f"[PASS] {name} ({time.time() - t:.1f}s) {msg or ''}", flush=True)
        except Exception as e:  # noqa: BLE001
            RESULTS.append((name, "FAIL", repr(e)))
            print(f"[FAIL] {name} ({time.time() - t:.1f}s): {e!r}", flush=True)
            traceback.print_exc()
        return fn
    return deco


ctx = c.Context(0)
print("device:", ctx.device_info(), flush=True)


@check("index small fp32")
def _():
    db = synth.global_db(3000, seed=3)
    q, rows = synth.queries_from_db(db, 8, seed=4)
    idx = c.IndexFlatIP(ctx, 4096)
    idx.add(db)
    D, I = idx.search(q, 10)
    Dr, Ir = match_ref.ip_search(db, q, 10)
    assert np.array_equal(I, Ir), (I[:2], Ir[:2])
    return f"max score err {np.abs(D - Dr).max():.2e}"


@check("bf match")
def _():
    a, b, _ = synth.local_descriptors(200, 64, seed=5, pair_noise=0.2)
    for mode in (0, 1):
        got, ref = c.bf_match(ctx, a, b, mode), match_ref.bf_match(a, b, mode)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), (mode, len(got[0]), len(ref[0]))
        assert np.array_equal(got[2], ref[2]), np.abs(got[2] - ref[2]).max()
    return f"{len(got[0])} matches"


def layer_report(sp, weights, imgs, prec_name):
    import torch
    import torch.nn.functional as F
    x = superpoint_ref.preprocess_u8(imgs)
    semi_r, desc_r, inter = superpoint_ref.forward(weights, x, return_intermediates=True)
    b = imgs.shape[0]
    bad = []
    for n in ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "heads"]:
        if n == "heads":
            ref = np.concatenate([inter["convPa"], inter["convDa"]], 1)
        else:
            a = torch.from_numpy(inter[n])
            ref = (F.max_pool2d(a, 2, 2) if n in ("conv1b", "conv2b", "conv3b") else a).numpy()
        got = sp.debug_layer(n, b)
        err = np.abs(got - ref)
        scale = max(1.0, np.abs(ref).max())
        print(f"   [{prec_name}] {n:7s} shape {got.shape} max|err| {err.max():.3e} mean {err.mean():.3e} ref max {np.abs(ref).max():.2f} "
              f"frac>1e-3: {(err > 1e-3 * scale).mean():.4f}", flush=True)
        if err.max() > 1e-2 * scale and len(bad) == 0:
            bad.append(n)
            bi, ci, yi, xi = np.unravel_index(np.argmax(err), err.shape)
            print(f"      worst at b={bi} c={ci} y={yi} x={xi}: got {got[bi, ci, yi, xi]:.5f} ref {ref[bi, ci, yi, xi]:.5f}")
            per_c = err.max(axis=(0, 2, 3))
            print("      per-channel max err (first 16):", np.round(per_c[:16], 4), " bad channels:", int((per_c > 1e-2 * scale).sum()), "/", len(per_c))
            per_x = err.max(axis=(0, 1, 2))
            per_y = err.max(axis=(0, 1, 3))
            print("      bad columns:", np.nonzero(per_x > 1e-2 * scale)[0][:40], " bad rows:", np.nonzero(per_y > 1e-2 * scale)[0][:40])
            print("      got[0,:4,:3,:6]:\n", np.round(got[0, :4, :3, :6], 4), "\n      ref:\n", np.round(ref[0, :4, :3, :6], 4))
    semi, desc = sp.get_dense(b)
    print(f"   [{prec_name}] semi max|err| {np.abs(semi - semi_r).max():.3e}   desc max|err| {np.abs(desc - desc_r).max():.3e}", flush=True)
    return semi, desc, semi_r, desc_r


for prec, pname in ((c.PREC_F32, "f32"), (c.PREC_F16, "f16")):
    for (h, w) in ((64, 96), (480, 600)):
        @check(f"superpoint {pname} {w}x{h}")
        def _(prec=prec, pname=pname, h=h, w=w):
            weights = superpoint_ref.synth_weights(0)
            comp, mean = synth.pca()
            imgs = np.stack([synth.image_u8(400 + i, h, w, n_shapes=60 if h < 100 else 200) for i in range(2)])
            sp = c.SuperPoint(ctx, weights, comp, mean, w, h, 0.015, 200, prec, 2)
            res = sp.inference(imgs)
            semi, desc, semi_r, desc_r = layer_report(sp, weights, imgs, pname)
            out = []
            for b in range(2):
                xy_r, conf_r, nc, ns = postproc_ref.get_keypoints(semi_r[b], 0.015, 200)
                xy_g, conf_g, _, _ = postproc_ref.get_keypoints(semi[b], 0.015, 200)
                kp = res[b][0].astype(np.int32)
                same_own = np.array_equal(kp, xy_g) and np.array_equal(res[b][2], conf_g)
                same_ref = np.array_equal(kp, xy_r)
                ov = len({tuple(p) for p in kp.tolist()} & {tuple(p) for p in xy_r.tolist()})
                d_r, _ = postproc_ref.compute_descriptors(desc_r[b], xy_r, w, h, comp, mean)
                derr = np.abs(res[b][1] - d_r).max() if same_ref and len(kp) else float("nan")
                out.append(f"img{b}: n={len(kp)} post-exact-on-own-map={same_own} kps==oracle={same_ref} overlap={ov}/{len(xy_r)} desc64 err={derr:.2e}")
                if not same_own:
                    print("      GPU kps[:8]", kp[:8].tolist(), "oracle-on-GPU-map[:8]", xy_g[:8].tolist())
            rel = np.linalg.norm(desc - desc_r, axis=1) / np.linalg.norm(desc_r, axis=1)
            out.append(f"dense desc rel-L2 p50={np.percentile(rel, 50):.2e} p99={np.percentile(rel, 99):.2e} max={rel.max():.2e}")
            sp.close()
            return " | ".join(out)


@check("sp_post golden small")
def _():
    g = np.load(os.path.join(ROOT, "tests", "golden", "sp_small.npz"))
    comp, mean = synth.pca()
    sp = c.SuperPoint(ctx, superpoint_ref.synth_weights(0), comp, mean, 96, 64, float(g["thres"]), 200, c.PREC_F32, 1)
    (kps, d, sc), = sp.postprocess_dense(g["semi"], g["desc"])
    assert np.array_equal(kps.astype(np.int32), g["kps"]), (kps[:5], g["kps"][:5])
    assert np.array_equal(sc, g["conf"])
    return f"desc err {np.abs(d - g['desc64']).max():.2e}"


@check("mobilenetvlad small + full")
def _():
    vw = mobilenetvlad_ref.synth_weights()
    out = []
    for (h, w, n) in ((96, 128, 2), (480, 600, 4)):
        net = c.MobileNetVLAD(ctx, vw, mobilenetvlad_ref.layer_specs(), 32, 112, 4096, w, h, n)
        imgs = np.stack([synth.image_u8(300 + i, h, w, n_shapes=60) for i in range(n)])
        y = net.inference(imgs)
        yr = mobilenetvlad_ref.forward(vw, imgs)
        rel = np.linalg.norm(y - yr, axis=1) / np.linalg.norm(yr, axis=1)
        out.append(f"{w}x{h} rel err {rel.max():.2e}")
        net.close()
    return " ".join(out)


@check("timings")
def _():
    out = []
    # loop-match latency, 100k rows fp32
    n = 100_000
    idx = c.IndexFlatIP(ctx, 4096, capacity=n)
    rng = np.random.default_rng(0)
    for s in range(0, n, 10000):
        blk = rng.standard_normal((10000, 4096), dtype=np.float32)
        blk /= np.linalg.norm(blk, axis=1, keepdims=True)
        idx.add(blk)
    q = blk[:1]
    lat, scan = [], []
    for i in range(30):
        t = time.perf_counter(); idx.search(q, 10); lat.append((time.perf_counter() - t) * 1e3); scan.append(idx.last_scan_ms())
    gb = n * 4096 * 4 / 1e9
    out.append(f"search 100k fp32: p50 {np.median(lat):.3f} ms (scan kernel {np.median(scan):.3f} ms = {gb / np.median(scan) * 1e3:.0f} GB/s)")
    idx.close()
    # SuperPoint per-stage, batch 8, 600x480
    weights = superpoint_ref.synth_weights(0)
    comp, mean = synth.pca()
    imgs = np.stack([synth.image_u8(i, 480, 600) for i in range(8)])
    gdev = ctx.to_device(imgs)
    for prec, pname in ((c.PREC_F16, "f16"), (c.PREC_SPLIT, "split"), (c.PREC_F32, "f32")):
        sp = c.SuperPoint(ctx, weights, comp, mean, 600, 480, 0.015, 200, prec, 8)
        sp.profile(gdev, 600, 8, 2)
        prof = sp.profile(gdev, 600, 8, 5)
        tot = sum(p["ms"] for p in prof)
        print(f"   [{pname}] batch-8 600x480 total {tot:.3f} ms -> {8 / tot * 1e3:.0f} img/s")
        for p in prof:
            tf = p["flops_per_image"] * 8 / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0
            print(f"      {p['stage']:22s} {p['ms']:8.3f} ms  {tf:8.1f} TFLOP/s", flush=True)
        out.append(f"SP {pname} b8: {tot:.2f} ms")
        sp.close()
    ctx.free(gdev)
    return " | ".join(out)


print("\n==== SUMMARY ====")
for n, s, m in RESULTS:
    print(f"{s:5s} {n}: {m[:300]}")
