#!/usr/bin/env python
"""Pins the oracle against the reference and writes the golden fixtures under tests/golden/.

Runs ONLY in the build container (it reads /root/reference, which does not exist on the GPU box):

  1. exec()s cell 3 of /root/reference/swarm_loop/superpoint.ipynb -- the reference's own PyTorch definition of
     the graph its TensorRT engine runs (superpoint.ipynb:135-205) -- loads the seeded synthetic weights
     through load_state_dict exactly like the notebook loads superpoint_v1.pth (:270), and checks that
     oracle/superpoint_ref.forward reproduces its (semi, desc) outputs;
  2. writes small fixtures the GPU tests compare against:
       sp_small.npz   64x96 frame : reference semi/desc + oracle key points / descriptors
       sp_full.npz    600x480 frames (one fisheye-masked): key points, confidences, 64-d descriptors,
                      checksums of the reference dense outputs
       match.npz      exact IP top-k results on a seeded DB, BFMatcher results on seeded descriptor sets
       detector.npz   LoopDetector decision trace on a seeded two-drone descriptor stream
     The reference holds no golden vectors of its own, so these are the pinned outputs of
     the reference's importable code (the net) and of the literal restatements (everything else).

Usage: python tools/gen_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import match_ref, mobilenetvlad_ref, postproc_ref, superpoint_ref  # noqa: E402
import omni_loader  # noqa: E402

omni_loader.load()
from omni_swarm_amd import synth  # noqa: E402  (seeded synthetic inputs: data, shared with bench.py)

REF_NB = "/root/reference/swarm_loop/superpoint.ipynb"
OUT = os.path.join(ROOT, "tests", "golden")


def reference_net(weights):
    nb = json.load(open(REF_NB))
    ns = {"torch": torch}
    exec("".join(nb["cells"][3]["source"]), ns)          # the export-variant SuperPointNet, unmodified
    model = ns["SuperPointNet"]()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    model.eval()
    return model


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    w = superpoint_ref.synth_weights(0)
    model = reference_net(w)
    comp, mean = synth.pca()

    def ref_forward(img01):
        with torch.no_grad():
            semi, desc = model(torch.from_numpy(img01)[None, None])
        return semi.numpy(), desc.numpy()

    # ---- 1. pin the network restatement -------------------------------------------------------------------
    worst = 0.0
    for idx, (h, wd, mask) in enumerate([(64, 96, False), (480, 600, True), (480, 640, False), (208, 400, False)]):
        img = synth.image_u8(200 + idx, h, wd, n_shapes=40 if h < 100 else 200, fisheye_mask=mask)
        x = superpoint_ref.preprocess_u8(img)
        rs, rd = ref_forward(x)
        os_, od = superpoint_ref.forward(w, x)
        e1, e2 = np.abs(rs - os_).max(), np.abs(rd - od).max()
        worst = max(worst, e1, e2)
        print(f"net restatement vs reference notebook  {wd}x{h}: max|dsemi|={e1:.3e} max|ddesc|={e2:.3e}")
    assert worst <= 1e-6, "oracle/superpoint_ref.py deviates from the reference notebook"

    # ---- 2a. small frame: dense outputs committed -----------------------------------------------------------
    img = synth.image_u8(100, 64, 96, n_shapes=40)
    semi, desc = ref_forward(superpoint_ref.preprocess_u8(img))
    thr = 0.015
    xy, conf, nc, ns = postproc_ref.get_keypoints(semi[0], thr, 200)
    d64, d256 = postproc_ref.compute_descriptors(desc[0], xy, 96, 64, comp, mean)
    np.savez_compressed(os.path.join(OUT, "sp_small.npz"), image=img, semi=semi[0], desc=desc[0], thres=np.float32(thr),
                        kps=xy, conf=conf, n_cand=nc, n_surv=ns, desc64=d64, desc256=d256)
    print(f"sp_small: {nc} candidates, {ns} survivors, {len(xy)} key points")

    # ---- 2b. full-size frames: sparse outputs + checksums --------------------------------------------------
    full = {}
    for i, (idx, mask) in enumerate([(0, True), (1, False)]):
        img = synth.image_u8(idx, 480, 600, fisheye_mask=False)     # mask applied by the pipeline itself
        x = superpoint_ref.preprocess_u8(img, fisheye_mask=mask)
        semi, desc = ref_forward(x)
        for thr in (0.015, 0.2):
            xy, conf, nc, ns = postproc_ref.get_keypoints(semi[0], thr, 200)
            d64, _ = postproc_ref.compute_descriptors(desc[0], xy, 600, 480, comp, mean)
            tag = f"img{i}_thr{int(thr * 1000)}"
            full[tag + "_kps"], full[tag + "_conf"], full[tag + "_desc64"] = xy, conf, d64
            full[tag + "_counts"] = np.array([nc, ns])
            print(f"sp_full {tag}: {nc} candidates, {ns} survivors")
        full[f"img{i}_semi_sum"] = np.float64(semi.astype(np.float64).sum())
        full[f"img{i}_desc_abs_sum"] = np.float64(np.abs(desc.astype(np.float64)).sum())
        full[f"img{i}_semi_rows"] = semi[0, ::48, ::60].copy()        # 10x10 probe grid of the heat map
        full[f"img{i}_desc_probe"] = desc[0, :, ::12, ::15].copy()    # 256x5x5 probe grid of the coarse descriptors
    full["image_index"] = np.array([0, 1])
    full["image_mask"] = np.array([1, 0])
    np.savez_compressed(os.path.join(OUT, "sp_full.npz"), **full)

    # ---- 2c. matcher ------------------------------------------------------------------------------------------
    db = synth.global_db(3000, seed=3)
    q, rows = synth.queries_from_db(db, 8, seed=4)
    D, I = match_ref.ip_search(db, q, 10)
    a, b, perm = synth.local_descriptors(150, 64, seed=5, pair_noise=0.2)
    b = b[:130]
    m0 = match_ref.bf_match(a, b, 0)
    m1 = match_ref.bf_match(a, b, 1)
    np.savez_compressed(os.path.join(OUT, "match.npz"), ip_D=D, ip_I=I, ip_rows=rows, bf_a=a, bf_b=b,
                        bf0_q=m0[0], bf0_t=m0[1], bf0_d=m0[2], bf1_q=m1[0], bf1_t=m1[1], bf1_d=m1[2])
    print(f"match: ip top-1 hits {int((I[:, 0] == rows).sum())}/8, bf opencv {len(m0[0])} matches, mutual {len(m1[0])}")

    # ---- 2d. LoopDetector decision trace ------------------------------------------------------------------------
    from tests.detector_stream import make_stream, run_oracle       # shared with the tests
    frames = make_stream(seed=11)
    log = run_oracle(frames)
    np.savez_compressed(os.path.join(OUT, "detector.npz"),
                        trace=np.array([[r["msg_id"], int(r["added"]), int(r["queried"]), r["image_id"], r["old_msg_id"],
                                         r["dir_old"], int(r["loop"])] for r in log], np.int64))
    print(f"detector: {len(log)} frames, {sum(1 for r in log if r['old_msg_id'] >= 0)} candidates")

    # ---- 2e. MobileNetVLAD (assumed architecture; oracle-only fixture) -----------------------------------------
    vw = mobilenetvlad_ref.synth_weights()
    imgs = np.stack([synth.image_u8(300 + i, 96, 128, n_shapes=60) for i in range(2)])
    y = mobilenetvlad_ref.forward(vw, imgs)
    np.savez_compressed(os.path.join(OUT, "vlad_small.npz"), images=imgs, out=y)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
