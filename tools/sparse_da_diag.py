"""Diagnostic (GPU): OMNI_PREC_SPLIT descriptors with convDa at the key points only (conv_split_c128_sparse) against the dense convDa, per key point and per
channel, without PCA: which key points / channels differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import omni_loader
omni = omni_loader.load()
from oracle import superpoint_ref as S
from omni_swarm_amd import synth
c = omni.capi
ctx = c.Context(0)
weights = S.synth_weights(0)
for (h, w, nb, thr, maxn) in ((480, 600, 2, 0.015, 200), (96, 128, 3, 0.015, 37)):
    imgs = np.stack([synth.image_u8(310 + i, h, w, n_shapes=60) for i in range(nb)])
    res = {}
    for flag in ("1", "0"):
        os.environ["OMNI_SP_SPARSE_DA"] = flag
        sp = c.SuperPoint(ctx, weights, None, None, w, h, thr, maxn, c.PREC_SPLIT, nb)
        res[flag] = sp.inference(imgs, fisheye_mask=(h == 480))
        sp.close()
    for b in range(nb):
        (k1, d1, s1), (k0, d0, s0) = res["1"][b], res["0"][b]
        print(h, w, "image", b, "kps", len(k1), len(k0), "same kps", np.array_equal(k1, k0))
        if d1.shape != d0.shape:
            continue
        bad = np.abs(d1 - d0) > 0
        print("  differing key points:", np.nonzero(bad.any(1))[0].tolist()[:80], "of", len(k1))
        print("  differing channels  :", np.nonzero(bad.any(0))[0].tolist()[:80], "n =", int(bad.any(0).sum()))
        if bad.any():
            i = int(np.nonzero(bad.any(1))[0][0])
            print("  first bad kp", i, k1[i], "sparse", d1[i, :8], "dense", d0[i, :8], "max abs diff", float(np.abs(d1 - d0).max()))
