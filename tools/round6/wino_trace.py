"""OMNI_WINO_TRACE=1: the in-kernel phase trace of the Winograd split kernels at the bench's launch shape (64 images, 600x480, mask on)."""
import os, sys
import numpy as np
os.environ["OMNI_WINO_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import omni_loader
omni = omni_loader.load()
from oracle import superpoint_ref as S
from omni_swarm_amd import synth
weights = S.synth_weights(0)
comp, mean = synth.pca()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
imgs = np.stack([synth.image_u8(400 + i % 8, 480, 600) for i in range(B)])
ctx = omni.capi.Context(0)
sp = omni.capi.SuperPoint(ctx, weights, comp, mean, 600, 480, 0.02, 200, omni.capi.PREC_SPLIT, B)
for _ in range(6):
    sp.inference(imgs, fisheye_mask=True)
sp.close()
