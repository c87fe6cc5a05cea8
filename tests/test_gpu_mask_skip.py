"""The constant region of the fisheye mask (fp16 path, on by default, OMNI_SP_MASK_SKIP=0 = the dense pass; omni_sp::MaskSkip in csrc/superpoint.hip): LoopCam blanks the lower
quarter of every image before the networks see it (loop_cam.cpp:536-539); a few pixels inside that band every activation is one vector per
layer, read once from a pass over an all-zero image and written once into the tile rectangle the persistent cin = 64 kernel then leaves out of
its walk.  The results must be BIT-IDENTICAL to the dense pass -- every layer, the dense heat map and descriptor map, key points, scores and
descriptors -- on every image, at every size (also where the band is too thin for a whole tile: nothing is skipped there), and after passes
without the mask in between (they overwrite the rectangles: the next masked pass calibrates again)."""
import numpy as np
import pytest

from oracle import superpoint_ref as S
from omni_swarm_amd import synth

pytestmark = pytest.mark.gpu
LAYERS = ["conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "heads"]


def _same(a, b):
    assert len(a) == len(b)
    for (k0, d0, s0), (k1, d1, s1) in zip(a, b):
        assert np.array_equal(k0, k1) and np.array_equal(s0, s1) and np.array_equal(d0, d1)


@pytest.mark.parametrize("prec", ["PREC_F16", "PREC_SPLIT", "PREC_SPLIT_UNFUSED"])
@pytest.mark.parametrize("shape,batch", [((480, 600), 3), ((480, 640), 2), ((240, 320), 2), ((64, 96), 2)])
def test_mask_skip_is_bit_identical_to_the_dense_pass(omni, ctx, shape, batch, prec, monkeypatch):
    """(OMNI_PREC_SPLIT: the four cin = 64 layers of the split kernel, 4 x 32 tiles; with OMNI_SPLIT_FUSE1A=0 conv1a is a tensor of its own and has its
    own constant band; OMNI_SP_MASK_SKIP_SPLIT=0 = the dense pass)"""
    h, w = shape
    env = "OMNI_SP_MASK_SKIP" if prec == "PREC_F16" else "OMNI_SP_MASK_SKIP_SPLIT"
    monkeypatch.setenv("OMNI_SPLIT_FUSE1A", "0" if prec == "PREC_SPLIT_UNFUSED" else "1")
    unfused = prec == "PREC_SPLIT_UNFUSED"
    prec = prec.replace("_UNFUSED", "")
    weights = S.synth_weights(0)
    comp, mean = synth.pca()
    imgs = np.stack([synth.image_u8(900 + i, h, w, n_shapes=60 if h < 100 else 200) for i in range(batch)])
    imgs2 = np.stack([synth.image_u8(950 + i, h, w, n_shapes=60 if h < 100 else 200) for i in range(batch + 1)])
    sps = []
    for flag in ("0", "1"):
        monkeypatch.setenv(env, flag)
        sps.append(omni.capi.SuperPoint(ctx, weights, comp, mean, w, h, 0.015, 200, getattr(omni.capi, prec), batch + 1))
    dense, skip = sps
    # 1. masked pass: every output and every layer
    _same(dense.inference(imgs, True), skip.inference(imgs, True))
    for n in LAYERS + (["conv1a"] if unfused else []):
        a, b = dense.debug_layer(n, batch), skip.debug_layer(n, batch)
        assert np.array_equal(a, b), (n, int((a != b).sum()), np.argwhere(a != b)[:4].tolist())
    (s0, d0), (s1, d1) = dense.get_dense(batch), skip.get_dense(batch)
    assert np.array_equal(s0, s1) and np.array_equal(d0, d1)
    # 2. a pass WITHOUT the mask overwrites the rectangles; a larger masked batch afterwards (the spare image slot was filled too)
    _same(dense.inference(imgs[:1], False), skip.inference(imgs[:1], False))
    _same(dense.inference(imgs2, True), skip.inference(imgs2, True))
    for n in ("conv1b", "conv3a", "conv4b"):
        assert np.array_equal(dense.debug_layer(n, batch + 1), skip.debug_layer(n, batch + 1)), n
    # 3. and the masked band really is constant where the plan says so (480 x 600: conv1b's pooled rows 184-235, columns 16-287)
    if (h, w) == (480, 600) and prec == "PREC_F16":
        a = skip.debug_layer("conv1b", batch)
        band = a[:, :, 184:236, 16:288]
        assert np.array_equal(band, np.broadcast_to(band[:1, :, :1, :1], band.shape))
    for sp in sps:
        sp.close()
