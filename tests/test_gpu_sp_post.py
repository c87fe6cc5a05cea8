"""GPU parity: getKeyPoints + NMS2 + computeDescriptors on the GPU (omni_sp_postprocess_dense) vs the literal oracle,
fed the SAME engine outputs.  Bar: key points and confidences bit-exact in the fixed order; descriptors 1e-5."""
import numpy as np
import pytest

from oracle import postproc_ref as P
from oracle import superpoint_ref as S
from omni_swarm_amd import synth          # seeded synthetic inputs (data generators; shared by bench.py)

pytestmark = pytest.mark.gpu


def _sp(omni, ctx, w, h, thr, max_num=200, pca=True):
    comp, mean = synth.pca()
    return omni.capi.SuperPoint(ctx, S.synth_weights(0), comp if pca else None, mean if pca else None, w, h, thr, max_num,
                                omni.capi.PREC_F32, 1), comp, mean


def _compare(sp, semi, desc, w, h, thr, max_num, comp, mean, atol=2e-5):
    (kps, d, sc), = sp.postprocess_dense(semi, desc)
    xy, conf, nc, ns = P.get_keypoints(semi, thr, max_num)
    assert np.array_equal(kps.astype(np.int32), xy), (len(kps), len(xy))
    assert np.array_equal(sc, conf)
    ref, _ = P.compute_descriptors(desc, xy, w, h, comp, mean)
    if len(xy):
        assert np.abs(d - ref).max() < atol, np.abs(d - ref).max()
    return len(xy)


def test_golden_small(omni, ctx, golden):
    g = golden("sp_small.npz")
    sp, comp, mean = _sp(omni, ctx, 96, 64, float(g["thres"]))
    (kps, d, sc), = sp.postprocess_dense(g["semi"], g["desc"])
    assert np.array_equal(kps.astype(np.int32), g["kps"]) and np.array_equal(sc, g["conf"])
    assert np.abs(d - g["desc64"]).max() < 2e-5


def test_random_maps_with_ties_and_borders(omni, ctx):
    rng = np.random.default_rng(0)
    w, h = 96, 64
    for t in range(12):
        thr = [0.05, 0.3, 0.6][t % 3]
        sp, comp, mean = _sp(omni, ctx, w, h, thr, max_num=[200, 16, 64][t % 3])
        semi = rng.random((h, w)).astype(np.float32) ** 4
        if t % 2 == 0:
            semi = (np.round(semi * 12) / 12).astype(np.float32)      # plateaus: equal confidences never suppress
        semi[0, :] = rng.random(w) ** 2                               # activity on the image border
        semi[:, -1] = rng.random(h) ** 2
        desc = rng.standard_normal((256, h // 8, w // 8)).astype(np.float32)
        desc /= np.linalg.norm(desc, axis=0, keepdims=True)
        _compare(sp, semi, desc, w, h, thr, sp.max_num, comp, mean)
        sp.close()


def test_no_candidates_and_all_candidates(omni, ctx):
    w, h = 64, 32
    sp, comp, mean = _sp(omni, ctx, w, h, 0.5, max_num=50)
    desc = np.ones((256, 4, 8), np.float32) / 16
    (kps, d, sc), = sp.postprocess_dense(np.zeros((h, w), np.float32), desc)
    assert len(kps) == 0
    semi = np.full((h, w), 0.9, np.float32)                          # 2048 candidates, all tied -> all survive
    (kps, d, sc), = sp.postprocess_dense(semi, desc)
    xy, conf, nc, ns = P.get_keypoints(semi, 0.5, 50)
    assert nc == w * h and ns == w * h and np.array_equal(kps.astype(np.int32), xy)
    assert kps[:3].tolist() == [[0, 0], [1, 0], [2, 0]]              # ties -> row-major order


def test_more_survivors_than_the_lds_sort_capacity(omni, ctx):
    w, h = 160, 96                                                    # 15360 tied candidates > 8192-key LDS batch
    sp, comp, mean = _sp(omni, ctx, w, h, 0.1, max_num=300)
    rng = np.random.default_rng(4)
    semi = np.full((h, w), 0.5, np.float32)
    hot = rng.choice(w * h, 40, replace=False)
    semi.ravel()[hot] = 0.5 + rng.random(40).astype(np.float32) * 0.4
    desc = rng.standard_normal((256, h // 8, w // 8)).astype(np.float32)
    _compare(sp, semi, desc, w, h, 0.1, 300, comp, mean, atol=1e-4)


def test_full_size_frame_from_oracle_net(omni, ctx):
    w, h = 600, 480
    weights = S.synth_weights(0)
    img = synth.image_u8(0, h, w)
    semi, desc = S.forward(weights, S.preprocess_u8(img, fisheye_mask=True))
    for thr in (0.015, 0.2):
        sp, comp, mean = _sp(omni, ctx, w, h, thr)
        n = _compare(sp, semi[0], desc[0], w, h, thr, 200, comp, mean)
        assert n == 200
        sp.close()
    sp, comp, mean = _sp(omni, ctx, w, h, 0.015, pca=False)          # #undef USE_PCA path: 256-d
    (kps, d, sc), = sp.postprocess_dense(semi[0], desc[0])
    xy, _, _, _ = P.get_keypoints(semi[0], 0.015, 200)
    _, raw = P.compute_descriptors(desc[0], xy, w, h, comp, mean)
    assert d.shape[1] == 256 and np.abs(d - raw).max() < 2e-5


@pytest.mark.parametrize("prec", ["PREC_F16", "PREC_SPLIT", "PREC_F32"])
def test_threshold_inside_the_detector_head_equals_the_separate_candidate_kernel(omni, ctx, monkeypatch, prec):
    """getKeyPoints' threshold (superpoint_tensorrt.cpp:167-173) fused into the detector head's epilogue + sp_mask_kernel (one thread per candidate)
    against sp_cand_kernel re-reading the heat map (OMNI_SP_FUSED_CAND=0; a handle keeps the setting it was created with): identical key points, scores,
    descriptors -- batches whose 32-cell fragments straddle two images (117 cells per image), activity on every image border, the full-size frame; and
    the GPU's own heat map through the literal oracle."""
    c = omni.capi
    comp, mean = synth.pca()
    weights = S.synth_weights(0)
    for (w, h, batch, thr, max_num) in ((104, 72, 3, 0.015, 200), (600, 480, 2, 0.02, 200), (96, 64, 1, 0.5, 30)):
        imgs = np.stack([synth.image_u8(40 + i, h, w, n_shapes=80 if w > 200 else 40) for i in range(batch)])
        imgs[:, 0, :] = 255 - imgs[:, 1, :]                               # texture right at the borders: windows that leave the image
        imgs[:, :, -1] = 255 - imgs[:, :, -2]
        monkeypatch.setenv("OMNI_SP_FUSED_CAND", "1")
        fused = c.SuperPoint(ctx, weights, comp, mean, w, h, thr, max_num, getattr(c, prec), batch)
        monkeypatch.setenv("OMNI_SP_FUSED_CAND", "0")
        plain = c.SuperPoint(ctx, weights, comp, mean, w, h, thr, max_num, getattr(c, prec), batch)
        monkeypatch.delenv("OMNI_SP_FUSED_CAND")
        for mask in (False, True):
            ra, rb = fused.inference(imgs, fisheye_mask=mask), plain.inference(imgs, fisheye_mask=mask)
            semi, _ = fused.get_dense(batch)
            for b in range(batch):
                assert np.array_equal(ra[b][0], rb[b][0]) and np.array_equal(ra[b][2], rb[b][2]) and np.array_equal(ra[b][1], rb[b][1]), (w, b, mask)
                xy, conf, _, _ = P.get_keypoints(semi[b], thr, max_num)
                assert np.array_equal(ra[b][0].astype(np.int32), xy) and np.array_equal(ra[b][2], conf)
            assert sum(len(r[0]) for r in ra) > 0 or thr > 0.4
        # a heat map handed in from outside: threshold kernel + mask kernel vs sp_cand_kernel
        rng = np.random.default_rng(w)
        semi = (rng.random((h, w)).astype(np.float32) ** 3)
        semi[:, 0] = rng.random(h) ** 2
        semi[-1, :] = rng.random(w) ** 2
        desc = rng.standard_normal((256, h // 8, w // 8)).astype(np.float32)
        (ka, da, sa), = fused.postprocess_dense(semi, desc)
        (kb, db, sb), = plain.postprocess_dense(semi, desc)
        assert np.array_equal(ka, kb) and np.array_equal(sa, sb) and np.array_equal(da, db)
        xy, conf, _, _ = P.get_keypoints(semi, thr, max_num)
        assert np.array_equal(ka.astype(np.int32), xy) and np.array_equal(sa, conf)
        fused.close(); plain.close()


def test_near_ties_at_the_top_k_cut(omni, ctx):
    """The cut at rank max_num (superpoint_tensorrt.cpp:173-189: sort by confidence, keep max_num) is a discontinuity: confidences one fp32 ulp apart -- and exactly
    equal -- placed ACROSS the cut must be ordered by the product exactly as by the oracle (confidence descending, ties in row-major order), whatever the value
    differences are; what the north-star gate "same key points up to confidence ties below the fp32 noise" leaves open is the NET's noise, not the sort's."""
    w, h, max_num, thr = 160, 96, 24, 0.1
    sp, comp, mean = _sp(omni, ctx, w, h, thr, max_num=max_num)
    rng = np.random.default_rng(11)
    desc = rng.standard_normal((256, h // 8, w // 8)).astype(np.float32)
    desc /= np.linalg.norm(desc, axis=0, keepdims=True)
    # isolated peaks on a 12-pixel lattice (NMS2's radius is 4: none suppresses another), 20 clear winners, then a cluster of 12 around the cut
    sites = [(6 + 12 * (i // 12), 6 + 12 * (i % 12)) for i in range(60)]
    rng.shuffle(sites)
    base = np.float32(0.5)
    up = lambda v, k: np.nextafter(v, np.float32(1.0), dtype=np.float32) if k == 0 else up(np.nextafter(v, np.float32(1.0), dtype=np.float32), k - 1)
    for variant in range(4):
        semi = np.zeros((h, w), np.float32)
        for i, (y, x) in enumerate(sites[:20]):
            semi[y, x] = 0.9 - 0.01 * i
        cluster = sites[20:32]
        if variant == 0:      # all twelve exactly equal: row-major order decides who makes the cut
            vals = [base] * 12
        elif variant == 1:    # one ulp apart, in shuffled positions
            vals = [up(base, k) for k in range(12)]
        elif variant == 2:    # pairs of equal values one ulp apart from the next pair
            vals = [up(base, k // 2) for k in range(12)]
        else:                 # one ulp apart, descending against the row-major order of the sites
            order = np.argsort([y * w + x for (y, x) in cluster])
            vals = [None] * 12
            for rank, idx in enumerate(order):
                vals[idx] = up(base, rank)
        for (y, x), v in zip(cluster, vals):
            semi[y, x] = v
        for (y, x) in sites[32:]:
            semi[y, x] = 0.2 + 0.001 * rng.random()
        n = _compare(sp, semi, desc, w, h, thr, max_num, comp, mean)
        assert n == max_num
        xy, conf, _, _ = P.get_keypoints(semi, thr, max_num)
        assert conf[19] > base and conf[20] <= up(base, 12) and conf[-1] >= base       # the cut runs through the cluster
    sp.close()
