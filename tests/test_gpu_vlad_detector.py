"""GPU parity: MobileNetVLAD (ASSUMED architecture, parity-unpinned -- oracle/mobilenetvlad_ref.py) and the LoopDetector
host logic running on the HIP index vs the oracle's decision trace."""
import numpy as np
import pytest

from oracle import mobilenetvlad_ref as V
from omni_swarm_amd import synth          # seeded synthetic inputs (data generators; shared by bench.py)
from tests import detector_stream as DS

pytestmark = pytest.mark.gpu


def _vlad(omni, ctx, w, h, max_batch):
    return omni.capi.MobileNetVLAD(ctx, V.synth_weights(), V.layer_specs(), V.N_CLUSTERS, V.FEAT_DIM, V.OUT_DIM, w, h, max_batch)


def test_vlad_golden_small(omni, ctx, golden):
    g = golden("vlad_small.npz")
    net = _vlad(omni, ctx, 128, 96, 2)
    y = net.inference(g["images"])
    assert np.allclose(np.linalg.norm(y, axis=1), 1, atol=1e-5)
    rel = np.linalg.norm(y - g["out"], axis=1) / np.linalg.norm(g["out"], axis=1)
    assert rel.max() < 1e-3, rel                                     # north_star: VLAD vectors within 1e-3 relative


def test_vlad_full_size_batch_and_mask(omni, ctx):
    imgs = np.stack([synth.image_u8(20 + i, 480, 600) for i in range(4)])
    net = _vlad(omni, ctx, 600, 480, 4)
    y = net.inference(imgs, fisheye_mask=True)
    masked = imgs.copy()
    masked[:, 360:] = 0                                              # loop_cam.cpp:536-539
    ref = V.forward(V.synth_weights(), masked)
    rel = np.linalg.norm(y - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert rel.max() < 1e-3, rel
    y1 = net.inference(imgs[2], fisheye_mask=True)
    assert np.array_equal(y1[0], y[2])


def test_fused_stem_block0_is_bit_identical(omni, ctx, monkeypatch):
    """vlad_stem_b0_kernel (stem + block 0 in one pass, the stem map never reaches HBM) keeps the FMA order of the two separate kernels:
    same descriptors bit for bit, odd sizes (partial tiles, image borders) and the fisheye mask included."""
    vw = V.synth_weights()
    for (h, w, nb, mask) in ((96, 128, 2, False), (104, 136, 1, True), (480, 600, 2, True)):
        imgs = np.stack([synth.image_u8(700 + i, h, w, n_shapes=80) for i in range(nb)])
        outs = []
        for flag in ("1", "0"):
            monkeypatch.setenv("OMNI_VLAD_STEM_FUSE", flag)
            net = omni.capi.MobileNetVLAD(ctx, vw, V.layer_specs(), V.N_CLUSTERS, V.FEAT_DIM, V.OUT_DIM, w, h, nb)
            outs.append(net.inference(imgs, fisheye_mask=mask))
            net.close()
        assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("h,w", [(480, 600), (480, 640), (360, 488), (240, 320), (104, 136)])
def test_masked_passes_skip_the_constant_region_bit_identically(omni, ctx, monkeypatch, h, w):
    """LoopCam blanks the bottom quarter of a fisheye frame before BOTH networks run (loop_cam.cpp:536-539, then :556-558 netvlad_net.inference): inside that
    band the stem's and the first blocks' outputs are one constant vector, which a masked pass leaves out of its tile walk (csrc/vlad.hip, omni_vlad::MaskSkip;
    OMNI_VLAD_MASK_SKIP=0 = the dense pass).  Same descriptors bit for bit: every image slot, a partial batch, an unmasked pass in between, sizes whose tiles
    overhang the map and a size too small to hold a rectangle."""
    vw = V.synth_weights()
    imgs = np.stack([synth.image_u8(900 + i, h, w, n_shapes=120) for i in range(3)])
    imgs[:, h * 3 // 4:] = 200                                       # the band's content must not matter
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("OMNI_VLAD_MASK_SKIP", flag)
        net = omni.capi.MobileNetVLAD(ctx, vw, V.layer_specs(), V.N_CLUSTERS, V.FEAT_DIM, V.OUT_DIM, w, h, 3)
        fr = net.mask_skip_layers()
        if flag == "0":
            assert fr == []
        elif h >= 240:
            assert len(fr) >= 2 and all(0.02 < f < 0.25 for f in fr), fr
        a = net.inference(imgs, fisheye_mask=True)
        b = net.inference(imgs[::-1], fisheye_mask=False)             # the rotating buffers: the rectangles stay as they are
        c = net.inference(imgs[1:], fisheye_mask=True)
        d = net.inference(imgs, fisheye_mask=True)
        outs[flag] = (a, b, c, d)
        net.close()
    for x, y in zip(outs["1"], outs["0"]):
        assert np.array_equal(x, y)
    assert np.array_equal(outs["1"][0], outs["1"][3]) and np.array_equal(outs["1"][2], outs["1"][0][1:])


@pytest.mark.parametrize("size", [(96, 128, 2, False), (104, 136, 1, True), (150, 210, 3, False), (480, 600, 2, True)])
def test_split_fp16_blocks_are_fp32_class(omni, ctx, monkeypatch, size):
    """vlad_sblock_kernel (fp16 matrix cores, every operand carried as hi + lo, fp32 everywhere else) against the exact-f32 kernels it
    replaces and against the oracle: 1e-4 relative on the descriptor (measured 2e-6, as the f32 kernels), odd sizes (partial tiles on every
    edge, borders inside a tile), the fisheye mask, persistent and one-tile-per-workgroup launches bit-identical, batch-1 == batch-n."""
    h, w, nb, mask = size
    vw = V.synth_weights()
    imgs = np.stack([synth.image_u8(900 + i, h, w, n_shapes=80) for i in range(nb)])
    outs = {}
    for key, env in (("split", {"OMNI_VLAD_SBLOCK": "1", "OMNI_VLAD_SB_PERSIST": "1"}), ("split_np", {"OMNI_VLAD_SBLOCK": "1", "OMNI_VLAD_SB_PERSIST": "0"}),
                     ("f32", {"OMNI_VLAD_SBLOCK": "0", "OMNI_VLAD_SB_PERSIST": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = omni.capi.MobileNetVLAD(ctx, vw, V.layer_specs(), V.N_CLUSTERS, V.FEAT_DIM, V.OUT_DIM, w, h, nb)
        outs[key] = net.inference(imgs, fisheye_mask=mask)
        if key == "split":
            one = net.inference(imgs[nb - 1], fisheye_mask=mask)
            assert np.array_equal(one[0], outs[key][nb - 1])
        net.close()
    assert np.array_equal(outs["split"], outs["split_np"])
    masked = imgs.copy()
    if mask:
        masked[:, h * 3 // 4:] = 0
    ref = V.forward(vw, masked)
    for key in ("split", "f32"):
        rel = np.linalg.norm(outs[key] - ref, axis=1) / np.linalg.norm(ref, axis=1)
        assert rel.max() < 1e-4, (key, rel)
    rel = np.linalg.norm(outs["split"] - outs["f32"], axis=1) / np.linalg.norm(outs["f32"], axis=1)
    assert rel.max() < 1e-4, rel


def test_fp16_operand_mode_error_and_batch_invariance(omni, ctx):
    """omni_vlad_set_precision(OMNI_PREC_F16): plain fp16 operands in the inverted-residual blocks (what the reference's fp16 TensorRT plan
    does, launch/realsense.launch:10-11), fp32 accumulation and residual stream.  Gate: 1e-2 relative on the descriptor (measured 4e-3, cosine
    to the fp32 descriptor >= 0.9999), every image bit-identical to its batch-1 result; switching back restores the fp32 result bit for bit."""
    c = omni.capi
    imgs = np.stack([synth.image_u8(40 + i, 480, 600) for i in range(4)])
    net = _vlad(omni, ctx, 600, 480, 4)
    y32 = net.inference(imgs, fisheye_mask=True)
    net.set_precision(c.PREC_F16)
    y16 = net.inference(imgs, fisheye_mask=True)
    rel = np.linalg.norm(y16 - y32, axis=1) / np.linalg.norm(y32, axis=1)
    assert 1e-4 < rel.max() < 1e-2, rel
    assert (y16 * y32).sum(1).min() > 0.9999
    assert np.array_equal(net.inference(imgs[1], fisheye_mask=True)[0], y16[1])
    net.set_precision(c.PREC_F32)
    assert np.array_equal(net.inference(imgs, fisheye_mask=True), y32)
    with pytest.raises(c.OmniError):
        net.set_precision(7)


def test_loop_detector_trace_equals_oracle(omni, ctx, golden):
    from omni_swarm_amd import detector
    frames = DS.make_stream(seed=11)
    tr = DS.trace(DS.run_product(frames, ctx, detector))
    assert np.array_equal(tr, golden("detector.npz")["trace"])
    frames2 = DS.make_stream(seed=12, n_frames=60)
    assert np.array_equal(DS.trace(DS.run_product(frames2, ctx, detector)), DS.trace(DS.run_oracle(frames2)))


def test_long_replay_match_ids_equal_oracle(omni, ctx):
    """BASELINE config 3 in miniature: a 1500-key-frame, 3-drone replay (up to 6000 index rows, ~1000 queries) through the HIP
    index + host rules gives the same decision trace (added / queried / image id / matched frame / direction / loop) as the
    oracle's literal LoopDetector, including the recency rule and the init-mode thresholds."""
    from omni_swarm_amd import detector
    frames = DS.make_stream(seed=21, n_frames=1500, n_places=60)
    got, ref = DS.trace(DS.run_product(frames, ctx, detector)), DS.trace(DS.run_oracle(frames))
    assert np.array_equal(got, ref)
    assert (ref[:, 4] != -1).sum() > 100                              # the stream does close loops


@pytest.mark.parametrize("batch,on_device", [(1, False), (4, False), (4, True), (7, True)])
def test_batched_detector_equals_frame_by_frame(omni, ctx, batch, on_device):
    """on_images_recv_batch (all appends + prefix-restricted searches of several frames enqueued ahead, one host sync per batch; rows
    taken from HBM when on_device) gives the decision trace of on_image_recv called frame by frame -- and of the oracle."""
    from omni_swarm_amd import detector
    frames = DS.make_stream(seed=31, n_frames=240, n_places=20)
    seq = DS.trace(DS.run_product(frames, ctx, detector))
    got = DS.trace(DS.run_product_batched(frames, ctx, detector, batch=batch, rows_on_device=on_device))
    assert np.array_equal(got, seq)
    assert np.array_equal(got, DS.trace(DS.run_oracle(frames)))
    assert (seq[:, 4] != -1).sum() > 10



def test_c3_replay_10k_keyframes_5_drones_match_ids_equal_oracle(omni, ctx):
    """BASELINE configs[2] at size: a 10 000-key-frame, 5-drone replay (about 30 000 database rows over the local and the remote index,
    about 10 000 searches) through the HIP index + host rules -- frame by frame AND through the batched entry point (micro-batches of 8, rows
    handed over in HBM) -- gives the decision trace (added / queried / image id / matched frame / direction / loop) of the oracle's literal
    LoopDetector.  The oracle answers its searches from float64 GEMMs (tests/detector_stream.run_oracle_fast, equal to the scalar oracle on
    the short streams)."""
    from omni_swarm_amd import detector
    frames = DS.make_stream(seed=77, n_frames=10000, n_places=800, n_drones=5)
    ref = DS.trace(DS.run_oracle_fast(frames))
    got_b = DS.trace(DS.run_product_batched(frames, ctx, detector, batch=8, rows_on_device=True))
    bad = np.nonzero((got_b != ref).any(1))[0]
    assert len(bad) == 0, (len(bad), bad[:5], got_b[bad[:3]], ref[bad[:3]])
    got_s = DS.trace(DS.run_product(frames, ctx, detector))
    assert np.array_equal(got_s, ref)
    assert (ref[:, 1] == 1).sum() > 7000 and (ref[:, 4] != -1).sum() > 5000 and len({f["drone_id"] for f in frames}) == 5


def test_c3_replay_with_fp16_rows_counts_the_decisions_that_differ_from_fp32_rows(omni, ctx):
    """The same 10 000-key-frame replay with the database stored as fp16 rows (OMNI_STORE_F16: half the HBM bytes per search, what
    `db100k.f16_rows` times) against the fp32-row trace: row scores move by ~1e-4 relative, so a candidate within that of the acceptance
    threshold (`INNER_PRODUCT_THRES`, loop_detector.cpp:232) or of its runner-up can flip.  Counted here and bounded: the storage format is not
    allowed to change more than 1 decision in 1 000."""
    from omni_swarm_amd import detector
    frames = DS.make_stream(seed=77, n_frames=10000, n_places=800, n_drones=5)
    ref = DS.trace(DS.run_oracle_fast(frames))
    got16 = DS.trace(DS.run_product_batched(frames, ctx, detector, batch=8, rows_on_device=True, storage=omni.capi.STORE_F16))
    differ = int((got16 != ref).any(1).sum())
    print(f"fp16 rows: {differ} of {len(ref)} key-frame decisions differ from fp32 rows")
    assert differ <= len(ref) // 1000, differ
