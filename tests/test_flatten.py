"""Fisheye flattening (FisheyeUndist, swarm_localization/test/fisheye_undist.hpp:57-215): undistortion maps from the C++ host header against
the numpy oracle (CPU), the HIP remap against the oracle's cv::cuda::remap restatement (GPU, bit-exact), and the flattened views feeding
SuperPoint without leaving the GPU."""
import numpy as np
import pytest

from oracle import flatten_ref as F
from omni_swarm_amd import synth          # seeded synthetic inputs (data generators; shared by bench.py)

MEI = (1.8, -0.2, 0.05, 0.001, -0.002, 1100.0, 1098.0, 640.0, 512.0)       # a DJI-class fisheye in camodocal's MEI parametrisation


def test_undist_maps_cpp_equal_oracle(omni):
    from omni_swarm_amd import flatten
    for fov, cam_id, width in ((235.0, 0, 600), (235.0, 1, 600), (190.0, 0, 400), (170.0, 0, 320)):
        got = flatten.generate_undist_maps(MEI, width, fov, cam_id)
        ref = F.generate_all_undist_maps(MEI, width, fov, cam_id)
        assert len(got) == len(ref) == (5 if fov > 180 else 1)
        for a, b in zip(got, ref):
            assert a.shape == b.shape and np.abs(a - b).max() < 1e-3          # f64 trigonometry on both sides, rounded to f32
    top = flatten.generate_undist_maps(MEI, 600, 235.0, 0)[0]
    assert np.allclose(top[300, 300], [640.0, 512.0])                          # the optical axis lands on the principal point
    side = flatten.generate_undist_maps(MEI, 600, 235.0, 0)[1]
    assert side.shape == (312, 600, 2)                                         # sideImgHeight = 2 * 300 * tan(27.5 deg)


def test_undist_maps_are_pinned_to_the_reference_text():
    """oracle.flatten_ref.generate_all_undist_maps against the reference's own FisheyeUndist::generateAllUndistMap / genOneUndistMap compiled from
    their own text (oracle/_ref/libref_flatten.so; camodocal's MEI projection, Eigen's quaternion algebra and cv::Mat as stand-ins): the same
    number of views, the same side-image height (integer truncation of 2 f tan(fov_side / 2)), the same maps -- the rotation sequence (incl. the
    cam_id == 1 flip), the two focal lengths and the pixel -> ray map -- to float32 rounding (the oracle composes rotation matrices, the
    reference quaternions)."""
    import ctypes
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"])
    path = os.path.join(root, "oracle", "_ref", "libref_flatten.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_flatten.so not built (needs /root/reference once)")
    L = ctypes.CDLL(path)
    L.ref_flatten_maps.restype = ctypes.c_int
    L.ref_flatten_maps.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
    mei = (ctypes.c_double * 9)(*MEI)
    for fov, cam_id, width in ((235.0, 0, 600), (235.0, 1, 600), (190.0, 0, 400), (170.0, 0, 320), (200.0, 1, 256)):
        out = np.zeros(5 * width * width * 2, np.float32)
        heights = (ctypes.c_int * 5)()
        n = L.ref_flatten_maps(mei, width, fov, cam_id, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), heights)
        ref = F.generate_all_undist_maps(MEI, width, fov, cam_id)
        assert n == len(ref)
        o = 0
        for m in range(n):
            h = heights[m]
            assert (h, width, 2) == ref[m].shape
            got = out[o:o + h * width * 2].reshape(h, width, 2)
            o += h * width * 2
            assert np.abs(got - ref[m]).max() <= 2e-4 * max(1.0, np.abs(ref[m]).max() / 1000), (fov, cam_id, m, np.abs(got - ref[m]).max())
            assert (got != ref[m]).mean() < 0.05                                     # float32 rounding of 1e-13 differences: a few last-place flips at most


@pytest.mark.gpu
def test_remap_bit_exact_and_feeds_superpoint(omni, ctx):
    from omni_swarm_amd import flatten
    from oracle import superpoint_ref as S
    c = omni.capi
    rng = np.random.default_rng(8)
    fish = np.stack([synth.image_u8(900 + i, 1024, 1280, n_shapes=400) for i in range(2)])
    fu = flatten.FisheyeUndist(ctx, MEI, 1280, 1024, 235.0, 600, 0)
    got = fu.flatten(fish)
    for b in range(2):
        for v, m in enumerate(fu.maps):
            ref = F.remap_linear(fish[b], m)
            assert np.array_equal(got[b][v], ref), (b, v, np.abs(got[b][v].astype(int) - ref.astype(int)).max())
    assert got[0][1].std() > 5 and (got[0][0] == 0).mean() < 0.5                # real content; the top view is inside the image circle
    # maps that leave the image: constant-0 border, no wild reads
    far = [np.full((16, 32, 2), -50.0, np.float32), np.stack(np.meshgrid(np.linspace(1270, 1290, 32), np.linspace(1015, 1030, 16)), -1).astype(np.float32)]
    fl = c.Flatten(ctx, 1280, 1024, far)
    out = fl(fish[0])[0]
    assert (out[0] == 0).all() and np.array_equal(out[1], F.remap_linear(fish[0], far[1]))
    fl.close()
    # the four side views stay in HBM and go straight into SuperPoint (600 x 312 each)
    src = ctx.to_device(fish[0])
    flat = ctx.alloc(fu.flatten.out_bytes)
    fu.flatten.enqueue_dev(src, 1280, 1, flat)
    ctx.sync()
    weights = S.synth_weights(0)
    comp, mean = synth.pca()
    sp = c.SuperPoint(ctx, weights, comp, mean, 600, 312, 0.015, 200, c.PREC_F32, 4)
    sp.enqueue_dev(flat + 600 * 600, 600, 4)                                   # skip the 600 x 600 top view
    res = sp.fetch(4)
    for v in range(4):
        (k1, d1, s1), = sp.inference(got[0][1 + v])
        assert np.array_equal(res[v][0], k1) and np.array_equal(res[v][1], d1) and len(k1) > 50
    sp.close(); fu.close()
    ctx.free(src); ctx.free(flat)
