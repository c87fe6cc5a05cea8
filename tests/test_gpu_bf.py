"""GPU parity: omni_bf_match (HIP) vs the cv::BFMatcher(NORM_L2, crossCheck=true) oracle.
Bar: indices bit-exact AND distances bit-exact (same fp32 operation order as oracle/csrc/oracle.c)."""
import numpy as np
import pytest

from oracle import match_ref as M
from omni_swarm_amd import synth          # seeded synthetic inputs (data generators; shared by bench.py)

pytestmark = pytest.mark.gpu


def _same(omni, ctx, a, b, mode):
    q, t, d = omni.capi.bf_match(ctx, a, b, mode)
    qr, tr, dr = M.bf_match(a, b, mode)
    assert np.array_equal(q, qr) and np.array_equal(t, tr)
    assert np.array_equal(d, dr), np.abs(d - dr).max()
    return q


def test_golden(omni, ctx, golden):
    g = golden("match.npz")
    for mode, tag in ((0, "bf0"), (1, "bf1")):
        q, t, d = omni.capi.bf_match(ctx, g["bf_a"], g["bf_b"], mode)
        assert np.array_equal(q, g[tag + "_q"]) and np.array_equal(t, g[tag + "_t"]) and np.array_equal(d, g[tag + "_d"])


@pytest.mark.parametrize("nq,nt,dim", [(1, 1, 64), (1, 50, 64), (50, 1, 64), (63, 65, 64), (64, 64, 64), (200, 200, 64),
                                       (200, 137, 64), (129, 300, 64), (90, 70, 256), (33, 47, 32)])
@pytest.mark.parametrize("mode", [0, 1])
def test_random_sets(omni, ctx, nq, nt, dim, mode):
    rng = np.random.default_rng(nq * 1000 + nt + dim)
    a = rng.standard_normal((nq, dim)).astype(np.float32)
    b = rng.standard_normal((nt, dim)).astype(np.float32)
    m = min(nq, nt) // 2
    b[:m] = a[rng.permutation(nq)[:m]] + 0.3 * rng.standard_normal((m, dim)).astype(np.float32)
    _same(omni, ctx, a, b, mode)


def test_opencv_vs_mutual_counter_example(omni, ctx):
    a = np.zeros((2, 4), np.float32); a[1, 0] = 1.5
    b = np.zeros((2, 4), np.float32); b[0, 0] = 1.0; b[1, 0] = -2.0
    q0, t0, d0 = omni.capi.bf_match(ctx, a, b, 0)
    q1, t1, _ = omni.capi.bf_match(ctx, a, b, 1)
    assert list(zip(q0, t0)) == [(0, 1), (1, 0)] and np.allclose(d0, [2.0, 0.5])
    assert list(zip(q1, t1)) == [(1, 0)]


def test_ties_first_minimum_wins_and_empty(omni, ctx):
    a = np.zeros((5, 64), np.float32)
    b = np.zeros((3, 64), np.float32)
    for mode in (0, 1):
        _same(omni, ctx, a, b, mode)
    q, t, d = omni.capi.bf_match(ctx, a[:0], b, 0)
    assert q.size == 0
    q, t, d = omni.capi.bf_match(ctx, a, b[:0], 0)
    assert q.size == 0
    dup = synth.local_descriptors(40, 64, seed=3)
    dup2 = np.concatenate([dup, dup[:10]])              # exact duplicates in the train set
    for mode in (0, 1):
        _same(omni, ctx, dup, dup2, mode)


def test_multi_pair_call_equals_single_pair_calls(omni, ctx):
    """omni_bf_match_multi (one upload / launch pair / download for several descriptor pairs: the geometric verification's four direction
    pairs) == omni_bf_match pair by pair, ragged sizes and empty sets included."""
    rng = np.random.default_rng(11)
    sizes = [(200, 200), (37, 190), (200, 1), (0, 50), (120, 0), (64, 64)]
    pairs = []
    for nq, nt in sizes:
        q = rng.standard_normal((nq, 64)).astype(np.float32)
        t = rng.standard_normal((nt, 64)).astype(np.float32)
        if nq and nt:
            t[: min(nq, nt) // 2] = q[: min(nq, nt) // 2] + 0.01 * rng.standard_normal((min(nq, nt) // 2, 64)).astype(np.float32)
        pairs.append((q, t))
    for mode in (omni.capi.BF_OPENCV, omni.capi.BF_MUTUAL):
        multi = omni.capi.bf_match_multi(ctx, pairs, mode)
        for (q, t), got in zip(pairs, multi):
            ref = omni.capi.bf_match(ctx, q, t, mode)
            assert all(np.array_equal(a, b) for a, b in zip(got, ref))
    one = omni.capi.bf_match_multi(ctx, pairs[:1])
    assert all(np.array_equal(a, b) for a, b in zip(one[0], omni.capi.bf_match(ctx, *pairs[0])))
