"""CPU checks of the packed constants of two SuperPoint kernels (host code inside libomni_hip.so, reached through the test hook omni_sp_pack_constants; the
kernels themselves are checked on the GPU against the torch oracle, tests/test_gpu_superpoint.py):

* conv1a inside the fp16 conv1b kernel with its matrix-core operands taken straight from the image bytes (csrc/conv.hip conv1a_pack_u8_weights, round 6):
  the B operand of a tap is half(4 + p / 256) = 0x4400 | p, the A operand the (hi, lo) split of w * 256 / 255, and the bias slot takes the offset back --
  the algebra  sum_t W_t (4 + p_t / 256) + bias' = sum_t w_t (p_t / 255) + bias  of the reference's first layer on `convertTo(CV_32F, 1/255.0)` input
  (superpoint_tensorrt.cpp:104-110), replayed here in exact arithmetic on the packed halfs;
* the Winograd F(2x2,3x3) fragments of a cin = 64 layer (csrc/conv_wino.hip conv_pack_weights_wino): U = G g G^T per (cout, cin), scaled by a power of two
  into fp16's range and split into (hi, lo) halfs, laid out [cout group][i][hi | lo][j][k group][m][lane][8] with the output channels rotated by 16 per
  position row i (the exchange of the kernel's four waves)."""
import numpy as np


def _halfs(u16):
    return u16.view(np.float16).astype(np.float64)


def test_conv1a_byte_operand_fragments_reproduce_the_layer(omni):
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((64, 9)) * 0.4).astype(np.float32)
    w[5] = 0.0
    w[7, :] = np.float32(1.75)                                   # a large sum of taps: the offset 4 sum(W) is 60 x the bias
    bias = (rng.standard_normal(64) * 0.1).astype(np.float32)
    frag, scale = omni.capi.sp_pack_constants(0, w, bias)
    assert scale == 1.0 and frag.shape == (2048,)
    f = _halfs(frag).reshape(2, 2, 64, 8)                        # [k step j][m][lane][slot]
    # the B operand the kernel builds: lanes 0-31 own taps 0-4, lanes 32-63 taps 5-8 and the bias slot (conv.hip build_finish, fz.lut_hl == nullptr)
    for p in (np.zeros(9, np.int64), np.full(9, 255), rng.integers(0, 256, 9), rng.integers(0, 256, 9)):
        P = 4.0 + p / 256.0
        assert all(np.float16(x) == x for x in P)               # exact in fp16: 0x4400 | p
        for co in range(64):
            m, n = divmod(co, 32)
            acc = 0.0
            for hh in (0, 1):
                a0, a1 = f[0, m, hh * 32 + n], f[1, m, hh * 32 + n]
                taps = [0, 1, 2, 3] if hh == 0 else [5, 6, 7, 8]
                b0 = np.repeat(P[taps], 2)                       # [P P] per tap against [Wh Wl]
                b1 = np.zeros(8)
                if hh == 0:
                    b1[:2] = P[4]
                else:
                    b1[:2] = 1.0                                 # the bias slot
                acc += float(a0 @ b0 + a1 @ b1)
            ref = float(np.dot(w[co].astype(np.float64), p / 255.0) + bias[co])
            assert abs(acc - ref) <= 2.0 ** -21 * (np.abs(w[co]).sum() * 5 + abs(bias[co]) + 1e-30), (co, acc, ref)
    # unused slots are zero (they meet zero operands, but NaN x 0 would not be)
    assert np.all(f[1, :, :32, 2:] == 0) and np.all(f[1, :, 32:, 2:] == 0)


def test_winograd_fragments_are_the_split_transformed_weights(omni):
    rng = np.random.default_rng(5)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    for cout in (64, 128):
        g = (rng.standard_normal((cout, 64, 3, 3)) * 0.05).astype(np.float32)
        frag, inv = omni.capi.sp_pack_constants(1, g, None, cout)
        U = np.einsum("ik,ockl,jl->ijoc", G, g.astype(np.float64), G)       # [i][j][cout][cin]
        k = int(round(-np.log2(inv)))
        assert inv == 2.0 ** -k and 256 <= np.abs(U).max() * 2.0 ** k < 512  # scaled into [256, 512): far from fp16's subnormals and its overflow
        f = _halfs(frag).reshape(cout // 64, 4, 2, 4, 4, 2, 64, 8)           # [cg][i][hl][j][kg][m][lane][e]
        got = np.zeros_like(U)
        seen = np.zeros(U.shape, bool)
        for cg in range(cout // 64):
            for i in range(4):
                for m in range(2):
                    for lane in range(64):
                        co = cg * 64 + ((m * 32 + (lane & 31) + 16 * i) & 63)
                        for kg in range(4):
                            ci = kg * 16 + (lane >> 5) * 8 + np.arange(8)
                            for j in range(4):
                                got[i, j, co, ci] = (f[cg, i, 0, j, kg, m, lane] + f[cg, i, 1, j, kg, m, lane]) * inv
                                seen[i, j, co, ci] = True
        assert seen.all()                                                      # every (position, cout, cin) exactly once per wave row: the rotation is a permutation
        err = np.abs(got - U.astype(np.float32).astype(np.float64)).max()     # (the packer rounds U to fp32 first, as the kernel's scale is applied in fp32)
        assert err <= 2.0 ** -21 * np.abs(U).max(), err
        # the hi part alone is fp16-accurate only: the lo part is what makes the scheme fp32-class
        hi_only = np.abs(f[:, :, 0] * inv).max()
        assert hi_only > 0 and np.abs(f[:, :, 1]).max() > 0
