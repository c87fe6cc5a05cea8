"""Numerics of the Winograd F(2x2,3x3) form of the split convolutions, simulated on the CPU (test infrastructure / design evidence, not product):
fp32 input and output transforms, (hi, lo) fp16 split of the TRANSFORMED tiles and of the transformed weights, three fp32-accumulated
products per term, against the torch fp32 oracle's layers (the per-layer gate of tests/test_gpu_superpoint.py: 2e-5 of the layer's magnitude)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import omni_loader; omni_loader.load()
from oracle import superpoint_ref as S
from omni_swarm_amd import synth

f32 = np.float32
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], f32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], f32)

def split(v):
    hi = v.astype(np.float16)
    lo = (v - hi.astype(f32)).astype(np.float16)
    return hi.astype(f32), lo.astype(f32)

def wino_layer(x, w, b, act_scale=32.0, mode="split"):
    """x [C,H,W] f32 (H, W even), w [Co,C,3,3], b [Co] -> relu(conv) [Co,H,W] f32"""
    C, H, W = x.shape
    Co = w.shape[0]
    U = np.einsum("ik,ockl,jl->ijoc", G, w.astype(np.float64), G)          # [4,4,Co,C] exact-ish in f64
    mx = np.abs(w).max()
    k = 9 - int(np.frexp(mx)[1])
    Us = (U * 2.0 ** k).astype(f32)
    xp = np.zeros((C, H + 2, W + 2), f32); xp[:, 1:-1, 1:-1] = x * f32(act_scale)
    th, tw = H // 2, W // 2
    # d[c, ty, tx, r, s]
    d = np.empty((C, th, tw, 4, 4), f32)
    for r in range(4):
        for s in range(4):
            d[:, :, :, r, s] = xp[:, r:r + 2 * th:2, s:s + 2 * tw:2]
    # fp32 transforms, one rounding per add: vertical then horizontal
    Wv = np.empty_like(d)
    Wv[..., 0, :] = d[..., 0, :] - d[..., 2, :]; Wv[..., 1, :] = d[..., 1, :] + d[..., 2, :]
    Wv[..., 2, :] = d[..., 2, :] - d[..., 1, :]; Wv[..., 3, :] = d[..., 1, :] - d[..., 3, :]
    V = np.empty_like(d)
    V[..., 0] = Wv[..., 0] - Wv[..., 2]; V[..., 1] = Wv[..., 1] + Wv[..., 2]
    V[..., 2] = Wv[..., 2] - Wv[..., 1]; V[..., 3] = Wv[..., 1] - Wv[..., 3]
    M = np.empty((Co, th, tw, 4, 4), f32)
    for i in range(4):
        for j in range(4):
            v = V[:, :, :, i, j].reshape(C, -1)
            u = Us[i, j]
            if mode == "split":
                vh, vl = split(v); uh, ul = split(u)
                m = (uh @ vh + ul @ vh) + uh @ vl          # fp32 accumulate (BLAS order, not the MFMA's)
            elif mode == "f16":
                m = u.astype(np.float16).astype(f32) @ v.astype(np.float16).astype(f32)
            else:
                m = u @ v
            M[:, :, :, i, j] = m.reshape(Co, th, tw)
    T = np.empty((Co, th, tw, 4, 2), f32)
    T[..., 0] = (M[..., 0] + M[..., 1]) + M[..., 2]; T[..., 1] = (M[..., 1] - M[..., 2]) - M[..., 3]
    Y = np.empty((Co, th, tw, 2, 2), f32)
    Y[..., 0, :] = (T[..., 0, :] + T[..., 1, :]) + T[..., 2, :]; Y[..., 1, :] = (T[..., 1, :] - T[..., 2, :]) - T[..., 3, :]
    inv = f32(2.0 ** -k / act_scale)
    y = Y.transpose(0, 1, 3, 2, 4).reshape(Co, H, W) * inv + b[:, None, None]
    return np.maximum(y, 0).astype(f32), float(np.abs(V).max()), float(np.abs(Us).max())

def direct_split_layer(x, w, b):
    import torch, torch.nn.functional as F
    xh, xl = split(x * f32(32.0))
    mx = np.abs(w).max(); k = 9 - int(np.frexp(mx)[1])
    wh, wl = split(w * f32(2.0 ** k))
    t = lambda a: torch.from_numpy(a[None] if a.ndim == 3 else a)
    y = F.conv2d(t(xh), t(wh), padding=1) + F.conv2d(t(xh), t(wl), padding=1) + F.conv2d(t(xl), t(wh), padding=1)
    y = y[0].numpy() * f32(2.0 ** -k / 32.0) + b[:, None, None]
    return np.maximum(y, 0).astype(f32)

def main():
    import torch, torch.nn.functional as F
    h, w = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (96, 128)))
    wt = S.synth_weights(0)
    img = synth.image_u8(400, h, w, n_shapes=60 if h < 100 else 200)
    semi, desc, inter = S.forward(wt, S.preprocess_u8(img), return_intermediates=True)
    pool = lambda a: F.max_pool2d(torch.from_numpy(a), 2, 2).numpy()
    chain = [("conv1b", inter["conv1a"][0]), ("conv2a", pool(inter["conv1b"])[0]), ("conv2b", inter["conv2a"][0]),
             ("conv3a", pool(inter["conv2b"])[0]), ("conv3b", inter["conv3a"][0]), ("conv4a", pool(inter["conv3b"])[0]), ("conv4b", inter["conv4a"][0])]
    print(f"{'layer':8s} {'|ref|max':>9s} {'wino-split':>11s} {'direct-split':>12s} {'wino-f32':>10s} {'gate':>9s}  |V|max  |U|max")
    for name, x in chain:
        W_, b_ = wt[name + ".weight"], wt[name + ".bias"]
        ref = inter[name][0]
        if x.shape[1] % 2 or x.shape[2] % 2:
            x = np.pad(x, ((0, 0), (0, x.shape[1] % 2), (0, x.shape[2] % 2)))
        y, vmax, umax = wino_layer(x, W_, b_)
        y32, _, _ = wino_layer(x, W_, b_, mode="f32")
        yd = direct_split_layer(x, W_, b_)
        H0, W0 = ref.shape[1:]
        e = lambda a: np.abs(a[:, :H0, :W0] - ref).max()
        print(f"{name:8s} {np.abs(ref).max():9.3f} {e(y):11.3e} {e(yd):12.3e} {e(y32):10.3e} {2e-5 * max(1, np.abs(ref).max()):9.2e}  {vmax:7.1f} {umax:7.1f}")

if __name__ == "__main__":
    main()
