"""MobileNetVLAD oracle (CPU, fp32) -- test infrastructure only.  PARITY UNPINNED.

The reference ships only the I/O contract of this network
(/root/reference/swarm_loop/src/mobilenetvlad_tensorrt.cpp:4-14, include/swarm_loop/mobilenetvlad_tensorrt.h:9-21:
input blob ``image:0`` = H x W float32 in [0,255] (cv::Mat::convertTo(CV_32F), NO scaling), output blob
``descriptor:0`` = 4096 float32).  The graph itself is the HF-Net ``mobilenetvlad`` TF saved-model
(ethz-asl/hfnet, un-vendored, un-pinned; loaded by swarm_loop/scripts/HFNet_server_TF2.py:155-181), which is
absent from /root/reference and cannot be fetched.  What follows is therefore an ASSUMED architecture, restated
from the published HF-Net design:

  x = (image - 128) / 128, tiled to 3 channels
  MobileNetV2 (width multiplier 0.35) stem + 17 inverted-residual blocks -> 112 channels at stride 32
      (ReLU6, batch-norm folded into conv weight+bias, symmetric padding 1 -- torch convention, the TF model
       uses SAME padding)
  NetVLAD, K=32 clusters: soft-assign = softmax(1x1 conv), V[k] = sum_p a_k(p) * (c_k - f(p)),
      intra-normalise each V[k], flatten (k-major), L2-normalise
  FC 3584 -> 4096 (+bias), L2-normalise

The layer table is data (``BLOCKS``) so it can be swapped if the real model becomes available.  Every report that
quotes numbers from this net must carry the "parity-unpinned, assumed architecture" flag.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

N_CLUSTERS = 32
OUT_DIM = 4096
STEM_OUT = 16
# (expand t, cout, stride) per inverted-residual block; MobileNetV2 (t,c,n,s) table at alpha=0.35, make_divisible 8
BLOCKS = (
    [(1, 8, 1)] +
    [(6, 8, 2), (6, 8, 1)] +
    [(6, 16, 2), (6, 16, 1), (6, 16, 1)] +
    [(6, 24, 2), (6, 24, 1), (6, 24, 1), (6, 24, 1)] +
    [(6, 32, 1), (6, 32, 1), (6, 32, 1)] +
    [(6, 56, 2), (6, 56, 1), (6, 56, 1)] +
    [(6, 112, 1)]
)
FEAT_DIM = BLOCKS[-1][1]          # 112
VLAD_DIM = N_CLUSTERS * FEAT_DIM  # 3584


def layer_specs():
    """Flat list of (name, kind, cin, cout, stride) -- the same table the HIP side walks."""
    specs = [("stem", "conv3x3", 3, STEM_OUT, 2)]
    cin = STEM_OUT
    for i, (t, c, s) in enumerate(BLOCKS):
        hid = cin * t
        if t != 1:
            specs.append((f"b{i}.expand", "pw_relu6", cin, hid, 1))
        specs.append((f"b{i}.dw", "dw3x3_relu6", hid, hid, s))
        specs.append((f"b{i}.project", "pw_linear_res" if (s == 1 and cin == c) else "pw_linear", hid, c, 1))
        cin = c
    return specs


def synth_weights(seed: int = 10) -> dict[str, np.ndarray]:
    g = torch.Generator().manual_seed(seed)
    w = {}

    def u(shape, bound):
        return ((torch.rand(*shape, generator=g) * 2 - 1) * bound).numpy().astype(np.float32)

    for name, kind, cin, cout, stride in layer_specs():
        if kind == "conv3x3":
            w[name + ".weight"] = u((cout, cin, 3, 3), np.sqrt(6.0 / (cin * 9)))
        elif kind == "dw3x3_relu6":
            w[name + ".weight"] = u((cout, 1, 3, 3), np.sqrt(6.0 / 9))
        else:
            w[name + ".weight"] = u((cout, cin, 1, 1), np.sqrt(6.0 / cin) * (0.7 if "linear" in kind else 1.0))
        w[name + ".bias"] = u((cout,), 0.1)
    w["vlad.assign.weight"] = u((N_CLUSTERS, FEAT_DIM, 1, 1), 1.0)
    w["vlad.assign.bias"] = u((N_CLUSTERS,), 0.5)
    w["vlad.clusters"] = u((N_CLUSTERS, FEAT_DIM), 1.0)
    w["fc.weight"] = u((OUT_DIM, VLAD_DIM), np.sqrt(3.0 / VLAD_DIM) * 4)
    w["fc.bias"] = u((OUT_DIM,), 0.01)
    return w


@torch.no_grad()
def forward(weights: dict[str, np.ndarray], gray_u8: np.ndarray, return_features=False):
    """gray_u8 [H,W] or [N,H,W] uint8 -> [N,4096] float32 (unit norm)."""
    x = torch.from_numpy(np.ascontiguousarray(gray_u8)).float()
    if x.dim() == 2:
        x = x[None]
    x = (x - 128.0) / 128.0
    x = x[:, None].repeat(1, 3, 1, 1)
    t = lambda n: torch.from_numpy(weights[n])
    for name, kind, cin, cout, stride in layer_specs():
        wt, bs = t(name + ".weight"), t(name + ".bias")
        if kind == "conv3x3":
            x = F.relu6(F.conv2d(x, wt, bs, stride=stride, padding=1))
        elif kind == "pw_relu6":
            block_in = x
            x = F.relu6(F.conv2d(x, wt, bs))
        elif kind == "dw3x3_relu6":
            x = F.relu6(F.conv2d(x, wt, bs, stride=stride, padding=1, groups=cin))
        elif kind == "pw_linear":
            x = F.conv2d(x, wt, bs)
        elif kind == "pw_linear_res":
            x = F.conv2d(x, wt, bs) + block_in
    feat = x                                                   # [N,112,h,w]
    n = feat.shape[0]
    a = torch.softmax(F.conv2d(feat, t("vlad.assign.weight"), t("vlad.assign.bias")), dim=1)   # [N,K,h,w]
    f = feat.flatten(2)                                        # [N,D,P]
    a = a.flatten(2)                                           # [N,K,P]
    c = t("vlad.clusters")                                     # [K,D]
    # V[k,d] = sum_p a[k,p] * (c[k,d] - f[d,p])
    v = a.sum(-1, keepdim=True) * c[None] - torch.einsum("nkp,ndp->nkd", a, f)
    v = v / torch.norm(v, dim=2, keepdim=True)                 # intra-normalisation
    v = v.reshape(n, VLAD_DIM)
    v = v / torch.norm(v, dim=1, keepdim=True)
    y = v @ t("fc.weight").T + t("fc.bias")
    y = y / torch.norm(y, dim=1, keepdim=True)
    if return_features:
        return y.numpy(), feat.numpy(), v.numpy()
    return y.numpy()
