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

import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import omni_loader as _omni_loader  # noqa: E402

_omni_loader.load()
from omni_swarm_amd import weights as _W  # noqa: E402

# the assumed layer table and the seeded weights are shared with the HIP side (omni-swarm_amd/weights.py)
N_CLUSTERS, OUT_DIM, STEM_OUT, BLOCKS = _W.VLAD_N_CLUSTERS, _W.VLAD_OUT_DIM, _W.VLAD_STEM_OUT, _W.VLAD_BLOCKS
FEAT_DIM, VLAD_DIM = _W.VLAD_FEAT_DIM, _W.VLAD_DIM
layer_specs = _W.mobilenetvlad_layer_specs
synth_weights = _W.mobilenetvlad_synth_weights


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
