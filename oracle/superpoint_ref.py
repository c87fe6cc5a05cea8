"""SuperPoint network oracle (CPU, fp32) -- test infrastructure only.

Restates the graph the reference exports to ONNX/TensorRT:
  /root/reference/swarm_loop/superpoint.ipynb:135-205 (export variant, cell 3)
    encoder  : conv1a,conv1b,pool, conv2a,conv2b,pool, conv3a,conv3b,pool,
               conv4a,conv4b  (all 3x3 pad 1 + ReLU)            (:170-181)
    detector : convPa 3x3+ReLU, convPb 1x1 -> 65                 (:183-184)
               softmax over the 65 channels, drop channel 64,
               depth-to-space 8x8 -> [H, W] probability map      (:190-199)
    descriptor: convDa 3x3+ReLU, convDb 1x1 -> 256,
               divide by the channel-wise L2 norm                (:186-188)
Weights use the reference checkpoint's state_dict names
(``conv1a.weight`` ... ``convDb.bias``, superpoint.ipynb:270), OIHW fp32.

Pinned by tools/gen_golden.py, which exec()s the notebook cell itself in the
build container and checks this restatement against it bit-for-bit.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

# (name, cin, cout, ksize) in execution order -- superpoint.ipynb:143-160
LAYERS = [
    ("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3),
    ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
    ("conv3a", 64, 128, 3), ("conv3b", 128, 128, 3),
    ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
    ("convPa", 128, 256, 3), ("convPb", 256, 65, 1),
    ("convDa", 128, 256, 3), ("convDb", 256, 256, 1),
]
SP_DESC_RAW_LEN = 256  # superpoint_tensorrt.h:10


def synth_weights(seed: int = 0, convPb_gain: float = 4.0, dustbin_bias: float = 12.0) -> dict[str, np.ndarray]:
    """Seeded synthetic weights (the real checkpoint superpoint_v1.pth is not in the reference tree).

    torch's default Conv2d init shrinks the signal ~sqrt(1/6) per layer, so after ten layers every image
    gives the same (bias-driven) heat map.  Use He-uniform weights (bound sqrt(6/fan_in)) with small biases
    so activations stay O(1) and image dependent, scale ``convPb`` by ``convPb_gain`` and lift the dustbin
    logit by ``dustbin_bias`` so thresholds 0.012-0.02 select ~5e3-1e4 candidates per 600x480 frame and NMS
    leaves ~2000 survivors ("No weights for SuperPoint either").
    """
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, cin, cout, k in LAYERS:
        fan_in = cin * k * k
        bound_w = np.sqrt(6.0 / fan_in)
        wt = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * bound_w
        bs = (torch.rand(cout, generator=g) * 2 - 1) * 0.05
        if name == "convPb":
            wt = wt * convPb_gain
            bs = bs * convPb_gain
            bs[64] += dustbin_bias
        w[name + ".weight"] = wt.numpy().astype(np.float32)
        w[name + ".bias"] = bs.numpy().astype(np.float32)
    return w


def _t(w, name):
    return torch.from_numpy(np.ascontiguousarray(w[name]))


@torch.no_grad()
def forward(weights: dict[str, np.ndarray], image01: np.ndarray, return_intermediates=False):
    """image01: [H, W] or [N, H, W] float32 in [0,1] -> (semi [N,H,W], desc [N,256,H/8,W/8])."""
    x = torch.from_numpy(np.ascontiguousarray(image01, dtype=np.float32))
    if x.dim() == 2:
        x = x[None]
    x = x[:, None]  # N,1,H,W
    inter = {}

    def conv(x, name, relu=True):
        k = weights[name + ".weight"].shape[-1]
        y = F.conv2d(x, _t(weights, name + ".weight"), _t(weights, name + ".bias"), padding=k // 2)
        y = F.relu(y) if relu else y
        if return_intermediates:
            inter[name] = y.numpy().copy()
        return y

    x = conv(x, "conv1a"); x = conv(x, "conv1b"); x = F.max_pool2d(x, 2, 2)
    x = conv(x, "conv2a"); x = conv(x, "conv2b"); x = F.max_pool2d(x, 2, 2)
    x = conv(x, "conv3a"); x = conv(x, "conv3b"); x = F.max_pool2d(x, 2, 2)
    x = conv(x, "conv4a"); x = conv(x, "conv4b")
    cPa = conv(x, "convPa")
    semi = conv(cPa, "convPb", relu=False)
    cDa = conv(x, "convDa")
    desc = conv(cDa, "convDb", relu=False)
    dn = torch.norm(desc, p=2, dim=1)
    desc = desc / dn.unsqueeze(1)

    semi = torch.softmax(semi, 1)[:, :64]           # drop the dustbin AFTER softmax
    n, _, hc, wc = semi.shape
    semi = semi.permute(0, 2, 3, 1).reshape(n, hc, wc, 8, 8)
    semi = semi.permute(0, 1, 3, 2, 4).reshape(n, hc * 8, wc * 8)
    if return_intermediates:
        return semi.numpy(), desc.numpy(), inter
    return semi.numpy(), desc.numpy()


def preprocess_u8(gray_u8: np.ndarray, fisheye_mask: bool = False) -> np.ndarray:
    """u8 -> f32 * (1/255)  (superpoint_tensorrt.cpp:123-128, cv::Mat::convertTo).

    OpenCV computes ``saturate_cast<float>(src * alpha)`` with alpha = 1/255.0 as a
    double, i.e. float(double(u8) * (1.0/255.0)).
    ``fisheye_mask`` zeroes rows [3H/4, H) first (loop_cam.cpp:536-539).
    """
    g = np.array(gray_u8, dtype=np.uint8, copy=True)
    if fisheye_mask:
        h = g.shape[-2]
        g[..., h * 3 // 4: h * 3 // 4 + h // 4, :] = 0
    return (g.astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
