"""Weights for the two networks: layer tables, seeded synthetic initialisation, and loaders for real checkpoints.

The reference loads TensorRT engines plus two PCA CSV files (swarm_loop/src/superpoint_tensorrt.cpp:91-115,14-89); the
engines' weights come from ``superpoint_v1.pth`` (swarm_loop/superpoint.ipynb:270) and the HF-Net mobilenetvlad
saved-model, neither of which is in the reference tree or reachable offline.  Here a "weights dict" maps the
checkpoint's state_dict names to float32 numpy arrays (OIHW), which capi.SuperPoint / capi.MobileNetVLAD hand to the
C ABI; ``load_superpoint_pth`` accepts the real checkpoint unchanged, ``load_pca_csv`` reads the reference's CSVs.
"""
from __future__ import annotations

import numpy as np
import torch

# (name, cin, cout, ksize) in execution order -- swarm_loop/superpoint.ipynb:143-160
SUPERPOINT_LAYERS = [
    ("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3),
    ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
    ("conv3a", 64, 128, 3), ("conv3b", 128, 128, 3),
    ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
    ("convPa", 128, 256, 3), ("convPb", 256, 65, 1),
    ("convDa", 128, 256, 3), ("convDb", 256, 256, 1),
]


def superpoint_synth_weights(seed: int = 0, convPb_gain: float = 4.0, dustbin_bias: float = 12.0) -> dict:
    """Seeded synthetic SuperPoint weights (the real checkpoint superpoint_v1.pth is not in the reference tree).

    torch's default Conv2d init shrinks the signal ~sqrt(1/6) per layer, so after ten layers every image gives the same
    (bias-driven) heat map.  He-uniform weights (bound sqrt(6/fan_in)) with small biases keep activations O(1) and image
    dependent; ``convPb`` is scaled by ``convPb_gain`` and the dustbin logit lifted by ``dustbin_bias`` so thresholds
    0.012-0.02 select ~5e3-1e4 candidates per 600x480 frame and NMS leaves ~2000 survivors.
    """
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, cin, cout, k in SUPERPOINT_LAYERS:
        fan_in = cin * k * k
        wt = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * np.sqrt(6.0 / fan_in)
        bs = (torch.rand(cout, generator=g) * 2 - 1) * 0.05
        if name == "convPb":
            wt = wt * convPb_gain
            bs = bs * convPb_gain
            bs[64] += dustbin_bias
        w[name + ".weight"] = wt.numpy().astype(np.float32)
        w[name + ".bias"] = bs.numpy().astype(np.float32)
    return w


def load_superpoint_pth(path: str) -> dict:
    """The reference checkpoint (state_dict of SuperPointNet, superpoint.ipynb:270) -> weights dict."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return {k: v.detach().numpy().astype(np.float32) for k, v in sd.items()}


def load_pca_csv(comp_csv: str, mean_csv: str):
    """components_.csv (64 rows x 256, comma separated) and mean_.csv (256 lines) as written by swarm_loop/pca.ipynb and
    parsed by load_csv_mat_eigen / load_csv_vec_eigen (superpoint_tensorrt.cpp:14-89)."""
    comp = np.loadtxt(comp_csv, delimiter=",", dtype=np.float32, ndmin=2)
    mean = np.loadtxt(mean_csv, delimiter=",", dtype=np.float32).reshape(-1)
    assert comp.shape[1] == 256 and mean.shape[0] == 256
    return comp, mean


# ---- MobileNetVLAD: ASSUMED architecture (the reference ships only the I/O contract) ----------
VLAD_N_CLUSTERS = 32
VLAD_OUT_DIM = 4096
VLAD_STEM_OUT = 16
# (expand t, cout, stride) per inverted-residual block; MobileNetV2 (t,c,n,s) table at width 0.35, make_divisible 8
VLAD_BLOCKS = (
    [(1, 8, 1)] +
    [(6, 8, 2), (6, 8, 1)] +
    [(6, 16, 2), (6, 16, 1), (6, 16, 1)] +
    [(6, 24, 2), (6, 24, 1), (6, 24, 1), (6, 24, 1)] +
    [(6, 32, 1), (6, 32, 1), (6, 32, 1)] +
    [(6, 56, 2), (6, 56, 1), (6, 56, 1)] +
    [(6, 112, 1)]
)
VLAD_FEAT_DIM = VLAD_BLOCKS[-1][1]                    # 112
VLAD_DIM = VLAD_N_CLUSTERS * VLAD_FEAT_DIM            # 3584


def mobilenetvlad_layer_specs():
    """Flat list of (name, kind, cin, cout, stride): the table both the HIP side and the oracle walk."""
    specs = [("stem", "conv3x3", 3, VLAD_STEM_OUT, 2)]
    cin = VLAD_STEM_OUT
    for i, (t, c, s) in enumerate(VLAD_BLOCKS):
        hid = cin * t
        if t != 1:
            specs.append((f"b{i}.expand", "pw_relu6", cin, hid, 1))
        specs.append((f"b{i}.dw", "dw3x3_relu6", hid, hid, s))
        specs.append((f"b{i}.project", "pw_linear_res" if (s == 1 and cin == c) else "pw_linear", hid, c, 1))
        cin = c
    return specs


def mobilenetvlad_synth_weights(seed: int = 10) -> dict:
    g = torch.Generator().manual_seed(seed)
    w = {}

    def u(shape, bound):
        return ((torch.rand(*shape, generator=g) * 2 - 1) * bound).numpy().astype(np.float32)

    for name, kind, cin, cout, stride in mobilenetvlad_layer_specs():
        if kind == "conv3x3":
            w[name + ".weight"] = u((cout, cin, 3, 3), np.sqrt(6.0 / (cin * 9)))
        elif kind == "dw3x3_relu6":
            w[name + ".weight"] = u((cout, 1, 3, 3), np.sqrt(6.0 / 9))
        else:
            w[name + ".weight"] = u((cout, cin, 1, 1), np.sqrt(6.0 / cin) * (0.7 if "linear" in kind else 1.0))
        w[name + ".bias"] = u((cout,), 0.1)
    w["vlad.assign.weight"] = u((VLAD_N_CLUSTERS, VLAD_FEAT_DIM, 1, 1), 1.0)
    w["vlad.assign.bias"] = u((VLAD_N_CLUSTERS,), 0.5)
    w["vlad.clusters"] = u((VLAD_N_CLUSTERS, VLAD_FEAT_DIM), 1.0)
    w["fc.weight"] = u((VLAD_OUT_DIM, VLAD_DIM), np.sqrt(3.0 / VLAD_DIM) * 4)
    w["fc.bias"] = u((VLAD_OUT_DIM,), 0.01)
    return w


# ---- OMNW1 weight files for the C++ host adapters (host/omni_swarm.hpp: omni::load_omnw) -----------------------------------------
def write_omnw(path, tensors: dict):
    """OMNW1 = "OMNW1\\0\\0\\0", u32 n, then per tensor: u32 name_len, name, u32 ndim, u32 dims[ndim], float32 data (little endian)."""
    import struct
    with open(path, "wb") as f:
        f.write(b"OMNW1\0\0\0")
        f.write(struct.pack("<I", len(tensors)))
        for name, arr in tensors.items():
            a = np.ascontiguousarray(arr, dtype="<f4")
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb)
            f.write(struct.pack("<I", a.ndim) + struct.pack(f"<{a.ndim}I", *a.shape))
            f.write(a.tobytes())


def vlad_omnw_tensors(weights, specs, kinds):
    """The MobileNetVLAD file carries the (assumed) layer table as tensor "layers" [n][4] = (kind, cin, cout, stride) and the per-layer
    tensors as "layer<i>.weight" / "layer<i>.bias"."""
    t = {"layers": np.array([[kinds[k], ci, co, s] for (_, k, ci, co, s) in specs], np.float32)}
    for i, (name, *_rest) in enumerate(specs):
        t[f"layer{i}.weight"] = weights[name + ".weight"]
        t[f"layer{i}.bias"] = weights[name + ".bias"]
    for k in ("vlad.assign.weight", "vlad.assign.bias", "vlad.clusters", "fc.weight", "fc.bias"):
        t[k] = weights[k]
    t["vlad.assign.weight"] = np.asarray(weights["vlad.assign.weight"]).reshape(weights["vlad.clusters"].shape)
    return t


def write_pipeline_files(dirpath, sp_w, pca_comp, pca_mean, vlad_w, vlad_specs, vlad_kinds):
    """Everything host/keyframe_pipeline.hpp loads from disk: sp.omnw, vlad.omnw and the reference's two PCA CSVs (%.9g round-trips fp32)."""
    import os
    paths = {k: os.path.join(dirpath, v) for k, v in (("sp", "sp.omnw"), ("comp", "components_.csv"), ("mean", "mean_.csv"), ("vlad", "vlad.omnw"))}
    write_omnw(paths["sp"], sp_w)
    write_omnw(paths["vlad"], vlad_omnw_tensors(vlad_w, vlad_specs, vlad_kinds))
    np.savetxt(paths["comp"], pca_comp, delimiter=",", fmt="%.9g")
    np.savetxt(paths["mean"], pca_mean, fmt="%.9g")
    return paths
