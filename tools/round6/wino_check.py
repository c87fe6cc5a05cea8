"""GPU check of the Winograd split kernels (OMNI_SPLIT_WINO bit mask) layer by layer against the torch oracle and against the direct split kernels."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import omni_loader
omni = omni_loader.load()
from oracle import superpoint_ref as S
from omni_swarm_amd import synth
import torch, torch.nn.functional as F

def run(mask, h, w, imgs, weights, comp, mean, fisheye=False):
    os.environ["OMNI_SPLIT_WINO"] = str(mask)
    ctx = omni.capi.Context(0)
    sp = omni.capi.SuperPoint(ctx, weights, comp, mean, w, h, 0.015, 200, omni.capi.PREC_SPLIT, len(imgs))
    res = sp.inference(imgs, fisheye_mask=fisheye)
    layers = {n: sp.debug_layer(n, len(imgs)) for n in ["conv1b", "conv2a", "conv2b", "conv3a", "conv4b", "heads"]}
    semi, desc = sp.get_dense(len(imgs))
    sp.close()
    return res, layers, semi, desc

def main():
    weights = S.synth_weights(0)
    comp, mean = synth.pca()
    for (h, w) in [(64, 96), (72, 104), (480, 600)]:
        imgs = np.stack([synth.image_u8(400 + i, h, w, n_shapes=60 if h < 100 else 200) for i in range(2)])
        semi_r, desc_r, inter = S.forward(weights, S.preprocess_u8(imgs), return_intermediates=True)
        ref = {}
        for n in ["conv1b", "conv2a", "conv2b", "conv3a", "conv4b"]:
            a = torch.from_numpy(inter[n])
            ref[n] = (F.max_pool2d(a, 2, 2) if n in ("conv1b", "conv2b", "conv3b") else a).numpy()
        ref["heads"] = np.concatenate([inter["convPa"], inter["convDa"]], 1)
        base = None
        for mask in (0, 8, 12, 4, 2, 6, 1, 7, 15):
            try:
                res, layers, semi, desc = run(mask, h, w, imgs, weights, comp, mean)
            except Exception as e:
                print(f"{h}x{w} mask {mask}: FAILED {e}")
                continue
            errs = {n: float(np.abs(layers[n] - ref[n]).max()) for n in layers}
            gate = {n: 2e-5 * max(1.0, float(np.abs(ref[n]).max())) for n in layers}
            ok = all(errs[n] < gate[n] for n in errs) and np.abs(semi - semi_r).max() < 2e-5 and np.abs(desc - desc_r).max() < 2e-5
            nan = any(not np.isfinite(layers[n]).all() for n in layers)
            print(f"{h}x{w} mask {mask}: {'OK ' if ok else 'BAD'} nan={nan} " + " ".join(f"{n}={errs[n]:.2e}" for n in errs) +
                  f" semi={np.abs(semi - semi_r).max():.2e} desc={np.abs(desc - desc_r).max():.2e}", flush=True)
            if mask == 0:
                base = res
            else:
                same = all(np.array_equal(a[0], b[0]) for a, b in zip(res, base))
                print(f"      key points identical to the direct kernels' : {same}")

if __name__ == "__main__":
    main()
