"""v5 register-stationary fp16 kernel (OMNI_CONV_RS=2) against v4 (=1): layer outputs must be bit-identical (separate processes: the option is process-wide)."""
import os, sys, subprocess, pickle
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import omni_loader
    omni = omni_loader.load()
    from oracle import superpoint_ref as S
    from omni_swarm_amd import synth
    out = {}
    for (h, w) in ((64, 96), (72, 104), (480, 600), (208, 400)):
        imgs = np.stack([synth.image_u8(400 + i, h, w, n_shapes=60 if h < 100 else 200) for i in range(2)])
        comp, mean = synth.pca()
        sp = omni.capi.SuperPoint(omni.capi.Context(0), S.synth_weights(0), comp, mean, w, h, 0.015, 200, omni.capi.PREC_F16, 2)
        sp.inference(imgs, fisheye_mask=(h == 480))
        for n in ("conv3b", "conv4a", "conv4b", "heads"):
            out[(h, w, n)] = sp.debug_layer(n, 2)
        sp.close()
    pickle.dump(out, open(sys.argv[2], "wb"))
else:
    res = {}
    for v in ("1", "2"):
        env = dict(os.environ, OMNI_CONV_RS=v)
        subprocess.check_call([sys.executable, __file__, "child", f"/tmp/rs_{v}.pkl"], env=env)
        res[v] = pickle.load(open(f"/tmp/rs_{v}.pkl", "rb"))
    for k in res["1"]:
        a, b = res["1"][k], res["2"][k]
        d = np.abs(a - b)
        bad = np.argwhere(d > 0)
        print(k, "identical" if not len(bad) else f"DIFFERENT: {len(bad)} of {a.size}, max {d.max():.3e}; first at {bad[:3].tolist()}; channels {sorted(set(bad[:,1].tolist()))[:8]} rows {sorted(set(bad[:,2].tolist()))[:8]} cols {sorted(set(bad[:,3].tolist()))[:12]}")
