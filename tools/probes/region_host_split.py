#!/usr/bin/env python
"""Where the host's time goes around a 20-key-frame region of the C++ pipeline (the driver's bench shape): run() itself, then the synchronisations the bench
brackets it with.  python tools/probes/region_host_split.py"""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import omni_loader
omni_loader.load()
from omni_swarm_amd import capi, pipeline, synth, weights
W, H, MB = 600, 480, 8
ictx = capi.Context(0)
sp_w = weights.superpoint_synth_weights(0); comp, mean = synth.pca(); vl_w = weights.mobilenetvlad_synth_weights(); specs = weights.mobilenetvlad_layer_specs()
td = tempfile.TemporaryDirectory()
files = weights.write_pipeline_files(td.name, sp_w, comp, mean, vl_w, specs, capi.VLAD_KINDS)
def block(seed, mb):
    kf = [[synth.image_u8(seed + 8 * m + i, H, W) for i in range(8)] for m in range(mb)]
    a = ictx.host_alloc((8 * mb, H, W), np.uint8)
    a[:] = np.stack([kf[m][i] for m in range(mb) for i in range(4)] + [kf[m][4 + i] for m in range(mb) for i in range(4)])
    return a
pool = [block(8 * MB * p, MB) for p in range(4)]
tail = block(8 * MB * 4, 4)
pl = pipeline.KeyframePipeline(0, files["sp"], files["comp"], files["mean"], files["vlad"], W, H, 0.02, 200, capi.PREC_F16, MB, 0, capi.STORE_F32, 1, 0.3, 0.2, 5, 30, 3)
rng = np.random.default_rng(3)
db = rng.standard_normal((4000, 4096), dtype=np.float32); db /= np.linalg.norm(db, axis=1, keepdims=True)
pl.preload(db)
pl.prepare(20)
ptrs = [a.ctypes.data for a in pool]
def sync_all():
    pl.sync(); ictx.sync(); torch.cuda.synchronize()
kid = 0
for _ in range(3):
    pl.run(20, kid, ptrs, 0, tail.ctypes.data, True); kid += 20
sync_all()
rows = []
for it in range(60):
    sync_all()
    t0 = time.perf_counter()
    pl.run(20, kid, ptrs, (kid // MB) % 4, tail.ctypes.data, True); kid += 20
    t1 = time.perf_counter()
    pl.sync(); t2 = time.perf_counter()
    ictx.sync(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    rows.append([(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3])
r = np.median(np.array(rows[10:]), axis=0)
print(f"20-key-frame region: run() {r[0]:.3f} ms, pipeline sync {r[1]:.3f}, context sync {r[2]:.3f}, torch synchronize {r[3]:.3f}: total {r[4]:.3f} ms = {20 / r[4] * 1e3:.0f} keyframes/s")
print("host ms per unit inside run():", pl.host_times())
