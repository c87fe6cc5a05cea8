"""Probe (GPU): how long does a small blocking search on its own stream take while another stream runs SuperPoint launches back to back?
Answers whether short work on a second stream gets onto the GPU between the kernels of a saturating stream (hardware queues, stream priority)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import omni_loader
omni = omni_loader.load()
from oracle import superpoint_ref as S
from omni_swarm_amd import synth
c = omni.capi
prio = int(os.environ.get("PROBE_PRIO", "0"))
n_extra = int(os.environ.get("PROBE_EXTRA_STREAMS", "3"))
ctx_a = c.Context(0)
extra = [c.Context(0) for _ in range(n_extra)]            # the pipeline's other streams exist (and take hardware queues) even when idle
ctx_b = c.Context(0, high_priority=bool(prio))
weights = S.synth_weights(0)
comp, mean = synth.pca()
W, H, NB = 600, 480, 64
sp = c.SuperPoint(ctx_a, weights, comp, mean, W, H, 0.02, 200, c.PREC_F16, NB)
imgs = np.stack([synth.image_u8(i, H, W) for i in range(NB)])
gdev = ctx_a.to_device(imgs)
db = synth.global_db(4000, seed=3)
idx = c.IndexFlatIP(ctx_b, 4096)
idx.add(db)
q = db[:8].copy()
idx.search(q, 10)
sp.enqueue_dev(gdev, W, NB, True); ctx_a.sync()
t = time.perf_counter(); sp.enqueue_dev(gdev, W, NB, True); ctx_a.sync(); one = (time.perf_counter() - t) * 1e3
lat_idle = []
for _ in range(5):
    t = time.perf_counter(); idx.search(q, 10); lat_idle.append((time.perf_counter() - t) * 1e3)
for r in range(3):
    for _ in range(6):
        sp.enqueue_dev(gdev, W, NB, True)                   # ~6 x 2.4 ms of saturating work queued on stream A
    time.sleep(0.001)
    lat = []
    for _ in range(4):
        t = time.perf_counter(); idx.search(q, 10); lat.append((time.perf_counter() - t) * 1e3)
    t = time.perf_counter(); ctx_a.sync(); rest = (time.perf_counter() - t) * 1e3
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} prio={prio} extra={n_extra}: SuperPoint pass {one:.2f} ms; search idle {np.median(lat_idle):.3f} ms; "
          f"4 searches while 6 passes are queued: {[round(x, 3) for x in lat]} ms; stream A still had {rest:.2f} ms to go", flush=True)
