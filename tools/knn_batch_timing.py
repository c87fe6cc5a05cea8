#!/usr/bin/env python
"""BASELINE config 5 per-GPU shard: NQ concurrent queries against an fp16 shard (run on the GPU box).
ROWS (default 125000 = 1 M key frames / 8 GPUs), NQ (default 64).  Prints the scan time of the matrix-core kernel and, with
OMNI_MQ_MIN=0 in the environment, of the 8-queries-per-pass VALU kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omni_loader
omni = omni_loader.load()
from omni_swarm_amd import capi
ctx = capi.Context(0)
rng = np.random.default_rng(7)
N = int(os.environ.get("ROWS", 125000)); NQ = int(os.environ.get("NQ", 64)); K = int(os.environ.get("K", 10))
idx = capi.IndexFlatIP(ctx, 4096, capi.STORE_F16, N)
planted = []
for s in range(0, N, 8192):
    x = rng.standard_normal((min(8192, N - s), 4096), dtype=np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
    idx.add(x); planted.append((s + 17, x[17].copy()))
rows = [planted[i % len(planted)][0] for i in range(NQ)]
q = np.stack([planted[i % len(planted)][1] for i in range(NQ)])
lat, scan = [], []
for i in range(40):
    t = time.perf_counter(); D, I = idx.search(q, K); lat.append((time.perf_counter() - t) * 1e3); scan.append(idx.last_scan_ms())
lat, scan = np.array(lat[10:]), np.array(scan[10:])
ok = I[:, 0].tolist() == rows
gb = N * 8192 / np.median(scan) / 1e6
print(f"rows {N} nq {NQ} mq_min {os.environ.get('OMNI_MQ_MIN', 'default')}: search p50 {np.median(lat):.3f} ms  scan {np.median(scan):.4f} ms "
      f"= {gb:.0f} GB/s of DB bytes ({NQ / np.median(lat) * 1e3:.0f} queries/s)  planted-first {ok}  D0 {D[0, 0]:.5f}")
