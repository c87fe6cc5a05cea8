#!/usr/bin/env python
"""p50 loop-match latency + scan GB/s for a 100k-row fp32 DB (run on the GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omni_loader
omni = omni_loader.load()
from omni_swarm_amd import capi
ctx = capi.Context(0)
rng = np.random.default_rng(7)
N = int(os.environ.get("ROWS", 100000))
F16 = os.environ.get("STORAGE", "f32") == "f16"
idx = capi.IndexFlatIP(ctx, 4096, capi.STORE_F16 if F16 else capi.STORE_F32, N)
for s in range(0, N, 8192):
    x = rng.standard_normal((min(8192, N - s), 4096), dtype=np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
    idx.add(x)
q = x[:1].copy()
lat, scan = [], []
for i in range(80):
    t = time.perf_counter(); D, I = idx.search(q, 10); lat.append((time.perf_counter() - t) * 1e3); scan.append(idx.last_scan_ms())
lat, scan = np.array(lat[20:]), np.array(scan[20:])
print(f"rows {N}: p50 {np.median(lat):.4f} ms  scan {np.median(scan):.4f} ms = {N * (8192 if F16 else 16384) / np.median(scan) / 1e6:.0f} GB/s ({'f16' if F16 else 'f32'})  top hit {I[0][0]} {D[0][0]:.4f}")
