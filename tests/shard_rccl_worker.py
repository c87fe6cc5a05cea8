"""Worker of tests/test_gpu_shard_rccl.py: one rank of an omni_shard group (RCCL inside libomni_hip.so, no torch).
usage: shard_rccl_worker.py rank world device id_file out_npz seed"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omni_loader  # noqa: E402

omni_loader.load()
from omni_swarm_amd import capi, shard  # noqa: E402


def main():
    rank, world, device, id_file, out, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6])
    ctx = capi.Context(device)
    if rank == 0:
        uid = capi.shard_unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 60:
                raise SystemExit("no unique id from rank 0")
            time.sleep(0.05)
        uid = open(id_file, "rb").read()
    idx = capi.IndexFlatIP(ctx, 4096)
    sw = shard.NativeSwarmIndex(ctx, idx, rank, world, uid)
    rng = np.random.default_rng(seed)                       # the SAME stream on every rank: everybody knows everybody's rows
    base = rng.standard_normal((world * 40, 4096)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    sw.preload_local(base[rank::world], len(base))
    F, m, k = 3, 4, 10
    res_D, res_I = [], []
    for it in range(4):
        rows_all = rng.standard_normal((world, F, m, 4096)).astype(np.float32)
        rows_all[:, :, 1] = base[rng.integers(0, len(base), (world, F))] + 0.3 * rows_all[:, :, 1]      # queries near known rows
        rows_all /= np.linalg.norm(rows_all, axis=-1, keepdims=True)
        for D, I in sw.step_batch(rows_all[rank], 1, k):
            res_D.append(D[0]); res_I.append(I[0])
    q = base[[3, 17]] + 0.01
    Ds, Is = sw.search(q, k)
    np.savez(out, D=np.stack(res_D), I=np.stack(res_I), Ds=Ds, Is=Is, ntotal=sw.ntotal, librccl=capi.shard_library_path())
    sw.close()


if __name__ == "__main__":
    main()
