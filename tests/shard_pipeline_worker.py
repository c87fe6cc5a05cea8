"""Worker of tests/test_gpu_shard_rccl.py::test_sharded_pipeline_*: one rank of the SHARDED C++ key-frame pipeline (host/keyframe_pipeline.hpp
attach_shard -> omni_shard_step_batch_dev per micro-batch), toy image size.  Also writes the global descriptors of its key frames (a second,
stand-alone MobileNetVLAD instance) so that the parent can replay the global insertion order on the unsharded oracle.
usage: shard_pipeline_worker.py rank world device id_file out_npz seed"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omni_loader  # noqa: E402

omni_loader.load()
from omni_swarm_amd import capi, pipeline, synth, weights  # noqa: E402

W, H, MB, N_UNITS = 128, 96, 2, 4


def microbatch(seed0):
    """[up cameras of the MB frames (4 each) | down cameras] of MB key frames"""
    kf = [[synth.image_u8(seed0 + 8 * m + i, H, W, n_shapes=60) for i in range(8)] for m in range(MB)]
    return np.stack([kf[m][i] for m in range(MB) for i in range(4)] + [kf[m][4 + i] for m in range(MB) for i in range(4)])


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
            if time.time() - t0 > 120:
                raise SystemExit("no unique id from rank 0")
            time.sleep(0.05)
        uid = open(id_file, "rb").read()
    sp_w, vw = weights.superpoint_synth_weights(0), weights.mobilenetvlad_synth_weights()
    comp, mean = synth.pca()
    specs = weights.mobilenetvlad_layer_specs()
    rng = np.random.default_rng(seed)                       # the same stream on every rank
    db = rng.standard_normal((world * 40, 4096)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    # every rank cycles the same two micro-batch blocks, rank r starting at block r: from the second exchange on, key frames revisit what
    # ANOTHER rank inserted one exchange earlier (score ~1 across shards)
    blocks = [microbatch(7000), microbatch(7000 + 64)]
    pins = []
    for b in blocks:
        p = ctx.host_alloc(b.shape, np.uint8)
        p[:] = b
        pins.append(p)
    with tempfile.TemporaryDirectory() as td:
        files = weights.write_pipeline_files(td, sp_w, comp, mean, vw, specs, capi.VLAD_KINDS)
        pl = pipeline.KeyframePipeline(device, files["sp"], files["comp"], files["mean"], files["vlad"], W, H, 0.015, 100, capi.PREC_F16, MB, 2,
                                       capi.STORE_F32, 1, 0.3, 0.2, 5, 10, 3)
        pl.attach_shard(rank, world, uid)
        pl.preload(db[rank::world])
        hits = []
        for u in range(N_UNITS):                             # one run() per exchange unit so that the hits come back per unit
            hits.append(pl.run(MB, u * MB, [pins[(rank + u) % 2].ctypes.data], 0, None, True))
        rows_total = pl.db_rows
        pl.close()
    net = capi.MobileNetVLAD(ctx, vw, specs, 32, 112, 4096, W, H, 4 * MB)
    g = [net.inference(b[:4 * MB], fisheye_mask=True) for b in blocks]          # [block][4 MB][4096]: the rows the pipeline appended
    np.savez(out, hits=np.array(hits), rows_total=rows_total, g=np.stack(g), librccl=capi.shard_library_path())
    for p in pins:
        ctx.host_free(p)


if __name__ == "__main__":
    main()
