"""Seeded synthetic multi-drone key-frame descriptor stream for LoopDetector match-ID parity tests.

Stands in for the 5-drone bag of BASELINE.json config 3 (no bags, no ROS offline): drones revisit a fixed set of
"places"; a (place, direction) pair owns a unit 4096-d global descriptor and every visit perturbs it, so revisits score
~0.75-0.9 and unrelated frames ~0.  Landmark counts, empty directions, non-keyframes (prevent_adding_db) and remote
drones exercise every branch of LoopDetector::on_image_recv (loop_detector.cpp:11-137).
"""
import numpy as np

SELF_ID = 1


def make_stream(seed=11, n_frames=90, n_places=14, dim=4096, n_dirs=4):
    rng = np.random.default_rng(seed)
    place = rng.standard_normal((n_places, n_dirs, dim)).astype(np.float32)
    place /= np.linalg.norm(place, axis=-1, keepdims=True)
    frames = []
    pos = {1: 0, 2: 5, 3: 9}
    for f in range(n_frames):
        drone = 1 if (f % 3 != 2) else (2 if (f // 3) % 2 == 0 else 3)
        step = rng.integers(0, 3)
        pos[drone] = (pos[drone] + step) % n_places if rng.random() < 0.8 else int(rng.integers(0, n_places))
        pl = pos[drone]
        imgs = []
        total = 0
        for d in range(n_dirs):
            noise = rng.standard_normal(dim).astype(np.float32)
            noise /= np.linalg.norm(noise)
            c = rng.uniform(0.72, 0.95)
            v = c * place[pl, d] + np.sqrt(1 - c * c) * noise
            v = (v / np.linalg.norm(v)).astype(np.float32)
            lm = 0 if rng.random() < 0.12 else int(rng.integers(5, 200))
            total += lm
            imgs.append({"drone_id": drone, "landmark_num": lm, "image_desc": v})
        frames.append({"msg_id": 1000 * drone + f, "drone_id": drone, "landmark_num": total,
                       "prevent_adding_db": bool(rng.random() < 0.15), "images": imgs})
    return frames


def loop_ok(new_msg_id, old_msg_id):
    """Deterministic stand-in for compute_loop's geometric verdict."""
    return (new_msg_id + old_msg_id) % 3 != 0


PARAMS = dict(inner_product_thres=0.6, init_mode_product_thres=0.3, match_index_dist=5, min_loop_num=30,
              min_direction_loop=3, inter_drone_init_frames=3)


def run_oracle(frames):
    from oracle import match_ref as M
    det = M.LoopDetectorRef(SELF_ID, compute_loop=lambda n, o, dn, do, im: loop_ok(n.msg_id, o.msg_id), **PARAMS)
    for fr in frames:
        det.on_image_recv(M.FisheyeFrameDesc(
            msg_id=fr["msg_id"], drone_id=fr["drone_id"], landmark_num=fr["landmark_num"],
            prevent_adding_db=fr["prevent_adding_db"],
            images=[M.ImageDesc(drone_id=i["drone_id"], landmark_num=i["landmark_num"], image_desc=i["image_desc"])
                    for i in fr["images"]]))
    return det.log


def run_product(frames, ctx, det_mod, **kw):
    det = det_mod.LoopDetector(ctx, SELF_ID, compute_loop=lambda n, o, dn, do, im: loop_ok(n.msg_id, o.msg_id), **PARAMS, **kw)
    for fr in frames:
        det.on_image_recv(det_mod.FisheyeFrameDescriptor(
            msg_id=fr["msg_id"], drone_id=fr["drone_id"], landmark_num=fr["landmark_num"],
            prevent_adding_db=fr["prevent_adding_db"],
            images=[det_mod.ImageDescriptor(drone_id=i["drone_id"], landmark_num=i["landmark_num"], image_desc=i["image_desc"])
                    for i in fr["images"]]))
    return det.log


def run_product_batched(frames, ctx, det_mod, batch=4, rows_on_device=False, **kw):
    """The same stream through LoopDetector.on_images_recv_batch, `batch` frames per call (one host synchronisation each)."""
    det = det_mod.LoopDetector(ctx, SELF_ID, compute_loop=lambda n, o, dn, do, im: loop_ok(n.msg_id, o.msg_id), **PARAMS, **kw)
    for s in range(0, len(frames), batch):
        chunk = [det_mod.FisheyeFrameDescriptor(
            msg_id=fr["msg_id"], drone_id=fr["drone_id"], landmark_num=fr["landmark_num"], prevent_adding_db=fr["prevent_adding_db"],
            images=[det_mod.ImageDescriptor(drone_id=i["drone_id"], landmark_num=i["landmark_num"], image_desc=i["image_desc"])
                    for i in fr["images"]]) for fr in frames[s:s + batch]]
        if rows_on_device:      # descriptors handed over in HBM (as MobileNetVLAD leaves them); the host copies are poisoned
            dev = ctx.to_device(np.stack([i.image_desc for f in chunk for i in f.images]))
            for f in chunk:
                for i in f.images:
                    i.image_desc = None
            det.on_images_recv_batch(chunk, rows_dev=dev)
            ctx.free(dev)
        else:
            det.on_images_recv_batch(chunk)
    return det.log


def trace(log):
    return np.array([[r["msg_id"], int(r["added"]), int(r["queried"]), r["image_id"], r["old_msg_id"], r["dir_old"],
                      int(r["loop"])] for r in log], np.int64)
