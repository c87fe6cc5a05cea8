"""Seeded synthetic multi-drone key-frame descriptor stream for LoopDetector match-ID parity tests.

Stands in for the 5-drone bag of BASELINE.json config 3 (no bags, no ROS offline): drones revisit a fixed set of
"places"; a (place, direction) pair owns a unit 4096-d global descriptor and every visit perturbs it, so revisits score
~0.75-0.9 and unrelated frames ~0.  Landmark counts, empty directions, non-keyframes (prevent_adding_db) and remote
drones exercise every branch of LoopDetector::on_image_recv (loop_detector.cpp:11-137).
"""
import numpy as np

SELF_ID = 1


def make_stream(seed=11, n_frames=90, n_places=14, dim=4096, n_dirs=4, n_drones=3):
    """n_drones = 3 is the stream the golden fixtures were generated with; n_drones = 5 (self + 4 remote drones, every third frame remote,
    round robin) is the 5-drone replay of BASELINE configs[2]."""
    rng = np.random.default_rng(seed)
    place = rng.standard_normal((n_places, n_dirs, dim)).astype(np.float32)
    place /= np.linalg.norm(place, axis=-1, keepdims=True)
    frames = []
    pos = {1: 0, 2: 5, 3: 9}
    for d in range(4, n_drones + 1):
        pos[d] = (4 * d) % n_places
    for f in range(n_frames):
        if n_drones == 3:
            drone = 1 if (f % 3 != 2) else (2 if (f // 3) % 2 == 0 else 3)
        else:
            drone = 1 if (f % 3 != 2) else 2 + (f // 3) % (n_drones - 1)
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


# ---- fast oracle for long replays: the same LoopDetectorRef, its two indexes answered from score matrices computed by two big float64
# GEMMs instead of one scalar scan per query (which rows exist when a query runs does not depend on any query result) -----------------------
class _PlanIndex:
    def __init__(self, log, name):
        self.rows, self.log, self.name = [], log, name

    @property
    def ntotal(self):
        return len(self.rows)

    def add(self, x):
        self.rows.append(np.asarray(x, np.float32).reshape(-1))

    def search(self, q, k, use_numpy=False):
        self.log.append((self.name, np.asarray(q, np.float32).reshape(-1), len(self.rows), k))
        return np.full((1, k), -3.4028235e38, np.float32), np.full((1, k), -1, np.int64)


class _ReplayIndex:
    def __init__(self, results):
        self.results, self.n = results, 0

    @property
    def ntotal(self):
        return self.n

    def add(self, x):
        self.n += 1

    def search(self, q, k, use_numpy=False):
        D, I, n_seen, kk = self.results.pop(0)
        assert n_seen == self.n and kk == k
        return D, I


def _frames_ref(frames):
    from oracle import match_ref as M
    return [M.FisheyeFrameDesc(msg_id=fr["msg_id"], drone_id=fr["drone_id"], landmark_num=fr["landmark_num"], prevent_adding_db=fr["prevent_adding_db"],
                               images=[M.ImageDesc(drone_id=i["drone_id"], landmark_num=i["landmark_num"], image_desc=i["image_desc"]) for i in fr["images"]])
            for fr in frames]


def run_oracle_fast(frames, params=None):
    """Decision trace of oracle/match_ref.LoopDetectorRef over a long stream; exact inner products in float64 (torch GEMM), rounded to f32,
    ties -> lower row.  Returns (log, searches) where searches[i] = (index name, n rows seen, sorted f64 scores of the top candidates)."""
    import torch
    from oracle import match_ref as M
    params = params or PARAMS
    loop = lambda n, o, dn, do, im: loop_ok(n.msg_id, o.msg_id)
    plan = []
    det = M.LoopDetectorRef(SELF_ID, compute_loop=loop, **params)
    det.local_index, det.remote_index = _PlanIndex(plan, "local"), _PlanIndex(plan, "remote")
    ref_frames = _frames_ref(frames)
    for f in ref_frames:
        det.on_image_recv(f)
    results = {"local": [], "remote": []}
    info = []
    for name, idx in (("local", det.local_index), ("remote", det.remote_index)):
        qs = [(q, n, k) for (nm, q, n, k) in plan if nm == name]
        if not qs:
            continue
        rows = torch.from_numpy(np.stack(idx.rows)).double() if idx.rows else torch.zeros((0, 4096), dtype=torch.float64)
        Q = torch.from_numpy(np.stack([q for q, _, _ in qs])).double()
        for s in range(0, len(qs), 512):                         # blocks of queries: bounded memory
            S = (Q[s:s + 512] @ rows.T).numpy() if len(rows) else np.zeros((len(qs[s:s + 512]), 0))
            for j, (q, n, k) in enumerate(qs[s:s + 512]):
                sc = S[j, :n].astype(np.float32)
                D = np.full((1, k), -3.4028235e38, np.float32)
                I = np.full((1, k), -1, np.int64)
                if n:
                    kk = min(k, n)
                    cand = np.argpartition(-sc, kk - 1)[:kk] if n > kk else np.arange(n)
                    thr = sc[cand].min()
                    cand = np.nonzero(sc >= thr)[0]               # everything tied with the cut-off takes part in the id-ordered sort
                    order = cand[np.lexsort((cand, -sc[cand].astype(np.float64)))][:kk]
                    D[0, :kk], I[0, :kk] = sc[order], order
                results[name].append((D, I, n, k))
    for nm, q, n, k in plan:
        info.append((nm, n))
    det2 = M.LoopDetectorRef(SELF_ID, compute_loop=loop, **params)
    det2.local_index, det2.remote_index = _ReplayIndex(results["local"]), _ReplayIndex(results["remote"])
    for f in ref_frames:
        det2.on_image_recv(f)
    assert not results["local"] and not results["remote"]
    return det2.log
