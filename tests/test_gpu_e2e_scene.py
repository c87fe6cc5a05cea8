"""End to end on a RENDERED scene: images -> SuperPoint -> up/down matching -> triangulation -> MobileNetVLAD -> database / query rule ->
compute_correspond_features -> homography mask -> PnP -> LoopEdge, through the C++ key-frame pipeline on the GPU (host/keyframe_pipeline.hpp,
the flow of SwarmLoop::VIOKF_callback, swarm_loop.cpp:140-170: LoopCam::on_flattened_images -> LoopDetector::on_image_recv -> compute_loop,
loop_detector.cpp:627-836) against the oracle chain on the SAME images (superpoint_ref -> postproc_ref -> mobilenetvlad_ref -> bf_match ->
geometry_ref.stereo_landmarks -> LoopDetectorRef -> geometry_ref.compute_loop) and against the scene's ground truth.

Scene (omni_swarm_amd.synth.room_keyframe): a stereo rig (up / down cameras 10 cm apart, four 90-degree pinhole views each) at the centre of a
square room whose textured walls are 1.875 m away, so that a wall point moves by exactly 16 rows between the up and the down view.  16 places
(16 rooms), each visited twice: the second visit is the loop closure -- the rig is back at the same spot, half of the revisits with sensor
noise on every camera, all of them with a DRIFTED odometry pose (what a loop closure is for): the true relative pose of the two visits is the
identity whatever the odometry says.

The pipeline runs in OMNI_PREC_SPLIT (the mode that meets north_star's tolerance: key points identical to the fp32 graph, descriptors to
2e-6), so the discrete decisions downstream -- matches, flags, masks, inliers -- are the oracle's except where one sits within fp32 rounding
of its threshold (a BF distance tie, a reprojection error at the RANSAC bound): at most 2 of an edge's ~600 correspondences may differ, and
the pose -- a least-squares refit over all inliers -- then agrees to 1e-4 instead of 1e-6."""
import math

import numpy as np
import pytest

from oracle import geometry_ref as G
from oracle import match_ref as M
from oracle import mobilenetvlad_ref as V
from oracle import postproc_ref as P
from oracle import superpoint_ref as S
from omni_swarm_amd import synth

pytestmark = pytest.mark.gpu
W, H, THR, MAXN, MB = 600, 480, 0.02, 200, 4
FX = FY = W / 2.0
CX, CY = W / 2.0, H / 2.0
N_PLACES = 16
PARAMS = dict(inner_product_thres=0.3, init_mode_product_thres=0.2, match_index_dist=5, min_loop_num=30, min_direction_loop=3)
R_BC = np.array([[0.0, 0, 1], [-1, 0, 0], [0, -1, 0]])       # camera (x right, y down, z forward) in the body (x forward, y left, z up)


def extrinsic(direction, up, baseline=0.10):
    """KeyframePipeline::view_extrinsic: direction d looks along the body x axis rotated by 90 deg * d, up / down cameras +- baseline / 2 along z"""
    yaw = math.pi / 2 * direction
    Rz = np.array([[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1]])
    return G.pose([0.0, 0.0, (0.5 if up else -0.5) * baseline], G.q_from_R(Rz @ R_BC))


def schedule():
    """key frame i -> (place, revisit, noise sigma, reported odometry pose).  Second visits come in another order than the first ones."""
    out = []
    for p in range(N_PLACES):
        out.append((p, 0, 0.0, G.pose([10.0 * p, 2.0 * (p % 3), 1.0], G.q_from_yaw(0.03 * p))))
    for i in range(N_PLACES):
        p = (5 * i + 3) % N_PLACES
        true = out[p][3]
        drift = G.pose(true[0] + np.array([0.30, -0.20, 0.05]), G.qmul(G.q_from_yaw(0.02), true[1]))      # what the VIO reports after drifting
        out.append((p, 1, 0.0 if i % 2 == 0 else 1.5, drift))
    return out


def oracle_frame(sp_w, vw, comp, mean, imgs, msg_id, pose):
    """One key frame through the oracle chain, as a geometry_ref frame dict (+ the 4 global descriptors)."""
    semi, desc = S.forward(sp_w, S.preprocess_u8(imgs, True))
    per = []
    for b in range(8):
        xy, _, _, _ = P.get_keypoints(semi[b], THR, MAXN)
        d64, _ = P.compute_descriptors(desc[b], xy, W, H, comp, mean)
        per.append((xy.astype(np.float64), d64))
    masked = imgs[:4].copy()
    masked[:, H * 3 // 4:] = 0
    g = V.forward(vw, masked)
    images = []
    for d in range(4):
        (xu, du), (xd, dd) = per[d], per[4 + d]
        lift = lambda x: np.stack([((x[:, 0] - CX) / FX), ((x[:, 1] - CY) / FY)], 1).astype(np.float32).astype(np.float64)     # float in the message
        nu, nd = lift(xu), lift(xd)
        lift64 = lambda x: np.stack([((x[:, 0] - CX) / FX), ((x[:, 1] - CY) / FY)], 1)             # the triangulation lifts the pixels again, in double (loop_cam.cpp:403-407)
        qi, ti, _ = M.bf_match(du, dd, 0)
        l3u, fu = np.zeros((len(xu), 3)), np.zeros(len(xu), np.uint8)
        if len(xu) > 50:                                                       # ACCEPT_MIN_3D_PTS (loop_cam.cpp:385)
            _, l3u, fu, _, _ = G.stereo_landmarks(pose, extrinsic(d, True), extrinsic(d, False), lift64(xu), lift64(xd), qi, ti, 0.006)
        images.append({"landmark_num": len(xu), "landmarks_2d": xu, "landmarks_2d_norm": nu, "feature_descriptor": du, "camera_extrinsic": extrinsic(d, True),
                       "landmarks_3d": l3u.astype(np.float32).astype(np.float64), "landmarks_flag": fu})
    return {"msg_id": msg_id, "drone_id": 1, "timestamp": float(msg_id), "pose_drone": pose, "images": images,
            "landmark_num": int(sum(i["landmark_num"] for i in images))}, g


def test_rendered_scene_images_to_loop_edges_equal_the_oracle_chain_and_the_ground_truth(omni, ctx, tmp_path):
    c = omni.capi
    from omni_swarm_amd import pipeline, weights
    sp_w, vw = S.synth_weights(0), V.synth_weights()
    comp, mean = synth.pca()
    files = weights.write_pipeline_files(str(tmp_path), sp_w, comp, mean, vw, V.layer_specs(), c.VLAD_KINDS)
    plan = schedule()
    n = len(plan)
    frames = [synth.room_keyframe(p, H, W, rv, sg) for (p, rv, sg, _) in plan]
    # ---- the product: C++ key-frame pipeline, GPU front end in OMNI_PREC_SPLIT, geometry stage on --------------------------------------------
    pins = []
    for s in range(0, n, MB):
        kf = frames[s:s + MB]
        blk = np.stack([kf[m][i] for m in range(MB) for i in range(4)] + [kf[m][4 + i] for m in range(MB) for i in range(4)])
        p = ctx.host_alloc(blk.shape, np.uint8)
        p[:] = blk
        pins.append(p)
    pl = pipeline.KeyframePipeline(0, files["sp"], files["comp"], files["mean"], files["vlad"], W, H, THR, MAXN, c.PREC_SPLIT, MB, 2, c.STORE_F32, 1,
                                   PARAMS["inner_product_thres"], PARAMS["init_mode_product_thres"], PARAMS["match_index_dist"], PARAMS["min_loop_num"],
                                   PARAMS["min_direction_loop"], geometry=True)
    pl.set_poses(0, np.array([np.concatenate([q[3][0], q[3][1]]) for q in plan]))
    hits = pl.run(n, 0, [p.ctypes.data for p in pins], 0, None, True)
    cand, edges = pl.candidates(), pl.edges()
    calls, n_edges = pl.geometry_stats()
    rows = pl.db_rows
    pl.close()
    for p in pins:
        ctx.host_free(p)
    # ---- the oracle chain on the same images ---------------------------------------------------------------------------------------------------
    geo, ref_edges = {}, []
    bf = lambda a, b: M.bf_match(a, b, 0)

    def compute_loop(new, old, dn, do, init_mode):
        r = G.compute_loop(geo[new.msg_id], geo[old.msg_id], dn, do, init_mode, bf, is_4dof=True, min_loop_num=PARAMS["min_loop_num"], init_min=10)
        if r is not None:
            ref_edges.append((old.msg_id, new.msg_id, r))
        return r is not None

    det = M.LoopDetectorRef(1, compute_loop=compute_loop, **PARAMS)
    for i, (p, rv, sg, pose) in enumerate(plan):
        geo[i], g = oracle_frame(sp_w, vw, comp, mean, frames[i], i, pose)
        det.on_image_recv(M.FisheyeFrameDesc(msg_id=i, drone_id=1, landmark_num=geo[i]["landmark_num"], prevent_adding_db=False,
                                             images=[M.ImageDesc(drone_id=1, landmark_num=im["landmark_num"], image_desc=g[d]) for d, im in enumerate(geo[i]["images"])]))
    ref_cand = np.array([[r["msg_id"], r["old_msg_id"], r["dir_new"], r["dir_old"]] for r in det.log if r["old_msg_id"] != -1], np.int64).reshape(-1, 4)
    # ---- same candidates, same edges, the oracle's poses, the scene's ground truth ------------------------------------------------------------------
    assert rows == det.database_size() == 4 * n
    assert hits == len(cand) and np.array_equal(cand, ref_cand), (cand, ref_cand)
    revisit_of = {i: plan[i][0] for i in range(N_PLACES, n)}                  # second visit i closes on first visit plan[i][0]
    assert {(int(a), int(b)) for a, b in cand[:, :2]} >= {(i, p) for i, p in revisit_of.items()}          # every revisit finds its first visit
    assert calls == len(cand)
    # A stereo MISMATCH between two key points on the same image row triangulates to a point at infinity (w ~ 0: 1e15 m) that passes the
    # reference's acceptance test (reprojection error <= TRIANGLE_THRES and z > 0, loop_cam.cpp:397-444) and is flagged as a valid landmark;
    # every solver downstream (the all-inlier DLT that starts the PnP refit, here as in cv::solvePnP) is ill-conditioned on such an input, and two
    # correct implementations may legitimately differ on it.  Key frames holding one are compared on candidates only, not on edges / poses.
    ill = {i for i, f in geo.items() if any((np.abs(im["landmarks_3d"][im["landmarks_flag"] > 0] - f["pose_drone"][0]).max(initial=0) > 1e3) for im in f["images"])}
    assert len(ill) <= 2, ill
    edges = np.array([e for e in edges if int(e[1]) not in ill]).reshape(-1, 12)
    ref_edges = [x for x in ref_edges if x[1] not in ill]
    got_list = [(int(e[0]), int(e[1]), int(e[4])) for e in edges]
    ref_list = [(a, b, r["inliers"]) for a, b, r in ref_edges]
    if got_list != ref_list:                                                  # leave the evidence where a GPU-box run can be read back from
        import json, os
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump({"product": got_list, "oracle": ref_list, "candidates": cand.tolist(), "plan": [(p, rv, sg) for p, rv, sg, _ in plan]}, open("gpurun_out/e2e_debug.json", "w"))
    assert n_edges >= len(edges) == len(ref_edges) >= N_PLACES - 2, (got_list, ref_list, cand.tolist())
    n_exact = 0
    for e, (old_id, new_id, r) in zip(edges, ref_edges):
        assert (int(e[0]), int(e[1]), int(e[2]), int(e[3])) == (old_id, new_id, 1, 1)
        assert abs(int(e[4]) - r["inliers"]) <= 2 and r["inliers"] > 100, (int(e[4]), r["inliers"])
        n_exact += int(e[4]) == r["inliers"]
        tol = 1e-6 if int(e[4]) == r["inliers"] else 1e-4
        pos, att = r["relative_pose"]
        assert np.abs(e[5:8] - pos).max() < tol, (e[5:8], pos)
        assert min(np.abs(e[8:12] - att).max(), np.abs(e[8:12] + att).max()) < tol
        # ground truth: the two visits are the same physical pose -> identity, whatever the drifted odometry says
        assert revisit_of.get(new_id) == old_id
        assert np.linalg.norm(e[5:8]) < 0.10 and abs(G.wrap_angle(G.quat2eulers(e[8:12])[2])) < math.radians(1.0), e
    assert n_exact >= len(edges) - 2, (n_exact, len(edges))                   # the inlier SET is the oracle's on all but at most two edges
