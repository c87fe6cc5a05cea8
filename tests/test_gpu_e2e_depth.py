"""End to end in CameraConfig::PINHOLE_DEPTH (launch/realsense.launch; BASELINE.json configs[0]: ONE 640 x 480 gray image + its depth image per key
frame): image -> SuperPoint + MobileNetVLAD (no rows blanked: loop_cam.cpp:536 masks STEREO_FISHEYE only) -> landmarks read from the depth image
(generate_gray_depth_image_descriptor, loop_cam.cpp:231-339) -> database / query rule on direction 0 (loop_detector.cpp:252-258) ->
compute_correspond_features with MAX_DIRS = 1 (swarm_loop.cpp:279-280) -> homography mask -> PnP -> LoopEdge, through the C++ key-frame pipeline
on the GPU (omni_cam_create_mono + host/keyframe_pipeline.hpp) against the oracle chain on the SAME images and against the scene's ground truth.

Scene (omni_swarm_amd.synth.depth_keyframe): a forward-looking camera in front of a textured wall with a step (2.0 m / 3.5 m), 8 places, each
visited twice -- the second visit is the loop closure: same spot, sensor noise on the image and the depth, a DRIFTED odometry pose."""
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
W, H, THR, MAXN, MB = 640, 480, 0.02, 200, 4
FX = FY = 320.0
CX, CY = 320.0, 240.0
NEAR, FAR, ACCEPT_MIN = 0.3, 7.0, 30                                  # accept_min_3d_pts: 30 in realsense.launch
N_PLACES = 8
PARAMS = dict(inner_product_thres=0.3, init_mode_product_thres=0.2, match_index_dist=3, min_loop_num=30, min_direction_loop=1)   # min_direction_loop: 1 (realsense.launch)
R_BC = np.array([[0.0, 0, 1], [-1, 0, 0], [0, -1, 0]])
EXT = G.pose([0.0, 0.0, 0.0], G.q_from_R(R_BC))                        # KeyframePipeline::view_extrinsic in this mode


def schedule():
    out = []
    for p in range(N_PLACES):
        out.append((p, 0, 0.0, G.pose([6.0 * p, 1.5 * (p % 3), 1.0], G.q_from_yaw(0.05 * p))))
    for i in range(N_PLACES):
        p = (3 * i + 2) % N_PLACES
        true = out[p][3]
        drift = G.pose(true[0] + np.array([0.25, -0.15, 0.04]), G.qmul(G.q_from_yaw(0.015), true[1]))
        out.append((p, 1, 0.0 if i % 2 == 0 else 1.5, drift))
    return out


def oracle_frame(sp_w, vw, comp, mean, gray, depth, msg_id, pose):
    semi, desc = S.forward(sp_w, S.preprocess_u8(gray[None], False))
    xy, _, _, _ = P.get_keypoints(semi[0], THR, MAXN)
    d64, _ = P.compute_descriptors(desc[0], xy, W, H, comp, mean)
    g = V.forward(vw, gray[None])
    lift64 = lambda x: np.stack([((x[:, 0] - CX) / FX), ((x[:, 1] - CY) / FY)], 1)
    xyf = xy.astype(np.float32)
    _, l3, fl = G.depth_landmarks(pose, EXT, xyf, lift64, depth, NEAR, FAR, ACCEPT_MIN)
    img = {"landmark_num": len(xy), "landmarks_2d": xy.astype(np.float64), "landmarks_2d_norm": lift64(xyf.astype(np.float64)).astype(np.float32).astype(np.float64),
           "feature_descriptor": d64, "camera_extrinsic": EXT, "landmarks_3d": l3.astype(np.float32).astype(np.float64), "landmarks_flag": fl}
    return {"msg_id": msg_id, "drone_id": 1, "timestamp": float(msg_id), "pose_drone": pose, "images": [img], "landmark_num": len(xy)}, g


def test_pinhole_depth_images_to_loop_edges_equal_the_oracle_chain_and_the_ground_truth(omni, ctx, tmp_path):
    c = omni.capi
    from omni_swarm_amd import pipeline, weights
    sp_w, vw = S.synth_weights(0), V.synth_weights()
    comp, mean = synth.pca()
    files = weights.write_pipeline_files(str(tmp_path), sp_w, comp, mean, vw, V.layer_specs(), c.VLAD_KINDS)
    plan = schedule()
    n = len(plan)
    frames = [synth.depth_keyframe(p, H, W, rv, sg) for (p, rv, sg, _) in plan]
    pins = []
    for s in range(0, n, MB):
        p = ctx.host_alloc((MB, H, W), np.uint8)
        p[:] = np.stack([frames[s + m][0] for m in range(MB)])
        pins.append(p)
    pl = pipeline.KeyframePipeline(0, files["sp"], files["comp"], files["mean"], files["vlad"], W, H, THR, MAXN, c.PREC_SPLIT, MB, 2, c.STORE_F32, 1,
                                   PARAMS["inner_product_thres"], PARAMS["init_mode_product_thres"], PARAMS["match_index_dist"], PARAMS["min_loop_num"],
                                   PARAMS["min_direction_loop"], geometry=True,
                                   pinhole_depth=dict(fx=FX, fy=FY, cx=CX, cy=CY, depth_near=NEAR, depth_far=FAR, accept_min_3d_pts=ACCEPT_MIN))
    pl.set_poses(0, np.array([np.concatenate([q[3][0], q[3][1]]) for q in plan]))
    pl.set_depth(0, np.stack([f[1] for f in frames]))
    hits = pl.run(n, 0, [p.ctypes.data for p in pins], 0, None, True)
    cand, edges = pl.candidates(), pl.edges()
    calls, n_edges = pl.geometry_stats()
    rows = pl.db_rows
    pl.close()
    for p in pins:
        ctx.host_free(p)
    # the same key frames one at a time through the streaming intake (push_keyframe: what a ROS callback calls; the stamp is the key frame's own,
    # the last micro-batch is flushed as a partial one): the same candidates and the same edges
    ps = pipeline.KeyframePipeline(0, files["sp"], files["comp"], files["mean"], files["vlad"], W, H, THR, MAXN, c.PREC_SPLIT, 3, 2, c.STORE_F32, 1,
                                   PARAMS["inner_product_thres"], PARAMS["init_mode_product_thres"], PARAMS["match_index_dist"], PARAMS["min_loop_num"],
                                   PARAMS["min_direction_loop"], geometry=True,
                                   pinhole_depth=dict(fx=FX, fy=FY, cx=CX, cy=CY, depth_near=NEAR, depth_far=FAR, accept_min_3d_pts=ACCEPT_MIN))
    hits_s = 0
    for i, (gray, depth) in enumerate(frames):
        hits_s += ps.push_keyframe([gray], i, float(i), np.concatenate([plan[i][3][0], plan[i][3][1]]), False, depth=depth)
    hits_s += ps.flush()                                                      # 16 key frames in micro-batches of 3: five full units and one of 1
    cand_s, edges_s = ps.candidates(), ps.edges()
    assert ps.db_rows == rows
    ps.close()
    assert hits_s == hits and np.array_equal(cand_s, cand)
    assert np.array_equal(edges_s[:, :5], edges[:, :5]) and np.abs(edges_s[:, 5:] - edges[:, 5:]).max(initial=0) < 1e-9
    # ---- the oracle chain ------------------------------------------------------------------------------------------------------------------------
    geo, ref_edges = {}, []
    bf = lambda a, b: M.bf_match(a, b, 0)

    def compute_loop(new, old, dn, do, init_mode):
        r = G.compute_loop(geo[new.msg_id], geo[old.msg_id], dn, do, init_mode, bf, is_4dof=True, min_loop_num=PARAMS["min_loop_num"], init_min=10,
                           max_dirs=1, min_direction_loop=1)
        if r is not None:
            ref_edges.append((old.msg_id, new.msg_id, r))
        return r is not None

    det = M.LoopDetectorRef(1, compute_loop=compute_loop, camera_configuration=M.PINHOLE_DEPTH, **PARAMS)
    for i, (p, rv, sg, pose) in enumerate(plan):
        geo[i], g = oracle_frame(sp_w, vw, comp, mean, frames[i][0], frames[i][1], i, pose)
        det.on_image_recv(M.FisheyeFrameDesc(msg_id=i, drone_id=1, landmark_num=geo[i]["landmark_num"], prevent_adding_db=False,
                                             images=[M.ImageDesc(drone_id=1, landmark_num=geo[i]["images"][0]["landmark_num"], image_desc=g[0])]))
    ref_cand = np.array([[r["msg_id"], r["old_msg_id"], r["dir_new"], r["dir_old"]] for r in det.log if r["old_msg_id"] != -1], np.int64).reshape(-1, 4)
    assert rows == det.database_size() == n                                   # one row per key frame
    assert hits == len(cand) and np.array_equal(cand, ref_cand), (cand, ref_cand)
    assert (cand[:, 2:] == 0).all()                                           # direction 0 on both sides
    revisit_of = {i: plan[i][0] for i in range(N_PLACES, n)}
    assert {(int(a), int(b)) for a, b in cand[:, :2]} >= {(i, p) for i, p in revisit_of.items()}
    assert calls == len(cand)
    got_list = [(int(e[0]), int(e[1]), int(e[4])) for e in edges]
    ref_list = [(a, b, r["inliers"]) for a, b, r in ref_edges]
    assert n_edges == len(edges) == len(ref_edges) >= N_PLACES - 1, (got_list, ref_list, cand.tolist())
    n_exact = 0
    for e, (old_id, new_id, r) in zip(edges, ref_edges):
        assert (int(e[0]), int(e[1]), int(e[2]), int(e[3])) == (old_id, new_id, 1, 1)
        assert abs(int(e[4]) - r["inliers"]) <= 2 and r["inliers"] > 60, (int(e[4]), r["inliers"])
        n_exact += int(e[4]) == r["inliers"]
        tol = 1e-6 if int(e[4]) == r["inliers"] else 1e-4
        pos, att = r["relative_pose"]
        assert np.abs(e[5:8] - pos).max() < tol, (e[5:8], pos)
        assert min(np.abs(e[8:12] - att).max(), np.abs(e[8:12] + att).max()) < tol
        assert revisit_of.get(new_id) == old_id
        assert np.linalg.norm(e[5:8]) < 0.10 and abs(G.wrap_angle(G.quat2eulers(e[8:12])[2])) < math.radians(1.0), e      # ground truth: the same physical pose
    assert n_exact >= len(edges) - 2, (n_exact, len(edges))
