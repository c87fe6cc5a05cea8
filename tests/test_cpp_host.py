"""The C++ host adapters (omni-swarm_amd/host/omni_swarm.hpp): g++ compile/link check on CPU, full run vs the oracle on GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "omni-swarm_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "host_smoke")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_smoke.cpp"),
                           "-L", LIBDIR, "-lomni_hip", f"-Wl,-rpath,{LIBDIR}"])
    return exe


def test_cpp_adapters_compile_with_plain_gxx(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr          # runs, links libomni_hip.so, needs arguments


def test_omnw_roundtrip(tmp_path, omni):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_weights
    from omni_swarm_amd import weights
    w = weights.superpoint_synth_weights(0)
    p = str(tmp_path / "sp.omnw")
    export_weights.write_omnw(p, w)
    raw = open(p, "rb").read()
    assert raw[:8] == b"OMNW1\0\0\0" and int.from_bytes(raw[8:12], "little") == 24
    assert len(raw) > sum(v.size for v in w.values()) * 4


@pytest.mark.gpu
def test_cpp_host_matches_oracle(tmp_path, omni, ctx, golden):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_weights
    from omni_swarm_amd import capi, synth, weights
    from oracle import match_ref as M, mobilenetvlad_ref as V, postproc_ref as P, superpoint_ref as S
    from tests import detector_stream as DS
    exe = _build(tmp_path)
    W, H = 96, 64
    sp_w = weights.superpoint_synth_weights(0)
    export_weights.write_omnw(str(tmp_path / "sp.omnw"), sp_w)
    vw = weights.mobilenetvlad_synth_weights()
    export_weights.write_omnw(str(tmp_path / "vlad.omnw"), export_weights.vlad_tensors(vw, weights.mobilenetvlad_layer_specs(), capi.VLAD_KINDS))
    comp, mean = synth.pca()
    np.savetxt(tmp_path / "comp.csv", comp, delimiter=",", fmt="%.9g")
    np.savetxt(tmp_path / "mean.csv", mean, fmt="%.9g")
    img = synth.image_u8(100, H, W, n_shapes=40)
    img.tofile(tmp_path / "img.u8")
    db = synth.global_db(500, seed=3)
    q, _ = synth.queries_from_db(db, 1, seed=4)
    db.tofile(tmp_path / "db.f32"); q.tofile(tmp_path / "q.f32")
    a, b, _ = synth.local_descriptors(90, 64, seed=5, pair_noise=0.2)
    b = b[:70]
    a.tofile(tmp_path / "a.f32"); b.tofile(tmp_path / "b.f32")
    frames = DS.make_stream(seed=11, n_frames=40)
    rows = []
    for fr in frames:
        rows.append(np.array([fr["msg_id"], fr["drone_id"], fr["landmark_num"], float(fr["prevent_adding_db"])], np.float32))
        for im in fr["images"]:
            rows.append(np.concatenate([[np.float32(im["landmark_num"])], im["image_desc"]]).astype(np.float32))
    np.concatenate(rows).tofile(tmp_path / "stream.bin")
    t = lambda n: str(tmp_path / n)
    r = subprocess.run([exe, t("sp.omnw"), t("comp.csv"), t("mean.csv"), t("vlad.omnw"), t("img.u8"), str(W), str(H), t("db.f32"), "500",
                        t("q.f32"), t("a.f32"), "90", t("b.f32"), "70", t("stream.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {ln.split(" ", 1)[0]: ln.split(" ")[1:] for ln in r.stdout.strip().split("\n")}
    assert "OK" in out
    # SuperPoint (f32 path) vs oracle; CSV round trip keeps 9 significant digits
    semi, desc = S.forward(sp_w, S.preprocess_u8(img))
    xy, conf, _, _ = P.get_keypoints(semi[0], 0.015, 200)
    d64, _ = P.compute_descriptors(desc[0], xy, W, H, comp, mean)
    n, dim = int(out["SP_N"][0]), int(out["SP_N"][1])
    assert n == len(xy) and dim == 64
    # enable_perf: the reference's line (superpoint_tensorrt.cpp:130-162: "Inference Time .. from_blob .. getKeyPoints+computeDescriptors .. inference all .. features .. desc size ..")
    perf = out["Inference"]
    assert perf[0] == "Time" and perf[2] == "from_blob" and perf[4] == "getKeyPoints+computeDescriptors" and perf[6:8] == ["inference", "all"] and perf[9] == "features"
    assert float(perf[1]) > 0 and float(perf[5]) > 0 and float(perf[8]) >= float(perf[1]) and int(perf[10]) == n and int(perf[13]) == n * dim
    stages = [ln for ln in r.stdout.split("\n") if ln.startswith("  stages (ms):")]
    assert len(stages) == 1 and "conv1b+pool" in stages[0] and "nms+topk+describe" in stages[0]
    assert out["PERF_SAME"] == ["1"]
    assert np.array_equal(np.array(out["SP_KPS"], int).reshape(-1, 2), xy)
    assert np.abs(np.array(out["SP_DESC"], np.float32).reshape(n, 64) - d64).max() < 1e-4
    y = V.forward(vw, img)[0]
    yc = np.array(out["VLAD"], np.float32)
    assert np.linalg.norm(yc - y) / np.linalg.norm(y) < 1e-3
    D, I = M.ip_search(db, q, 15)
    assert int(out["IP_NTOTAL"][0]) == 500 and np.array_equal(np.array(out["IP_I"], np.int64), I[0])
    assert np.allclose(np.array(out["IP_D"], np.float32), D[0], rtol=1e-5, atol=1e-6)
    qi, ti, dd = M.bf_match(a, b, 0)
    bf = np.array(out["BF"], np.float64).reshape(-1, 3)
    assert np.array_equal(bf[:, 0].astype(int), qi) and np.array_equal(bf[:, 1].astype(int), ti) and np.allclose(bf[:, 2], dd, rtol=1e-7)
    tr = DS.trace(DS.run_oracle(frames))
    assert np.array_equal(np.array(out["DET"], np.int64).reshape(-1, 7), tr)
    assert np.array_equal(np.array(out["DETB"], np.int64).reshape(-1, 7), tr)      # on_images_recv_batch: same decisions
    # omni::LoopCamHIP (omni_cam): same key points / descriptors / global descriptor as the blocking calls, self-match diagonal
    assert out["CAM"][:2] == ["1", "1"] and int(out["CAM"][2]) == n, out["CAM"]


PARAMS_PIN = os.path.join(ROOT, "oracle", "_ref", "params_pin")
LAUNCH_DIR = os.path.join(ROOT, "oracle", "_ref", "launch")


def _norm(t, v):
    """one text per value whatever printed it (the pin prints %.17g doubles, Python's reader repr()s them)"""
    return repr(float(v)) if t == "D" else v


def test_launch_parameter_table_is_pinned_to_the_reference_text(omni):
    """omni::SwarmLoopParams (host/swarm_loop_params.hpp): the 50 parameters of SwarmLoop::Init == the reference's own nh.param<T>(name, variable, default)
    calls, compiled verbatim (swarm_loop.cpp:205-270 into oracle/_ref/params_pin): same names in the same order, same types, same defaults."""
    from omni_swarm_amd import pipeline
    assert os.path.exists(PARAMS_PIN), "make -C oracle ref"
    ref = [tuple(l.split("\t")) for l in subprocess.run([PARAMS_PIN, "defaults"], capture_output=True, text=True, check=True).stdout.split("\n") if l]
    ref = [r if len(r) == 3 else r + ("",) for r in ref]
    mine = pipeline.swarm_params_table()
    assert len(ref) == len(mine) == 50
    assert [(n, t, _norm(t, d)) for n, t, d in ref] == [(n, t, _norm(t, d)) for n, t, d in mine]
    # a launch file that sets nothing = the defaults
    vals, mism, unk = pipeline.swarm_params_from_launch('<launch><node pkg="swarm_loop" type="swarm_loop_node" name="swarm_loop"/></launch>')
    assert not mism and not unk and [(n, _norm(t, vals[n])) for n, t, _ in mine] == [(n, _norm(t, d)) for n, t, d in mine]


@pytest.mark.parametrize("launch,args", [("realsense.launch", {}), ("realsense.launch", {"self_id": "7", "show": "true", "max_freq": "2.5", "match_index_dist": "3"}),
                                         ("nodelet-sfisheye.launch", {}), ("node-sfisheye.launch", {"self_id": "2"}), ("pc-outdoor-fisheye.launch", {})])
def test_reference_launch_files_configure_the_node_as_roslaunch_would(omni, launch, args):
    """The reference's own launch files through omni::SwarmLoopParams::from_launch against the reference's own parameter block: an independent reader
    (oracle/launch_ref.py: xml.etree + PyYAML, the YAML library roslaunch calls; roslaunch's convert_value for <param>) produces the typed content of the
    parameter server, the verbatim block runs on it with roscpp's param<T> conversions, and every variable it sets equals the loader's field.  Includes the
    quirk the files themselves hold: `loop_cov_pos: 1e-2` is a STRING in YAML 1.1, nh.param<double> refuses it, the node runs with the default 0.013."""
    from omni_swarm_amd import pipeline
    from oracle import launch_ref
    path = os.path.join(LAUNCH_DIR, launch)
    assert os.path.exists(PARAMS_PIN) and os.path.exists(path), "make -C oracle ref"
    xml = open(path).read()
    server = launch_ref.parameter_server(xml, "swarm_loop", args)
    r = subprocess.run([PARAMS_PIN, "apply"], input="".join(f"{n}\t{t}\t{v}\n" for n, t, v in server), capture_output=True, text=True, check=True)
    ref = dict(tuple(l.split("\t")) if l.count("\t") == 1 else (l.rstrip("\t"), "") for l in r.stdout.split("\n") if l)
    vals, mism, unk = pipeline.swarm_params_from_launch(xml, "swarm_loop", args)
    table = pipeline.swarm_params_table()
    assert set(ref) == set(vals) == {n for n, _, _ in table}
    for n, t, _ in table:
        assert _norm(t, vals[n]) == _norm(t, ref[n]), (n, vals[n], ref[n])
    names = {n for n, _, _ in table}
    types = {n: t for n, t, _ in table}
    ok = {"I": "ID", "D": "ID", "B": "B", "S": "S"}
    assert sorted(unk) == sorted(n for n, _, _ in server if n not in names)
    assert sorted(mism) == sorted(n for n, t, _ in server if n in names and t not in ok[types[n]])
    if "sfisheye" in launch:
        assert mism == ["loop_cov_pos"] and float(vals["loop_cov_pos"]) == 0.013          # the reference's own files: the YAML-1.1 string
    if args.get("self_id"):
        assert vals["self_id"] == args["self_id"]
    assert "enable_pub_remote_img" in unk                                                  # set by every file, read by no code
