"""LoopNet's wire side (omni-swarm_amd/host/loop_net_wire.hpp; swarm_loop/src/loop_net.cpp:19-120,143-324): header + per-landmark packets,
LCM-style big-endian encoding, reassembly under shuffling and loss with the reference's time-outs, self-sent filter, corrupt packets rejected.
CPU only: g++ builds tests/cpp/wire_check.cpp.

test_wire_is_pinned_to_the_reference_text: the reference's OWN LoopNet (loop_net.h's class + loop_net.cpp, extracted at build time into
oracle/_ref/, never committed) compiled verbatim against stand-in ROS / LCM / swarm_msgs types runs next to LoopNetWire
(tests/cpp/wire_pin.cpp, built by oracle/Makefile)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("wire") / "wire_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", out, os.path.join(ROOT, "tests", "cpp", "wire_check.cpp")])
    return out


def run(exe, group, seed):
    r = subprocess.run([exe, str(group), str(seed)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    return [ln.split() for ln in r.stdout.strip().split("\n")]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_split_and_reassemble(exe, seed):
    for group in (0, 1):
        out = run(exe, group, seed)
        tags = [ln[0] for ln in out]
        assert "SELF_LEAK" not in tags                                  # a drone ignores its own packets
        sent = next(ln for ln in out if ln[0] == "SENT")
        n_pkts, n_hdr, n_bytes, flagged = map(int, sent[1:])
        assert n_hdr == 3 and n_pkts == 3 + flagged                     # one header per non-empty direction, one packet per 3-D landmark
        assert n_bytes == 3 * (8 + 8 + 4 + 8 + 8 + 4 + 4 + 1 + 2 * 56 + 4 + 4096 * 4) + flagged * (8 + 8 + 8 + 4 + 4 + 4 + 7 * 4 + 4 + 64 * 4)
        assert next(ln for ln in out if ln[0] == "JUNK")[1] == "0"      # a packet that does not parse is rejected
        assert next(ln for ln in out if ln[0] == "MALFORMED")[1:] == ["0", "0", "0"]     # right framing, wrong array lengths / direction: dropped
        assert next(ln for ln in out if ln[0] == "ORPHAN")[1:] == ["1", "1", "1", "0", "1"]   # header-less landmarks expire, their id is black-listed
        assert next(ln for ln in out if ln[0] == "BEFORE_TIMEOUT")[1] == "0"    # lossy images only complete by time-out
        lost = int(next(ln for ln in out if ln[0] == "LOST")[1])
        frames = [ln for ln in out if ln[0] == "FRAME"]
        # the reference keys frames by the image's msg_id: one frame per direction; group_by_frame_id reunites the key frame
        assert len(frames) == (1 if group else 3)
        total = 0
        for fr in frames:
            assert fr[1] == "4242" and fr[2] == "2" and fr[4] == "4"
            total += int(fr[3])
            imgs = " ".join(fr[5:-2]).replace("[", "").split("]")[:4]
            for im in imgs:
                d, n, bad, desc_ok, prevent = map(int, im.split())
                assert bad == 0 and desc_ok == 1                         # descriptors, 3-D points and global descriptor arrive bit-exact
                if n:
                    assert prevent == (1 if d == 1 else 0)
            assert fr[-2] == "1.500" and fr[-1] == "0.600"
        assert total == flagged - lost


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_wire_is_pinned_to_the_reference_text(seed):
    """Same key frames through LoopNet::broadcast_fisheye_desc and LoopNetWire::broadcast_fisheye_desc -> the same header / landmark messages
    field for field (also with SEND_ALL_FEATURES, where the header keeps announcing only the landmarks with a 3-D point, loop_net.cpp:62);
    the same packets into both receivers -- in order, with lost landmarks, with lost headers, shuffled -- under the reference's time-outs ->
    the same FisheyeFrameDescriptor sequence and the same receive rates; a landmark overtaking its header: the reference never delivers that
    image, LoopNetWire does (the one deliberate difference)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "wire_pin")
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/wire_pin"])
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/wire_pin is built from /root/reference, which is absent here")
    r = subprocess.run([exe, str(seed)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    assert lines[-1] == "OK 0"
    tags = [ln.split()[0] for ln in lines]
    assert tags.count("SEND") == 2 and tags.count("RECV") == 4 and tags.count("OVERTAKE") == 1
    for ln in lines:
        if ln.startswith("RECV"):
            assert int(ln.split()[3]) >= 3                                     # frames were actually delivered in every schedule
