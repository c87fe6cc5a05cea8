"""omni_swarm_amd -- MI355X-native (gfx950) swarm_loop hot path: Python host layer over the C ABI.

The directory is named ``omni-swarm_amd`` (not importable as-is); ``omni_loader.load()`` at the repo root registers it
as the module ``omni_swarm_amd``.  Everything numerical happens in ``lib/libomni_hip.so`` (hand-written HIP,
include/omni_hip.h); this package only mirrors the reference's class surfaces on top of it:

    capi.SuperPoint      <-> SuperPointTensorRT      (swarm_loop/include/swarm_loop/superpoint_tensorrt.h:12-29)
    capi.MobileNetVLAD   <-> MobileNetVLADTensorRT   (swarm_loop/include/swarm_loop/mobilenetvlad_tensorrt.h:6-22)
    capi.IndexFlatIP     <-> faiss::IndexFlatIP      (swarm_loop/src/loop_detector.cpp:166-170,213)
    capi.bf_match        <-> cv::BFMatcher(NORM_L2, crossCheck=true).match
    detector.LoopDetector<-> LoopDetector DB + decision rules (swarm_loop/src/loop_detector.cpp:11-287)
    frontend.LoopCam     <-> LoopCam::on_flattened_images (swarm_loop/src/loop_cam.cpp:178-229,341-585), CNN part
    shard.ShardedIndex   row-sharded index + all-gather top-k merge (new)

There is no CPU fallback: importing works anywhere, but creating a Context without a HIP device raises.
"""
from . import capi  # noqa: F401
