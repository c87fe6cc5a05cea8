"""Fisheye flattening: binding of host/fisheye_flatten.hpp (undistortion maps, C++) + capi.Flatten (the GPU remap).

Mirrors swarm_detector_pkg::FisheyeUndist (swarm_localization/test/fisheye_undist.hpp:17-215): FisheyeUndist(camera, fov, width, cam_id) builds
five undistortion maps (top / down view + four side views) and undist_all_cuda remaps a fisheye image into them (cv::cuda::remap, INTER_LINEAR).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi, pipeline


def generate_undist_maps(mei, img_width: int, fov_deg: float, cam_id: int = 0):
    """mei = (xi, k1, k2, p1, p2, gamma1, gamma2, u0, v0) of the camodocal MEI / CataCamera model -> list of [h][w][2] float32 maps."""
    L = pipeline.lib()
    fp = C.POINTER(C.c_float)
    L.omni_fisheye_maps.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(fp)]
    m = (C.c_double * 9)(*[float(x) for x in mei])
    n = C.c_int(0)
    vw, vh = (C.c_int * 8)(), (C.c_int * 8)()
    if L.omni_fisheye_maps(m, img_width, fov_deg, cam_id, C.byref(n), vw, vh, None):
        raise capi.OmniError("omni_fisheye_maps failed")
    maps = [np.empty((vh[v], vw[v], 2), np.float32) for v in range(n.value)]
    ptrs = (fp * n.value)(*[a.ctypes.data_as(fp) for a in maps])
    if L.omni_fisheye_maps(m, img_width, fov_deg, cam_id, C.byref(n), vw, vh, ptrs):
        raise capi.OmniError("omni_fisheye_maps failed")
    return maps


class FisheyeUndist:
    """FisheyeUndist(camera_config, fov, enable_cuda, imgWidth, cam_id): maps on the host once, remap on the GPU per image."""

    def __init__(self, ctx: capi.Context, mei, src_width: int, src_height: int, fov_deg: float, img_width: int = 600, cam_id: int = 0):
        self.maps = generate_undist_maps(mei, img_width, fov_deg, cam_id)
        self.flatten = capi.Flatten(ctx, src_width, src_height, self.maps)

    def undist_all(self, fisheye_u8: np.ndarray, enable_rear: bool = True):
        views = self.flatten(fisheye_u8)[0]
        return views if enable_rear else views[:-1]

    def close(self):
        self.flatten.close()
