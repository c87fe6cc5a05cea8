// fisheye_flatten.hpp -- the undistortion maps of FisheyeUndist (swarm_localization/test/fisheye_undist.hpp:118-215): for a fisheye camera and a
// field of view, one top view (W x W, focal length f_center) and four side views (W x sideImgHeight, focal length W/2) as virtual pinhole
// cameras; map(x, y) = where the pinhole pixel's ray lands in the fisheye image.  The remap itself runs on the GPU (omni_flatten_*).
// camodocal is un-vendored: the camera here is its MEI / CataCamera model (unit-sphere projection with mirror parameter xi, radial-tangential
// distortion, generalised focal lengths), restated from the published model -- PARITY UNPINNED.
#pragma once
#include <cmath>
#include <vector>

#include "geometry.hpp"

namespace omni {

struct MeiCamera {                 // camodocal::CataCamera parameters (mirror_parameters.xi, distortion k1 k2 p1 p2, projection gamma1 gamma2 u0 v0)
    double xi = 0, k1 = 0, k2 = 0, p1 = 0, p2 = 0, gamma1 = 1, gamma2 = 1, u0 = 0, v0 = 0;
    geom::Vec2 spaceToPlane(geom::Vec3 P) const {
        const double n = geom::norm(P);
        const double z = P.z / n + xi;
        double mx = P.x / n / z, my = P.y / n / z;
        const double mx2 = mx * mx, my2 = my * my, mxy = mx * my, rho2 = mx2 + my2, rad = k1 * rho2 + k2 * rho2 * rho2;
        const double dx = mx * rad + 2 * p1 * mxy + p2 * (rho2 + 2 * mx2), dy = my * rad + 2 * p2 * mxy + p1 * (rho2 + 2 * my2);
        mx += dx; my += dy;
        return {gamma1 * mx + u0, gamma2 * my + v0};
    }
};

struct FlattenMaps {
    std::vector<int> w, h;
    std::vector<std::vector<float>> xy;        // per view [h][w][2]
    double f_center = 0, f_side = 0;
    int side_height = 0;
};

// genOneUndistMap (:188-215)
inline std::vector<float> gen_one_undist_map(const MeiCamera& cam, const geom::Quat& rotation, unsigned w, unsigned h, double f) {
    std::vector<float> map((size_t)w * h * 2);
    for (unsigned x = 0; x < w; ++x)
        for (unsigned y = 0; y < h; ++y) {
            const geom::Vec3 obj = rotation * geom::Vec3{(double)x - (double)w / 2, (double)y - (double)h / 2, f};
            const geom::Vec2 p = cam.spaceToPlane(obj);
            map[((size_t)y * w + x) * 2] = (float)p.x; map[((size_t)y * w + x) * 2 + 1] = (float)p.y;
        }
    return map;
}

inline geom::Quat angle_axis(double angle, double ax, double ay, double az) {
    const double s = std::sin(angle / 2);
    return {std::cos(angle / 2), ax * s, ay * s, az * s};
}

// generateAllUndistMap (:118-186): view 0 = top (or down, cam_id == 1: the rig is flipped about X), views 1..4 = the side views
inline FlattenMaps generate_all_undist_maps(const MeiCamera& cam, unsigned img_width, double fov_deg, int cam_id) {
    FlattenMaps m;
    double side_fov = (fov_deg - 180) * M_PI / 180.0;
    if (side_fov < 0) side_fov = 0;
    const double center_fov = fov_deg * M_PI / 180.0 - side_fov * 2;
    m.f_center = (double)img_width / 2 / std::tan(center_fov / 2);
    m.f_side = (double)img_width / 2;
    m.side_height = (int)(2 * m.f_side * std::tan(side_fov / 2));
    geom::Quat t;
    auto push = [&](unsigned w, unsigned h, double f) { m.w.push_back((int)w); m.h.push_back((int)h); m.xy.push_back(gen_one_undist_map(cam, t, w, h, f)); };
    push(img_width, img_width, m.f_center);
    if (cam_id == 1) t = angle_axis(M_PI, 1, 0, 0);
    if (m.side_height > 0) {
        t = t * angle_axis(-M_PI / 2, 1, 0, 0);
        push(img_width, (unsigned)m.side_height, m.f_side);
        for (int i = 0; i < 3; ++i) { t = t * angle_axis(M_PI / 2, 0, 1, 0); push(img_width, (unsigned)m.side_height, m.f_side); }
    }
    return m;
}

}  // namespace omni
