// TEST INFRASTRUCTURE ONLY (oracle/).  C entry point around the reference's OWN map generation -- FisheyeUndist::generateAllUndistMap and
// ::genOneUndistMap (swarm_localization/test/fisheye_undist.hpp:118-225: from the first to the end of the class), extracted at build time into
// oracle/_ref/flatten_snippet.inc and compiled VERBATIM into a class that carries the members those two functions touch.
// Stand-ins, not pinned: camodocal's camera (spaceToPlane = the MEI / CataCamera projection as published, the same formula as
// oracle/flatten_ref.space_to_plane; PinholeCamera only stores its parameters), Eigen (Vector3d / Vector2d, Quaterniond x AngleAxis, quaternion
// x vector: Eigen's own formulas), cv::Mat with Vec2f elements, ROS_INFO / ROS_DEBUG.
// Pinned: the field-of-view arithmetic (side / centre FOV, f_center, f_side, the integer side-image height), the sequence of view rotations
// (incl. the cam_id == 1 flip), the pixel -> ray map, the float32 map layout.  Pins oracle/flatten_ref.generate_all_undist_maps (tests/test_flatten.py).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#define ROS_INFO(...) ((void)0)
#define ROS_DEBUG(...) ((void)0)
#define CV_32FC2 13

namespace Eigen {
struct Vector3d {
    double v[3];
    Vector3d() : v{0, 0, 0} {}
    Vector3d(double a, double b, double c) : v{a, b, c} {}
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    static Vector3d UnitX() { return Vector3d(1, 0, 0); }
    Vector3d cross(const Vector3d& o) const { return Vector3d(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]); }
};
inline Vector3d operator+(const Vector3d& a, const Vector3d& b) { return Vector3d(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]); }
inline Vector3d operator*(double s, const Vector3d& a) { return Vector3d(s * a.v[0], s * a.v[1], s * a.v[2]); }
struct Vector2d {
    double v[2];
    Vector2d() : v{0, 0} {}
    double& operator[](int i) { return v[i]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
};
template <typename T>
struct AngleAxis {
    T angle; Vector3d axis;
    AngleAxis(T a, const Vector3d& ax) : angle(a), axis(ax) {}
};
typedef AngleAxis<double> AngleAxisd;
struct Quaterniond {
    double w, x, y, z;
    Quaterniond() : w(1), x(0), y(0), z(0) {}
    Quaterniond(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
    Quaterniond(const AngleAxisd& a) { const double h = 0.5 * a.angle, s = std::sin(h); w = std::cos(h); x = s * a.axis[0]; y = s * a.axis[1]; z = s * a.axis[2]; }   // Eigen: QuaternionBase::operator=(AngleAxis)
    static Quaterniond Identity() { return Quaterniond(); }
    Quaterniond operator*(const Quaterniond& b) const {                      // Eigen's quat_product
        return Quaterniond(w * b.w - x * b.x - y * b.y - z * b.z, w * b.x + x * b.w + y * b.z - z * b.y, w * b.y + y * b.w + z * b.x - x * b.z,
                           w * b.z + z * b.w + x * b.y - y * b.x);
    }
    Quaterniond operator*(const AngleAxisd& a) const { return *this * Quaterniond(a); }
    Vector3d operator*(const Vector3d& v) const {                            // Eigen's _transformVector: v + w * uv + u x uv, uv = 2 u x v
        const Vector3d u(x, y, z), uv = 2.0 * u.cross(v);
        return v + w * uv + u.cross(uv);
    }
};
}  // namespace Eigen

namespace camodocal {
struct Camera {                                   // the catadioptric (MEI) model of camodocal::CataCamera::spaceToPlane
    double xi, k1, k2, p1, p2, g1, g2, u0, v0;
    void spaceToPlane(const Eigen::Vector3d& P, Eigen::Vector2d& p) const {
        const double n = std::sqrt(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]);
        const double z = P[2] / n + xi, mx = P[0] / n / z, my = P[1] / n / z;
        const double rho2 = mx * mx + my * my, rad = k1 * rho2 + k2 * rho2 * rho2;
        const double dx = mx * rad + 2 * p1 * mx * my + p2 * (rho2 + 2 * mx * mx), dy = my * rad + 2 * p2 * mx * my + p1 * (rho2 + 2 * my * my);
        p[0] = g1 * (mx + dx) + u0; p[1] = g2 * (my + dy) + v0;
    }
};
typedef std::shared_ptr<Camera> CameraPtr;
struct PinholeCamera {
    PinholeCamera(const std::string&, int, int, double, double, double, double, double, double, double, double) {}
};
typedef std::shared_ptr<PinholeCamera> PinholeCameraPtr;
}  // namespace camodocal

namespace cv {
struct Point { int x, y; Point(int a, int b) : x(a), y(b) {} };
struct Vec2f { float v[2]; Vec2f() : v{0, 0} {} Vec2f(float a, float b) : v{a, b} {} float operator[](int i) const { return v[i]; } };
class Mat {
public:
    int rows = 0, cols = 0;
    int size[2] = {0, 0};
    std::vector<Vec2f> d;
    Mat() {}
    Mat(int r, int c, int) : rows(r), cols(c), d((size_t)r * c) { size[0] = r; size[1] = c; }
    template <typename T> T& at(Point p) { return d[(size_t)p.y * cols + p.x]; }
};
}  // namespace cv

namespace swarm_detector_pkg {
class FisheyeUndist {
public:
    camodocal::PinholeCameraPtr cam_top, cam_side;
    double f_side = 0, f_center = 0, cx_side = 0, cy_side = 0;
    int cam_id = 0, sideImgHeight = 0;
#define DEG_TO_RAD (M_PI / 180.0)
#include REF_FLATTEN_SNIPPET
// (the snippet closes the class and the namespace)

extern "C" int ref_flatten_maps(const double* mei9, int img_width, double fov_deg, int cam_id, float* out /* 5 maps, [h][w][2] each, top first */, int* heights /* 5 */) {
    auto cam = std::make_shared<camodocal::Camera>();
    cam->xi = mei9[0]; cam->k1 = mei9[1]; cam->k2 = mei9[2]; cam->p1 = mei9[3]; cam->p2 = mei9[4]; cam->g1 = mei9[5]; cam->g2 = mei9[6]; cam->u0 = mei9[7]; cam->v0 = mei9[8];
    swarm_detector_pkg::FisheyeUndist fu;
    fu.cam_id = cam_id;
    const unsigned w = (unsigned)img_width;
    std::vector<cv::Mat> maps = fu.generateAllUndistMap(cam, Eigen::Vector3d(0, 0, 0), w, fov_deg);
    size_t o = 0;
    for (size_t m = 0; m < maps.size(); ++m) {
        heights[m] = maps[m].rows;
        for (size_t i = 0; i < maps[m].d.size(); ++i) { out[o++] = maps[m].d[i][0]; out[o++] = maps[m].d[i][1]; }
    }
    return (int)maps.size();
}
