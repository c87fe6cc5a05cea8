// Drives the host geometry (omni-swarm_amd/host/geometry.hpp, loop_geometry.hpp) from a text protocol on stdin so that
// tests/test_geometry_cpu.py can compare it with the numpy oracle (oracle/geometry_ref.py).  CPU only: the descriptor matcher is the
// oracle's C restatement of cv::BFMatcher (test infrastructure; the product plugs in the HIP matcher, omni::BFMatcherL2X).
#include <cstdio>
#include <iostream>

#include "../../omni-swarm_amd/host/loop_geometry.hpp"

extern "C" int oracle_bf_match(const float* q, int nq, const float* t, int nt, int dim, int mode, int* q_idx, int* t_idx, float* dist_out);

using namespace omni;
using namespace omni::geom;

static Pose read_pose() { Pose p; std::cin >> p.pos.x >> p.pos.y >> p.pos.z >> p.att.w >> p.att.x >> p.att.y >> p.att.z; return p; }
static void print_pose(const Pose& p) { std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g", p.pos.x, p.pos.y, p.pos.z, p.att.w, p.att.x, p.att.y, p.att.z); }

static FisheyeFrameDescriptor read_frame() {
    FisheyeFrameDescriptor f;
    int n_img;
    std::cin >> f.msg_id >> f.drone_id >> f.timestamp >> f.landmark_num;
    f.pose_drone = to_msg(read_pose());
    std::cin >> n_img;
    f.images.resize(n_img);
    for (auto& im : f.images) {
        std::cin >> im.landmark_num;
        im.camera_extrinsic = to_msg(read_pose());
        im.pose_drone = f.pose_drone;
        im.drone_id = f.drone_id;
        const int n = im.landmark_num;
        im.landmarks_2d.resize(n); im.landmarks_2d_norm.resize(n); im.landmarks_3d.resize(n); im.landmarks_flag.resize(n); im.feature_descriptor.resize((size_t)n * 64);
        for (int i = 0; i < n; ++i) {
            int flag;
            std::cin >> im.landmarks_2d[i].x >> im.landmarks_2d[i].y >> im.landmarks_2d_norm[i].x >> im.landmarks_2d_norm[i].y >> im.landmarks_3d[i].x >>
                im.landmarks_3d[i].y >> im.landmarks_3d[i].z >> flag;
            im.landmarks_flag[i] = (uint8_t)flag;
            for (int k = 0; k < 64; ++k) std::cin >> im.feature_descriptor[(size_t)i * 64 + k];
        }
    }
    return f;
}

int main() {
    std::string cmd;
    while (std::cin >> cmd) {
        if (cmd == "tri") {
            int n; std::cin >> n;
            for (int i = 0; i < n; ++i) {
                Pose a = read_pose(), b = read_pose();
                Vec2 p0, p1; std::cin >> p0.x >> p0.y >> p1.x >> p1.y;
                Vec3 X;
                const double err = triangulate_point(a.att, a.pos, b.att, b.pos, p0, p1, X);
                std::printf("TRI %.17g %.17g %.17g %.17g\n", err, X.x, X.y, X.z);
            }
        } else if (cmd == "stereo") {
            Pose pd = read_pose(), eu = read_pose(), ed = read_pose();
            int nu, nd, nm; double thres;
            std::cin >> nu >> nd >> nm >> thres;
            std::vector<Vec2> a(nu), b(nd);
            for (auto& p : a) std::cin >> p.x >> p.y;
            for (auto& p : b) std::cin >> p.x >> p.y;
            std::vector<int> iu(nm), id(nm);
            for (int i = 0; i < nm; ++i) std::cin >> iu[i] >> id[i];
            std::vector<Vec3> l3u, l3d; std::vector<uint8_t> fu, fd;
            const int c = stereo_landmarks(pd, eu, ed, a, b, iu.data(), id.data(), nm, thres, l3u, fu, l3d, fd);
            std::printf("STEREO %d", c);
            for (int i = 0; i < nu; ++i) std::printf(" %d %.17g %.17g %.17g", fu[i], l3u[i].x, l3u[i].y, l3u[i].z);
            std::printf("\nSTEREO_DOWN");
            for (int i = 0; i < nd; ++i) std::printf(" %d", fd[i]);
            std::printf("\n");
        } else if (cmd == "homo") {
            int n; double thr; std::cin >> n >> thr;
            std::vector<Vec2> s(n), d(n);
            for (int i = 0; i < n; ++i) std::cin >> s[i].x >> s[i].y >> d[i].x >> d[i].y;
            std::vector<uint8_t> mask; double H[9] = {0};
            const bool ok = find_homography_ransac(s, d, thr, mask, H);
            std::printf("HOMO %d", ok ? 1 : 0);
            for (int i = 0; i < n; ++i) std::printf(" %d", mask[i]);
            std::printf("\nHOMO_H");
            for (double h : H) std::printf(" %.17g", h);
            std::printf("\n");
        } else if (cmd == "pnp") {
            int n, iters; double thr, conf; std::cin >> n >> iters >> thr >> conf;
            std::vector<Vec3> X(n); std::vector<Vec2> u(n);
            for (int i = 0; i < n; ++i) std::cin >> X[i].x >> X[i].y >> X[i].z >> u[i].x >> u[i].y;
            Rt rt; std::vector<int> inl;
            const bool ok = solve_pnp_ransac(X, u, iters, thr, conf, rt, inl);
            std::printf("PNP %d %zu", ok ? 1 : 0, inl.size());
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) std::printf(" %.17g", rt.R.m[r][c]);
            std::printf(" %.17g %.17g %.17g\n", rt.t.x, rt.t.y, rt.t.z);
        } else if (cmd == "epnp") {                      // the minimal solver alone: n points (X, u), all of them used
            int n; std::cin >> n;
            std::vector<Vec3> X(n); std::vector<Vec2> u(n); std::vector<int> idx(n);
            for (int i = 0; i < n; ++i) { std::cin >> X[i].x >> X[i].y >> X[i].z >> u[i].x >> u[i].y; idx[i] = i; }
            Rt rt;
            const bool ok = epnp(X, u, idx.data(), n, rt);
            std::printf("EPNP %d", ok ? 1 : 0);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) std::printf(" %.17g", rt.R.m[r][c]);
            std::printf(" %.17g %.17g %.17g\n", rt.t.x, rt.t.y, rt.t.z);
        } else if (cmd == "rng") {
            int n; std::cin >> n;
            CvRng r;
            std::printf("RNG");
            for (int i = 0; i < n; ++i) std::printf(" %d", r.uniform(0, 1000));
            std::printf("\n");
        } else if (cmd == "loop") {
            int dn, dold, init_mode, is4;
            std::cin >> dn >> dold >> init_mode >> is4;
            FisheyeFrameDescriptor nw = read_frame(), old = read_frame();
            LoopGeometry g;
            g.is_4dof = is4 != 0; g.self_id = old.drone_id;
            g.match = [](const float* q, int nq, const float* t, int nt, int dim, std::vector<DMatch>& out) {
                out.clear();
                if (nq <= 0 || nt <= 0) return;
                std::vector<int> qi(nq), ti(nq); std::vector<float> dd(nq);
                const int n = oracle_bf_match(q, nq, t, nt, dim, 0, qi.data(), ti.data(), dd.data());
                for (int i = 0; i < n; ++i) out.push_back({qi[i], ti[i], dd[i]});
            };
            LoopEdge e; LoopGeometry::Correspondence c;
            const bool ok = g.compute_loop(nw, old, dn, dold, e, init_mode != 0, &c);
            std::printf("LOOP %d %zu %d %lld %lld %d %d ", ok ? 1 : 0, c.new_norm_2d.size(), e.pnp_inlier_num, (long long)e.keyframe_id_a, (long long)e.keyframe_id_b,
                        e.drone_id_a, e.drone_id_b);
            print_pose(e.relative_pose);
            std::printf("\n");
        } else {
            std::fprintf(stderr, "unknown command %s\n", cmd.c_str());
            return 2;
        }
    }
    return 0;
}
