// Minimal C++ user of the drop-in mirror (include/o3dslam_icp.hpp): registers a noisy, displaced copy of a
// synthetic "two walls + floor" cloud against the original, the way o3d_slam::Mapper drives its icp_ member
// (initReference once, compute per scan).  Build: make -C examples ; run on an MI355X: ./examples/icp_demo
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "../include/o3dslam_icp.hpp"

int main() {
    std::mt19937 rng(42);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    std::normal_distribution<float> noise(0.f, 0.005f);
    std::vector<float> ref, refn, rd, rdn;
    const int M = 60000, N = 8000;
    auto sample = [&](std::vector<float>& p, std::vector<float>& n) {
        const int f = (int)(u(rng) * 3.f);
        float x = 0, y = 0, z = 0, nx = 0, ny = 0, nz = 0;
        if (f == 0) { x = 10 * u(rng); y = 8 * u(rng); z = 0; nz = 1; }
        else if (f == 1) { x = 10 * u(rng); y = 0; z = 3 * u(rng); ny = 1; }
        else { x = 0; y = 8 * u(rng); z = 3 * u(rng); nx = 1; }
        p.insert(p.end(), {x + noise(rng), y + noise(rng), z + noise(rng), 1.f});
        n.insert(n.end(), {nx, ny, nz});
    };
    for (int i = 0; i < M; ++i) sample(ref, refn);
    for (int i = 0; i < N; ++i) sample(rd, rdn);
    // displace the reading by the inverse of (yaw 1 deg, t = (0.05, -0.03, 0.02))
    const float a = 1.0f * 3.14159265f / 180.f, c = std::cos(a), s = std::sin(a);
    for (int i = 0; i < N; ++i) {
        const float x = rd[4 * i] - 0.05f, y = rd[4 * i + 1] + 0.03f, z = rd[4 * i + 2] - 0.02f;
        rd[4 * i] = c * x + s * y;
        rd[4 * i + 1] = -s * x + c * y;
        rd[4 * i + 2] = z;
        const float nx = rdn[3 * i], ny = rdn[3 * i + 1];
        rdn[3 * i] = c * nx + s * ny;
        rdn[3 * i + 1] = -s * nx + c * ny;
    }
    try {
        o3dreg::ICP icp;
        icp.setShippedChain();
        o3dreg::DataPointsView reference{ref.data(), 4, M, refn.data(), 3};
        o3dreg::DataPointsView reading{rd.data(), 4, N, rdn.data(), 3};
        // SurfaceNormalDataPointsFilter on the device: the estimated reference normals must agree with the analytic
        // ones (up to sign) away from the wall/floor edges
        o3dreg::SurfaceNormalFilter sn;
        sn.knn = 10;
        sn.maxDist = 0.5f;
        std::vector<float> est(3 * (size_t)M);
        sn.compute(reference, est.data());
        int agree = 0;
        for (int i = 0; i < M; ++i) {
            const float d = est[3 * i] * refn[3 * i] + est[3 * i + 1] * refn[3 * i + 1] + est[3 * i + 2] * refn[3 * i + 2];
            if (std::fabs(d) > 0.95f) ++agree;
        }
        std::printf("estimated normals agree with the analytic ones on %.1f %% of the reference\n", 100.0 * agree / M);
        if (agree < 0.9 * M) return 4;
        if (!icp.initReference(reference)) return 2;
        const auto T = icp.compute(reading, reference, o3dreg::identity4(), false);
        const auto& r = icp.lastResult();
        std::printf("iterations %d converged %d inliers %lld  yaw %.4f deg  t = (%.4f, %.4f, %.4f)\n", r.iterations,
                    r.converged, (long long)r.n_inliers, std::atan2(T[1], T[0]) * 180.0 / 3.14159265, T[12], T[13], T[14]);
        const bool ok = std::fabs(std::atan2(T[1], T[0]) - a) < 2e-3 && std::fabs(T[12] - 0.05f) < 5e-3 &&
                        std::fabs(T[13] + 0.03f) < 5e-3 && std::fabs(T[14] - 0.02f) < 5e-3;
        // the same scan as Open3D holds it (fp64), cast on the device: the same transform
        std::vector<double> rd64(3 * (size_t)N), rdn64(3 * (size_t)N);
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < 3; ++k) {
                rd64[3 * i + k] = rd[4 * i + k];
                rdn64[3 * i + k] = rdn[3 * i + k];
            }
        const auto T64 = icp.computeF64(rd64.data(), rdn64.data(), N, false, o3dreg::identity4());
        bool same64 = true;
        for (int k = 0; k < 16; ++k) same64 = same64 && T64[k] == T[k];
        // the multi-GPU loop (C++ steering + RCCL collectives) with a group of ONE rank: the same transform again
        icp.joinGroup(o3dreg::ICP::makeGroupId(), 0, 1);
        const auto Tp = icp.computePartitioned(reading, o3dreg::identity4());
        icp.leaveGroup();
        float dmax = 0.f;
        for (int k = 0; k < 16; ++k) dmax = std::fmax(dmax, std::fabs(Tp[k] - T[k]));
        std::printf("fp64 reading: %s; partitioned loop (1 rank, RCCL): max |dT| = %.2e\n", same64 ? "identical" : "DIFFERENT", dmax);
        const bool ok2 = same64 && dmax < 1e-5f;
        std::printf(ok && ok2 ? "OK\n" : "MISMATCH\n");
        return ok && ok2 ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
}
