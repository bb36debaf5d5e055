// reg_kernels.hpp -- gfx950 device code of the registration hot path.
//
// Replaces, on the device (paths relative to the reference tree):
//   R3  RigidTransformation::inPlaceCompute      libpointmatcher/pointmatcher/TransformationsImpl.cpp:60-102
//   R4  KDTreeMatcher::findClosests (libnabo knn) MatchersImpl.cpp:86-101
//   R5  Trimmed / SurfaceNormal / MaxDist filters OutlierFiltersImpl.cpp:74-81,139-147,235-288, Matches.cpp:60-87
//   R6  ErrorElements compaction                  ErrorMinimizer.cpp:59-193   (implicit: masked lanes)
//   R7  ICP::calculateOptimizationHessian         ICP.cpp:1512-1566
// Numeric contract: every fp32 expression is evaluated in the written order with one rounding per
// operation (this translation unit is compiled with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace o3dreg {

constexpr int kBrickLog2 = 3;                 // 8x8x8 voxel bins per brick
constexpr int kBrickDim = 1 << kBrickLog2;
constexpr int kBrickCells = kBrickDim * kBrickDim * kBrickDim;  // 512
constexpr int kBrickBits = 18;                // brick coordinate bits per axis in the sort key
constexpr int kMaxLevels = 16;
constexpr int kSums = 32;                     // reduction payload (doubles)
constexpr uint64_t kEmptyKey = ~0ull;

struct HashEntry {
    uint64_t key;
    uint32_t val;
    uint32_t pad;
};

// Voxel-bin search structure ("brick table"): target points sorted by (brick, bin-in-brick);
// bricks located through an open-addressing hash, bins through a dense per-brick start table.
struct Grid {
    float ox, oy, oz;      // origin of bin (0,0,0) (centred target frame)
    float inv_c;           // 1 / bin edge
    float dimx, dimy, dimz;  // number of bins per axis, as float (for clamping before int conversion)
    uint32_t hash_mask;
    const HashEntry* hash;
    const uint32_t* cell_start;  // [n_bricks*512 + 1], monotone; bin r holds sorted points [cell_start[r], cell_start[r+1])
    const float4* pts;           // sorted {x, y, z, bits(original index)}
    int n_levels;
    float rho[kMaxLevels];       // search radii; the last one is max_dist (or +inf)
    float rho_box[kMaxLevels];   // radius used for the bin box (rho + safety margin)
    float max_d2;                // fl(max_dist*max_dist) (+inf allowed)
};

struct Xf {  // row-major 3x4
    float m[12];
};

__device__ __forceinline__ float3 xf_point(const Xf& T, float x, float y, float z) {
    float3 r;
    float a, b, s;
    a = T.m[0] * x; b = T.m[1] * y; s = a + b; a = T.m[2] * z; s = s + a; r.x = s + T.m[3];
    a = T.m[4] * x; b = T.m[5] * y; s = a + b; a = T.m[6] * z; s = s + a; r.y = s + T.m[7];
    a = T.m[8] * x; b = T.m[9] * y; s = a + b; a = T.m[10] * z; s = s + a; r.z = s + T.m[11];
    return r;
}
__device__ __forceinline__ float3 xf_rot(const Xf& T, float x, float y, float z) {
    float3 r;
    float a, b, s;
    a = T.m[0] * x; b = T.m[1] * y; s = a + b; a = T.m[2] * z; r.x = s + a;
    a = T.m[4] * x; b = T.m[5] * y; s = a + b; a = T.m[6] * z; r.y = s + a;
    a = T.m[8] * x; b = T.m[9] * y; s = a + b; a = T.m[10] * z; r.z = s + a;
    return r;
}

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

__host__ __device__ __forceinline__ uint64_t brick_key(uint32_t bx, uint32_t by, uint32_t bz) {
    return ((uint64_t)bz << (2 * kBrickBits)) | ((uint64_t)by << kBrickBits) | (uint64_t)bx;
}

// bin coordinate of a (centred) coordinate: monotone non-decreasing in v (NC: fl(fl(v-o)*inv_c), floor).
__device__ __forceinline__ float bin_coord_f(float v, float o, float inv_c) {
    float d = v - o;
    float s = d * inv_c;
    return floorf(s);
}

__device__ __forceinline__ int find_brick(const Grid& g, uint64_t bk) {
    uint32_t h = (uint32_t)mix64(bk) & g.hash_mask;
    for (;;) {
        const HashEntry e = g.hash[h];
        if (e.key == bk) return (int)e.val;
        if (e.key == kEmptyKey) return -1;
        h = (h + 1) & g.hash_mask;
    }
}

struct Best {
    float d2;
    uint32_t idx;  // original target index (tie-break: lowest wins)
    int pos;       // position in the sorted arrays
};

// Scan every target point in bins [lo, hi] (inclusive, already clamped to the grid).
__device__ __forceinline__ void scan_box(const Grid& g, float3 p, int lox, int loy, int loz, int hix, int hiy,
                                         int hiz, Best& best) {
    for (int bz = loz >> kBrickLog2; bz <= (hiz >> kBrickLog2); ++bz)
        for (int by = loy >> kBrickLog2; by <= (hiy >> kBrickLog2); ++by)
            for (int bx = lox >> kBrickLog2; bx <= (hix >> kBrickLog2); ++bx) {
                const int bid = find_brick(g, brick_key((uint32_t)bx, (uint32_t)by, (uint32_t)bz));
                if (bid < 0) continue;
                const int x0 = max(lox, bx << kBrickLog2) & (kBrickDim - 1);
                const int x1 = min(hix, (bx << kBrickLog2) + kBrickDim - 1) & (kBrickDim - 1);
                const int y0 = max(loy, by << kBrickLog2) & (kBrickDim - 1);
                const int y1 = min(hiy, (by << kBrickLog2) + kBrickDim - 1) & (kBrickDim - 1);
                const int z0 = max(loz, bz << kBrickLog2) & (kBrickDim - 1);
                const int z1 = min(hiz, (bz << kBrickLog2) + kBrickDim - 1) & (kBrickDim - 1);
                const uint32_t* cs = g.cell_start + (size_t)bid * kBrickCells;
                for (int lz = z0; lz <= z1; ++lz)
                    for (int ly = y0; ly <= y1; ++ly) {
                        const int row = (lz << (2 * kBrickLog2)) | (ly << kBrickLog2);
                        const uint32_t s = cs[row + x0];
                        const uint32_t e = cs[row + x1 + 1];
                        for (uint32_t j = s; j < e; ++j) {
                            const float4 t = g.pts[j];
                            const float dx = p.x - t.x, dy = p.y - t.y, dz = p.z - t.z;
                            float a = dx * dx;
                            float b = dy * dy;
                            float d2 = a + b;
                            a = dz * dz;
                            d2 = d2 + a;
                            const uint32_t idx = __float_as_uint(t.w);
                            if (d2 <= g.max_d2 && (d2 < best.d2 || (d2 == best.d2 && idx < best.idx))) {
                                best.d2 = d2;
                                best.idx = idx;
                                best.pos = (int)j;
                            }
                        }
                    }
            }
}

// Exact nearest neighbour of p within max_dist (lowest original index on ties).
__device__ __forceinline__ Best nearest(const Grid& g, float3 p) {
    Best best;
    best.d2 = INFINITY;
    best.idx = 0xffffffffu;
    best.pos = -1;
    for (int l = 0; l < g.n_levels; ++l) {
        const float rb = g.rho_box[l];
        // clamp in float before converting: handles +-inf radii and far-away queries
        const float fx0 = fminf(fmaxf(bin_coord_f(p.x - rb, g.ox, g.inv_c), 0.f), g.dimx - 1.f);
        const float fy0 = fminf(fmaxf(bin_coord_f(p.y - rb, g.oy, g.inv_c), 0.f), g.dimy - 1.f);
        const float fz0 = fminf(fmaxf(bin_coord_f(p.z - rb, g.oz, g.inv_c), 0.f), g.dimz - 1.f);
        const float fx1 = fminf(fmaxf(bin_coord_f(p.x + rb, g.ox, g.inv_c), 0.f), g.dimx - 1.f);
        const float fy1 = fminf(fmaxf(bin_coord_f(p.y + rb, g.oy, g.inv_c), 0.f), g.dimy - 1.f);
        const float fz1 = fminf(fmaxf(bin_coord_f(p.z + rb, g.oz, g.inv_c), 0.f), g.dimz - 1.f);
        scan_box(g, p, (int)fx0, (int)fy0, (int)fz0, (int)fx1, (int)fy1, (int)fz1, best);
        const float r = g.rho[l];
        float r2 = r * r;
        if (best.pos >= 0 && best.d2 <= r2) break;  // every point within rho was inside the box: exact
    }
    return best;
}

}  // namespace o3dreg
