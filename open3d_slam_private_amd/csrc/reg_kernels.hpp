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
    // Level-0 accelerator ("halo bins"): a dense grid of bins of edge c_h; bin B lists every reference point
    // inside B's box grown by rho_h on each side, contiguously, as {x, y, z, bits(sorted position)}.  A query
    // in bin B therefore finds every point within rho_h in ONE contiguous run (no neighbour-bin lookups).
    int use_halo;
    float hox, hoy, hoz, hinv_c;
    int hdimx, hdimy, hdimz;
    const uint32_t* halo_start;  // [hdimx*hdimy*hdimz + 1]
    const float4* halo_pts;
    float rho_h;                 // exactness radius of the halo level
    int level_after_halo;        // first regular level with rho > rho_h
    // Dense brick directory (brick id or -1 per brick coordinate): replaces the hash probe when the brick grid is
    // small enough to be stored densely (null otherwise).
    const int32_t* brick_dir;
    int bdx, bdy, bdz;
    // Row occupancy per brick coordinate (same index as brick_dir; null without it): bit (z_local * 8 + y_local) is set
    // when that x-row of 8 bins holds at least one point.  A bin box of a large radius is mostly empty rows (a surface
    // crosses ~n of the n^2 rows of a brick): the wide level scan enumerates only the set bits.
    const unsigned long long* brick_rows;
    int dyn_prune;   // 1: the ball of a level scan shrinks with the group's running best (O3D_NO_DYNPRUNE: A/B)
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

// brick id of brick coordinate (bx, by, bz) (inside the grid), or -1
__device__ __forceinline__ int brick_lookup(const Grid& g, int bx, int by, int bz) {
    if (g.brick_dir) return g.brick_dir[((size_t)bz * g.bdy + by) * g.bdx + bx];
    return find_brick(g, brick_key((uint32_t)bx, (uint32_t)by, (uint32_t)bz));
}

struct Best {
    float d2;
    uint32_t idx;  // original target index (tie-break: lowest wins)
    int pos;       // position in the sorted arrays
    float second;  // smallest d2 among the scanned points at a DIFFERENT position (group search only; +inf: none)
    // kTop2 searches additionally keep WHICH point the runner-up is and the distance of the best point that is neither:
    int pos2;      // position of the runner-up (-1: none)
    float third;   // smallest d2 among the scanned points that are neither the winner nor the runner-up (+inf: none)
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
    best.second = INFINITY;
    best.pos2 = -1;
    best.third = INFINITY;
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


// -------------------------------------------------------------------------------------------------
// Cooperative search: kGroup (= 8) adjacent lanes share one query.  Work items of a box are spread
// over the lanes of the group (level 0: the <= 2x2x2 bins, one bin per lane; wider boxes: one
// (brick, y, z) row segment per lane), every lane scans its own contiguous run of sorted points,
// and the group's best candidate is combined with three xor-shuffles.  All lanes of a group take the
// same control flow, so divergence is limited to the 8 groups of a wave.
// -------------------------------------------------------------------------------------------------
constexpr int kGroup = 8;   // default group width (lanes per reading point); kernels are templated on it

#ifndef O3D_SEARCH_STATS
#define O3D_SEARCH_STATS 0   // diagnostic A/B builds only (tools/build_ab.sh -DO3D_SEARCH_STATS=1, tools/tools_search_stats.py)
#endif
#if O3D_SEARCH_STATS
// [0] level scans [1] bricks of their boxes [2] non-empty rows inside the boxes [3] rows kept by the ball [4] candidates
// looked at [5] batches (phase 1 + compaction + flattened scan) [6] halo candidates [7] searches
__device__ unsigned long long g_search_stats[64];   // [8..23] level scans by log2(candidates + 1), [24..39] by log2(rows kept + 1), [40..55] by log2(bricks + 1)
#define SEARCH_STAT(i, v) atomicAdd(&g_search_stats[i], (unsigned long long)(v))
// wave time by section of the search ([56 + i], s_memtime ticks of 10 ns; whoever is executing stamps: divergent groups of a wave
// take their turns, each turn is charged to the section it runs): 0 halo run, 1 level box + brick directory, 2 row slots
// (phase 1), 3 compaction, 4 flattened candidate scan, 5 group minimum + level logic, 6 ball update
__device__ __forceinline__ void search_sec(int i) {
    __shared__ unsigned long long last_w[16];
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    const int w = (int)(threadIdx.x >> 6);
    if ((int)(threadIdx.x & 63) == __ffsll((unsigned long long)__ballot(1)) - 1) {
        if (i >= 0) atomicAdd(&g_search_stats[56 + i], now - last_w[w]);
        last_w[w] = now;
    }
}
#define SEARCH_SEC(i) search_sec(i)
#else
#define SEARCH_STAT(i, v)
#define SEARCH_SEC(i)
#endif

template <bool kTop2 = false>
__device__ __forceinline__ void consider(const Grid& g, float3 p, const float4 t, int j, Best& best) {
    const float dx = p.x - t.x, dy = p.y - t.y, dz = p.z - t.z;
    float a = dx * dx;
    float b = dy * dy;
    float d2 = a + b;
    a = dz * dz;
    d2 = d2 + a;
    const uint32_t idx = __float_as_uint(t.w);
    if (d2 <= g.max_d2 && j != best.pos && (!kTop2 || j != best.pos2)) {   // (the clamped tail of a flattened scan re-reads its last point)
        if (d2 < best.d2 || (d2 == best.d2 && idx < best.idx)) {
            if (kTop2) {
                best.third = best.second;
                best.pos2 = best.pos;
            }
            best.second = best.d2;           // the previous best becomes the runner-up
            best.d2 = d2;
            best.idx = idx;
            best.pos = j;
        } else if (kTop2) {
            if (d2 < best.second) {
                best.third = best.second;
                best.second = d2;
                best.pos2 = j;
            } else {
                best.third = fminf(best.third, d2);
            }
        } else {
            best.second = fminf(best.second, d2);
        }
    }
}

__device__ __forceinline__ void scan_run(const Grid& g, float3 p, uint32_t s, uint32_t e, Best& best) {
    // two loads in flight per step; the clamped duplicate of the last point is harmless
    for (uint32_t j = s; j < e; j += 2) {
        const uint32_t j1 = min(j + 1, e - 1);
        const float4 t0 = g.pts[j];
        const float4 t1 = g.pts[j1];
        consider(g, p, t0, (int)j, best);
        consider(g, p, t1, (int)j1, best);
    }
}

// Runner-up of the group: every lane offers the smallest d2 it has seen at a position other than the winner's.
template <int G>
__device__ __forceinline__ float group_second(const Best& mine, int winner_pos) {
    float c = (mine.pos == winner_pos) ? mine.second : fminf(mine.d2, mine.second);
#pragma unroll
    for (int m = 1; m < G; m <<= 1) c = fminf(c, __shfl_xor(c, m));
    return c;
}

// kTop2: runner-up (distance AND position) and the best of the rest, from the lanes' own {winner, runner-up, third}.
// `mine` = this lane's state before the merge, r.pos = the group's winner.
template <int G>
__device__ __forceinline__ void group_top2(const Best& mine, Best& r, int gbase) {
    // every lane offers its best point other than the group's winner
    const bool own = mine.pos == r.pos;
    const float cd = own ? mine.second : mine.d2;
    const int cp = own ? mine.pos2 : mine.pos;
    float m2 = cd;
#pragma unroll
    for (int m = 1; m < G; m <<= 1) m2 = fminf(m2, __shfl_xor(m2, m));
    const unsigned who = (unsigned)((__ballot(cd == m2 && cp >= 0) >> gbase) & ((1ull << G) - 1ull));
    const int p2 = who ? __shfl(cp, gbase + __ffs((int)who) - 1) : -1;
    // ... and its best point that is neither the winner nor the group's runner-up
    float t = mine.third;
    if (mine.pos >= 0 && mine.pos != r.pos && mine.pos != p2) t = fminf(t, mine.d2);
    if (mine.pos2 >= 0 && mine.pos2 != r.pos && mine.pos2 != p2) t = fminf(t, mine.second);
#pragma unroll
    for (int m = 1; m < G; m <<= 1) t = fminf(t, __shfl_xor(t, m));
    r.second = p2 >= 0 ? m2 : INFINITY;
    r.pos2 = p2;
    r.third = t;
}

template <int G, bool kTop2 = false>
__device__ __forceinline__ Best group_min(Best b) {
    const Best mine = b;
#pragma unroll
    for (int m = 1; m < G; m <<= 1) {
        const float od2 = __shfl_xor(b.d2, m);
        const uint32_t oidx = __shfl_xor(b.idx, m);
        const int opos = __shfl_xor(b.pos, m);
        if (od2 < b.d2 || (od2 == b.d2 && oidx < b.idx)) {
            b.d2 = od2;
            b.idx = oidx;
            b.pos = opos;
        }
    }
    if (kTop2)
        group_top2<G>(mine, b, (int)(threadIdx.x & 63) & ~(G - 1));
    else
        b.second = group_second<G>(mine, b.pos);
    return b;
}

// Candidate from a halo record {x, y, z, bits(sorted position)}.  The original index (tie-break) is only
// fetched when two candidates are exactly equidistant.
template <bool kTop2 = false>
__device__ __forceinline__ void consider_pos(const Grid& g, float3 p, const float4 t, Best& best) {
    const float dx = p.x - t.x, dy = p.y - t.y, dz = p.z - t.z;
    float a = dx * dx;
    float b = dy * dy;
    float d2 = a + b;
    a = dz * dz;
    d2 = d2 + a;
    const int pos = (int)__float_as_uint(t.w);
    if (d2 <= g.max_d2 && pos != best.pos && (!kTop2 || pos != best.pos2)) {
        if (d2 < best.d2) {
            if (kTop2) {
                best.third = best.second;
                best.pos2 = best.pos;
            }
            best.second = best.d2;
            best.d2 = d2;
            best.pos = pos;
            best.idx = 0xffffffffu;  // not fetched
        } else if (d2 == best.d2) {
            // an exact tie: the runner-up is as close as the winner
            if (best.idx == 0xffffffffu) best.idx = __float_as_uint(g.pts[best.pos].w);
            const uint32_t idx = __float_as_uint(g.pts[pos].w);
            if (kTop2) best.third = best.second;
            best.second = d2;
            if (idx < best.idx) {
                if (kTop2) best.pos2 = best.pos;
                best.idx = idx;
                best.pos = pos;
            } else if (kTop2) {
                best.pos2 = pos;
            }
        } else if (kTop2) {
            if (d2 < best.second) {
                best.third = best.second;
                best.second = d2;
                best.pos2 = pos;
            } else {
                best.third = fminf(best.third, d2);
            }
        } else {
            best.second = fminf(best.second, d2);
        }
    }
}

// Group minimum of (d2, original index) where idx may not have been fetched yet (0xffffffff).
template <int G, bool kTop2 = false>
__device__ __forceinline__ Best group_min_lazy(const Grid& g, Best b, int gbase) {
    float m = b.d2;
#pragma unroll
    for (int k = 1; k < G; k <<= 1) m = fminf(m, __shfl_xor(m, k));
    const bool cand = b.pos >= 0 && b.d2 == m;
    const unsigned mask = (unsigned)((__ballot(cand) >> gbase) & ((1ull << G) - 1ull));
    Best r;
    r.d2 = INFINITY;
    r.idx = 0xffffffffu;
    r.pos = -1;
    r.second = INFINITY;
    r.pos2 = -1;
    r.third = INFINITY;
    if (mask == 0) return r;
    if ((mask & (mask - 1)) == 0) {  // exactly one lane holds the minimum
        r.d2 = m;
        r.pos = __shfl(b.pos, gbase + __ffs((int)mask) - 1);
        r.idx = (uint32_t)__shfl((int)b.idx, gbase + __ffs((int)mask) - 1);
        if (kTop2)
            group_top2<G>(b, r, gbase);
        else
            r.second = group_second<G>(b, r.pos);
        return r;
    }
    // tie between lanes: lowest original index wins
    uint32_t idx = 0xffffffffu;
    if (cand) idx = b.idx != 0xffffffffu ? b.idx : __float_as_uint(g.pts[b.pos].w);
    uint32_t mi = idx;
#pragma unroll
    for (int k = 1; k < G; k <<= 1) mi = min(mi, (uint32_t)__shfl_xor((int)mi, k));
    const unsigned win = (unsigned)((__ballot(cand && idx == mi) >> gbase) & ((1ull << G) - 1ull));
    r.d2 = m;
    r.idx = mi;
    r.pos = __shfl(b.pos, gbase + __ffs((int)win) - 1);
    if (kTop2)
        group_top2<G>(b, r, gbase);
    else
        r.second = group_second<G>(b, r.pos);
    return r;
}

// Wide level scan: the same bin box as below, but every lane looks up kSegPerLane row segments at once (the brick
// directory loads, then the bin-start loads, are all in flight together) and the group scans the concatenation of up
// to G * kSegPerLane segments in one flattened pass.  The compacted segment list {first point, exclusive offset}
// lives in LDS (`seg`: 2 * G * kSegPerLane + 2 words owned by this group); each lane walks it monotonically.
// Large radii (first iterations of a registration) are bound by dependent-load rounds: this cuts them roughly in half.
#ifndef O3D_SEG_PER_LANE
#define O3D_SEG_PER_LANE 4
#endif
#ifndef O3D_SCAN_UNROLL
#define O3D_SCAN_UNROLL 4
#endif
#ifndef O3D_HALO_UNROLL
#define O3D_HALO_UNROLL 4
#endif
constexpr int kSegPerLane = O3D_SEG_PER_LANE;

// a / b and a % b for 0 <= a < 2^20, 1 <= b <= a + 1: fp32 reciprocal estimate (off by at most one), corrected exactly
__device__ __forceinline__ void fast_divmod(int a, int b, int& q, int& r) {
    q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
    r = a - q * b;
    if (r < 0) {
        --q;
        r += b;
    } else if (r >= b) {
        ++q;
        r -= b;
    }
}

// index of the k-th (0-based) set bit of the 64-bit mask {lo, hi}; k < popcount
__device__ __forceinline__ int nth_set_bit64(uint32_t lo, uint32_t hi, int k) {
    int c = __popc(lo), base = 0;
    uint32_t w = lo;
    if (k >= c) { k -= c; w = hi; base = 32; }
    c = __popc(w & 0xffffu);
    if (k >= c) { k -= c; w >>= 16; base += 16; }
    c = __popc(w & 0xffu);
    if (k >= c) { k -= c; w >>= 8; base += 8; }
    c = __popc(w & 0xfu);
    if (k >= c) { k -= c; w >>= 4; base += 4; }
    c = __popc(w & 0x3u);
    if (k >= c) { k -= c; w >>= 2; base += 2; }
    if (k >= (int)(w & 1u)) base += 1;
    return base;
}

// Wide level scan, row-mask driven.  A box is cut into row segments (one per (brick, y, z): a contiguous run of sorted
// points).  The bricks overlapping the bin box are taken G at a time (one brick per lane: directory entry and
// row-occupancy mask arrive in ONE round trip); the set bits of the masks inside the box's (y, z) range are the
// non-empty rows -- only those get a slot in the segment batches: every lane looks up kSegPerLane segments at once (bin
// starts, independent loads), the non-empty ones are compacted into a per-group LDS list {first point, exclusive offset}
// (`seg`: 2 * G * kSegPerLane + 2 words), and the group scans their concatenation in one flattened pass, each lane
// walking the list monotonically.  A radius of 8 bins used to cost 17 x 17 x 3 = 867 row segments = 27 batches
// of three dependent round trips each; a surface crossing the box leaves ~50 non-empty rows = 2 batches.
// The scanned SET is unchanged (every point of every non-empty row inside the pruned box), hence the same exactness.
// Returns the radius up to which every reference point is guaranteed to have been scanned (the level's rho, or -- when
// the box was shrunk to a known candidate -- that candidate's distance plus `slack`).
template <int G, bool kPrune, bool kTop2 = false>
__device__ __forceinline__ float scan_level_rows(const Grid& g, const float3 p, int sub, int gbase, int l,
                                                 uint32_t* seg, Best& best, float slack) {
    constexpr int S = kSegPerLane, CAP = G * S;
    uint32_t* const seg_st = seg;         // [CAP]
    uint32_t* const seg_ex = seg + CAP;   // [CAP + 1]
    // Radius of the box: the level's rho_box, or -- when a candidate is already known (halo run, previous level) -- the
    // candidate's own distance (+ slack) grown by the same margins: no point farther than the candidate can win, and
    // every point at most as far (ties included) stays inside the box.
    float rb = g.rho_box[l];
    float cover = g.rho[l];
    if (best.pos >= 0 && g.rho[l] < INFINITY) {
        const float dc = __builtin_amdgcn_sqrtf(best.d2) + slack;
        const float rs = dc * 1.001f + (rb - g.rho[l]);
        if (rs < rb) {
            rb = rs;
            cover = dc;
        }
    }
    const int lox = (int)fminf(fmaxf(bin_coord_f(p.x - rb, g.ox, g.inv_c), 0.f), g.dimx - 1.f);
    const int loy = (int)fminf(fmaxf(bin_coord_f(p.y - rb, g.oy, g.inv_c), 0.f), g.dimy - 1.f);
    const int loz = (int)fminf(fmaxf(bin_coord_f(p.z - rb, g.oz, g.inv_c), 0.f), g.dimz - 1.f);
    const int hix = (int)fminf(fmaxf(bin_coord_f(p.x + rb, g.ox, g.inv_c), 0.f), g.dimx - 1.f);
    const int hiy = (int)fminf(fmaxf(bin_coord_f(p.y + rb, g.oy, g.inv_c), 0.f), g.dimy - 1.f);
    const int hiz = (int)fminf(fmaxf(bin_coord_f(p.z + rb, g.oz, g.inv_c), 0.f), g.dimz - 1.f);
    const int bx0 = lox >> kBrickLog2, by0 = loy >> kBrickLog2, bz0 = loz >> kBrickLog2;
    const int nbx = (hix >> kBrickLog2) - bx0 + 1, nby = (hiy >> kBrickLog2) - by0 + 1,
              nbz = (hiz >> kBrickLog2) - bz0 + 1;
    const int nbxy = nbx * nby;
    const int n_bricks = nbxy * nbz;
    const bool pk_ok = nbx <= 1024 && nby <= 1024 && nbz <= 1024;
    const bool small_box = n_bricks < (1 << 20);
    const unsigned gmask = (1u << G) - 1u;
    // Ball pruning (exact): a bin box holds ~2x the volume of the ball it covers.  Rows whose (y, z) bin interval lies
    // farther than the box radius from the query are skipped and the others are cut to the x-range the ball reaches, in BIN
    // units with a slack of kPruneSlack bins per axis: the bin coordinate fl(fl(v - o) * 1/c) of a reference point and of
    // the query each carry a relative error of 2 * 2^-24, i.e. < 1e-3 bins below 8192 bins per axis -- the slack is twice
    // their sum (larger grids: no pruning).  Every reference point within rho_box of the query stays inside the scanned
    // set, which is all the termination test (best d2 <= rho^2 < rho_box^2) relies on.
    constexpr float kPruneSlack = 4e-3f;
    const bool prune = kPrune && g.dimx <= 8192 && g.dimy <= 8192 && g.dimz <= 8192;
#if O3D_SEARCH_STATS
    if (sub == 0) SEARCH_STAT(0, 1);
    unsigned stat_pts = 0, stat_rows = 0;
#endif
    const float fxq = (p.x - g.ox) * g.inv_c, fyq = (p.y - g.oy) * g.inv_c, fzq = (p.z - g.oz) * g.inv_c;
    float rbb = rb * g.inv_c + kPruneSlack;
    float rb2 = rbb * rbb;
    for (int cb = 0; cb < n_bricks; cb += G) {
        SEARCH_SEC(5);
        // ---- this lane's brick of the chunk: directory entry + row mask (one batch of independent loads)
        const int bi = cb + sub;
        int bid = -1;
        uint32_t m_lo = 0, m_hi = 0;
        uint32_t bpk = 0;   // this lane's brick inside the box, packed (ix | iy << 8 | iz << 16): the slots below fetch it from
                            // the owner lane by ONE shuffle instead of decoding the brick number with two integer divisions
                            // each (~45 of the ~140 VALU instructions a slot cost: the level scans of the first iterations are
                            // bound by this enumeration, not by their candidates -- profiles/r03_counters_c3_per_dispatch.txt)
        if (bi < n_bricks) {
            int iz, rem, iy, ix;
            if (small_box) {   // (group-uniform) quotient by a float reciprocal, corrected: two integer divisions cost ~70 instructions
                fast_divmod(bi, nbxy, iz, rem);
                fast_divmod(rem, nbx, iy, ix);
            } else {
                iz = bi / nbxy;
                rem = bi - iz * nbxy;
                iy = rem / nbx;
                ix = rem - iy * nbx;
            }
            bpk = (uint32_t)ix | ((uint32_t)iy << 10) | ((uint32_t)iz << 20);   // (boxes wider than 1024 bricks: decoded below)
            const int bx = bx0 + ix, by = by0 + iy, bz = bz0 + iz;
            const size_t at = ((size_t)bz * g.bdy + by) * g.bdx + bx;
            bid = g.brick_dir[at];
            const unsigned long long rows = g.brick_rows[at];
            // rows of this brick inside the box: y in [y0, y1], z in [z0, z1] (brick-local)
            const int y0 = max(loy, by << kBrickLog2) & (kBrickDim - 1), y1 = min(hiy, (by << kBrickLog2) + kBrickDim - 1) & (kBrickDim - 1);
            const int z0 = max(loz, bz << kBrickLog2) & (kBrickDim - 1), z1 = min(hiz, (bz << kBrickLog2) + kBrickDim - 1) & (kBrickDim - 1);
            const unsigned long long ybits = (unsigned long long)(((1u << (y1 - y0 + 1)) - 1u) << y0);
            const unsigned long long zhi = z1 == 7 ? ~0ull : ((1ull << (8 * (z1 + 1))) - 1ull);
            const unsigned long long zsel = zhi & ~((1ull << (8 * z0)) - 1ull) & 0x0101010101010101ull;
            const unsigned long long m = bid >= 0 ? (rows & (ybits * zsel)) : 0ull;
            m_lo = (uint32_t)m;
            m_hi = (uint32_t)(m >> 32);
        }
        // exclusive prefix of the row counts over the lanes of the group
        const uint32_t cnt = (uint32_t)(__popc(m_lo) + __popc(m_hi));
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
            if (sub >= o) incl += v;
        }
        const int total = (int)(uint32_t)__shfl((int)incl, gbase + G - 1);
#if O3D_SEARCH_STATS
        if (sub == 0) {
            SEARCH_STAT(1, min(G, n_bricks - cb));
            SEARCH_STAT(2, total);
        }
#endif
        if (total == 0) continue;
        const uint32_t excl = incl - cnt;
        SEARCH_SEC(1);
        for (int base = 0; base < total; base += CAP) {
            if (prune && g.dyn_prune && g.rho[l] < INFINITY) {
                // The ball shrinks with the best candidate the GROUP has seen so far (the box was sized before the level
                // started; rows come in brick order, not by distance): same rule as the candidate-bounded box above -- no
                // point farther than a known candidate can win, ties stay inside -- so the rows still to come are cut to
                // the ball of the current best.  The covered radius shrinks with it.
                float gb = best.d2;
#pragma unroll
                for (int o = G / 2; o >= 1; o >>= 1) gb = fminf(gb, __shfl_xor(gb, o));
                if (gb < INFINITY) {
                    const float dc = __builtin_amdgcn_sqrtf(gb) + slack;
                    const float rs = dc * 1.001f + (g.rho_box[l] - g.rho[l]);
                    if (rs < rb) {
                        rb = rs;
                        cover = fminf(cover, dc);
                        rbb = rb * g.inv_c + kPruneSlack;
                        rb2 = rbb * rbb;
                    }
                }
            }
            SEARCH_SEC(6);
            // phase 1: this lane's row segments of the batch -> bin starts (independent loads)
            uint32_t s[S], e[S];
#pragma unroll
            for (int u = 0; u < S; ++u) {
                s[u] = 0;
                e[u] = 0;
                // (most level scans keep fewer than G rows: slots no group of the wave has are skipped by the whole wave)
                if (u > 0 && __ballot(base + u * G < total) == 0ull) continue;
                const int t = base + u * G + sub;
                // owner lane of slot t (branch-free), then its mask / brick id by lane-indexed shuffles; lanes without a
                // slot take part in the shuffles with a clamped slot
                const int tc = min(t, total - 1);
                int j = 0;   // = number of lanes whose inclusive prefix is <= tc (binary search over the group, by shuffles)
#pragma unroll
                for (int step = G / 2; step >= 1; step >>= 1) {
                    const uint32_t v = (uint32_t)__shfl((int)incl, gbase + j + step - 1);
                    if ((uint32_t)tc >= v) j += step;
                }
                const uint32_t o_lo = (uint32_t)__shfl((int)m_lo, gbase + j), o_hi = (uint32_t)__shfl((int)m_hi, gbase + j);
                const int o_bid = __shfl(bid, gbase + j);
                const uint32_t o_ex = (uint32_t)__shfl((int)excl, gbase + j);
                const uint32_t o_pk = (uint32_t)__shfl((int)bpk, gbase + j);
                if (t < total) {
                    const int r = nth_set_bit64(o_lo, o_hi, (int)((uint32_t)tc - o_ex));   // row = z_local * 8 + y_local
                    int ix = (int)(o_pk & 1023u), iy = (int)((o_pk >> 10) & 1023u), iz = (int)(o_pk >> 20);
                    if (!pk_ok) {   // group-uniform; an unbounded search over a very large grid
                        const int bj = cb + j;
                        iz = bj / nbxy;
                        const int rem = bj - iz * nbxy;
                        iy = rem / nbx;
                        ix = rem - iy * nbx;
                    }
                    const int bx = bx0 + ix, cy = ((by0 + iy) << kBrickLog2) + (r & 7), cz = ((bz0 + iz) << kBrickLog2) + (r >> 3);
                    int gx0 = max(lox, bx << kBrickLog2), gx1 = min(hix, (bx << kBrickLog2) + kBrickDim - 1);
                    bool keep = true;
                    if (prune) {
                        const float dyb = fmaxf(fmaxf((float)cy - fyq, fyq - (float)(cy + 1)) - kPruneSlack, 0.f);
                        const float dzb = fmaxf(fmaxf((float)cz - fzq, fzq - (float)(cz + 1)) - kPruneSlack, 0.f);
                        const float r2 = rb2 - dyb * dyb - dzb * dzb;
                        keep = !(r2 < 0.f);   // else: the whole row lies outside the ball
                        if (keep) {
                            const float hxb = __builtin_amdgcn_sqrtf(r2) * 1.0001f + kPruneSlack;
                            gx0 = max(gx0, (int)fminf(fmaxf(floorf(fxq - hxb), 0.f), g.dimx - 1.f));
                            gx1 = min(gx1, (int)fminf(fmaxf(floorf(fxq + hxb), 0.f), g.dimx - 1.f));
                            keep = gx0 <= gx1;
                        }
                    }
                    if (keep) {
                        const uint32_t* cs = g.cell_start + (size_t)o_bid * kBrickCells + (r << kBrickLog2);
                        s[u] = cs[gx0 & (kBrickDim - 1)];
                        e[u] = cs[(gx1 & (kBrickDim - 1)) + 1];
                    }
                }
            }
            // compact the non-empty segments into the group's LDS list, in segment order
            SEARCH_SEC(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            uint32_t run_pts = 0, run_seg = 0;
#pragma unroll
            for (int u = 0; u < S; ++u) {
                const uint32_t c2 = e[u] - s[u];
                if (u > 0 && __ballot(c2 != 0u) == 0ull) continue;
                uint32_t in2 = c2;
#pragma unroll
                for (int o = 1; o < G; o <<= 1) {
                    const uint32_t v = (uint32_t)__shfl_up((int)in2, o);
                    if (sub >= o) in2 += v;
                }
                const uint32_t tot_u = (uint32_t)__shfl((int)in2, gbase + G - 1);
                const unsigned m = (unsigned)(__ballot(c2 != 0) >> gbase) & gmask;
                if (c2 != 0) {
                    const uint32_t my = run_seg + (uint32_t)__popc(m & ((1u << sub) - 1u));
                    seg_st[my] = s[u];
                    seg_ex[my] = run_pts + in2 - c2;
                }
                run_pts += tot_u;
                run_seg += (uint32_t)__popc(m);
            }
#if O3D_SEARCH_STATS
            if (sub == 0) {
                SEARCH_STAT(3, run_seg);
                SEARCH_STAT(4, run_pts);
                SEARCH_STAT(5, 1);
            }
            stat_pts += run_pts;
            stat_rows += run_seg;
#endif
            if (run_pts == 0) continue;
            if (sub == 0) seg_ex[run_seg] = run_pts;   // sentinel: end of the last segment
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            SEARCH_SEC(3);
            // phase 2: flattened scan, 4 independent 16-byte loads in flight per lane
            uint32_t k = 0, cur_ex = 0, cur_st = seg_st[0], next_ex = seg_ex[1];
            constexpr int kUnroll = O3D_SCAN_UNROLL;
            for (uint32_t f0 = 0; f0 < run_pts; f0 += kUnroll * G) {
                float4 tv[kUnroll];
                uint32_t jv[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const uint32_t f = min(f0 + (uint32_t)(u * G + sub), run_pts - 1);   // clamp: duplicates are harmless
                    while (f >= next_ex) {
                        ++k;
                        cur_ex = next_ex;
                        cur_st = seg_st[k];
                        next_ex = seg_ex[k + 1];
                    }
                    jv[u] = cur_st + (f - cur_ex);
                    tv[u] = g.pts[jv[u]];
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) consider<kTop2>(g, p, tv[u], (int)jv[u], best);
            }
            SEARCH_SEC(4);
        }
    }
    SEARCH_SEC(1);
#if O3D_SEARCH_STATS
    if (sub == 0) {
        SEARCH_STAT(8 + min(15, 32 - __clz((int)stat_pts)), 1);
        SEARCH_STAT(24 + min(15, 32 - __clz((int)stat_rows)), 1);
        SEARCH_STAT(40 + min(15, 32 - __clz(n_bricks)), 1);
    }
#endif
    return cover;
}

// `sub` = lane index inside the group (0..7); `first_level` < 0 tries the halo level first; `after_halo` is the regular
// level to continue at when the halo level cannot answer (the previous iteration's terminating level minus one: radii
// that were too small then are skipped; any starting level is exact).  Returns the level at
// which the search terminated through *level_out.
//
// A box is cut into row segments (one per (brick, y, z): a contiguous run of sorted points).  Phase 1:
// each lane of the group looks up one segment (brick hash probe + two bin-start loads).  Phase 2: the
// group scans every non-empty segment TOGETHER, lane k reading point s+k, s+k+8, ... -- consecutive
// 16-byte records, i.e. one or two cache lines per group step instead of one line per lane.
// kPrune: ball pruning of the wide level scan (search kernel of the select-based iterations, whose first launches scan
// large boxes); the fused kernel, whose queries almost always end in the halo level, does without (it would cost it
// registers beyond the 80 of 6 waves/SIMD).
// *cov2_out (may be null): squared radius around p inside which EVERY reference point has been looked at; together with
// Best::second this bounds how close any point other than the winner can be: d2(x) >= min(second, cov2) for all x != best
// (the temporal-coherence shortcut of the next iterations relies on it; `slack` widens candidate-bounded boxes a little so
// that the bound is not just the winner's own distance).
// Halo part of the search: one dense-grid lookup, one contiguous run.  Returns true when the halo level answered the query
// (level -1); otherwise `best` (the group's merged candidate of the halo run, if any, the same in every lane), `cov` and the regular
// level `l` to continue at are what nearest_levels() needs -- by the same lanes or, handed over through LDS, by another group
// (k_match_g8 regroups the unanswered queries of a workgroup so that whole waves are done after the halo part).
template <int G, bool kTop2>
__device__ __forceinline__ bool nearest_halo(const Grid& g, float3 p, int sub, int gbase, int first_level, int after_halo, Best& best,
                                             float& cov, int& l, int* level_out, float* cov2_out) {
    best.d2 = INFINITY;
    best.idx = 0xffffffffu;
    best.pos = -1;
    best.second = INFINITY;
    best.pos2 = -1;
    best.third = INFINITY;
    cov = 0.f;   // radius covered so far
#if O3D_SEARCH_STATS
    if (sub == 0) SEARCH_STAT(7, 1);
    SEARCH_SEC(-1);
#endif
    l = min(first_level, g.n_levels - 1);
    if (g.use_halo && first_level < 0) {
        // halo level: one dense-grid lookup, one contiguous run
        const float fx = bin_coord_f(p.x, g.hox, g.hinv_c), fy = bin_coord_f(p.y, g.hoy, g.hinv_c),
                    fz = bin_coord_f(p.z, g.hoz, g.hinv_c);
        const bool inside = fx >= 0.f && fy >= 0.f && fz >= 0.f && fx < (float)g.hdimx && fy < (float)g.hdimy &&
                            fz < (float)g.hdimz;
        l = 0;
        if (inside) {
            const size_t B = ((size_t)(int)fz * g.hdimy + (int)fy) * g.hdimx + (int)fx;
            const uint32_t s = g.halo_start[B], e = g.halo_start[B + 1];
#if O3D_SEARCH_STATS
            if (sub == 0) SEARCH_STAT(6, e - s);
#endif
            constexpr int kU = O3D_HALO_UNROLL;
            for (uint32_t j0 = s; j0 < e; j0 += kU * G) {
                float4 tv[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) tv[u] = g.halo_pts[min(j0 + (uint32_t)(u * G + sub), e - 1)];
#pragma unroll
                for (int u = 0; u < kU; ++u) consider_pos<kTop2>(g, p, tv[u], best);
            }
            best = group_min_lazy<G, kTop2>(g, best, gbase);
            SEARCH_SEC(0);
            const float rh = g.rho_h;
            const float rh2 = rh * rh;
            cov = rh;
            if (best.pos >= 0 && best.d2 <= rh2) {
                *level_out = -1;
                // (capped at max_dist: points beyond it are not candidates, hence not in `second` either -- found by the
                // round-2 fuzz: halo radius 0.24 m with maxDist 0.2 m let a point 0.22 m away go unnoticed by the bound)
                if (cov2_out) *cov2_out = fminf(rh2, g.max_d2);
                return true;
            }
            if (best.pos >= 0 && best.idx == 0xffffffffu) best.idx = __float_as_uint(g.pts[best.pos].w);
            if (best.pos >= 0) {
                // the halo run produced a candidate: the first radius that covers it terminates the search
                l = g.level_after_halo;
                while (l + 1 < g.n_levels && g.rho[l] * g.rho[l] < best.d2) ++l;
            } else {
                l = max(g.level_after_halo, after_halo);
            }
        }
    }
    l = max(max(l, after_halo), 0);   // no halo level (or the query lies outside its grid): the hinted level
    return false;
}

// Regular levels, from level `l` on, continuing from the state nearest_halo() left.
template <int G, bool kPrune, bool kTop2>
__device__ __forceinline__ Best nearest_levels(const Grid& g, float3 p, int sub, int gbase, int l, Best best, float cov, uint32_t* seg,
                                               float slack, int* level_out, float* cov2_out) {
    for (; l < g.n_levels; ++l) {
        cov = fmaxf(cov, scan_level_rows<G, kPrune, kTop2>(g, p, sub, gbase, l, seg, best, slack));
        best = group_min<G, kTop2>(best);
        const float rw = g.rho[l];
        if (best.pos >= 0 && best.d2 <= rw * rw) break;   // every point within rho was inside the box: exact
        if (best.pos >= 0) {
            // a candidate at distance sqrt(best.d2) exists, so the true neighbour is no farther: jump straight to the
            // first radius that covers it (the levels in between could not terminate either)
            while (l + 2 < g.n_levels) {
                const float rn = g.rho[l + 1];
                if (rn * rn >= best.d2) break;
                ++l;
            }
        }
    }
    SEARCH_SEC(5);
    *level_out = min(l, g.n_levels - 1);
    if (cov2_out) *cov2_out = fminf(cov * cov, g.max_d2);
    return best;
}

template <int G, bool kPrune = false, bool kTop2 = false>
__device__ __forceinline__ Best nearest_group(const Grid& g, float3 p, int sub, int first_level, int* level_out,
                                              uint32_t* seg, int after_halo = -1, float* cov2_out = nullptr,
                                              float slack = 0.f) {
    Best best;
    float cov;
    int l;
    const int gbase = (int)(threadIdx.x & 63) & ~(G - 1);  // first lane of this group in the wave
    if (nearest_halo<G, kTop2>(g, p, sub, gbase, first_level, after_halo, best, cov, l, level_out, cov2_out)) return best;
    return nearest_levels<G, kPrune, kTop2>(g, p, sub, gbase, l, best, cov, seg, slack, level_out, cov2_out);
}

}  // namespace o3dreg
