// reg_core.hip -- C ABI (include/o3dslam_reg.h) + HIP kernels of the MI355X registration path.
// gfx950 only.  The product never falls back to a CPU path: every entry point that needs the
// device returns REG_DEVICE_ERROR when HIP fails.
#include "../../include/o3dslam_reg.h"
#include "host_math.hpp"
#include "reg_kernels.hpp"

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

using namespace o3dreg;

// minimum waves per SIMD the search kernels are compiled for (2nd __launch_bounds__ argument): caps their VGPRs
#ifndef O3D_MATCH_WAVES
#define O3D_MATCH_WAVES 1
#endif

// Iteration state living in device memory: the pose the kernels read, the checker history and the
// termination flags.  The update kernel (last kernel of an iteration) is its only writer, so a whole
// registration can be enqueued without a host round trip per Gauss-Newton iteration.
struct IterState {
    float T[16];          // T_iter, row-major (P2PL: centred frames; GICP: reading -> reference)
    double Td[16];        // GICP: the same in double
    Checkers chk;         // DifferentialTransformationChecker / CounterTransformationChecker state
    int iterations;
    int done;             // 1: the remaining enqueued kernels return immediately
    int status;           // reg_status of the loop (REG_OK / REG_NO_CORRESPONDENCES)
    int rank_last;
    int cost;
    int fixed_iters;
    int max_iter;
    int update;           // 0: reduce only (reg_linearize / distributed halves), 1: solve + update + check
    float gicp_rot_eps, gicp_trans_eps;
    double sums[kSums];
    // fused path (k_iter_fused): predicted band [band_lo, band_hi) around the trimmed-quantile limit
    float band_lo, band_hi;   // +inf / +inf: no trimming (every finite match is inside)
    float trim_ratio;
    int use_trim;             // 1: TrimmedDistOutlierFilter active
    int stall;                // 1: the band prediction failed; enqueued fused kernels return until the host repairs
    float limit_last;         // trimmed limit of the last completed iteration (+inf: none)
    float limit_prev;         // ... and of the one before
    unsigned int band_count;  // records appended to the band buffer in this iteration
    unsigned int band_cap;
    int debug_narrow_band;
    // R8x (X-ICP localizability, OptimizedEqualityConstraints)
    int xicp_stage;           // 0: off / analysed, 1: analysis pending (first iteration), 2: sums being collected
    int xicp_nc;              // number of non-localizable directions (constraints)
    int xicp_flags[6];        // 1 = localizable; rotation eigen-directions 0-2, translation 3-5
    float xicp_enough, xicp_insufficient, xicp_cos_min, xicp_cos_strong;
    float xicp_Trd[12];       // T_refMean_dataIn (row-major 3x4): its inverse takes the matched data to the frame it came from
    double xicp_comb[6], xicp_high[6];   // the information sums of the analysis (reported with every mirror)
};

// Scratch of the first-iteration localizability analysis.
struct XicpState {
    float vr[9], vt[9];       // eigenvectors in the data frame, [k*3 + r]
    int pad[2];
    double center[4];         // sum of the matched reading points (data frame) + their count
    double comb[6], high[6];  // information sums: rotation 0-2, translation 3-5
};

// What the update kernel mirrors into mapped host memory (the host polls `seq`).
struct HostMirror {
    double sums[kSums];
    float T[16];
    int iterations, done, status, rank_last, converged, max_iter_reached, stall, band_count;
    float limit_last, limit_prev, band_lo, band_hi;
    int pad_nband, pad2;
    int localizable[6];
    int n_constraints, pad3;
    double xicp_comb[6], xicp_high[6];
    unsigned long long stamps[8];   // s_memtime stamps of the update kernel (diagnostics only; nothing reads them)
    unsigned long long seq;
};

// XCD-aware workgroup order: the dispatcher deals workgroups round-robin over the 8 XCDs (blockIdx % 8
// shares an XCD).  With a Morton-ordered reading, giving each XCD ONE contiguous eighth of the reading means
// its private 4 MB L2 only has to hold that region's slice of the reference cloud and tables.
// Launch with gridDim.x = 8 * ceil(n_blocks / 8); returns the logical block (>= n_blocks: nothing to do).
__device__ __forceinline__ int xcd_block(int n_blocks) {
    const int chunk = (n_blocks + 7) >> 3;
    return (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
}

struct Xf4 {
    float m[16];
};

__device__ __forceinline__ Xf load_xf(const IterState* it) {
    Xf x;
#pragma unroll
    for (int k = 0; k < 12; ++k) x.m[k] = it->T[k];
    return x;
}

// =================================================================================================
// kernels: target preparation (R1)
// =================================================================================================

// Order-independent centroid: integer sum of llrint(x * 2^16) (numeric contract NC1).
__global__ void k_centroid_sums(const float* __restrict__ xyz, int64_t stride, int64_t n, unsigned long long* sums) {
    __shared__ long long sh[3][4];
    long long s0 = 0, s1 = 0, s2 = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float* p = xyz + i * stride;
        s0 += llrint((double)p[0] * 65536.0);
        s1 += llrint((double)p[1] * 65536.0);
        s2 += llrint((double)p[2] * 65536.0);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_down(s0, o);
        s1 += __shfl_down(s1, o);
        s2 += __shfl_down(s2, o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sh[0][wave] = s0;
        sh[1][wave] = s1;
        sh[2][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        long long t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[threadIdx.x][w];
        atomicAdd(&sums[threadIdx.x], (unsigned long long)t);
    }
}

__device__ __forceinline__ int f2ord(float f) {
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float ord2f(int i) {
    int j = i >= 0 ? i : i ^ 0x7fffffff;
    float f;
    memcpy(&f, &j, 4);
    return f;
}

// centred = fl(x - c); bounding box of the centred cloud (ordered-int atomics).
__global__ void k_center_bbox(const float* __restrict__ xyz, int64_t stride, int64_t n, float cx, float cy, float cz,
                              float4* __restrict__ out, int* bbox /* min xyz, max xyz as ordered ints */) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float* p = xyz + i * stride;
        float4 q;
        q.x = p[0] - cx;
        q.y = p[1] - cy;
        q.z = p[2] - cz;
        q.w = __uint_as_float((uint32_t)i);
        out[i] = q;
        mn[0] = fminf(mn[0], q.x); mx[0] = fmaxf(mx[0], q.x);
        mn[1] = fminf(mn[1], q.y); mx[1] = fmaxf(mx[1], q.y);
        mn[2] = fminf(mn[2], q.z); mx[2] = fmaxf(mx[2], q.z);
    }
    for (int o = 32; o > 0; o >>= 1)
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(mn[k], __shfl_down(mn[k], o));
            mx[k] = fmaxf(mx[k], __shfl_down(mx[k], o));
        }
    // same-address atomics serialise (measured: 49 k of them on 6 words cost 0.5 ms): one set per workgroup only
    __shared__ float red[6][4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) {
            red[k][wave] = mn[k];
            red[3 + k][wave] = mx[k];
        }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        const int nw = (int)(blockDim.x >> 6);
        float v = red[k][0];
        for (int w = 1; w < nw; ++w) v = k < 3 ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
        if (k < 3)
            atomicMin(&bbox[k], f2ord(v));
        else
            atomicMax(&bbox[k], f2ord(v));
    }
}

// sort key = (brick z,y,x | bin-in-brick z,y,x)
__global__ void k_point_keys(const float4* __restrict__ pts, int64_t n, float ox, float oy, float oz, float inv_c,
                             uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const int cx = (int)bin_coord_f(p.x, ox, inv_c);
    const int cy = (int)bin_coord_f(p.y, oy, inv_c);
    const int cz = (int)bin_coord_f(p.z, oz, inv_c);
    const uint64_t bk = brick_key((uint32_t)(cx >> kBrickLog2), (uint32_t)(cy >> kBrickLog2), (uint32_t)(cz >> kBrickLog2));
    const uint32_t local = ((cz & (kBrickDim - 1)) << (2 * kBrickLog2)) | ((cy & (kBrickDim - 1)) << kBrickLog2) |
                           (cx & (kBrickDim - 1));
    keys[i] = (bk << (3 * kBrickLog2)) | local;
    vals[i] = (uint32_t)i;
}

__global__ void k_gather_target(const float4* __restrict__ centred, const uint32_t* __restrict__ order, int64_t n,
                                const float* __restrict__ nrm, int64_t nrm_stride, const float* __restrict__ cov,
                                float4* __restrict__ pts_sorted, float4* __restrict__ nrm_sorted,
                                float4* __restrict__ cov_sorted /* 2 float4 per point */) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = order[i];
    pts_sorted[i] = centred[src];
    if (nrm_sorted) {
        const float* q = nrm + (int64_t)src * nrm_stride;
        nrm_sorted[i] = make_float4(q[0], q[1], q[2], 0.f);
    }
    if (cov_sorted) {
        const float* q = cov + (int64_t)src * 6;
        cov_sorted[2 * i] = make_float4(q[0], q[1], q[2], q[3]);
        cov_sorted[2 * i + 1] = make_float4(q[4], q[5], 0.f, 0.f);
    }
}

__global__ void k_brick_heads(const uint64_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ flags) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || (keys[i] >> (3 * kBrickLog2)) != (keys[i - 1] >> (3 * kBrickLog2))) ? 1u : 0u;
}

// brick_id = inclusive_scan(flags) - 1.  Inserts brick heads into the hash and counts points per bin.
__global__ void k_fill_tables(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ flags,
                              const uint32_t* __restrict__ scan, int64_t n, HashEntry* hash, uint32_t mask,
                              uint32_t* __restrict__ counts, uint32_t* __restrict__ occupied,
                              int32_t* __restrict__ dir, int bdx, int bdy) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t bid = scan[i] - 1u;
    const uint64_t key = keys[i];
    const uint32_t local = (uint32_t)(key & (kBrickCells - 1));
    if (flags[i]) {
        const uint64_t bk = key >> (3 * kBrickLog2);
        uint32_t h = (uint32_t)mix64(bk) & mask;
        for (;;) {
            const unsigned long long prev =
                atomicCAS((unsigned long long*)&hash[h].key, (unsigned long long)kEmptyKey, (unsigned long long)bk);
            if (prev == kEmptyKey) {
                hash[h].val = bid;
                break;
            }
            h = (h + 1) & mask;
        }
        if (dir) {
            const uint32_t m18 = (1u << kBrickBits) - 1u;
            const uint32_t bx = (uint32_t)bk & m18, by = (uint32_t)(bk >> kBrickBits) & m18,
                           bz = (uint32_t)(bk >> (2 * kBrickBits)) & m18;
            dir[((size_t)bz * bdy + by) * bdx + bx] = (int32_t)bid;
        }
    }
    const uint32_t old = atomicAdd(&counts[(size_t)bid * kBrickCells + local], 1u);
    // one aggregated atomic per wave on the single "occupied bins" word (same-address atomics serialise)
    const unsigned long long first = __ballot(old == 0);
    if (first && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)first) - 1)) atomicAdd(occupied, (uint32_t)__popcll(first));
}

// Halo bins: every reference point is listed in each bin whose box, grown by rho_h, contains it.
struct HaloCfg {
    float ox, oy, oz, inv_c, r_ins;  // r_ins = rho_h + safety margin
    int dimx, dimy, dimz;
};
__device__ __forceinline__ void halo_range(float v, float o, float inv_c, float r, int dim, int& lo, int& hi) {
    lo = (int)fminf(fmaxf(bin_coord_f(v - r, o, inv_c), 0.f), (float)(dim - 1));
    hi = (int)fminf(fmaxf(bin_coord_f(v + r, o, inv_c), 0.f), (float)(dim - 1));
}
// pass 0: count, pass 1: fill (cursor = running insert position per bin)
__global__ void k_halo_insert(const float4* __restrict__ pts_sorted, int64_t n, HaloCfg c, int pass,
                              uint32_t* __restrict__ counts_or_cursor, float4* __restrict__ halo_pts) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts_sorted[i];
    int x0, x1, y0, y1, z0, z1;
    halo_range(p.x, c.ox, c.inv_c, c.r_ins, c.dimx, x0, x1);
    halo_range(p.y, c.oy, c.inv_c, c.r_ins, c.dimy, y0, y1);
    halo_range(p.z, c.oz, c.inv_c, c.r_ins, c.dimz, z0, z1);
    for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                const size_t B = ((size_t)z * c.dimy + y) * c.dimx + x;
                const uint32_t slot = atomicAdd(&counts_or_cursor[B], 1u);
                if (pass == 1) halo_pts[slot] = make_float4(p.x, p.y, p.z, __uint_as_float((uint32_t)i));
            }
}

// =================================================================================================
// kernels: reading preparation (R2)
// =================================================================================================

// Reading-preparation state computed on the device (no host round trip between the centroid reduction and the
// kernels that need it): centroid of the reading and T0 = T_refIn_refMean^-1 * T_init * T_readIn_readMean.
struct PrepState {
    float c_read[3];
    float pad;
    float T0[16];   // row-major
};
// sums: integer centroid sums (NC1); c_override != null: use the given (global, multi-GPU) centroid instead.
__global__ void k_make_T0(const unsigned long long* __restrict__ sums, int64_t n, float3 c_ref, Xf4 T_init, int centre,
                          int use_override, float3 c_override, PrepState* __restrict__ out,
                          PrepState* __restrict__ host_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float c[3] = {0.f, 0.f, 0.f};
    if (centre) {
        if (use_override) {
            c[0] = c_override.x; c[1] = c_override.y; c[2] = c_override.z;
        } else {
            for (int k = 0; k < 3; ++k) c[k] = (float)((double)(long long)sums[k] / (65536.0 * (double)n));
        }
    }
    float A[16], B[16], tmp[16], T0[16];
    m4_identity(A);
    m4_identity(B);
    if (centre) {
        A[3] = -c_ref.x; A[7] = -c_ref.y; A[11] = -c_ref.z;   // T_refIn_refMean^-1
        B[3] = c[0]; B[7] = c[1]; B[11] = c[2];               // T_readIn_readMean
        m4_mul(A, T_init.m, tmp);
        m4_mul(tmp, B, T0);
    } else {
        for (int i = 0; i < 16; ++i) T0[i] = T_init.m[i];
    }
    for (int k = 0; k < 3; ++k) out->c_read[k] = c[k];
    for (int i = 0; i < 16; ++i) out->T0[i] = T0[i];
    if (host_out) {   // mapped pinned copy for the final composition on the host (a D2H memcpy costs ~50 us of host time)
        for (int k = 0; k < 3; ++k) host_out->c_read[k] = c[k];
        for (int i = 0; i < 16; ++i) host_out->T0[i] = T0[i];
        __threadfence_system();
    }
}

// Morton key of the bin the (pre-transformed) reading point falls into: neighbouring lanes then search
// neighbouring bins (speed only -- results are reported in the caller's order).
__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// Morton key of a reading point in the reading's OWN frame (cells of edge `cell`, anchored at the first point, 10
// bits per axis): a rigid transform keeps neighbours together, so the order is computed once per reading
// (reg_set_source) and serves every initial guess.  Speed only -- results are reported in the caller's order.
__global__ void k_source_keys(const float* __restrict__ xyz, int64_t stride, int64_t n, float inv_cell,
                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = xyz + i * stride;
    const float ax = xyz[0], ay = xyz[1], az = xyz[2];
    const uint32_t bx = (uint32_t)fminf(fmaxf(floorf((p[0] - ax) * inv_cell) + 512.f, 0.f), 1023.f);
    const uint32_t by = (uint32_t)fminf(fmaxf(floorf((p[1] - ay) * inv_cell) + 512.f, 0.f), 1023.f);
    const uint32_t bz = (uint32_t)fminf(fmaxf(floorf((p[2] - az) * inv_cell) + 512.f, 0.f), 1023.f);
    keys[i] = spread10(bx) | (spread10(by) << 1) | (spread10(bz) << 2);
    vals[i] = (uint32_t)i;
}

// reading' = T0 * (p - c_read), normals' = R0 * n  (ICP.cpp:966-984); slot i holds input point perm[i]
// Also clears the per-registration scratch (level hints, trimmed-quantile histograms, accumulator replicas), so the
// registration needs no memset launches.
__global__ void k_prepare_source(const float* __restrict__ xyz, int64_t stride, const float* __restrict__ nrm,
                                 int64_t nrm_stride, int64_t n, const PrepState* __restrict__ ps, int centre,
                                 const uint32_t* __restrict__ perm, float4* __restrict__ out_xyz,
                                 float4* __restrict__ out_nrm, uint8_t* __restrict__ hint, uint32_t* __restrict__ hist,
                                 double* __restrict__ acc, int n_acc) {
    if (blockIdx.x == 0) {
        for (int k = threadIdx.x; k < 3 * 2048; k += blockDim.x) hist[k] = 0u;
        for (int k = threadIdx.x; k < n_acc; k += blockDim.x) acc[k] = 0.0;
    }
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    hint[i] = 0;
    Xf T0;
#pragma unroll
    for (int k = 0; k < 12; ++k) T0.m[k] = ps->T0[k];
    const float cx = ps->c_read[0], cy = ps->c_read[1], cz = ps->c_read[2];
    const int64_t src = perm ? (int64_t)perm[i] : i;
    const float* p = xyz + src * stride;
    float x = p[0], y = p[1], z = p[2];
    if (centre) {
        x = x - cx;
        y = y - cy;
        z = z - cz;
        const float3 q = xf_point(T0, x, y, z);
        x = q.x; y = q.y; z = q.z;
    }
    out_xyz[i] = make_float4(x, y, z, 1.f);
    if (out_nrm) {
        const float* q = nrm + src * nrm_stride;
        float3 r = make_float3(q[0], q[1], q[2]);
        if (centre) r = xf_rot(T0, r.x, r.y, r.z);
        out_nrm[i] = make_float4(r.x, r.y, r.z, 0.f);
    }
}

__global__ void k_pack_cov(const float* __restrict__ cov, int64_t n, const uint32_t* __restrict__ perm,
                           float4* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = cov + (perm ? (int64_t)perm[i] : i) * 6;
    out[2 * i] = make_float4(q[0], q[1], q[2], q[3]);
    out[2 * i + 1] = make_float4(q[4], q[5], 0.f, 0.f);
}

// =================================================================================================
// kernels: the iteration (R3-R7)
// =================================================================================================

// R3 + R4: transform the reading by T_iter, exact 1-NN through the voxel-bin table.
// Writes the sorted position of the match (-1 = none) and the squared distance (+inf = none), and the
// level-0 radix histogram (top 11 bits) of the finite distances for the trimmed-quantile select.
__global__ void __launch_bounds__(256)
k_match(const float4* __restrict__ src, int64_t n, const IterState* __restrict__ it, Grid g, int* __restrict__ pos,
        float* __restrict__ d2, uint32_t* __restrict__ hist0 /* 2048 or null */, uint32_t* __restrict__ hist2_to_zero,
        int shift0) {
    __shared__ uint32_t sh[2048];
    if (it->done) return;
    const Xf T = load_xf(it);
    if (hist2_to_zero && blockIdx.x == 0)
        for (int k = threadIdx.x; k < 2048; k += blockDim.x) hist2_to_zero[k] = 0;
    if (hist0) {
        for (int k = threadIdx.x; k < 2048; k += blockDim.x) sh[k] = 0;
        __syncthreads();
    }
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) {
        const float4 s = src[i];
        const float3 p = xf_point(T, s.x, s.y, s.z);
        const Best b = nearest(g, p);
        pos[i] = b.pos;
        d2[i] = b.pos >= 0 ? b.d2 : INFINITY;
        if (hist0 && b.pos >= 0) atomicAdd(&sh[__float_as_uint(b.d2) >> shift0], 1u);
    }
    if (hist0) {
        __syncthreads();
        for (int k = threadIdx.x; k < 2048; k += blockDim.x)
            if (sh[k]) atomicAdd(&hist0[k], sh[k]);
    }
}

// LDS words per group for the wide level scan (segment starts + exclusive offsets + sentinel)
template <int G>
constexpr int kSegWords = 2 * G * kSegPerLane + 2;

// Cooperative variant: G (8 or 4) lanes per reading point (256/G points per 256-thread workgroup).
// `hint` (one byte per point, may be null) carries the terminating level of the previous iteration.
template <int G>
__global__ void __launch_bounds__(256, O3D_MATCH_WAVES)
k_match_g8(const float4* __restrict__ src, int64_t n, const IterState* __restrict__ it, Grid g, int* __restrict__ pos,
           float* __restrict__ d2, uint32_t* __restrict__ hist0 /* 2048 or null */,
           uint32_t* __restrict__ hist2_to_zero, uint8_t* __restrict__ hint, int shift0, int debug, int n_blocks) {
    __shared__ uint32_t sh[2048];
    __shared__ uint32_t seg_lds[(256 / G) * kSegWords<G>];
    if (it->done) return;
    const Xf T = load_xf(it);
    if (hist2_to_zero && blockIdx.x == 0)
        for (int k = threadIdx.x; k < 2048; k += blockDim.x) hist2_to_zero[k] = 0;
    if (hist0) {
        for (int k = threadIdx.x; k < 2048; k += blockDim.x) sh[k] = 0;
        __syncthreads();
    }
    const int lb = xcd_block(n_blocks);
    const int64_t tid = lb * (int64_t)blockDim.x + threadIdx.x;
    const int64_t q = lb < n_blocks ? (tid / G) : n;
    const int sub = (int)(tid & (G - 1));
    if (q < n) {
        const float4 s = src[q];
        const float3 p = xf_point(T, s.x, s.y, s.z);
        if (debug & 4) {  // timing experiment: fixed cost of the launch + reading load only
            if (sub == 0) {
                pos[q] = -1;
                d2[q] = p.x;
            }
            return;
        }
        // hint h: 0/1 = the last search ended at the halo level / regular level 0 -> try the halo first;
        // h >= 2 = it ended at regular level h-1 -> skip the halo and start one regular level below.
        const int hv = hint ? (int)hint[q] : 0;
        const int first = hv >= 2 ? hv - 2 : -1;
        int lvl;
        const Best b = nearest_group<G>(g, p, sub, first, &lvl, seg_lds + (threadIdx.x / G) * kSegWords<G>);
        if (sub == 0) {
            pos[q] = b.pos;
            d2[q] = b.pos >= 0 ? b.d2 : INFINITY;
            if (hint) hint[q] = (uint8_t)(lvl + 1);
            if (hist0 && b.pos >= 0) atomicAdd(&sh[__float_as_uint(b.d2) >> shift0], 1u);
        }
    }
    if (hist0) {
        __syncthreads();
        for (int k = threadIdx.x; k < 2048; k += blockDim.x)
            if (sh[k]) atomicAdd(&hist0[k], sh[k]);
    }
}

// Radix-select state kept on the device between the passes of one iteration.
struct SelectState {
    uint32_t prefix;     // bits of the k-th smallest value fixed so far (left-aligned)
    uint32_t rank;       // remaining 0-based rank inside the selected bucket
    uint32_t n_finite;
    uint32_t done;
    float limit;         // result: k-th smallest finite d2 (or max when ratio == 1)
    uint32_t pad[3];
};

// Exact trimmed-quantile select (Matches.cpp:60-87) as a 3-level radix select on the fp32 bit pattern
// (d2 >= 0, so the pattern is monotone): level 0 = bits [31:21] (histogram built by the match kernel),
// level 1 = bits [20:10], level 2 = bits [9:0].  No single-workgroup pass: every workgroup of the
// NEXT kernel re-derives the bin picked at the previous level from the (tiny) global histogram.

// Block-wide (256 threads): bin b with cum[b] <= rank < cum[b+1] over hist[0..nb), nb <= 2048.
// Returns through LDS: out[0] = bin, out[1] = rank inside the bin, out[2] = total count.
__device__ __forceinline__ void block_pick256(const uint32_t* __restrict__ hist, int nb, uint32_t rank,
                                              uint32_t* wave_tot /*[4]*/, uint32_t* out /*[3]*/) {
    const int t = threadIdx.x;
    uint32_t loc[8];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int bin = t * 8 + k;
        loc[k] = bin < nb ? hist[bin] : 0u;
        sum += loc[k];
    }
    uint32_t incl = sum;
    const int lane = t & 63, wave = t >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    if (t == 0) {
        out[0] = 0;
        out[1] = 0;
    }
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t v = wave_tot[w];
        if (w < wave) base += v;
        total += v;
    }
    if (t == 0) out[2] = total;
    uint32_t run = base + incl - sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (loc[k] && rank >= run && rank < run + loc[k]) {
            out[0] = (uint32_t)(t * 8 + k);
            out[1] = rank - run;
        }
        run += loc[k];
    }
    __syncthreads();
}

// Matches.cpp:82-86: index = size()*quantile evaluated in float, truncated; quantile == 1 -> maximum.
__device__ __forceinline__ uint32_t trim_rank(uint32_t total, float ratio) {
    if (total == 0) return 0;
    if (ratio == 1.0f) return total - 1;
    const float posf = (float)total * ratio;
    uint32_t r = (uint32_t)posf;
    return r >= total ? total - 1 : r;
}

// level = 1: pick level 0 from hist_prev (= hist0), histogram bits [20:10] into hist_out, zero nothing.
// level = 2: pick level 1 from hist_prev (= hist1) using the state, histogram bits [9:0]; zero `to_zero`.
__global__ void __launch_bounds__(256)
k_select_level(const float* __restrict__ d2, int64_t n, int level, int shift0, float ratio, const uint32_t* __restrict__ hist_prev,
               uint32_t* __restrict__ hist_out, uint32_t* __restrict__ to_zero, SelectState* st,
               const IterState* __restrict__ it) {
    __shared__ uint32_t sh[2048];
    if (it->done) return;
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t pick[3];
    for (int k = threadIdx.x; k < 2048; k += blockDim.x) sh[k] = 0;
    uint32_t prefix, rank_in;
    if (level == 1) {
        // total count of finite distances = sum of hist0
        block_pick256(hist_prev, 2048, 0xffffffffu, wave_tot, pick);
        const uint32_t total = pick[2];
        __syncthreads();
        block_pick256(hist_prev, 2048, trim_rank(total, ratio), wave_tot, pick);
        prefix = pick[0] << shift0;
        rank_in = pick[1];
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->n_finite = total;
            st->prefix = prefix;
            st->rank = rank_in;
            if (total == 0) st->limit = INFINITY;
        }
    } else {
        block_pick256(hist_prev, 2048, st->rank, wave_tot, pick);
        prefix = st->prefix | (pick[0] << (shift0 - 11));
        rank_in = pick[1];
    }
    __syncthreads();
    const int s1 = shift0 - 11;  // low bit of the level-1 digit; level 2 = the s1 lowest bits
    const uint32_t mask = level == 1 ? ~((1u << shift0) - 1u) : ~((1u << s1) - 1u);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t u = __float_as_uint(d2[i]);
        if (u != 0x7f800000u && (u & mask) == prefix) {
            const uint32_t b = level == 1 ? ((u >> s1) & 2047u) : (u & ((1u << s1) - 1u));
            atomicAdd(&sh[b], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2048; k += blockDim.x)
        if (sh[k]) atomicAdd(&hist_out[k], sh[k]);
    if (blockIdx.x == 0) {
        if (to_zero)
            for (int k = threadIdx.x; k < 2048; k += blockDim.x) to_zero[k] = 0;
        if (level == 2 && threadIdx.x == 0) {
            // the state is only read by later kernels
        }
    }
    // publish the level-2 prefix/rank for the linearize kernel (kernel boundary orders it)
    if (level == 2 && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        st->pad[0] = prefix;    // 22 fixed bits
        st->pad[1] = rank_in;   // rank inside that bucket
    }
}

// Level-0 histogram of the single-GPU pipeline (digit = bits [shift0+10 : shift0]).  A separate pass with
// few workgroups: folding it into the match kernel costs ~10^5 global atomics (~25-40 us on MI355X).
__global__ void __launch_bounds__(256)
k_hist_level0(const float* __restrict__ d2, int64_t n, int shift0, uint32_t* __restrict__ hist,
              const IterState* __restrict__ it) {
    __shared__ uint32_t sh[2048];
    if (it->done) return;
    for (int k = threadIdx.x; k < 2048; k += blockDim.x) sh[k] = 0;
    __syncthreads();
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t u = __float_as_uint(d2[i]);
        if (u != 0x7f800000u) atomicAdd(&sh[u >> shift0], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2048; k += blockDim.x)
        if (sh[k]) atomicAdd(&hist[k], sh[k]);
}

// Histogram of one radix level for a caller-supplied prefix (distributed path: the ranks sum these).
__global__ void __launch_bounds__(256)
k_hist_prefix(const float* __restrict__ d2, int64_t n, int level, uint32_t prefix, uint32_t* __restrict__ hist) {
    __shared__ uint32_t sh[2048];
    for (int k = threadIdx.x; k < 2048; k += blockDim.x) sh[k] = 0;
    __syncthreads();
    const uint32_t mask = level == 0 ? 0u : (level == 1 ? 0xffe00000u : 0xfffffc00u);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t u = __float_as_uint(d2[i]);
        if (u != 0x7f800000u && (u & mask) == prefix) {
            const uint32_t b = level == 0 ? (u >> 21) : (level == 1 ? ((u >> 10) & 2047u) : (u & 1023u));
            atomicAdd(&sh[b], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2048; k += blockDim.x)
        if (sh[k]) atomicAdd(&hist[k], sh[k]);
}

struct FilterCfg {
    int use_trim, use_normal, use_maxdist;
    int debug;   // ablation switches for kernel timing experiments (0 in production)
    float cos_max_angle;
    float outlier_max_d2;
};

__device__ __forceinline__ float3 normalize3(float3 n) {
    float a = n.x * n.x;
    float b = n.y * n.y;
    float z = a + b;
    a = n.z * n.z;
    z = z + a;
    if (z > 0.f) {
        const float s = sqrtf(z);
        n.x = n.x / s;
        n.y = n.y / s;
        n.z = n.z / s;
    }
    return n;
}

// Reduce kSums (=32) doubles per lane over the 64 lanes of a wave with 32 shuffles instead of 192:
// at every step a lane keeps one half of its values and hands the other half to its xor-partner.
// On return lanes 2c and 2c+1 both hold the wave total of component c in v[0].
template <int HALF, int BIT>
__device__ __forceinline__ void wave_reduce_step(double (&v)[kSums], int lane) {
    const bool up = (lane & BIT) != 0;
#pragma unroll
    for (int k = 0; k < HALF; ++k) {
        const double keep = up ? v[k + HALF] : v[k];
        const double send = up ? v[k] : v[k + HALF];
        v[k] = keep + __shfl_xor(send, BIT);
    }
}
// (every index is a compile-time constant after unrolling: the array must stay in registers -- an earlier
// version with a runtime `half` put it in scratch: 272 B/lane, 51 MB of scratch writes per launch)
__device__ __forceinline__ void wave_reduce32(double (&v)[kSums]) {
    const int lane = threadIdx.x & 63;
    wave_reduce_step<16, 32>(v, lane);
    wave_reduce_step<8, 16>(v, lane);
    wave_reduce_step<4, 8>(v, lane);
    wave_reduce_step<2, 4>(v, lane);
    wave_reduce_step<1, 2>(v, lane);
    v[0] = v[0] + __shfl_xor(v[0], 1);
}

// block partial -> global: partials[blockIdx.x][kSums]  (256 threads = 4 waves)
__device__ __forceinline__ void block_reduce_store(double (&vals)[kSums], double* __restrict__ partials) {
    __shared__ double sh[4][kSums];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    wave_reduce32(vals);
    if ((lane & 1) == 0) sh[wave][lane >> 1] = vals[0];
    __syncthreads();
    if (threadIdx.x < kSums) {
        double t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w][threadIdx.x];
        partials[(size_t)blockIdx.x * kSums + threadIdx.x] = t;
    }
}

// R5 + R6 + R7 (point-to-plane): weights, F = [p x n ; n], A += w F F^T, b -= w F (n.(p-q)).
// Per-pair products in fp32 (as the reference computes them), summed in fp64 (numeric contract NC8).
__global__ void __launch_bounds__(256)
k_linearize_p2pl(const float4* __restrict__ src, const float4* __restrict__ src_nrm, int64_t n,
                 const IterState* __restrict__ it,
                 const int* __restrict__ pos, const float* __restrict__ d2, const float4* __restrict__ tgt,
                 const float4* __restrict__ tgt_nrm, FilterCfg f, SelectState* __restrict__ st,
                 const uint32_t* __restrict__ hist2, uint32_t* __restrict__ hist1_to_zero, int shift0,
                 float* __restrict__ w_out, double* __restrict__ partials) {
    if (it->done) return;
    const Xf T = load_xf(it);
    // trimmed-quantile limit: last radix level, re-derived by every workgroup (f.use_trim == 2),
    // or taken from the state as given by the caller (f.use_trim == 1: distributed path)
    float limit = INFINITY;
    if (f.use_trim == 2) {
        __shared__ uint32_t wave_tot[4];
        __shared__ uint32_t pick[3];
        if (st->n_finite != 0) {
            block_pick256(hist2, 1 << (shift0 - 11), st->pad[1], wave_tot, pick);
            limit = __uint_as_float(st->pad[0] | pick[0]);
        }
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) st->limit = limit;
            for (int k = threadIdx.x; k < 2048; k += blockDim.x) hist1_to_zero[k] = 0;
        }
    } else if (f.use_trim == 1) {
        limit = st->limit;
    }
    double v[kSums];
#pragma unroll
    for (int k = 0; k < kSums; ++k) v[k] = 0.0;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) {
        const int ps = pos[i];
        const float dd = d2[i];
        float w = 0.f;
        if (ps >= 0) {
            v[29] = 1.0;
            w = 1.f;
            if (f.use_maxdist && !(dd <= f.outlier_max_d2)) w = 0.f;
            if (f.use_trim && !(dd <= limit)) w = 0.f;
            const float4 s = src[i];
            const float3 p = xf_point(T, s.x, s.y, s.z);
            const float4 nn = (f.debug & 1) ? make_float4(0.f, 0.f, 1.f, 0.f) : tgt_nrm[ps];
            if (f.use_normal) {
                const float4 sn = src_nrm[i];
                const float3 nr = normalize3(xf_rot(T, sn.x, sn.y, sn.z));
                const float3 nt = normalize3(make_float3(nn.x, nn.y, nn.z));
                float a = nr.x * nt.x;
                float b = nr.y * nt.y;
                float val = a + b;
                a = nr.z * nt.z;
                val = val + a;
                if (val < f.cos_max_angle) w = 0.f;
            }
            if (w != 0.f) {
                const float4 q = (f.debug & 1) ? make_float4(s.x, s.y, s.z, 0.f) : tgt[ps];
                float F[6];
                float a = p.y * nn.z, b = p.z * nn.y;
                F[0] = a - b;
                a = p.z * nn.x; b = p.x * nn.z;
                F[1] = a - b;
                a = p.x * nn.y; b = p.y * nn.x;
                F[2] = a - b;
                F[3] = nn.x; F[4] = nn.y; F[5] = nn.z;
                const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
                float r = dx * nn.x;
                float t2 = dy * nn.y;
                r = r + t2;
                t2 = dz * nn.z;
                r = r + t2;
                int k = 0;
#pragma unroll
                for (int a6 = 0; a6 < 6; ++a6) {
                    const float wf = w * F[a6];
#pragma unroll
                    for (int c6 = a6; c6 < 6; ++c6) {
                        const float pr = wf * F[c6];
                        v[k++] = (double)pr;
                    }
                }
#pragma unroll
                for (int a6 = 0; a6 < 6; ++a6) {
                    const float wf = w * F[a6];
                    const float pr = wf * r;
                    v[21 + a6] = (double)pr;
                }
                const float rr = r * r;
                v[27] = (double)(w * rr);
                v[28] = 1.0;
                v[30] = (double)dd;
            }
        }
        if (w_out) w_out[i] = w;
    }
    if (f.debug & 2) {
        double t = 0;
        for (int k = 0; k < kSums; ++k) t += v[k];
        if (t == 1.2345) partials[0] = t;
        return;
    }
    block_reduce_store(v, partials);
}

// GICP factor (north-star cost): r = q - T p, M = (Cq + R Cp R^T)^-1, J = [R skew(p), -R];
// H += J^T M J, b += J^T M r, e += 0.5 r^T M r.  Per-point algebra in fp64 (inputs fp32).
__global__ void __launch_bounds__(256)
k_linearize_gicp(const float4* __restrict__ src, const float4* __restrict__ src_cov, int64_t n,
                 const IterState* __restrict__ it,
                 const int* __restrict__ pos, const float* __restrict__ d2, const float4* __restrict__ tgt,
                 const float4* __restrict__ tgt_cov, float* __restrict__ w_out, double* __restrict__ partials) {
    if (it->done) return;
    const Xf T = load_xf(it);
    double v[kSums];
#pragma unroll
    for (int k = 0; k < kSums; ++k) v[k] = 0.0;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) {
        const int ps = pos[i];
        float w = 0.f;
        if (ps >= 0) {
            w = 1.f;
            const float4 s = src[i];
            const float3 tp = xf_point(T, s.x, s.y, s.z);
            const float4 q = tgt[ps];
            const double r[3] = {(double)q.x - (double)tp.x, (double)q.y - (double)tp.y, (double)q.z - (double)tp.z};
            const float4 a0 = src_cov[2 * i], a1 = src_cov[2 * i + 1];
            const float4 b0 = tgt_cov[2 * (int64_t)ps], b1 = tgt_cov[2 * (int64_t)ps + 1];
            const double Cp[9] = {a0.x, a0.y, a0.z, a0.y, a0.w, a1.x, a0.z, a1.x, a1.y};
            const double Cq[9] = {b0.x, b0.y, b0.z, b0.y, b0.w, b1.x, b0.z, b1.x, b1.y};
            double R[9];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) R[3 * a + c] = (double)T.m[4 * a + c];
            double RC[9], S[9];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double t = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) t += R[3 * a + k] * Cp[3 * k + c];
                    RC[3 * a + c] = t;
                }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double t = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) t += RC[3 * a + k] * R[3 * c + k];
                    S[3 * a + c] = t + Cq[3 * a + c];
                }
            // symmetric 3x3 inverse
            double Mi[9];
            {
                const double a = S[0], b = S[1], c = S[2], d = S[4], e = S[5], ff = S[8];
                const double co00 = d * ff - e * e, co01 = c * e - b * ff, co02 = b * e - c * d;
                const double id = 1.0 / (a * co00 + b * co01 + c * co02);
                Mi[0] = co00 * id;
                Mi[1] = Mi[3] = co01 * id;
                Mi[2] = Mi[6] = co02 * id;
                Mi[4] = (a * ff - c * c) * id;
                Mi[5] = Mi[7] = (b * c - a * e) * id;
                Mi[8] = (a * d - b * b) * id;
            }
            const double px = s.x, py = s.y, pz = s.z;
            const double sk[9] = {0, -pz, py, pz, 0, -px, -py, px, 0};
            double J[18];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double t = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) t += R[3 * a + k] * sk[3 * k + c];
                    J[6 * a + c] = t;
                    J[6 * a + 3 + c] = -R[3 * a + c];
                }
            double MJ[18], Mr[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double t = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) t += Mi[3 * a + k] * J[6 * k + c];
                    MJ[6 * a + c] = t;
                }
                Mr[a] = Mi[3 * a] * r[0] + Mi[3 * a + 1] * r[1] + Mi[3 * a + 2] * r[2];
            }
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int c = a; c < 6; ++c) {
                    double t = 0;
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) t += J[6 * kk + a] * MJ[6 * kk + c];
                    v[k++] = t;
                }
#pragma unroll
            for (int a = 0; a < 6; ++a) v[21 + a] = J[a] * Mr[0] + J[6 + a] * Mr[1] + J[12 + a] * Mr[2];
            v[27] = 0.5 * (r[0] * Mr[0] + r[1] * Mr[1] + r[2] * Mr[2]);
            v[28] = 1.0;
            v[29] = 1.0;
            v[30] = (double)d2[i];
        }
        if (w_out) w_out[i] = w;
    }
    block_reduce_store(v, partials);
}

// =================================================================================================
// Fused iteration kernel (north star: KNN fused into the Jacobian / normal-equation accumulation)
// =================================================================================================
// One launch does R3 + R4 + R5 + R6 + R7 for a group's reading point: search, weights, F, r, the 27 products.
// The trimmed-quantile limit of THIS iteration is not known yet, so points are classified against a band
// [lo, hi) predicted from the previous iteration: d2 < lo -> certainly kept, d2 >= hi -> certainly trimmed,
// lo <= d2 < hi -> a 32-float record {d2, products, flags, point} is appended to a small buffer.  The update
// kernel verifies the prediction with exact counts (n_below <= k < n_below + n_band), picks the exact k-th
// smallest d2 inside the band, adds the surviving records, and only then solves.  A failed prediction stalls
// the queue; the host re-runs that iteration on the generic (select-based) path.  Results are identical to
// the generic path by construction: same products, same fp64 accumulation, exact quantile.
constexpr int kBandCap = 16384;
// Multi-GPU fused iteration: every rank contributes one fixed-size block {32 double sums, band count, up to
// kContribCap band records}; ONE all-gather per iteration hands every rank all blocks.
constexpr int kContribHdr = 128;                       // floats: [0..63] = 32 doubles, [64] = band count (uint32 bits)
constexpr int kContribCap = 512;                       // band records per rank
constexpr int kContribFloats = kContribHdr + kContribCap * 32;
constexpr int kAccRows = 64;   // replicas of the 32-double accumulator (spreads the fp64 atomics)
constexpr int kRec = 32;   // floats per band record
// Band buffer layout: record-major band[slot][kRec] (component-major, with or without a padded pitch, measured
// 2x slower for the single-workgroup reader: 19.5k vs 5.9k cycles for the add phase at 600 records)
__host__ __device__ __forceinline__ size_t band_at(int comp, size_t slot) { return slot * kRec + (size_t)comp; }

// products for one reading point: vals[0..20] = F_a F_c (upper triangle), [21..26] = F_a r, [27] = r^2
__device__ __forceinline__ void p2pl_products(float3 p, float4 q, float4 nn, float w, float* vals) {
    float F[6];
    float a = p.y * nn.z, b = p.z * nn.y;
    F[0] = a - b;
    a = p.z * nn.x; b = p.x * nn.z;
    F[1] = a - b;
    a = p.x * nn.y; b = p.y * nn.x;
    F[2] = a - b;
    F[3] = nn.x; F[4] = nn.y; F[5] = nn.z;
    const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
    float r = dx * nn.x;
    float t2 = dy * nn.y;
    r = r + t2;
    t2 = dz * nn.z;
    r = r + t2;
    int k = 0;
#pragma unroll
    for (int a6 = 0; a6 < 6; ++a6) {
        const float wf = w * F[a6];
#pragma unroll
        for (int c6 = a6; c6 < 6; ++c6) vals[k++] = wf * F[c6];
    }
#pragma unroll
    for (int a6 = 0; a6 < 6; ++a6) {
        const float wf = w * F[a6];
        vals[21 + a6] = wf * r;
    }
    const float rr = r * r;
    vals[27] = w * rr;
}

template <int G>
__global__ void __launch_bounds__(256, O3D_MATCH_WAVES)
k_iter_fused(const float4* __restrict__ src, const float4* __restrict__ src_nrm, int64_t n, IterState* __restrict__ it,
             Grid g, const float4* __restrict__ tgt_nrm, FilterCfg f, int* __restrict__ pos_out,
             float* __restrict__ d2_out, float* __restrict__ w_out, uint8_t* __restrict__ hint,
             float* __restrict__ band, int band_cap, double* __restrict__ partials, int n_blocks) {
    constexpr int CP = kSums / G;   // components owned by each lane of a group
    __shared__ double sh[4][kSums];
    __shared__ uint32_t seg_lds[(256 / G) * kSegWords<G>];
    if (it->done || it->stall) return;
    const Xf T = load_xf(it);
    const float band_lo = it->band_lo, band_hi = it->band_hi;
    const int lb = xcd_block(n_blocks);
    const int64_t tid = lb * (int64_t)blockDim.x + threadIdx.x;
    const int64_t q = lb < n_blocks ? (tid / G) : n;
    const int sub = (int)(tid & (G - 1));
    double mine[CP];
#pragma unroll
    for (int j = 0; j < CP; ++j) mine[j] = 0.0;
    if (q < n) {
        const float4 s = src[q];
        const float3 p = xf_point(T, s.x, s.y, s.z);
        const int hv = hint ? (int)hint[q] : 0;
        int lvl;
        const Best b = nearest_group<G>(g, p, sub, hv >= 2 ? hv - 2 : -1, &lvl,
                                        seg_lds + (threadIdx.x / G) * kSegWords<G>);
        float vals[kSums];
#pragma unroll
        for (int k = 0; k < kSums; ++k) vals[k] = 0.f;
        float w = 0.f;
        int cls = 2;  // 0: certainly kept, 1: band, 2: dropped / unmatched
        if (b.pos >= 0) {
            const float dd = b.d2;
            vals[29] = 1.f;
            w = 1.f;
            if (f.use_maxdist && !(dd <= f.outlier_max_d2)) w = 0.f;
            const float4 nn = tgt_nrm[b.pos];
            if (f.use_normal) {
                const float4 sn = src_nrm[q];
                const float3 nr = normalize3(xf_rot(T, sn.x, sn.y, sn.z));
                const float3 nt = normalize3(make_float3(nn.x, nn.y, nn.z));
                float a = nr.x * nt.x;
                float bb = nr.y * nt.y;
                float val = a + bb;
                a = nr.z * nt.z;
                val = val + a;
                if (val < f.cos_max_angle) w = 0.f;
            }
            cls = dd < band_lo ? 0 : (dd < band_hi ? 1 : 2);
            if (cls == 0) vals[31] = 1.f;   // counts towards n_below (rank bookkeeping is independent of w)
            if (w != 0.f && cls != 2) {
                const float4 tq = g.pts[b.pos];
                p2pl_products(p, tq, nn, w, vals);
                vals[28] = 1.f;
                vals[30] = dd;
            }
            if (cls == 1 && sub == 0) {
                // band record: decided by the update kernel
                const unsigned slot = atomicAdd(&it->band_count, 1u);
                if (slot < (unsigned)band_cap) {
#pragma unroll
                    for (int k = 0; k < 31; ++k)
                        if (k != 29) band[band_at(k, slot)] = vals[k];
                    band[band_at(29, slot)] = dd;   // [29] = d2 (the "matched" count is added from cls below)
                    band[band_at(31, slot)] = __int_as_float((int)q);
                }
            }
        }
        if (sub == 0) {
            pos_out[q] = b.pos;
            d2_out[q] = b.pos >= 0 ? b.d2 : INFINITY;
            if (hint) hint[q] = (uint8_t)(lvl + 1);
            if (w_out) w_out[q] = (cls == 2) ? 0.f : w;   // band points: provisional, patched by the update kernel
        }
        // certainly-kept contributions: lane `sub` owns components sub*CP .. sub*CP+CP-1
        if (cls == 1) {
#pragma unroll
            for (int k = 0; k < 29; ++k) vals[k] = 0.f;   // deferred
            vals[30] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < CP; ++j) {
            float v = 0.f;
#pragma unroll
            for (int sIdx = 0; sIdx < G; ++sIdx)
                if (sub == sIdx) v = vals[sIdx * CP + j];
            mine[j] = (double)v;
        }
    }
    // sum over the groups of the wave (same `sub` lanes), then over the 4 waves
#pragma unroll
    for (int m = G; m < 64; m <<= 1)
#pragma unroll
        for (int j = 0; j < CP; ++j) mine[j] += __shfl_xor(mine[j], m);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < G)
#pragma unroll
        for (int j = 0; j < CP; ++j) sh[wave][lane * CP + j] = mine[j];
    __syncthreads();
    if (threadIdx.x < kSums && lb < n_blocks) {
        const double t = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
        // one 256-byte fp64 atomic wave-instruction per workgroup into one of kAccRows replicas (memory-side
        // atomics; the update kernel sums the replicas in a fixed order and clears them)
        unsafeAtomicAdd(&partials[(size_t)(lb & (kAccRows - 1)) * kSums + threadIdx.x], t);
    }
}

// Block-wide (1024 threads): bin with cum[b] <= rank < cum[b+1] over h[0..nb), nb <= 2048 (2 bins per thread).
__device__ __forceinline__ void block_pick1024(const uint32_t* h, int nb, uint32_t rank, uint32_t* wave_tot /*[16]*/,
                                               uint32_t* out /*[0]=bin, [1]=rank inside the bin*/) {
    const int t = threadIdx.x;
    const uint32_t a = (2 * t < nb) ? h[2 * t] : 0u;
    const uint32_t b = (2 * t + 1 < nb) ? h[2 * t + 1] : 0u;
    const uint32_t sum = a + b;
    uint32_t incl = sum;
    const int lane = t & 63, wave = t >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    if (t == 0) {
        out[0] = 0;
        out[1] = 0;
    }
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    const uint32_t excl = base + incl - sum;
    if (a && rank >= excl && rank < excl + a) {
        out[0] = 2 * t;
        out[1] = rank - excl;
    }
    if (b && rank >= excl + a && rank < excl + a + b) {
        out[0] = 2 * t + 1;
        out[1] = rank - excl - a;
    }
    __syncthreads();
}

// Last kernel of an iteration: fixed-order sum of the workgroup partials; (fused path) verification of the
// predicted trimmed band + exact quantile inside it + the surviving band records; then (it->update) R8 + R9 on
// the device -- 6x6 solve in fp64, x -> 4x4, T_iter <- dT * T_iter, transformation checkers -- and a mirror of
// the outcome into mapped host memory followed by a sequence word the host polls.
// Multi-GPU fused iteration, between the fused kernel and the all-gather: reduce this rank's accumulator replicas
// into the header of its contribution block (and clear them), record its band count.
// ---- target-side preparation: crop (croppers.cpp:76-170) + fp64 -> fp32 (open3d_conversions.cpp:57-118)
struct CropCfg {
    int type;
    double cx, cy, cz, rmin, rmax, zmin, zmax;
};
__device__ __forceinline__ bool crop_inside(const CropCfg& c, double x, double y, double z) {
    if (c.type == REG_CROP_NONE) return true;
    const double dx = x - c.cx, dy = y - c.cy, dz = z - c.cz;
    if (c.type == REG_CROP_CYLINDER) {
        double a = dx * dx;
        double b = dy * dy;
        const double d = sqrt(a + b);
        return z >= c.zmin && z <= c.zmax && d <= c.rmax;
    }
    double a = dx * dx;
    double b = dy * dy;
    double s2 = a + b;
    a = dz * dz;
    s2 = s2 + a;
    const double d = sqrt(s2);
    if (c.type == REG_CROP_MAX_RADIUS) return d <= c.rmax;
    if (c.type == REG_CROP_MIN_RADIUS) return d >= c.rmin;
    return d <= c.rmax && d >= c.rmin;
}
__global__ void k_crop_flags(const double* __restrict__ xyz, int64_t m, CropCfg c, uint32_t* __restrict__ flags) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    flags[i] = crop_inside(c, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]) ? 1u : 0u;
}
// offs = exclusive scan of flags: order-preserving compaction + conversion
__global__ void k_crop_gather(const double* __restrict__ xyz, const double* __restrict__ nrm, const double* __restrict__ cov,
                              int64_t m, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ offs,
                              float* __restrict__ oxyz, float* __restrict__ onrm, float* __restrict__ ocov,
                              int32_t* __restrict__ oidx) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= m || !flags[i]) return;
    const size_t o = offs[i];
    oxyz[3 * o + 0] = (float)xyz[3 * i + 0];
    oxyz[3 * o + 1] = (float)xyz[3 * i + 1];
    oxyz[3 * o + 2] = (float)xyz[3 * i + 2];
    if (nrm) {
        onrm[3 * o + 0] = (float)nrm[3 * i + 0];
        onrm[3 * o + 1] = (float)nrm[3 * i + 1];
        onrm[3 * o + 2] = (float)nrm[3 * i + 2];
    }
    if (cov) {
        const double* c = cov + 9 * i;   // Matrix3d, symmetric: xx xy xz / . yy yz / . . zz
        ocov[6 * o + 0] = (float)c[0];
        ocov[6 * o + 1] = (float)c[1];
        ocov[6 * o + 2] = (float)c[2];
        ocov[6 * o + 3] = (float)c[4];
        ocov[6 * o + 4] = (float)c[5];
        ocov[6 * o + 5] = (float)c[8];
    }
    oidx[o] = (int32_t)i;
}

// ---- voxelizeWithinCroppingVolume (helpers.cpp:117-192) ----
constexpr int kVoxBits = 21;                       // voxel index bits per axis in the sort key (offset binary)
constexpr long long kVoxOff = 1ll << (kVoxBits - 1);
__global__ void k_vox_classify(const double* __restrict__ xyz, int64_t m, CropCfg c, double inv, uint32_t* __restrict__ f_in,
                               uint32_t* __restrict__ f_out, uint32_t* __restrict__ overflow) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const bool in = crop_inside(c, x, y, z);
    f_in[i] = in ? 1u : 0u;
    f_out[i] = in ? 0u : 1u;
    if (in) {
        const double vx = floor(x * inv), vy = floor(y * inv), vz = floor(z * inv);
        if (!(fabs(vx) < (double)kVoxOff && fabs(vy) < (double)kVoxOff && fabs(vz) < (double)kVoxOff)) atomicOr(overflow, 1u);
    }
}
__global__ void k_vox_scatter(const double* __restrict__ xyz, const double* __restrict__ nrm, const double* __restrict__ cov,
                              int64_t m, double inv, const uint32_t* __restrict__ f_in, const uint32_t* __restrict__ o_in,
                              const uint32_t* __restrict__ o_out, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                              double* __restrict__ oxyz, double* __restrict__ onrm, double* __restrict__ ocov) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (f_in[i]) {
        const long long vx = (long long)floor(xyz[3 * i] * inv) + kVoxOff;
        const long long vy = (long long)floor(xyz[3 * i + 1] * inv) + kVoxOff;
        const long long vz = (long long)floor(xyz[3 * i + 2] * inv) + kVoxOff;
        keys[o_in[i]] = ((uint64_t)vz << (2 * kVoxBits)) | ((uint64_t)vy << kVoxBits) | (uint64_t)vx;
        vals[o_in[i]] = (uint32_t)i;
    } else {
        const size_t o = o_out[i];
        for (int k = 0; k < 3; ++k) oxyz[3 * o + k] = xyz[3 * i + k];
        if (nrm)
            for (int k = 0; k < 3; ++k) onrm[3 * o + k] = nrm[3 * i + k];
        if (cov)
            for (int k = 0; k < 9; ++k) ocov[9 * o + k] = cov[9 * i + k];
    }
}
__global__ void k_vox_heads(const uint64_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ flags) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// one thread per voxel head: sequential sums in double over the voxel's points, which the stable sort left in
// ascending index order (= the insertion order of the reference's accumulator)
__global__ void k_vox_reduce(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t n,
                             const uint32_t* __restrict__ heads, const uint32_t* __restrict__ vox_id,
                             const double* __restrict__ xyz, const double* __restrict__ nrm, const double* __restrict__ cov,
                             int64_t base, double* __restrict__ oxyz, double* __restrict__ onrm, double* __restrict__ ocov) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n || !heads[i]) return;
    const uint64_t key = keys[i];
    double p[3] = {0, 0, 0}, nn[3] = {0, 0, 0}, cc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int cnt = 0;
    for (int64_t j = i; j < n && keys[j] == key; ++j) {
        const size_t s = vals[j];
        p[0] += xyz[3 * s];
        p[1] += xyz[3 * s + 1];
        p[2] += xyz[3 * s + 2];
        if (nrm) {
            const double a = nrm[3 * s], b = nrm[3 * s + 1], c = nrm[3 * s + 2];
            if (!(a != a) && !(b != b) && !(c != c)) {
                nn[0] += a;
                nn[1] += b;
                nn[2] += c;
            }
        }
        if (cov)
            for (int k = 0; k < 9; ++k) cc[k] += cov[9 * s + k];
        ++cnt;
    }
    const size_t o = (size_t)base + vox_id[i];
    const double dc = (double)cnt;
    for (int k = 0; k < 3; ++k) oxyz[3 * o + k] = p[k] / dc;
    if (nrm) {
        double a[3] = {nn[0] / dc, nn[1] / dc, nn[2] / dc};
        double u = a[0] * a[0];
        double v = a[1] * a[1];
        double z2 = u + v;
        u = a[2] * a[2];
        z2 = z2 + u;
        if (z2 > 0.0) {   // Eigen normalized(): the zero vector stays zero
            const double r = sqrt(z2);
            a[0] = a[0] / r;
            a[1] = a[1] / r;
            a[2] = a[2] / r;
        }
        for (int k = 0; k < 3; ++k) onrm[3 * o + k] = a[k];
    }
    if (cov)
        for (int k = 0; k < 9; ++k) ocov[9 * o + k] = cc[k] / dc;
}

// ---- R8x first-iteration analysis (ICP.cpp:2187-2444): matched pairs -> data frame, centre, alignment sums ----
// Vectors in fp32 with one rounding per operation (numeric contract), sums in fp64.
__device__ __forceinline__ float3 xicp_to_data_frame_point(const float* Trd, const float3 p) {
    const float q0 = p.x - Trd[3], q1 = p.y - Trd[7], q2 = p.z - Trd[11];
    float3 r;
    float a0, a1, a2, sacc;
    a0 = Trd[0] * q0; a1 = Trd[4] * q1; a2 = Trd[8] * q2; sacc = a0 + a1; r.x = sacc + a2;
    a0 = Trd[1] * q0; a1 = Trd[5] * q1; a2 = Trd[9] * q2; sacc = a0 + a1; r.y = sacc + a2;
    a0 = Trd[2] * q0; a1 = Trd[6] * q1; a2 = Trd[10] * q2; sacc = a0 + a1; r.z = sacc + a2;
    return r;
}
__device__ __forceinline__ float3 xicp_to_data_frame_vec(const float* Trd, const float x, const float y, const float z) {
    float3 r;
    float a0, a1, a2, sacc;
    a0 = Trd[0] * x; a1 = Trd[4] * y; a2 = Trd[8] * z; sacc = a0 + a1; r.x = sacc + a2;
    a0 = Trd[1] * x; a1 = Trd[5] * y; a2 = Trd[9] * z; sacc = a0 + a1; r.y = sacc + a2;
    a0 = Trd[2] * x; a1 = Trd[6] * y; a2 = Trd[10] * z; sacc = a0 + a1; r.z = sacc + a2;
    return r;
}

template <int NV>
__device__ __forceinline__ void xicp_block_add(double* v, double* dst) {
    __shared__ double red[4][NV];
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < NV) {
        const double t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (t != 0.0) unsafeAtomicAdd(&dst[threadIdx.x], t);
    }
}

__global__ void __launch_bounds__(256)
k_xicp_center(const float4* __restrict__ src, int64_t n, const IterState* __restrict__ it, const int* __restrict__ pos,
              const float* __restrict__ w, XicpState* __restrict__ xs) {
    if (it->done || it->xicp_stage != 2) return;
    const Xf T = load_xf(it);
    float Trd[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) Trd[k] = it->xicp_Trd[k];
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (pos[i] < 0 || w[i] == 0.f) continue;
        const float4 s = src[i];
        const float3 ps = xicp_to_data_frame_point(Trd, xf_point(T, s.x, s.y, s.z));
        v[0] += (double)ps.x;
        v[1] += (double)ps.y;
        v[2] += (double)ps.z;
        v[3] += 1.0;
    }
    xicp_block_add<4>(v, xs->center);
}

__global__ void __launch_bounds__(256)
k_xicp_detect(const float4* __restrict__ src, int64_t n, const IterState* __restrict__ it, const int* __restrict__ pos,
              const float* __restrict__ w, const float4* __restrict__ tgt_nrm, XicpState* __restrict__ xs) {
    if (it->done || it->xicp_stage != 2) return;
    const Xf T = load_xf(it);
    float Trd[12], vr[9], vt[9];
#pragma unroll
    for (int k = 0; k < 12; ++k) Trd[k] = it->xicp_Trd[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        vr[k] = xs->vr[k];
        vt[k] = xs->vt[k];
    }
    const double cnt = xs->center[3];
    float c[3] = {0.f, 0.f, 0.f};
    if (cnt > 0.0) {
        c[0] = (float)(xs->center[0] / cnt);
        c[1] = (float)(xs->center[1] / cnt);
        c[2] = (float)(xs->center[2] / cnt);
    }
    const float cos_min = it->xicp_cos_min, cos_strong = it->xicp_cos_strong;
    double v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = 0.0;   // comb[0..5], high[0..5]
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = pos[i];
        if (j < 0 || w[i] == 0.f) continue;
        const float4 s = src[i];
        float3 ps = xicp_to_data_frame_point(Trd, xf_point(T, s.x, s.y, s.z));
        ps.x = ps.x - c[0];
        ps.y = ps.y - c[1];
        ps.z = ps.z - c[2];
        const float4 nr = tgt_nrm[j];
        const float3 nn = xicp_to_data_frame_vec(Trd, nr.x, nr.y, nr.z);
        float cr[3];
        float u, q;
        u = ps.y * nn.z; q = ps.z * nn.y; cr[0] = u - q;
        u = ps.z * nn.x; q = ps.x * nn.z; cr[1] = u - q;
        u = ps.x * nn.y; q = ps.y * nn.x; cr[2] = u - q;
        float a = cr[0] * cr[0], b2 = cr[1] * cr[1];
        float s2 = a + b2;
        a = cr[2] * cr[2];
        s2 = s2 + a;
        const float nrm = sqrtf(s2);
        if (!(nrm < 1.0f)) {
            cr[0] = cr[0] / nrm;
            cr[1] = cr[1] / nrm;
            cr[2] = cr[2] / nrm;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a0 = cr[0] * vr[3 * k], a1 = cr[1] * vr[3 * k + 1], a2 = cr[2] * vr[3 * k + 2];
            float sacc = a0 + a1;
            const float ar = fabsf(sacc + a2);
            a0 = nn.x * vt[3 * k];
            a1 = nn.y * vt[3 * k + 1];
            a2 = nn.z * vt[3 * k + 2];
            sacc = a0 + a1;
            const float at = fabsf(sacc + a2);
            if (ar > cos_min) v[k] += (double)ar;
            if (ar > cos_strong) v[6 + k] += (double)ar;
            if (at > cos_min) v[3 + k] += (double)at;
            if (at > cos_strong) v[9 + k] += (double)at;
        }
    }
    xicp_block_add<12>(v, xs->comb);   // comb[6] and high[6] are contiguous
}

__global__ void __launch_bounds__(64)
k_pack_contrib(double* __restrict__ acc, const IterState* __restrict__ it, float* __restrict__ contrib) {
    const int c = threadIdx.x;
    double* hdr = reinterpret_cast<double*>(contrib);
    if (it->done || it->stall) {
        if (c < kSums) hdr[c] = 0.0;
        if (c == 0) reinterpret_cast<uint32_t*>(contrib)[64] = 0u;
        return;
    }
    if (c < kSums) {
        double t = 0;
        for (int r = 0; r < kAccRows; ++r) {
            t += acc[(size_t)r * kSums + c];
            acc[(size_t)r * kSums + c] = 0.0;
        }
        hdr[c] = t;
    }
    if (c == 0) reinterpret_cast<uint32_t*>(contrib)[64] = it->band_count;
}

// Workgroup barrier that only waits for LDS traffic: global loads issued earlier stay in flight across it
// (__syncthreads() drains vmcnt(0) first -- cdna_hip_programming.md, "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

__global__ void __launch_bounds__(1024)
k_reduce_update(const double* __restrict__ partials, int n_blocks, IterState* it, HostMirror* host,
                unsigned long long seq, int fused, const float* __restrict__ band, float* __restrict__ w_out,
                const SelectState* __restrict__ sel, const float* __restrict__ gathered, int n_ranks, int my_rank,
                XicpState* __restrict__ xs) {
    // fused: 0 = select-based iteration, 1 = fused iteration (band verification), 2 = R8x finish: the sums are
    // already in the state (first-iteration localizability analysis done in between), only solve + update
    const bool finish = fused == 2;
    if (finish) fused = 0;
    __shared__ double sh[32][kSums];
    __shared__ double tot[kSums];
    __shared__ uint32_t hist[2048 + 64];
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t pick[2];
    __shared__ float s_limit;
    __shared__ uint32_t bd2[kBandCap];
    __shared__ __attribute__((aligned(16))) uint32_t mir_w[(sizeof(HostMirror) + 3) / 4];
    __shared__ uint32_t small[64];
    __shared__ uint32_t s_cnt, s_csel, s_need_radix;
    __shared__ uint32_t rk_off[65];   // multi-GPU: first global band index of every rank's records (+ total)
    __shared__ uint32_t rk_bad;
    __shared__ int s_skip_mirror;
    // The whole iteration state is staged in LDS by one coalesced load (every separate `it->` access below would
    // cost an L2 round trip on a single lane); wave 0 writes the modified copy back at the end.  The accumulator
    // rows do not depend on the state, so their loads are issued in the same batch.
    constexpr int kStateWords = (int)(sizeof(IterState) / 4);
    static_assert(sizeof(IterState) % 4 == 0 && kStateWords <= 1024, "IterState must be a whole number of words");
    __shared__ __attribute__((aligned(16))) uint32_t s_state[kStateWords];
    IterState* const sit = reinterpret_cast<IterState*>(s_state);
    const int comp = threadIdx.x & (kSums - 1), part = threadIdx.x / kSums;  // 32 parts x 32 comps
    if (threadIdx.x < kStateWords) s_state[threadIdx.x] = reinterpret_cast<const uint32_t*>(it)[threadIdx.x];
    double t = 0;
    if (!gathered && !finish) {
        const int n_rows = fused ? kAccRows : n_blocks;
        for (int b = part; b < n_rows; b += 32) t += partials[(size_t)b * kSums + comp];
    }
    __syncthreads();
    const int s_done = sit->done, s_stall = sit->stall, s_use_trim = sit->use_trim;
    const float s_ratio = sit->trim_ratio, s_band_lo = sit->band_lo, s_band_hi = sit->band_hi;
    uint32_t s_band_count = sit->band_count;
    bool band_bad = false;   // a band buffer overflowed: the prediction cannot be verified
    if (s_done) return;
    if (fused && s_stall) return;
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
    if (gathered) {
        // every rank sees the same blocks in the same order -> identical results everywhere, no broadcast needed
        if (threadIdx.x == 0) {
            uint32_t off = 0, bad = 0;
            for (int r = 0; r < n_ranks; ++r) {
                rk_off[r] = off;
                const uint32_t cnt = reinterpret_cast<const uint32_t*>(gathered + (size_t)r * kContribFloats)[64];
                if (cnt > (uint32_t)kContribCap) bad = 1;
                off += min(cnt, (uint32_t)kContribCap);
            }
            rk_off[n_ranks] = off;
            rk_bad = bad;
        }
        __syncthreads();
        s_band_count = rk_off[n_ranks];
        band_bad = rk_bad != 0;
    }
    if (!gathered && s_band_count > (uint32_t)kBandCap) band_bad = true;
    // record (i, component c) of the band, whichever buffer holds it
    auto rec = [&](uint32_t i, int c) -> float {
        if (!gathered) return band[band_at(c, i)];
        int r = 0;
        while (r + 1 < n_ranks && i >= rk_off[r + 1]) ++r;
        return gathered[(size_t)r * kContribFloats + kContribHdr + (size_t)(i - rk_off[r]) * kRec + c];
    };
    // fused path: issue this thread's band-record loads right away (they only depend on the record count); the
    // barriers below are LDS-only, so the loads stay in flight behind the partial sums
    const bool trim = s_use_trim && s_ratio != 1.0f;
    const uint32_t n_band = (fused && trim && !band_bad) ? s_band_count : 0u;
    const bool add_comp = comp != 29 && comp != 31;
    float pre[16];
    uint32_t my_d2[kBandCap / 1024];
    if (n_band) {
#pragma unroll
        for (int u = 0; u < kBandCap / 1024; ++u) {
            const uint32_t i = threadIdx.x + 1024u * u;
            my_d2[u] = i < n_band ? __float_as_uint(rec(i, 29)) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t i = min((uint32_t)part + 32u * u, n_band - 1);
            pre[u] = add_comp ? rec(i, comp) : 0.f;
        }
    }
    if (gathered) {
        for (int r = part; r < n_ranks; r += 32)
            t += reinterpret_cast<const double*>(gathered + (size_t)r * kContribFloats)[comp];
    } else if (fused) {
        for (int b = part; b < kAccRows; b += 32)
            const_cast<double*>(partials)[(size_t)b * kSums + comp] = 0.0;   // ready for the next iteration
    }
    // 32 parts -> 1: the two parts of a wave by one shuffle, the 16 waves through LDS (fixed order: deterministic)
    t += __shfl_xor(t, 32);
    if ((threadIdx.x & 63) < 32) sh[threadIdx.x >> 6][comp] = t;
    if (threadIdx.x == 0) {
        s_limit = INFINITY;
    }
    lds_barrier();
    if (threadIdx.x < kSums) {
        double s = 0;
#pragma unroll
        for (int p = 0; p < 16; ++p) s += sh[p][threadIdx.x];
        tot[threadIdx.x] = finish ? sit->sums[threadIdx.x] : s;
    }
    if (threadIdx.x == 0) s_skip_mirror = 0;
    lds_barrier();
    const unsigned long long stA = __builtin_amdgcn_s_memtime();
    unsigned long long stB = stA, stC = stA, sx1 = stA, sx2 = stA, sx3 = stA;
    if (fused && trim) {
        // ---- verify the predicted band with exact counts, then select the exact quantile inside it
        const uint32_t n_finite = (uint32_t)llround(tot[29]), n_below = (uint32_t)llround(tot[31]);
        const uint32_t k = trim_rank(n_finite, s_ratio);
        const bool ok = n_finite == 0 || (!band_bad && n_below <= k && k < n_below + n_band);
        if (!ok) {
            if (threadIdx.x == 0) {
                it->stall = 1;
                it->band_count = 0;
                host->stall = 1;
                host->band_count = (int)s_band_count;
                host->iterations = sit->iterations;
                host->done = 0;
                __threadfence_system();
                __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        if (n_finite != 0) {
            // stage the band's d2 bit patterns (loaded at kernel start) in LDS
#pragma unroll
            for (int u = 0; u < kBandCap / 1024; ++u) {
                const uint32_t i = threadIdx.x + 1024u * u;
                if (i < n_band) bd2[i] = my_d2[u];
            }
            // One-level select: the band's values lie in [band_lo, band_hi), so the order-preserving key
            // (u - u_lo) * 2048 / (u_hi - u_lo) spreads them over 2048 bins (about one value per bin); the bin that
            // holds rank r is then resolved by direct ranking.  Crowded bin (> 64 equal-ish values): radix levels.
            const uint32_t u_lo = __float_as_uint(s_band_lo), u_hi = __float_as_uint(s_band_hi);
            // order-preserving key without integer division: trunc(double(u - u_lo) * 2048 / span) (monotone in u)
            const double kscale = 2048.0 / (double)(u_hi > u_lo ? u_hi - u_lo : 1u);
            uint32_t rank = k - n_below, prefix = 0;
            sx1 = __builtin_amdgcn_s_memtime();
            for (int i = threadIdx.x; i < 2048 + 64; i += 1024) hist[i] = 0;
            if (threadIdx.x == 0) {
                s_cnt = 0;
                s_need_radix = 0;
            }
            lds_barrier();   // LDS-only barrier: the record loads issued above stay in flight
            for (uint32_t i = threadIdx.x; i < n_band; i += 1024) {
                const uint32_t key = (uint32_t)((double)(bd2[i] - u_lo) * kscale);
                const uint32_t kk = min(key, 2047u);
                atomicAdd(&hist[kk + (kk >> 5)], 1u);   // +1 pad per 32 bins: lane-contiguous reads below are conflict-free
            }
            lds_barrier();   // LDS-only barrier: the record loads issued above stay in flight
            sx2 = __builtin_amdgcn_s_memtime();
            {
                // block-wide pick (2 bins per thread, padded index): exclusive scan of the 2048 counts
                const uint32_t b0 = 2u * threadIdx.x, b1 = b0 + 1u;
                const uint32_t h0 = hist[b0 + (b0 >> 5)], h1 = hist[b1 + (b1 >> 5)];
                const uint32_t loc = h0 + h1;
                uint32_t incl = loc;
                const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t v = __shfl_up(incl, o);
                    if (ln >= o) incl += v;
                }
                if (ln == 63) wave_tot[wv] = incl;
                lds_barrier();
                uint32_t base = 0;
#pragma unroll
                for (int w = 0; w < 16; ++w) base += (w < wv) ? wave_tot[w] : 0u;
                const uint32_t excl = base + incl - loc;
                if (loc && rank >= excl && rank < excl + loc) {   // exactly one thread
                    const bool first = rank < excl + h0;
                    pick[0] = first ? b0 : b1;
                    pick[1] = first ? rank - excl : rank - excl - h0;
                    s_csel = first ? h0 : h1;
                }
                lds_barrier();
                const uint32_t bsel = pick[0], rsel = pick[1], csel = s_csel;
                if (csel <= 64u) {
                    // gather the picked bin's values (all threads), rank them directly (wave 0)
                    for (uint32_t i = threadIdx.x; i < n_band; i += 1024) {
                        const uint32_t key = min((uint32_t)((double)(bd2[i] - u_lo) * kscale), 2047u);
                        if (key == bsel) small[atomicAdd(&s_cnt, 1u)] = bd2[i];
                    }
                    lds_barrier();
                    if (threadIdx.x < csel) {
                        const uint32_t e = small[threadIdx.x];
                        uint32_t rr = 0;
                        for (uint32_t j = 0; j < csel; ++j) {
                            const uint32_t o = small[j];
                            rr += (o < e || (o == e && j < threadIdx.x)) ? 1u : 0u;
                        }
                        if (rr == rsel) s_limit = __uint_as_float(e);
                    }
                } else if (threadIdx.x == 0) {
                    s_need_radix = 1;
                }
            }
            lds_barrier();   // LDS-only barrier: the record loads issued above stay in flight
            sx3 = __builtin_amdgcn_s_memtime();
            if (s_need_radix) {   // crowded bin (many equal distances): plain 3-level radix select, all threads
            for (int level = 0; level < 3; ++level) {
                for (int i = threadIdx.x; i < 2048; i += 1024) hist[i] = 0;
                __syncthreads();
                const uint32_t mask = level == 0 ? 0u : (level == 1 ? 0xffe00000u : 0xfffffc00u);
                for (uint32_t i = threadIdx.x; i < n_band; i += 1024) {
                    const uint32_t u = bd2[i];
                    if ((u & mask) == prefix)
                        atomicAdd(&hist[level == 0 ? (u >> 21) : (level == 1 ? ((u >> 10) & 2047u) : (u & 1023u))], 1u);
                }
                __syncthreads();
                block_pick1024(hist, level == 2 ? 1024 : 2048, rank, wave_tot, pick);
                prefix |= pick[0] << (level == 0 ? 21 : (level == 1 ? 10 : 0));
                rank = pick[1];
                __syncthreads();
            }
            if (threadIdx.x == 0) s_limit = __uint_as_float(prefix);
            __syncthreads();
            }
            stB = __builtin_amdgcn_s_memtime();
            const float limit = s_limit;
            // ---- add the band records that survive the trim (component-wise, 32 parts)
            double acc = 0;
            if (add_comp) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t i = (uint32_t)part + 32u * u;
                    if (i < n_band && __uint_as_float(bd2[i]) <= limit) acc += (double)pre[u];
                }
                for (uint32_t i0 = part + 32u * 16u; i0 < n_band; i0 += 32 * 16) {   // only when n_band > 512
                    float vv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const uint32_t i = min(i0 + 32u * u, n_band - 1);
                        vv[u] = rec(i, comp);
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const uint32_t i = i0 + 32u * u;
                        if (i < n_band && __uint_as_float(bd2[i]) <= limit) acc += (double)vv[u];
                    }
                }
            }
            acc += __shfl_xor(acc, 32);
            if ((threadIdx.x & 63) < 32) sh[threadIdx.x >> 6][comp] = acc;
            if (w_out)
                for (uint32_t i = threadIdx.x; i < n_band; i += 1024) {
                    if (!(__uint_as_float(bd2[i]) <= limit)) {
                        if (!gathered) {
                            w_out[__float_as_int(rec(i, 31))] = 0.f;
                        } else if (i >= rk_off[my_rank] && i < rk_off[my_rank + 1]) {
                            w_out[__float_as_int(rec(i, 31))] = 0.f;   // only this rank's own points
                        }
                    }
                }
            __syncthreads();
            if (threadIdx.x < kSums) {
                double s2 = 0;
#pragma unroll
                for (int p = 0; p < 16; ++p) s2 += sh[p][threadIdx.x];
                tot[threadIdx.x] += s2;
            }
            __syncthreads();
            stC = __builtin_amdgcn_s_memtime();
        }
    } else if (!fused && trim && sel) {
        if (threadIdx.x == 0) s_limit = sel->limit;
        __syncthreads();
    }
    if (threadIdx.x < kSums) sit->sums[threadIdx.x] = tot[threadIdx.x];
    if (threadIdx.x >= 64) return;   // the rest is wave 0 only (wave-synchronous: no workgroup barriers below)
    const int lane = threadIdx.x;
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
    unsigned long long st2 = st1, st3 = st1;
    const bool p2pl = sit->cost == REG_COST_P2PL;
    // ---- R8: 6x6 solve by Gauss-Jordan elimination on the augmented 6x7 system, one entry per lane (fp64).
    // P2PL: A, b are first rounded to fp32 (the reference hands fp32 matrices to its fp64 solver).
    const int r = lane >> 3, c = lane & 7;
    double a = 0.0;
    if (r < 6 && c < 7) {
        if (c < 6) {
            const int lo = r < c ? r : c, hi = r < c ? c : r;
            const int k = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);   // index into the packed upper triangle
            a = p2pl ? (double)(float)tot[k] : tot[k];
        } else {
            a = p2pl ? (double)(-(float)tot[21 + r]) : -tot[21 + r];
        }
    }
    const double a_orig = a;
    double dmax = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) dmax = fmax(dmax, fabs(__shfl(a_orig, j * 8 + j)));
    bool well = dmax > 0.0;
    const double piv_thr = (p2pl ? 1e-4 : 1e-10) * dmax;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double pj = __shfl(a, j * 8 + j);
        well = well && (pj > piv_thr);
        const double ajc = __shfl(a, j * 8 + c);
        const double arj = __shfl(a, r * 8 + j);
        const double q = ajc / pj;
        a = (r == j) ? q : a - arj * q;
    }
    double xsol[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) xsol[i] = __shfl(a, i * 8 + 6);
    st2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        // band for the next iteration from the limits seen so far
        const float limit = finish ? sit->limit_last : s_limit;
        if (!finish) {
            sit->limit_prev = sit->limit_last;
            sit->limit_last = limit;
        }
        if (finish) {
            // keep the band computed when the sums were reduced
        } else if (!trim || !(limit < INFINITY)) {
            sit->band_lo = INFINITY;   // no trimming / nothing to predict from: every finite match is "certainly kept"
            sit->band_hi = INFINITY;
        } else {
            const float prev = sit->limit_prev;
            float m = 0.3f;
            if (prev < INFINITY && prev > 0.f) m = fminf(fmaxf(2.0f * fabsf(limit - prev) / limit + 0.003f, 0.003f), 0.6f);
            if (sit->debug_narrow_band) m = 1e-7f;   // test hook: forces band mispredictions (stall + repair path)
            sit->band_lo = limit * (1.0f - m);
            sit->band_hi = limit * (1.0f + m);
        }
        const int nband_report = (int)sit->band_count;
        sit->band_count = 0;
        sit->stall = 0;
        bool do_update = sit->update != 0;
        if (!finish && do_update && p2pl && sit->xicp_stage == 1 && tot[28] != 0.0 && xs) {
            // R8x, first iteration: eigen-directions of the rotation / translation blocks of A, expressed in the frame
            // the data came from; the analysis kernels that follow collect the information sums, then this kernel
            // runs again (finish) to decide, solve and update.  Nothing is reported to the host yet.
            float H[36];
            int k = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 6; ++j) {
                    const float v = (float)tot[k++];
                    H[6 * i + j] = v;
                    H[6 * j + i] = v;
                }
            double Vr[9], Vt[9];
            xicp_eigvecs(H, Vr, Vt);
            for (int kk = 0; kk < 3; ++kk)
                for (int rr = 0; rr < 3; ++rr) {
                    float a0 = sit->xicp_Trd[rr] * (float)Vr[kk], a1 = sit->xicp_Trd[4 + rr] * (float)Vr[3 + kk];
                    float a2 = sit->xicp_Trd[8 + rr] * (float)Vr[6 + kk];
                    float sacc = a0 + a1;
                    xs->vr[3 * kk + rr] = sacc + a2;
                    a0 = sit->xicp_Trd[rr] * (float)Vt[kk];
                    a1 = sit->xicp_Trd[4 + rr] * (float)Vt[3 + kk];
                    a2 = sit->xicp_Trd[8 + rr] * (float)Vt[6 + kk];
                    sacc = a0 + a1;
                    xs->vt[3 * kk + rr] = sacc + a2;
                }
            for (int i = 0; i < 4; ++i) xs->center[i] = 0.0;
            for (int i = 0; i < 6; ++i) {
                xs->comb[i] = 0.0;
                xs->high[i] = 0.0;
            }
            sit->xicp_stage = 2;
            do_update = false;
            s_skip_mirror = 1;
        }
        if (finish && xs) {
            int nc = 0;
            for (int i = 0; i < 6; ++i) {
                const int ok = (xs->comb[i] >= (double)sit->xicp_enough || xs->high[i] >= (double)sit->xicp_insufficient) ? 1 : 0;
                sit->xicp_flags[i] = ok;
                sit->xicp_comb[i] = xs->comb[i];
                sit->xicp_high[i] = xs->high[i];
                nc += ok ? 0 : 1;
            }
            sit->xicp_nc = nc;
            sit->xicp_stage = 0;
        }
        if (do_update) {
            if (tot[28] == 0.0) {
                sit->status = REG_NO_CORRESPONDENCES;
                sit->done = 1;
            } else if (p2pl) {
                float x[6], dT[16], Tn[16];
                int rank = 6;
                if (sit->xicp_nc > 0) {
                    // R8x: no update along the non-localizable eigen-directions of the CURRENT A (PointToPlane.cpp:459-505)
                    float H[36], b6[6];
                    int k = 0;
                    for (int i = 0; i < 6; ++i)
                        for (int j = i; j < 6; ++j) {
                            const float v = (float)tot[k++];
                            H[6 * i + j] = v;
                            H[6 * j + i] = v;
                        }
                    for (int i = 0; i < 6; ++i) b6[i] = -(float)tot[21 + i];
                    rank = solve6_xicp(H, b6, sit->xicp_flags, x);
                } else if (well) {
                    for (int i = 0; i < 6; ++i) x[i] = (float)xsol[i];
                } else {
                    // ill-conditioned / rank deficient: eigen-solve with the fp32 rank threshold (minimum norm)
                    float H[36], b6[6];
                    int k = 0;
                    for (int i = 0; i < 6; ++i)
                        for (int j = i; j < 6; ++j) {
                            const float v = (float)tot[k++];
                            H[6 * i + j] = v;
                            H[6 * j + i] = v;
                        }
                    for (int i = 0; i < 6; ++i) b6[i] = -(float)tot[21 + i];
                    rank = solve6_p2pl(H, b6, x);
                }
                sit->rank_last = rank;
                x_to_T(x, dT);
                m4_mul(dT, sit->T, Tn);  // T_iter = real * T_iter (ICP.cpp:1213-1215)
                for (int i = 0; i < 16; ++i) sit->T[i] = Tn[i];
                sit->iterations += 1;
                bool iterate;
                if (sit->fixed_iters > 0)
                    iterate = sit->iterations < sit->fixed_iters;
                else
                    iterate = sit->chk.check(Tn);
                if (!iterate) sit->done = 1;
            } else {
                double dl[6], E[16], Tn[16];
                int rank = 6;
                if (well) {
                    for (int i = 0; i < 6; ++i) dl[i] = xsol[i];
                } else {
                    double Hd[36], g[6];
                    int k = 0;
                    for (int i = 0; i < 6; ++i)
                        for (int j = i; j < 6; ++j) Hd[6 * i + j] = Hd[6 * j + i] = tot[k++];
                    for (int i = 0; i < 6; ++i) g[i] = -tot[21 + i];
                    rank = solve_sym6(Hd, g, dl, 1e-12);
                }
                sit->rank_last = rank;
                se3_exp(dl, E);
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        double v = 0;
                        for (int kk = 0; kk < 4; ++kk) v += sit->Td[4 * i + kk] * E[4 * kk + j];
                        Tn[4 * i + j] = v;
                    }
                for (int i = 0; i < 16; ++i) {
                    sit->Td[i] = Tn[i];
                    sit->T[i] = (float)Tn[i];
                }
                sit->iterations += 1;
                if (sit->fixed_iters > 0) {
                    if (sit->iterations >= sit->fixed_iters) sit->done = 1;
                } else {
                    const double dr = sqrt(dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2]);
                    const double dt = sqrt(dl[3] * dl[3] + dl[4] * dl[4] + dl[5] * dl[5]);
                    if (dr < (double)sit->gicp_rot_eps && dt < (double)sit->gicp_trans_eps) {
                        sit->chk.converged = true;
                        sit->done = 1;
                    } else if (sit->iterations >= sit->max_iter) {
                        sit->chk.max_iter_reached = true;
                        sit->done = 1;
                    }
                }
            }
        }
        st3 = __builtin_amdgcn_s_memtime();
        // stage the host mirror in LDS (word layout of HostMirror); the whole wave then writes it out
        HostMirror* m = reinterpret_cast<HostMirror*>(mir_w);
        for (int i = 0; i < 16; ++i) m->T[i] = sit->T[i];
        m->iterations = sit->iterations;
        m->done = sit->done;
        m->status = sit->status;
        m->rank_last = sit->rank_last;
        m->converged = sit->chk.converged ? 1 : 0;
        m->max_iter_reached = sit->chk.max_iter_reached ? 1 : 0;
        m->stall = 0;
        m->band_count = 0;
        m->limit_last = sit->limit_last;
        m->limit_prev = sit->limit_prev;
        m->band_lo = sit->band_lo;
        m->band_hi = sit->band_hi;
        m->pad_nband = nband_report;
        for (int i = 0; i < 6; ++i) {
            m->localizable[i] = sit->xicp_flags[i];
            m->xicp_comb[i] = sit->xicp_comb[i];
            m->xicp_high[i] = sit->xicp_high[i];
        }
        m->n_constraints = sit->xicp_nc;
        m->pad3 = 0;
        m->stamps[0] = st1 - st0;
        m->stamps[1] = st2 - st1;
        m->stamps[2] = st3 - st2;
        m->stamps[4] = stA - st0;
        m->stamps[5] = stB - stA;
        m->stamps[6] = stC - stB;
        m->stamps[3] = sx1 - stA;
        m->stamps[7] = ((sx2 - sx1) << 32) | (sx3 - sx2);
    }
    if (lane < kSums) reinterpret_cast<HostMirror*>(mir_w)->sums[lane] = tot[lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // LDS writes of lane 0 visible to the wave
    __builtin_amdgcn_wave_barrier();
    // write the modified state back (coalesced); the next kernel of the stream reads it from global memory
    for (int w = lane; w < kStateWords; w += 64) reinterpret_cast<uint32_t*>(it)[w] = s_state[w];
    if (s_skip_mirror) return;   // R8x analysis pending: the finish pass reports
    constexpr int kMirrorWords = (int)(offsetof(HostMirror, seq) / 4);
    uint32_t* hw = reinterpret_cast<uint32_t*>(host);
    for (int w = lane; w < kMirrorWords; w += 64) hw[w] = mir_w[w];
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Stream-ordered distributed path: this rank's workgroup partials -> 32 doubles (summed over ranks by the caller's
// all-reduce before the update kernel runs).
__global__ void __launch_bounds__(1024)
k_partials_sum(const double* __restrict__ partials, int n_blocks, double* __restrict__ out, const IterState* __restrict__ it) {
    __shared__ double sh[32][kSums];
    if (it->done) {
        if (threadIdx.x < kSums) out[threadIdx.x] = 0.0;   // a finished rank contributes nothing
        return;
    }
    const int comp = threadIdx.x & (kSums - 1), part = threadIdx.x / kSums;
    double t = 0;
    for (int b = part; b < n_blocks; b += 32) t += partials[(size_t)b * kSums + comp];
    sh[part][comp] = t;
    __syncthreads();
    if (threadIdx.x < kSums) {
        double s = 0;
        for (int p = 0; p < 32; ++p) s += sh[p][threadIdx.x];
        out[threadIdx.x] = s;
    }
}

// results back into the caller's order: out[perm[i]] = value of slot i
__global__ void k_ids_from_pos(const int* __restrict__ pos, const float4* __restrict__ tgt, int64_t n,
                               const uint32_t* __restrict__ perm, int32_t* ids) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = pos[i];
    ids[perm ? (int64_t)perm[i] : i] = p >= 0 ? (int32_t)__float_as_uint(tgt[p].w) : -1;
}
__global__ void k_unpermute_f32(const float* __restrict__ in, int64_t n, const uint32_t* __restrict__ perm,
                                float* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[perm ? (int64_t)perm[i] : i] = in[i];
}

// =================================================================================================
// Next row: surface normals / covariances by exact k-NN + PCA on the voxel-bin table
//   libpointmatcher/pointmatcher/DataPointsFilters/SurfaceNormal.cpp:152-252 (self k-NN incl. the point itself,
//   mean, C = NN NN^T, eigenvector of the smallest eigenvalue, clamp to [-1,1]);
//   orientation towards the sensor: open3d_slam/src/CloudRegistration.cpp:37.
// 16 lanes per point.  Per radius level the group gathers every point of the bin box (within max_dist) into an
// LDS list, then extracts the k smallest (d2, original index) one by one; the k-th distance <= rho^2 proves the
// list held every closer point (same exactness argument as the 1-NN search).
// =================================================================================================
constexpr int kPcaGroup = 16;
constexpr int kPcaCap = 256;     // candidates per point and level held in LDS (8 B each)
constexpr int kPcaMaxK = 32;

// Calls f(j, target point j, d2) on the lanes of one 16-lane group for every target point inside the bin box of
// level l around p that lies within max_dist.
template <class F>
__device__ __forceinline__ void pca_scan_box(const Grid& g, const float3 p, int l, int sub, int gbase, F&& f) {
    const float rb = g.rho_box[l];
    const int lox = (int)fminf(fmaxf(bin_coord_f(p.x - rb, g.ox, g.inv_c), 0.f), g.dimx - 1.f);
    const int loy = (int)fminf(fmaxf(bin_coord_f(p.y - rb, g.oy, g.inv_c), 0.f), g.dimy - 1.f);
    const int loz = (int)fminf(fmaxf(bin_coord_f(p.z - rb, g.oz, g.inv_c), 0.f), g.dimz - 1.f);
    const int hix = (int)fminf(fmaxf(bin_coord_f(p.x + rb, g.ox, g.inv_c), 0.f), g.dimx - 1.f);
    const int hiy = (int)fminf(fmaxf(bin_coord_f(p.y + rb, g.oy, g.inv_c), 0.f), g.dimy - 1.f);
    const int hiz = (int)fminf(fmaxf(bin_coord_f(p.z + rb, g.oz, g.inv_c), 0.f), g.dimz - 1.f);
    const int ny = hiy - loy + 1, nz = hiz - loz + 1;
    const int bx0 = lox >> kBrickLog2;
    const int nbx = (hix >> kBrickLog2) - bx0 + 1;
    const int nrow = nbx * ny;
    const int64_t total = (int64_t)nrow * nz;
    for (int64_t base = 0; base < total; base += kPcaGroup) {
        uint32_t s = 0, e = 0;
        const int64_t t = base + sub;
        if (t < total) {
            const int iz = (int)(t / nrow), rem = (int)(t - (int64_t)iz * nrow);
            const int iy = rem / nbx, ix = rem - iy * nbx;
            const int bx = bx0 + ix, cy = loy + iy, cz = loz + iz;
            const int bid = brick_lookup(g, bx, cy >> kBrickLog2, cz >> kBrickLog2);
            if (bid >= 0) {
                const int x0 = max(lox, bx << kBrickLog2) & (kBrickDim - 1);
                const int x1 = min(hix, (bx << kBrickLog2) + kBrickDim - 1) & (kBrickDim - 1);
                const uint32_t* cs = g.cell_start + (size_t)bid * kBrickCells +
                                     (((cz & (kBrickDim - 1)) << (2 * kBrickLog2)) |
                                      ((cy & (kBrickDim - 1)) << kBrickLog2));
                s = cs[x0];
                e = cs[x1 + 1];
            }
        }
        unsigned mask = (unsigned)((__ballot(e > s) >> gbase) & 0xffffull);
        while (mask) {
            const int it = __ffs((int)mask) - 1;
            mask &= mask - 1;
            const uint32_t si = (uint32_t)__shfl((int)s, gbase + it);
            const uint32_t ei = (uint32_t)__shfl((int)e, gbase + it);
            for (uint32_t j = si + (uint32_t)sub; j < ei; j += kPcaGroup) {
                const float4 tpt = g.pts[j];
                const float dx = p.x - tpt.x, dy = p.y - tpt.y, dz = p.z - tpt.z;
                float a = dx * dx;
                float b = dy * dy;
                float d2 = a + b;
                a = dz * dz;
                d2 = d2 + a;
                if (d2 <= g.max_d2) f(j, tpt, d2);
            }
        }
    }
}

__global__ void __launch_bounds__(256)
k_knn_pca(Grid g, const float* __restrict__ raw_xyz, int64_t raw_stride, int64_t n, int k, int start_level, float vx,
          float vy, float vz, int has_vp, int regularise, float* __restrict__ normals, float* __restrict__ eigvals,
          float* __restrict__ covs, int32_t* __restrict__ ids_out, uint32_t* __restrict__ n_overflow) {
    constexpr int GP = 256 / kPcaGroup;   // points per workgroup
    __shared__ float l_d2[GP][kPcaCap];
    __shared__ uint32_t l_idx[GP][kPcaCap];
    __shared__ uint32_t l_cnt[GP];
    __shared__ uint32_t nb_idx[GP][kPcaMaxK];
    __shared__ float nb_xyz[GP][kPcaMaxK][3];
    const int grp = threadIdx.x / kPcaGroup, sub = threadIdx.x & (kPcaGroup - 1);
    const int gbase = (int)(threadIdx.x & 63) & ~(kPcaGroup - 1);
    const int64_t q = blockIdx.x * (int64_t)GP + grp;
    if (q >= n) return;   // whole groups leave together; nothing below synchronises across groups
    const float4 me = g.pts[q];
    const float3 p = make_float3(me.x, me.y, me.z);
    const uint32_t my_idx = __float_as_uint(me.w);
    int m = 0;
    bool overflow = false;
    for (int l = min(start_level, g.n_levels - 1); l < g.n_levels; ++l) {
        if (sub == 0) l_cnt[grp] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        pca_scan_box(g, p, l, sub, gbase, [&](uint32_t, const float4& tpt, float d2) {
            const uint32_t slot = atomicAdd(&l_cnt[grp], 1u);
            if (slot < (uint32_t)kPcaCap) {
                l_d2[grp][slot] = d2;
                l_idx[grp][slot] = __float_as_uint(tpt.w);
            }
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint32_t cnt = l_cnt[grp];
        const bool listed = cnt <= (uint32_t)kPcaCap;
        if (!listed) overflow = true;   // too many candidates for LDS: every extraction round rescans the box
        // extract the k smallest (d2, idx), ascending
        float last_d2 = -1.f;
        uint32_t last_idx = 0;
        m = 0;
        for (int r = 0; r < k; ++r) {
            float bd = INFINITY;
            uint32_t bi = 0xffffffffu;
            if (listed) {
                for (uint32_t t2 = sub; t2 < cnt; t2 += kPcaGroup) {
                    const float d = l_d2[grp][t2];
                    const uint32_t ix = l_idx[grp][t2];
                    const bool after = r == 0 || d > last_d2 || (d == last_d2 && ix > last_idx);
                    if (after && (d < bd || (d == bd && ix < bi))) {
                        bd = d;
                        bi = ix;
                    }
                }
            } else {
                pca_scan_box(g, p, l, sub, gbase, [&](uint32_t, const float4& tpt, float d) {
                    const uint32_t ix = __float_as_uint(tpt.w);
                    const bool after = r == 0 || d > last_d2 || (d == last_d2 && ix > last_idx);
                    if (after && (d < bd || (d == bd && ix < bi))) {
                        bd = d;
                        bi = ix;
                    }
                });
            }
#pragma unroll
            for (int x = 1; x < kPcaGroup; x <<= 1) {
                const float od = __shfl_xor(bd, x);
                const uint32_t oi = (uint32_t)__shfl_xor((int)bi, x);
                if (od < bd || (od == bd && oi < bi)) {
                    bd = od;
                    bi = oi;
                }
            }
            if (bi == 0xffffffffu) break;
            if (sub == 0) nb_idx[grp][r] = bi;
            last_d2 = bd;
            last_idx = bi;
            ++m;
        }
        const float r2 = g.rho[l] * g.rho[l];
        if ((m == k && last_d2 <= r2) || l == g.n_levels - 1) break;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // neighbour coordinates (input frame == table frame: the workspace table is not centred) and ids, in parallel
    for (int r = sub; r < k; r += kPcaGroup) {
        if (r < m) {
            const uint32_t ix = nb_idx[grp][r];
            const float* s = raw_xyz + (size_t)ix * raw_stride;
            nb_xyz[grp][r][0] = s[0];
            nb_xyz[grp][r][1] = s[1];
            nb_xyz[grp][r][2] = s[2];
            if (ids_out) ids_out[(size_t)my_idx * k + r] = (int32_t)ix;
        } else if (ids_out) {
            ids_out[(size_t)my_idx * k + r] = -1;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (sub != 0) return;
    if (overflow && n_overflow) atomicAdd(n_overflow, 1u);   // statistics only: the result is still exact
    // PCA: fp32 sequential sums in neighbour order (numeric contract), eigen-decomposition in fp64
    float mean[3] = {0.f, 0.f, 0.f};
    for (int r = 0; r < m; ++r) {
        mean[0] = mean[0] + nb_xyz[grp][r][0];
        mean[1] = mean[1] + nb_xyz[grp][r][1];
        mean[2] = mean[2] + nb_xyz[grp][r][2];
    }
    const float fm = (float)m;
    if (m > 0) {
        mean[0] = mean[0] / fm;
        mean[1] = mean[1] / fm;
        mean[2] = mean[2] / fm;
    }
    float C[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < m; ++r) {
        const float dx = nb_xyz[grp][r][0] - mean[0], dy = nb_xyz[grp][r][1] - mean[1], dz = nb_xyz[grp][r][2] - mean[2];
        float u;
        u = dx * dx; C[0] = C[0] + u;
        u = dx * dy; C[1] = C[1] + u;
        u = dx * dz; C[2] = C[2] + u;
        u = dy * dy; C[3] = C[3] + u;
        u = dy * dz; C[4] = C[4] + u;
        u = dz * dz; C[5] = C[5] + u;
    }
    double M[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]}, V[9], lam[3];
    jacobi_eig_sym3(M, V, lam);
    int o0 = 0, o1 = 1, o2 = 2;
    if (lam[o1] < lam[o0]) { const int t = o0; o0 = o1; o1 = t; }
    if (lam[o2] < lam[o0]) { const int t = o0; o0 = o2; o2 = t; }
    if (lam[o2] < lam[o1]) { const int t = o1; o1 = o2; o2 = t; }
    const double lmax = fabs(lam[o2]);
    int rank = 0;
    for (int a = 0; a < 3; ++a)
        if (lmax > 0 && fabs(lam[a]) > lmax * 3.0 * 1.1920929e-07) ++rank;
    float nv[3] = {0.f, 0.f, 0.f};
    if (m >= 3 && rank + 1 >= 3) {
        const double v[3] = {V[0 * 3 + o0], V[1 * 3 + o0], V[2 * 3 + o0]};
        double sgn = 1.0;
        if (has_vp) {
            const double dot = v[0] * ((double)vx - (double)p.x) + v[1] * ((double)vy - (double)p.y) + v[2] * ((double)vz - (double)p.z);
            if (dot < 0) sgn = -1.0;
        } else {
            int big = 0;
            if (fabs(v[1]) > fabs(v[big])) big = 1;
            if (fabs(v[2]) > fabs(v[big])) big = 2;
            if (v[big] < 0) sgn = -1.0;
        }
        for (int a = 0; a < 3; ++a) {
            const float f = (float)(sgn * v[a]);
            nv[a] = f > 1.f ? 1.f : (f < -1.f ? -1.f : f);
        }
    }
    const size_t oi = (size_t)my_idx;
    normals[3 * oi + 0] = nv[0];
    normals[3 * oi + 1] = nv[1];
    normals[3 * oi + 2] = nv[2];
    if (eigvals) {
        eigvals[3 * oi + 0] = (float)lam[o0];
        eigvals[3 * oi + 1] = (float)lam[o1];
        eigvals[3 * oi + 2] = (float)lam[o2];
    }
    if (covs) {
        double Cn[9];
        if (regularise) {
            const int oo[3] = {o0, o1, o2};
            const double w[3] = {1e-3, 1.0, 1.0};
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double t = 0;
                    for (int e2 = 0; e2 < 3; ++e2) t += w[e2] * V[a * 3 + oo[e2]] * V[b * 3 + oo[e2]];
                    Cn[3 * a + b] = t;
                }
        } else {
            const double inv = m > 0 ? 1.0 / (double)m : 0.0;
            const double Cd[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]};
            for (int a = 0; a < 9; ++a) Cn[a] = Cd[a] * inv;
        }
        covs[6 * oi + 0] = (float)Cn[0];
        covs[6 * oi + 1] = (float)Cn[1];
        covs[6 * oi + 2] = (float)Cn[2];
        covs[6 * oi + 3] = (float)Cn[4];
        covs[6 * oi + 4] = (float)Cn[5];
        covs[6 * oi + 5] = (float)Cn[8];
    }
}

// =================================================================================================
// host side
// =================================================================================================


struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const {
        return (T*)p;
    }
};

struct reg_handle {
    reg_params prm;
    std::string err;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool device_ok = false;   // false: reg_create could not get a HIP device (every entry point then fails loudly)
    bool structure_only = false;   // workspace handle of reg_estimate_normals: bin table only, no attributes
    reg_handle* normals_ws = nullptr;
    DevBuf n_out, n_eig, n_cov, n_ids;
    DevBuf i_xicp;                 // XicpState (R8x first-iteration analysis)
    DevBuf c_in_xyz, c_in_nrm, c_in_cov, c_flags, c_offs, c_xyz, c_nrm, c_cov, c_idx;   // reg_set_target_f64
    int64_t crop_kept = 0;
    DevBuf v_fout, v_oout, v_oxyz, v_onrm, v_ocov;   // reg_voxelize_within_volume
    bool xicp_pending = false;     // the next generic iteration is followed by the analysis kernels
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_iter = nullptr;
    bool iter_copy_pending = false;

    // target
    int64_t m = 0;
    bool has_tnrm = false, has_tcov = false;
    float c_ref[3] = {0, 0, 0};
    DevBuf t_raw, t_nrm_raw, t_cov_raw, t_centred, t_keys, t_keys2, t_vals, t_vals2, t_pts, t_nrm, t_cov, t_flags,
        t_scan, t_hash, t_cells, t_tmp, t_misc, t_dir;
    Grid grid;
    reg_target_info info;
    float target_build_ms = 0.f;

    // source
    int64_t n = 0;
    bool has_snrm = false, has_scov = false, prepared = false;
    int64_t s_stride = 3, s_nstride = 3;
    float c_read[3] = {0, 0, 0};
    DevBuf s_raw, s_nrm_raw, s_cov_raw, s_xyz, s_nrm, s_cov, s_misc;
    float T_init[16];              // row-major
    float T0[16];                  // T_refMean_readMean (row-major)
    // iteration buffers
    DevBuf i_pos, i_d2, i_w, i_hist, i_state, i_partials, i_sums, i_ids;
    HostMirror* h_mirror = nullptr;   // mapped pinned host memory written by the update kernel
    HostMirror* d_mirror = nullptr;   // device view of h_mirror
    IterState* h_iter = nullptr;      // pinned staging copy of the iteration state
    DevBuf i_iter;                    // IterState on the device
    unsigned long long seq = 0;
    DevBuf t_halo_start, t_halo_cursor, t_halo_pts, i_band, i_acc;
    DevBuf s_prep;
    PrepState* h_prep = nullptr;      // mapped pinned host copy of the device-side preparation state
    PrepState* d_prep_host = nullptr; // device view of h_prep
    bool prep_pending = false;        // h_prep not yet folded into c_read / T0
    DevBuf i_hint, s_keys, s_keys2, s_perm, s_perm2, s_tmp, i_tmpf;
    const uint32_t* perm = nullptr;   // slot -> input index (null: identity)
    int last_stalls = 0;
    unsigned long long dist_seq0 = 0;
    DevBuf d_contrib, d_gathered;     // multi-GPU fused iteration: this rank's block / all ranks' blocks
    int dist_ranks = 0, dist_rank = 0;
    // loop profiling (params.profile_loop): HIP events around the search kernels of every iteration
    std::vector<hipEvent_t> prof_ev;   // pairs (start, stop)
    std::vector<int> prof_kind;        // 0: k_match, 1: k_iter_fused
    bool profiling = false;
    int shift0 = 21;                  // low bit of the level-0 radix digit (19 when max_dist^2 < 2: bits 31,30 are 0)
    int n_blocks = 0;
    bool have_match = false;
};

#define HIPCHK(h, call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return REG_DEVICE_ERROR;                                                           \
        }                                                                                      \
    } while (0)

static inline void col_to_row(const float* c, float* r) { m4_transpose(c, r); }
static inline void row_to_col(const float* r, float* c) { m4_transpose(r, c); }

static inline int grid_for(int64_t n, int block = 256) { return (int)((n + block - 1) / block); }

extern "C" {

void reg_default_params(reg_params* p) {
    std::memset(p, 0, sizeof(*p));
    p->struct_size = (int32_t)sizeof(reg_params);
    p->cost = REG_COST_P2PL;
    p->knn = 1;
    p->max_dist = std::numeric_limits<float>::infinity();
    p->epsilon = 0.f;
    p->use_trimmed = 1;
    p->trim_ratio = 0.85f;
    p->use_surface_normal = 0;
    p->max_normal_angle = 1.57f;
    p->use_max_dist_filter = 0;
    p->outlier_max_dist = 1.f;
    p->max_iter = 40;
    p->min_diff_rot = 0.001f;
    p->min_diff_trans = 0.001f;
    p->smooth_len = 3;
    p->fixed_iters = 0;
    p->gicp_rot_eps = 0.1f * 3.14159265358979f / 180.f;
    p->gicp_trans_eps = 1e-3f;
    p->cell_size = 0.f;
    p->device = 0;
    p->sort_source = 1;
    p->use_xicp = 0;
    p->xicp_enough = 250.f;             // icp.yaml:50-55
    p->xicp_insufficient = 180.f;
    p->xicp_min_angle_deg = 80.f;
    p->xicp_strong_angle_deg = 45.f;
}

void reg_shipped_params(reg_params* p) {
    reg_default_params(p);
    p->max_dist = 0.5f;
    p->epsilon = 0.f;
    p->use_trimmed = 1;
    p->trim_ratio = 0.90f;
    p->use_surface_normal = 1;
    p->max_normal_angle = 1.57f;
    p->max_iter = 30;
    p->min_diff_rot = 0.001f;
    p->min_diff_trans = 0.008f;
    p->smooth_len = 3;
}

reg_status reg_create(const reg_params* p, reg_handle** out) {
    if (!p || !out) return REG_BAD_ARGUMENT;
    *out = nullptr;
    if (p->struct_size != (int32_t)sizeof(reg_params)) return REG_BAD_ARGUMENT;
    if (p->knn != 1) return REG_BAD_ARGUMENT;
    if (!(p->max_dist > 0.f)) return REG_BAD_ARGUMENT;
    if (p->cost != REG_COST_P2PL && p->cost != REG_COST_GICP) return REG_BAD_ARGUMENT;
    if (p->use_trimmed && !(p->trim_ratio >= 0.f && p->trim_ratio <= 1.f)) return REG_BAD_ARGUMENT;
    if (p->fixed_iters <= 0 && p->max_iter <= 0) return REG_BAD_ARGUMENT;
    if (p->use_xicp && p->cost != REG_COST_P2PL) return REG_BAD_ARGUMENT;   // the analysis expects point-to-plane (ICP.cpp:1118)
    reg_handle* h = new reg_handle();
    h->prm = *p;
    std::memset(&h->info, 0, sizeof(h->info));
    std::memset(&h->grid, 0, sizeof(h->grid));
    if (hipSetDevice(p->device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        // keep the handle so that reg_last_error can explain; entry points will fail loudly
        h->err = "no usable HIP device (hipSetDevice/hipStreamCreate failed): the HIP path is mandatory";
        h->stream = nullptr;
        *out = h;
        return REG_DEVICE_ERROR;
    }
    h->own_stream = true;
    h->device_ok = true;
    (void)hipEventCreate(&h->ev0);
    (void)hipEventCreate(&h->ev1);
    (void)hipEventCreateWithFlags(&h->ev_iter, hipEventDisableTiming);
    if (hipHostMalloc((void**)&h->h_mirror, sizeof(HostMirror), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->d_mirror, h->h_mirror, 0) != hipSuccess ||
        hipHostMalloc((void**)&h->h_iter, sizeof(IterState), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_prep, sizeof(PrepState), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->d_prep_host, h->h_prep, 0) != hipSuccess ||
        h->i_iter.reserve(sizeof(IterState)) != hipSuccess) {
        h->err = "hipHostMalloc / hipMalloc of the iteration state failed";
        *out = h;
        return REG_DEVICE_ERROR;
    }
    std::memset(h->h_mirror, 0, sizeof(HostMirror));
    *out = h;
    return REG_OK;
}

void reg_destroy(reg_handle* h) {
    if (!h) return;
    if (h->normals_ws) reg_destroy(h->normals_ws);
    h->n_out.release();
    h->i_xicp.release();
    for (DevBuf* b : {&h->c_in_xyz, &h->c_in_nrm, &h->c_in_cov, &h->c_flags, &h->c_offs, &h->c_xyz, &h->c_nrm, &h->c_cov, &h->c_idx, &h->v_fout, &h->v_oout, &h->v_oxyz, &h->v_onrm, &h->v_ocov}) b->release();
    h->n_eig.release();
    h->n_cov.release();
    h->n_ids.release();
    DevBuf* bufs[] = {&h->t_raw, &h->t_nrm_raw, &h->t_cov_raw, &h->t_centred, &h->t_keys, &h->t_keys2, &h->t_vals,
                      &h->t_vals2, &h->t_pts, &h->t_nrm, &h->t_cov, &h->t_flags, &h->t_scan, &h->t_hash, &h->t_cells,
                      &h->t_tmp, &h->t_misc, &h->t_dir, &h->s_raw, &h->s_nrm_raw, &h->s_cov_raw, &h->s_xyz, &h->s_nrm, &h->s_cov,
                      &h->s_misc, &h->i_pos, &h->i_d2, &h->i_w, &h->i_hist, &h->i_state, &h->i_partials, &h->i_sums,
                      &h->i_ids, &h->d_contrib, &h->d_gathered, &h->s_prep, &h->i_iter, &h->t_halo_start, &h->t_halo_cursor, &h->t_halo_pts, &h->i_band, &h->i_acc, &h->i_hint, &h->s_keys, &h->s_keys2, &h->s_perm, &h->s_perm2, &h->s_tmp, &h->i_tmpf};
    for (DevBuf* b : bufs) b->release();
    if (h->h_mirror) (void)hipHostFree(h->h_mirror);
    if (h->h_iter) (void)hipHostFree(h->h_iter);
    if (h->h_prep) (void)hipHostFree(h->h_prep);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_iter) (void)hipEventDestroy(h->ev_iter);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* reg_last_error(const reg_handle* h) { return h ? h->err.c_str() : "null handle"; }

reg_status reg_set_stream(reg_handle* h, void* hip_stream) {
    if (!h) return REG_BAD_ARGUMENT;
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    return REG_OK;
}

}  // extern "C"

// copy (host or device) -> device buffer
static reg_status upload(reg_handle* h, DevBuf& dst, const float* src, size_t bytes, int on_device) {
    HIPCHK(h, dst.reserve(bytes));
    HIPCHK(h, hipMemcpyAsync(dst.p, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    return REG_OK;
}

static reg_status device_centroid_sums(reg_handle* h, const float* d_xyz, int64_t stride, int64_t n, DevBuf& misc,
                                       long long s[3]) {
    HIPCHK(h, misc.reserve(256));
    HIPCHK(h, hipMemsetAsync(misc.p, 0, 3 * sizeof(unsigned long long), h->stream));
    const int blocks = std::min<int64_t>(1024, (n + 255) / 256);
    k_centroid_sums<<<blocks, 256, 0, h->stream>>>(d_xyz, stride, n, misc.as<unsigned long long>());
    HIPCHK(h, hipMemcpyAsync(s, misc.p, 3 * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return REG_OK;
}

static reg_status device_centroid(reg_handle* h, const float* d_xyz, int64_t stride, int64_t n, DevBuf& misc, float out[3]) {
    HIPCHK(h, misc.reserve(256));
    HIPCHK(h, hipMemsetAsync(misc.p, 0, 3 * sizeof(unsigned long long), h->stream));
    const int blocks = std::min<int64_t>(1024, (n + 255) / 256);
    k_centroid_sums<<<blocks, 256, 0, h->stream>>>(d_xyz, stride, n, misc.as<unsigned long long>());
    long long s[3];
    HIPCHK(h, hipMemcpyAsync(s, misc.p, sizeof(s), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < 3; ++k) out[k] = (float)((double)s[k] / (65536.0 * (double)n));
    return REG_OK;
}

// Build the brick table for bin edge c.  Returns occupied-bin count through *occupied.
static reg_status build_grid(reg_handle* h, float c, const float bmin[3], const float bmax[3], uint32_t* occupied) {
    const int64_t m = h->m;
    Grid& g = h->grid;
    const float inv_c = 1.0f / c;
    g.ox = bmin[0];
    g.oy = bmin[1];
    g.oz = bmin[2];
    g.inv_c = inv_c;
    float dims[3];
    for (int k = 0; k < 3; ++k) {
        volatile float d = bmax[k] - bmin[k];
        volatile float s = d * inv_c;
        dims[k] = std::floor((float)s) + 1.0f;
    }
    const double max_dim = (double)(1u << (kBrickBits + kBrickLog2));
    if (dims[0] > max_dim || dims[1] > max_dim || dims[2] > max_dim) {
        h->err = "cell_size too small for the target extent (bin coordinates overflow the sort key)";
        return REG_BAD_ARGUMENT;
    }
    g.dimx = dims[0];
    g.dimy = dims[1];
    g.dimz = dims[2];
    h->info.dims[0] = (int32_t)dims[0];
    h->info.dims[1] = (int32_t)dims[1];
    h->info.dims[2] = (int32_t)dims[2];

    HIPCHK(h, h->t_keys.reserve(m * 8));
    HIPCHK(h, h->t_keys2.reserve(m * 8));
    HIPCHK(h, h->t_vals.reserve(m * 4));
    HIPCHK(h, h->t_vals2.reserve(m * 4));
    k_point_keys<<<grid_for(m), 256, 0, h->stream>>>(h->t_centred.as<float4>(), m, g.ox, g.oy, g.oz, inv_c,
                                                      h->t_keys.as<uint64_t>(), h->t_vals.as<uint32_t>());
    // significant key bits
    auto bits_for = [](double v) { int b = 1; while ((double)(1ull << b) < v) ++b; return b; };
    const int bz_bits = bits_for(std::ceil(dims[2] / kBrickDim) + 1);
    const int end_bit = std::min(64, 3 * kBrickLog2 + 2 * kBrickBits + bz_bits);
    size_t tmp_bytes = 0;
    HIPCHK(h, rocprim::radix_sort_pairs(nullptr, tmp_bytes, h->t_keys.as<uint64_t>(), h->t_keys2.as<uint64_t>(),
                                        h->t_vals.as<uint32_t>(), h->t_vals2.as<uint32_t>(), (size_t)m, 0, end_bit,
                                        h->stream));
    HIPCHK(h, h->t_tmp.reserve(tmp_bytes));
    HIPCHK(h, rocprim::radix_sort_pairs(h->t_tmp.p, tmp_bytes, h->t_keys.as<uint64_t>(), h->t_keys2.as<uint64_t>(),
                                        h->t_vals.as<uint32_t>(), h->t_vals2.as<uint32_t>(), (size_t)m, 0, end_bit,
                                        h->stream));
    // brick heads -> brick ids
    HIPCHK(h, h->t_flags.reserve(m * 4));
    HIPCHK(h, h->t_scan.reserve(m * 4));
    k_brick_heads<<<grid_for(m), 256, 0, h->stream>>>(h->t_keys2.as<uint64_t>(), m, h->t_flags.as<uint32_t>());
    size_t scan_bytes = 0;
    HIPCHK(h, rocprim::inclusive_scan(nullptr, scan_bytes, h->t_flags.as<uint32_t>(), h->t_scan.as<uint32_t>(),
                                      (size_t)m, rocprim::plus<uint32_t>(), h->stream));
    HIPCHK(h, h->t_tmp.reserve(scan_bytes));
    HIPCHK(h, rocprim::inclusive_scan(h->t_tmp.p, scan_bytes, h->t_flags.as<uint32_t>(), h->t_scan.as<uint32_t>(),
                                      (size_t)m, rocprim::plus<uint32_t>(), h->stream));
    uint32_t nb = 0;
    HIPCHK(h, hipMemcpyAsync(&nb, h->t_scan.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // tables
    uint32_t cap = 16;
    while (cap < 2 * nb) cap <<= 1;
    HIPCHK(h, h->t_hash.reserve((size_t)cap * sizeof(HashEntry)));
    HIPCHK(h, hipMemsetAsync(h->t_hash.p, 0xff, (size_t)cap * sizeof(HashEntry), h->stream));
    const size_t n_cells = (size_t)nb * kBrickCells + 1;
    HIPCHK(h, h->t_cells.reserve(n_cells * 4));
    HIPCHK(h, hipMemsetAsync(h->t_cells.p, 0, n_cells * 4, h->stream));
    HIPCHK(h, h->t_misc.reserve(256));
    HIPCHK(h, hipMemsetAsync(h->t_misc.p, 0, 64, h->stream));
    // dense brick directory (brick id per brick coordinate) when the brick grid is small enough: one 4-byte load
    // instead of a 64-bit hash + probe per row segment
    const int bdx = (int)std::ceil(dims[0] / kBrickDim), bdy = (int)std::ceil(dims[1] / kBrickDim),
              bdz = (int)std::ceil(dims[2] / kBrickDim);
    const size_t n_dir = (size_t)bdx * bdy * bdz;
    const bool use_dir = n_dir <= ((size_t)64 << 20) && !(h->prm.debug_flags & 32);
    if (use_dir) {
        HIPCHK(h, h->t_dir.reserve(n_dir * 4));
        HIPCHK(h, hipMemsetAsync(h->t_dir.p, 0xff, n_dir * 4, h->stream));
    }
    k_fill_tables<<<grid_for(m), 256, 0, h->stream>>>(h->t_keys2.as<uint64_t>(), h->t_flags.as<uint32_t>(),
                                                       h->t_scan.as<uint32_t>(), m, h->t_hash.as<HashEntry>(), cap - 1,
                                                       h->t_cells.as<uint32_t>(), h->t_misc.as<uint32_t>(),
                                                       use_dir ? h->t_dir.as<int32_t>() : nullptr, bdx, bdy);
    g.brick_dir = use_dir ? h->t_dir.as<int32_t>() : nullptr;
    g.bdx = bdx;
    g.bdy = bdy;
    g.bdz = bdz;
    g.wide_scan = (h->prm.debug_flags & 16) ? 0 : 1;
    size_t ex_bytes = 0;
    HIPCHK(h, rocprim::exclusive_scan(nullptr, ex_bytes, h->t_cells.as<uint32_t>(), h->t_cells.as<uint32_t>(), 0u,
                                      n_cells, rocprim::plus<uint32_t>(), h->stream));
    HIPCHK(h, h->t_tmp.reserve(ex_bytes));
    HIPCHK(h, rocprim::exclusive_scan(h->t_tmp.p, ex_bytes, h->t_cells.as<uint32_t>(), h->t_cells.as<uint32_t>(), 0u,
                                      n_cells, rocprim::plus<uint32_t>(), h->stream));
    HIPCHK(h, hipMemcpyAsync(occupied, h->t_misc.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    g.hash_mask = cap - 1;
    g.hash = h->t_hash.as<HashEntry>();
    g.cell_start = h->t_cells.as<uint32_t>();
    h->info.n_bricks = nb;
    h->info.n_cells_occupied = *occupied;
    h->info.table_bytes = (int64_t)((size_t)cap * sizeof(HashEntry) + n_cells * 4 + (use_dir ? n_dir * 4 : 0));
    h->info.cell_size = c;
    return REG_OK;
}

static void set_levels(reg_handle* h, float c, float max_abs) {
    Grid& g = h->grid;
    const float md = h->prm.max_dist;
    g.max_d2 = std::isinf(md) ? INFINITY : md * md;
    h->shift0 = (g.max_d2 < 2.0f) ? 19 : 21;
    int n = 0;
    float rho = 0.5f * c;
    const float abs_margin = 4e-7f * (1.0f + max_abs);
    while (n < kMaxLevels - 1 && rho < md) {
        g.rho[n] = rho;
        g.rho_box[n] = rho + 1e-3f * rho + abs_margin;
        ++n;
        rho *= 2.0f;
        if (std::isinf(md) && n >= 6) break;  // unbounded search: after 6 doublings fall through to a full scan
    }
    g.rho[n] = md;
    g.rho_box[n] = std::isinf(md) ? INFINITY : md + 1e-3f * md + abs_margin;
    ++n;
    g.n_levels = n;
}

// Level-0 accelerator: dense halo bins of edge c_h = 1.5 c with rho_h = c_h / 4 (each point is listed in
// 1-2 bins per axis: ~3.4 copies).  Skipped when the dense grid would be too large or on request.
static reg_status build_halo(reg_handle* h, float c, const float bmin[3], const float bmax[3], float max_abs) {
    Grid& g = h->grid;
    g.use_halo = 0;
    g.level_after_halo = 0;
    if (h->prm.disable_halo == 1) return REG_OK;  // A/B experiments
    const float ch = 1.5f * c;
    const float abs_margin = 4e-7f * (1.0f + max_abs);
    const float rho_h = 0.25f * ch * (1.0f - 4e-3f) - 2.f * abs_margin;
    if (!(rho_h > 0.f)) return REG_OK;
    const float r_ins = rho_h + 1e-3f * rho_h + abs_margin;
    const float inv = 1.0f / ch;
    double dims[3];
    for (int k = 0; k < 3; ++k) dims[k] = std::floor((double)(bmax[k] - bmin[k]) * inv) + 1.0;
    const double nb = dims[0] * dims[1] * dims[2];
    if (nb > 48e6) return REG_OK;
    const size_t nbins = (size_t)nb;
    HaloCfg hc;
    hc.ox = bmin[0];
    hc.oy = bmin[1];
    hc.oz = bmin[2];
    hc.inv_c = inv;
    hc.r_ins = r_ins;
    hc.dimx = (int)dims[0];
    hc.dimy = (int)dims[1];
    hc.dimz = (int)dims[2];
    HIPCHK(h, h->t_halo_start.reserve((nbins + 1) * 4));
    HIPCHK(h, h->t_halo_cursor.reserve((nbins + 1) * 4));
    HIPCHK(h, hipMemsetAsync(h->t_halo_start.p, 0, (nbins + 1) * 4, h->stream));
    k_halo_insert<<<grid_for(h->m), 256, 0, h->stream>>>(h->t_pts.as<float4>(), h->m, hc, 0,
                                                         h->t_halo_start.as<uint32_t>(), nullptr);
    size_t ex_bytes = 0;
    HIPCHK(h, rocprim::exclusive_scan(nullptr, ex_bytes, h->t_halo_start.as<uint32_t>(), h->t_halo_start.as<uint32_t>(),
                                      0u, nbins + 1, rocprim::plus<uint32_t>(), h->stream));
    HIPCHK(h, h->t_tmp.reserve(ex_bytes));
    HIPCHK(h, rocprim::exclusive_scan(h->t_tmp.p, ex_bytes, h->t_halo_start.as<uint32_t>(),
                                      h->t_halo_start.as<uint32_t>(), 0u, nbins + 1, rocprim::plus<uint32_t>(),
                                      h->stream));
    uint32_t total = 0;
    HIPCHK(h, hipMemcpyAsync(&total, h->t_halo_start.as<uint32_t>() + nbins, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->t_halo_cursor.p, h->t_halo_start.p, nbins * 4, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, h->t_halo_pts.reserve((size_t)std::max<uint32_t>(total, 1) * 16));
    k_halo_insert<<<grid_for(h->m), 256, 0, h->stream>>>(h->t_pts.as<float4>(), h->m, hc, 1,
                                                         h->t_halo_cursor.as<uint32_t>(), h->t_halo_pts.as<float4>());
    g.use_halo = 1;
    g.hox = hc.ox;
    g.hoy = hc.oy;
    g.hoz = hc.oz;
    g.hinv_c = inv;
    g.hdimx = hc.dimx;
    g.hdimy = hc.dimy;
    g.hdimz = hc.dimz;
    g.halo_start = h->t_halo_start.as<uint32_t>();
    g.halo_pts = h->t_halo_pts.as<float4>();
    g.rho_h = rho_h;
    g.level_after_halo = g.n_levels - 1;
    for (int l = 0; l < g.n_levels; ++l)
        if (g.rho[l] > rho_h) {
            g.level_after_halo = l;
            break;
        }
    h->info.table_bytes += (int64_t)((nbins + 1) * 4 + (size_t)total * 16);
    return REG_OK;
}

extern "C" {

reg_status reg_set_target(reg_handle* h, const float* xyz, int64_t xyz_stride, const float* nrm, int64_t nrm_stride,
                          const float* cov, int64_t m, int on_device) {
    if (!h) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    h->m = 0;
    h->crop_kept = 0;
    h->have_match = false;
    if (m <= 0) {
        h->err = "The reference point cloud is empty";
        return REG_EMPTY_TARGET;
    }
    if (!xyz || xyz_stride < 3 || (nrm && nrm_stride < 3) || m > 0x7fffffffLL) return REG_BAD_ARGUMENT;
    if (h->prm.cost == REG_COST_P2PL && !nrm && !h->structure_only) {
        h->err = "InvalidField: point-to-plane needs the `normals` descriptor on the reference";
        return REG_MISSING_FIELD;
    }
    if (h->prm.cost == REG_COST_GICP && !cov && !h->structure_only) {
        h->err = "InvalidField: GICP needs covariances on the reference";
        return REG_MISSING_FIELD;
    }
    HIPCHK(h, hipSetDevice(h->prm.device));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    const float* d_xyz = xyz;
    const float* d_nrm = nrm;
    const float* d_cov = cov;
    if (!on_device) {
        reg_status s = upload(h, h->t_raw, xyz, (size_t)m * xyz_stride * 4, 0);
        if (s != REG_OK) return s;
        d_xyz = h->t_raw.as<float>();
        if (nrm) {
            s = upload(h, h->t_nrm_raw, nrm, (size_t)m * nrm_stride * 4, 0);
            if (s != REG_OK) return s;
            d_nrm = h->t_nrm_raw.as<float>();
        }
        if (cov) {
            s = upload(h, h->t_cov_raw, cov, (size_t)m * 6 * 4, 0);
            if (s != REG_OK) return s;
            d_cov = h->t_cov_raw.as<float>();
        }
    }
    h->m = m;
    h->has_tnrm = nrm != nullptr;
    h->has_tcov = cov != nullptr;
    // R1: centroid (P2PL path only: GICP works in the input frame, as small_gicp does)
    float c[3] = {0, 0, 0};
    if (h->prm.cost == REG_COST_P2PL) {
        reg_status s = device_centroid(h, d_xyz, xyz_stride, m, h->t_misc, c);
        if (s != REG_OK) return s;
    }
    std::memcpy(h->c_ref, c, sizeof(c));
    std::memcpy(h->info.centroid, c, sizeof(c));
    // centred copy + bbox
    HIPCHK(h, h->t_centred.reserve((size_t)m * 16));
    HIPCHK(h, h->t_misc.reserve(256));
    int bb_init[6] = {0x7f800000, 0x7f800000, 0x7f800000, (int)0x80000000 ^ 0, 0, 0};
    // ordered-int encodings of +inf / -inf
    bb_init[0] = bb_init[1] = bb_init[2] = 0x7f800000;                    // +inf
    bb_init[3] = bb_init[4] = bb_init[5] = (int)(0xff800000u ^ 0x7fffffffu);  // -inf
    HIPCHK(h, hipMemcpyAsync(h->t_misc.p, bb_init, sizeof(bb_init), hipMemcpyHostToDevice, h->stream));
    const int blocks = (int)std::min<int64_t>(512, (m + 255) / 256);
    k_center_bbox<<<blocks, 256, 0, h->stream>>>(d_xyz, xyz_stride, m, c[0], c[1], c[2], h->t_centred.as<float4>(),
                                                 h->t_misc.as<int>());
    int bb[6];
    HIPCHK(h, hipMemcpyAsync(bb, h->t_misc.p, sizeof(bb), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float bmin[3], bmax[3], max_abs = 0.f;
    for (int k = 0; k < 3; ++k) {
        bmin[k] = ord2f(bb[k]);
        bmax[k] = ord2f(bb[3 + k]);
        if (!std::isfinite(bmin[k]) || !std::isfinite(bmax[k])) {
            h->err = "reference cloud contains non-finite coordinates";
            h->m = 0;
            return REG_BAD_ARGUMENT;
        }
        max_abs = std::max(max_abs, std::max(std::fabs(bmin[k]), std::fabs(bmax[k])));
    }
    // bin edge: user value, or adapt to ~8 points per occupied bin (surface-like clouds: count ~ c^2)
    float cs = h->prm.cell_size;
    uint32_t occupied = 0;
    if (cs > 0.f) {
        reg_status s = build_grid(h, cs, bmin, bmax, &occupied);
        if (s != REG_OK) return s;
    } else {
        const float ext = std::max(bmax[0] - bmin[0], std::max(bmax[1] - bmin[1], bmax[2] - bmin[2]));
        // start from the edge that would give 8 points per bin if the cloud were a single ext x ext sheet
        cs = std::max(ext * std::sqrt(8.0f / (float)m), 1e-4f * std::max(ext, 1e-3f));
        const float cs_min = std::max(ext / (float)(1u << 20), 1e-6f);
        cs = std::max(cs, cs_min);
        for (int pass = 0; pass < 3; ++pass) {
            reg_status s = build_grid(h, cs, bmin, bmax, &occupied);
            if (s != REG_OK) return s;
            const float per = (float)m / (float)std::max(1u, occupied);
            if (per <= 12.0f && per >= 5.0f) break;
            if (pass == 2) break;
            float next = cs * std::sqrt(8.0f / per);
            next = std::max(next, cs_min);
            if (std::fabs(next - cs) < 0.05f * cs) break;
            cs = next;
        }
    }
    // sorted arrays
    HIPCHK(h, h->t_pts.reserve((size_t)m * 16));
    if (d_nrm) HIPCHK(h, h->t_nrm.reserve((size_t)m * 16));
    if (d_cov) HIPCHK(h, h->t_cov.reserve((size_t)m * 32));
    k_gather_target<<<grid_for(m), 256, 0, h->stream>>>(h->t_centred.as<float4>(), h->t_vals2.as<uint32_t>(), m, d_nrm,
                                                         nrm_stride, d_cov, h->t_pts.as<float4>(),
                                                         d_nrm ? h->t_nrm.as<float4>() : nullptr,
                                                         d_cov ? h->t_cov.as<float4>() : nullptr);
    h->grid.pts = h->t_pts.as<float4>();
    set_levels(h, cs, max_abs);
    {
        reg_status hs = build_halo(h, cs, bmin, bmax, max_abs);
        if (hs != REG_OK) return hs;
    }
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    (void)hipEventElapsedTime(&h->target_build_ms, h->ev0, h->ev1);
    h->info.n_points = m;
    h->info.origin[0] = bmin[0];
    h->info.origin[1] = bmin[1];
    h->info.origin[2] = bmin[2];
    return REG_OK;
}

reg_status reg_set_target_f64(reg_handle* h, const double* xyz, const double* normals, const double* covs, int64_t m,
                              int on_device, const reg_crop* crop, int64_t* n_kept) {
    if (!h) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    if (n_kept) *n_kept = 0;
    h->crop_kept = 0;
    if (m <= 0) {
        h->m = 0;
        h->err = "The reference point cloud is empty";
        return REG_EMPTY_TARGET;
    }
    if (!xyz || m > 0x7fffffffLL) return REG_BAD_ARGUMENT;
    CropCfg c;
    std::memset(&c, 0, sizeof(c));
    if (crop) {
        if (crop->type < REG_CROP_NONE || crop->type > REG_CROP_CYLINDER) return REG_BAD_ARGUMENT;
        c.type = crop->type;
        c.cx = crop->center[0];
        c.cy = crop->center[1];
        c.cz = crop->center[2];
        c.rmin = crop->radius_min;
        c.rmax = crop->radius_max;
        c.zmin = crop->min_z;
        c.zmax = crop->max_z;
    }
    HIPCHK(h, hipSetDevice(h->prm.device));
    const double *d_xyz = xyz, *d_nrm = normals, *d_cov = covs;
    if (!on_device) {
        HIPCHK(h, h->c_in_xyz.reserve((size_t)m * 24));
        HIPCHK(h, hipMemcpyAsync(h->c_in_xyz.p, xyz, (size_t)m * 24, hipMemcpyHostToDevice, h->stream));
        d_xyz = h->c_in_xyz.as<double>();
        if (normals) {
            HIPCHK(h, h->c_in_nrm.reserve((size_t)m * 24));
            HIPCHK(h, hipMemcpyAsync(h->c_in_nrm.p, normals, (size_t)m * 24, hipMemcpyHostToDevice, h->stream));
            d_nrm = h->c_in_nrm.as<double>();
        }
        if (covs) {
            HIPCHK(h, h->c_in_cov.reserve((size_t)m * 72));
            HIPCHK(h, hipMemcpyAsync(h->c_in_cov.p, covs, (size_t)m * 72, hipMemcpyHostToDevice, h->stream));
            d_cov = h->c_in_cov.as<double>();
        }
    }
    HIPCHK(h, h->c_flags.reserve((size_t)m * 4));
    HIPCHK(h, h->c_offs.reserve((size_t)m * 4));
    k_crop_flags<<<grid_for(m), 256, 0, h->stream>>>(d_xyz, m, c, h->c_flags.as<uint32_t>());
    size_t tb = 0;
    HIPCHK(h, rocprim::exclusive_scan(nullptr, tb, h->c_flags.as<uint32_t>(), h->c_offs.as<uint32_t>(), 0u, (size_t)m,
                                      rocprim::plus<uint32_t>(), h->stream));
    HIPCHK(h, h->t_tmp.reserve(tb));
    HIPCHK(h, rocprim::exclusive_scan(h->t_tmp.p, tb, h->c_flags.as<uint32_t>(), h->c_offs.as<uint32_t>(), 0u, (size_t)m,
                                      rocprim::plus<uint32_t>(), h->stream));
    uint32_t last[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(&last[0], h->c_offs.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&last[1], h->c_flags.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int64_t kept = (int64_t)last[0] + (int64_t)last[1];
    if (n_kept) *n_kept = kept;
    if (kept == 0) {
        h->m = 0;
        h->err = "The reference point cloud is empty (no point inside the cropping volume)";   // ScanToMapRegistration.cpp:94
        return REG_EMPTY_TARGET;
    }
    HIPCHK(h, h->c_xyz.reserve((size_t)kept * 12));
    if (d_nrm) HIPCHK(h, h->c_nrm.reserve((size_t)kept * 12));
    if (d_cov) HIPCHK(h, h->c_cov.reserve((size_t)kept * 24));
    HIPCHK(h, h->c_idx.reserve((size_t)kept * 4));
    k_crop_gather<<<grid_for(m), 256, 0, h->stream>>>(d_xyz, d_nrm, d_cov, m, h->c_flags.as<uint32_t>(),
                                                      h->c_offs.as<uint32_t>(), h->c_xyz.as<float>(),
                                                      d_nrm ? h->c_nrm.as<float>() : nullptr,
                                                      d_cov ? h->c_cov.as<float>() : nullptr, h->c_idx.as<int32_t>());
    const reg_status s = reg_set_target(h, h->c_xyz.as<float>(), 3, d_nrm ? h->c_nrm.as<float>() : nullptr, 3,
                                        d_cov ? h->c_cov.as<float>() : nullptr, kept, 1);
    if (s == REG_OK) h->crop_kept = kept;
    return s;
}

reg_status reg_get_target_source_indices(reg_handle* h, int32_t* idx) {
    if (!h || !idx) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    if (h->crop_kept <= 0 || h->crop_kept != h->m) {
        h->err = "the current reference was not set through reg_set_target_f64";
        return REG_NOT_CONFIGURED;
    }
    HIPCHK(h, hipSetDevice(h->prm.device));
    HIPCHK(h, hipMemcpyAsync(idx, h->c_idx.p, (size_t)h->crop_kept * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return REG_OK;
}

reg_status reg_voxelize_within_volume(reg_handle* h, const double* xyz, const double* normals, const double* covs, int64_t m,
                                      int on_device, const reg_crop* volume, double voxel_size, double* out_xyz,
                                      double* out_normals, double* out_covs, int64_t* n_out, int64_t* n_outside) {
    if (!h) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    if (n_out) *n_out = 0;
    if (n_outside) *n_outside = 0;
    if (m < 0 || m > 0x7fffffffLL || (m > 0 && (!xyz || !out_xyz)) || (normals && !out_normals) || (covs && !out_covs))
        return REG_BAD_ARGUMENT;
    if (m == 0) return REG_OK;
    CropCfg c;
    std::memset(&c, 0, sizeof(c));
    if (volume) {
        if (volume->type < REG_CROP_NONE || volume->type > REG_CROP_CYLINDER) return REG_BAD_ARGUMENT;
        c.type = volume->type;
        c.cx = volume->center[0];
        c.cy = volume->center[1];
        c.cz = volume->center[2];
        c.rmin = volume->radius_min;
        c.rmax = volume->radius_max;
        c.zmin = volume->min_z;
        c.zmax = volume->max_z;
    }
    HIPCHK(h, hipSetDevice(h->prm.device));
    const hipMemcpyKind in_kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const hipMemcpyKind out_kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (!(voxel_size > 0.0)) {   // helpers.cpp:121-124: nothing to do
        HIPCHK(h, hipMemcpyAsync(out_xyz, xyz, (size_t)m * 24, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToHost, h->stream));
        if (normals) HIPCHK(h, hipMemcpyAsync(out_normals, normals, (size_t)m * 24, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToHost, h->stream));
        if (covs) HIPCHK(h, hipMemcpyAsync(out_covs, covs, (size_t)m * 72, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (n_out) *n_out = m;
        if (n_outside) *n_outside = m;
        return REG_OK;
    }
    const double *d_xyz = xyz, *d_nrm = normals, *d_cov = covs;
    double *d_oxyz = out_xyz, *d_onrm = out_normals, *d_ocov = out_covs;
    if (!on_device) {
        HIPCHK(h, h->c_in_xyz.reserve((size_t)m * 24));
        HIPCHK(h, hipMemcpyAsync(h->c_in_xyz.p, xyz, (size_t)m * 24, in_kind, h->stream));
        d_xyz = h->c_in_xyz.as<double>();
        HIPCHK(h, h->v_oxyz.reserve((size_t)m * 24));
        d_oxyz = h->v_oxyz.as<double>();
        if (normals) {
            HIPCHK(h, h->c_in_nrm.reserve((size_t)m * 24));
            HIPCHK(h, hipMemcpyAsync(h->c_in_nrm.p, normals, (size_t)m * 24, in_kind, h->stream));
            d_nrm = h->c_in_nrm.as<double>();
            HIPCHK(h, h->v_onrm.reserve((size_t)m * 24));
            d_onrm = h->v_onrm.as<double>();
        }
        if (covs) {
            HIPCHK(h, h->c_in_cov.reserve((size_t)m * 72));
            HIPCHK(h, hipMemcpyAsync(h->c_in_cov.p, covs, (size_t)m * 72, in_kind, h->stream));
            d_cov = h->c_in_cov.as<double>();
            HIPCHK(h, h->v_ocov.reserve((size_t)m * 72));
            d_ocov = h->v_ocov.as<double>();
        }
    }
    const double inv = 1.0 / voxel_size;   // fromVoxelSize (VoxelHashMap.hpp:43-45)
    HIPCHK(h, h->c_flags.reserve((size_t)m * 4));
    HIPCHK(h, h->c_offs.reserve((size_t)m * 4));
    HIPCHK(h, h->v_fout.reserve((size_t)m * 4));
    HIPCHK(h, h->v_oout.reserve((size_t)m * 4));
    HIPCHK(h, h->t_misc.reserve(256));
    HIPCHK(h, hipMemsetAsync(h->t_misc.p, 0, 4, h->stream));
    k_vox_classify<<<grid_for(m), 256, 0, h->stream>>>(d_xyz, m, c, inv, h->c_flags.as<uint32_t>(), h->v_fout.as<uint32_t>(),
                                                       h->t_misc.as<uint32_t>());
    size_t tb = 0;
    HIPCHK(h, rocprim::exclusive_scan(nullptr, tb, h->c_flags.as<uint32_t>(), h->c_offs.as<uint32_t>(), 0u, (size_t)m,
                                      rocprim::plus<uint32_t>(), h->stream));
    HIPCHK(h, h->t_tmp.reserve(tb));
    HIPCHK(h, rocprim::exclusive_scan(h->t_tmp.p, tb, h->c_flags.as<uint32_t>(), h->c_offs.as<uint32_t>(), 0u, (size_t)m,
                                      rocprim::plus<uint32_t>(), h->stream));
    HIPCHK(h, rocprim::exclusive_scan(h->t_tmp.p, tb, h->v_fout.as<uint32_t>(), h->v_oout.as<uint32_t>(), 0u, (size_t)m,
                                      rocprim::plus<uint32_t>(), h->stream));
    uint32_t tail[3] = {0, 0, 0};
    HIPCHK(h, hipMemcpyAsync(&tail[0], h->c_offs.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&tail[1], h->c_flags.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&tail[2], h->t_misc.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (tail[2]) {
        h->err = "voxel_size too small for the extent of the cloud (voxel index exceeds 2^20)";
        return REG_BAD_ARGUMENT;
    }
    const int64_t n_in = (int64_t)tail[0] + tail[1], n_outs = m - n_in;
    int64_t n_vox = 0;
    HIPCHK(h, h->t_keys.reserve((size_t)std::max<int64_t>(n_in, 1) * 8));
    HIPCHK(h, h->t_keys2.reserve((size_t)std::max<int64_t>(n_in, 1) * 8));
    HIPCHK(h, h->t_vals.reserve((size_t)std::max<int64_t>(n_in, 1) * 4));
    HIPCHK(h, h->t_vals2.reserve((size_t)std::max<int64_t>(n_in, 1) * 4));
    k_vox_scatter<<<grid_for(m), 256, 0, h->stream>>>(d_xyz, d_nrm, d_cov, m, inv, h->c_flags.as<uint32_t>(),
                                                      h->c_offs.as<uint32_t>(), h->v_oout.as<uint32_t>(),
                                                      h->t_keys.as<uint64_t>(), h->t_vals.as<uint32_t>(), d_oxyz, d_onrm,
                                                      d_ocov);
    if (n_in > 0) {
        size_t sb = 0;
        HIPCHK(h, rocprim::radix_sort_pairs(nullptr, sb, h->t_keys.as<uint64_t>(), h->t_keys2.as<uint64_t>(),
                                            h->t_vals.as<uint32_t>(), h->t_vals2.as<uint32_t>(), (size_t)n_in, 0,
                                            3 * kVoxBits, h->stream));
        HIPCHK(h, h->t_tmp.reserve(sb));
        HIPCHK(h, rocprim::radix_sort_pairs(h->t_tmp.p, sb, h->t_keys.as<uint64_t>(), h->t_keys2.as<uint64_t>(),
                                            h->t_vals.as<uint32_t>(), h->t_vals2.as<uint32_t>(), (size_t)n_in, 0,
                                            3 * kVoxBits, h->stream));
        HIPCHK(h, h->t_flags.reserve((size_t)n_in * 4));
        HIPCHK(h, h->t_scan.reserve((size_t)n_in * 4));
        k_vox_heads<<<grid_for(n_in), 256, 0, h->stream>>>(h->t_keys2.as<uint64_t>(), n_in, h->t_flags.as<uint32_t>());
        size_t eb = 0;
        HIPCHK(h, rocprim::exclusive_scan(nullptr, eb, h->t_flags.as<uint32_t>(), h->t_scan.as<uint32_t>(), 0u, (size_t)n_in,
                                          rocprim::plus<uint32_t>(), h->stream));
        HIPCHK(h, h->t_tmp.reserve(eb));
        HIPCHK(h, rocprim::exclusive_scan(h->t_tmp.p, eb, h->t_flags.as<uint32_t>(), h->t_scan.as<uint32_t>(), 0u, (size_t)n_in,
                                          rocprim::plus<uint32_t>(), h->stream));
        uint32_t lv[2] = {0, 0};
        HIPCHK(h, hipMemcpyAsync(&lv[0], h->t_scan.as<uint32_t>() + (n_in - 1), 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(&lv[1], h->t_flags.as<uint32_t>() + (n_in - 1), 4, hipMemcpyDeviceToHost, h->stream));
        k_vox_reduce<<<grid_for(n_in), 256, 0, h->stream>>>(h->t_keys2.as<uint64_t>(), h->t_vals2.as<uint32_t>(), n_in,
                                                            h->t_flags.as<uint32_t>(), h->t_scan.as<uint32_t>(), d_xyz, d_nrm,
                                                            d_cov, n_outs, d_oxyz, d_onrm, d_ocov);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        n_vox = (int64_t)lv[0] + lv[1];
    }
    const int64_t total = n_outs + n_vox;
    if (!on_device) {
        HIPCHK(h, hipMemcpyAsync(out_xyz, d_oxyz, (size_t)total * 24, out_kind, h->stream));
        if (normals) HIPCHK(h, hipMemcpyAsync(out_normals, d_onrm, (size_t)total * 24, out_kind, h->stream));
        if (covs) HIPCHK(h, hipMemcpyAsync(out_covs, d_ocov, (size_t)total * 72, out_kind, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (n_out) *n_out = total;
    if (n_outside) *n_outside = n_outs;
    return REG_OK;
}

reg_status reg_estimate_normals(reg_handle* h, const float* xyz, int64_t xyz_stride, int64_t n, int on_device, int k,
                                float max_dist, const float* viewpoint, int regularise, float* normals, float* eigvals,
                                float* covs, int32_t* ids, int64_t* n_rescanned) {
    if (!h) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    if (!xyz || xyz_stride < 3 || !normals || k < 1 || k > kPcaMaxK || !(max_dist > 0.f) || n > 0x7fffffffLL) {
        h->err = "reg_estimate_normals: bad argument (1 <= k <= 32, max_dist > 0, normals != NULL)";
        return REG_BAD_ARGUMENT;
    }
    if (n <= 0) {
        h->err = "The point cloud is empty";
        return REG_EMPTY_SOURCE;
    }
    if (!h->normals_ws) {
        reg_params p = h->prm;
        p.cost = REG_COST_GICP;   // no centring: neighbourhoods are formed in the input frame
        p.disable_halo = 1;
        reg_handle* w = nullptr;
        const reg_status cs = reg_create(&p, &w);
        if (cs != REG_OK) {
            h->err = std::string("reg_estimate_normals: workspace: ") + reg_last_error(w);
            reg_destroy(w);
            return cs;
        }
        w->structure_only = true;
        h->normals_ws = w;
    }
    reg_handle* w = h->normals_ws;
    (void)reg_set_stream(w, h->stream);
    w->prm.max_dist = max_dist;
    reg_status st = reg_set_target(w, xyz, xyz_stride, nullptr, 3, nullptr, n, on_device);
    if (st != REG_OK) {
        h->err = w->err;
        return st;
    }
    // first radius level expected to hold k neighbours on a surface-like cloud (exactness does not depend on it)
    const float per = (float)n / (float)std::max<int64_t>(1, w->info.n_cells_occupied);
    const float need = w->info.cell_size * std::sqrt(1.3f * (float)k / (3.14159265f * std::max(per, 1e-3f)));
    int start = 0;
    while (start < w->grid.n_levels - 1 && w->grid.rho[start] < need) ++start;
    const float* d_raw = on_device ? xyz : w->t_raw.as<float>();
    HIPCHK(h, hipSetDevice(h->prm.device));
    float *d_n = normals, *d_e = eigvals, *d_c = covs;
    int32_t* d_i = ids;
    if (!on_device) {
        HIPCHK(h, h->n_out.reserve((size_t)n * 12));
        d_n = h->n_out.as<float>();
        if (eigvals) {
            HIPCHK(h, h->n_eig.reserve((size_t)n * 12));
            d_e = h->n_eig.as<float>();
        }
        if (covs) {
            HIPCHK(h, h->n_cov.reserve((size_t)n * 24));
            d_c = h->n_cov.as<float>();
        }
        if (ids) {
            HIPCHK(h, h->n_ids.reserve((size_t)n * k * 4));
            d_i = h->n_ids.as<int32_t>();
        }
    }
    HIPCHK(h, w->t_misc.reserve(256));
    HIPCHK(h, hipMemsetAsync(w->t_misc.p, 0, 4, h->stream));
    const float vp[3] = {viewpoint ? viewpoint[0] : 0.f, viewpoint ? viewpoint[1] : 0.f, viewpoint ? viewpoint[2] : 0.f};
    const int64_t blocks = (n + (256 / kPcaGroup) - 1) / (256 / kPcaGroup);
    k_knn_pca<<<(unsigned)blocks, 256, 0, h->stream>>>(w->grid, d_raw, xyz_stride, n, k, start, vp[0], vp[1], vp[2], viewpoint ? 1 : 0,
                                                       regularise, d_n, d_e, d_c, d_i, w->t_misc.as<uint32_t>());
    uint32_t resc = 0;
    HIPCHK(h, hipMemcpyAsync(&resc, w->t_misc.p, 4, hipMemcpyDeviceToHost, h->stream));
    if (!on_device) {
        HIPCHK(h, hipMemcpyAsync(normals, d_n, (size_t)n * 12, hipMemcpyDeviceToHost, h->stream));
        if (eigvals) HIPCHK(h, hipMemcpyAsync(eigvals, d_e, (size_t)n * 12, hipMemcpyDeviceToHost, h->stream));
        if (covs) HIPCHK(h, hipMemcpyAsync(covs, d_c, (size_t)n * 24, hipMemcpyDeviceToHost, h->stream));
        if (ids) HIPCHK(h, hipMemcpyAsync(ids, d_i, (size_t)n * k * 4, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (n_rescanned) *n_rescanned = resc;
    return REG_OK;
}

reg_status reg_get_target_info(const reg_handle* h, reg_target_info* info) {
    if (!h || !info) return REG_BAD_ARGUMENT;
    if (h->m == 0) return REG_NOT_CONFIGURED;
    *info = h->info;
    return REG_OK;
}

reg_status reg_set_source(reg_handle* h, const float* xyz, int64_t xyz_stride, const float* nrm, int64_t nrm_stride,
                          const float* cov, int64_t n, int on_device) {
    if (!h) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    h->n = 0;
    h->prepared = false;
    h->have_match = false;
    if (n <= 0) {
        h->err = "The reading point cloud is empty.";
        return REG_EMPTY_SOURCE;
    }
    if (!xyz || xyz_stride < 3 || (nrm && nrm_stride < 3) || n > 0x7fffffffLL) return REG_BAD_ARGUMENT;
    if (h->prm.cost == REG_COST_P2PL && h->prm.use_surface_normal && !nrm) {
        h->err = "InvalidField: SurfaceNormalOutlierFilter needs the `normals` descriptor on the reading";
        return REG_MISSING_FIELD;
    }
    if (h->prm.cost == REG_COST_GICP && !cov) {
        h->err = "InvalidField: GICP needs covariances on the reading";
        return REG_MISSING_FIELD;
    }
    HIPCHK(h, hipSetDevice(h->prm.device));
    // packed private copies (the reference deep-copies the reading, ICP.cpp:952)
    reg_status s = upload(h, h->s_raw, xyz, (size_t)n * xyz_stride * 4, on_device);
    if (s != REG_OK) return s;
    if (nrm) {
        s = upload(h, h->s_nrm_raw, nrm, (size_t)n * nrm_stride * 4, on_device);
        if (s != REG_OK) return s;
    }
    if (cov) {
        s = upload(h, h->s_cov_raw, cov, (size_t)n * 24, on_device);
        if (s != REG_OK) return s;
    }
    h->n = n;
    h->has_snrm = nrm != nullptr;
    h->has_scov = cov != nullptr;
    // iteration buffers
    HIPCHK(h, h->s_xyz.reserve((size_t)n * 16));
    if (nrm) HIPCHK(h, h->s_nrm.reserve((size_t)n * 16));
    if (cov) HIPCHK(h, h->s_cov.reserve((size_t)n * 32));
    HIPCHK(h, h->i_pos.reserve((size_t)n * 4));
    HIPCHK(h, h->i_d2.reserve((size_t)n * 4));
    HIPCHK(h, h->i_w.reserve((size_t)n * 4));
    HIPCHK(h, h->i_hist.reserve(3 * 2048 * 4));
    HIPCHK(h, h->i_state.reserve(sizeof(SelectState)));
    h->n_blocks = grid_for(n);
    HIPCHK(h, h->i_partials.reserve((size_t)(grid_for(n * 8) + 8) * kSums * 8));
    HIPCHK(h, h->i_band.reserve((size_t)kBandCap * kRec * 4));
    HIPCHK(h, h->i_acc.reserve((size_t)kAccRows * kSums * 8));
    HIPCHK(h, h->i_sums.reserve(kSums * 8));
    HIPCHK(h, h->i_hint.reserve((size_t)n));
    h->s_stride = xyz_stride;
    h->s_nstride = nrm_stride;
    h->perm = nullptr;
    if (h->prm.sort_source) {
        // spatial (Morton) order of the reading, in its own frame: once per reading, not once per registration
        HIPCHK(h, h->s_keys.reserve((size_t)n * 4));
        HIPCHK(h, h->s_keys2.reserve((size_t)n * 4));
        HIPCHK(h, h->s_perm.reserve((size_t)n * 4));
        HIPCHK(h, h->s_perm2.reserve((size_t)n * 4));
        const float cell = h->m > 0 ? h->info.cell_size * (float)kBrickDim : 1.0f;
        k_source_keys<<<grid_for(n), 256, 0, h->stream>>>(h->s_raw.as<float>(), xyz_stride, n, 1.0f / cell,
                                                          h->s_keys.as<uint32_t>(), h->s_perm.as<uint32_t>());
        size_t tb = 0;
        HIPCHK(h, rocprim::radix_sort_pairs(nullptr, tb, h->s_keys.as<uint32_t>(), h->s_keys2.as<uint32_t>(),
                                            h->s_perm.as<uint32_t>(), h->s_perm2.as<uint32_t>(), (size_t)n, 0, 30,
                                            h->stream));
        HIPCHK(h, h->s_tmp.reserve(tb));
        HIPCHK(h, rocprim::radix_sort_pairs(h->s_tmp.p, tb, h->s_keys.as<uint32_t>(), h->s_keys2.as<uint32_t>(),
                                            h->s_perm.as<uint32_t>(), h->s_perm2.as<uint32_t>(), (size_t)n, 0, 30,
                                            h->stream));
        h->perm = h->s_perm2.as<uint32_t>();
    }
    return REG_OK;
}

}  // extern "C"

// =================================================================================================
// iteration driver (host)
// =================================================================================================

static reg_status check_ready(reg_handle* h, bool need_prepared) {
    if (!h) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    if (h->m == 0) {
        h->err = "no reference set (reg_set_target)";
        return REG_NOT_CONFIGURED;
    }
    if (h->n == 0) {
        h->err = "no reading set (reg_set_source)";
        return REG_NOT_CONFIGURED;
    }
    if (need_prepared && !h->prepared) {
        h->err = "reg_prepare has not been called for this reading";
        return REG_NOT_CONFIGURED;
    }
    return REG_OK;
}

// R2
static reg_status prepare_rowmajor(reg_handle* h, const float* T_init_row, const float* c_override = nullptr) {
    reg_status s = check_ready(h, false);
    if (s != REG_OK) return s;
    if (!m4_is_finite(T_init_row)) {
        h->err = "initial transformation contains non-finite values";
        return REG_BAD_TRANSFORM;
    }
    const bool ptrace = getenv("O3D_TRACE") != nullptr;
    const auto pt0 = std::chrono::steady_clock::now();
    auto pmark = [&](const char* what) {
        if (ptrace) fprintf(stderr, "[o3dreg] prepare %-18s t=%.1fus\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - pt0).count());
    };
    HIPCHK(h, hipSetDevice(h->prm.device));
    std::memcpy(h->T_init, T_init_row, 64);
    const int64_t n = h->n;
    const bool p2pl = h->prm.cost == REG_COST_P2PL;
    // centroid sums -> (device) centroid + T0; the host copy arrives later through the pinned staging buffer and is
    // only needed for the final composition (R10), so nothing here waits for the device
    HIPCHK(h, h->s_misc.reserve(256));
    HIPCHK(h, h->s_prep.reserve(sizeof(PrepState)));
    if (p2pl && !c_override) {
        HIPCHK(h, hipMemsetAsync(h->s_misc.p, 0, 3 * sizeof(unsigned long long), h->stream));
        const int blocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
        k_centroid_sums<<<blocks, 256, 0, h->stream>>>(h->s_raw.as<float>(), h->s_stride, n,
                                                       h->s_misc.as<unsigned long long>());
    }
    pmark("centroid");
    Xf4 Ti;
    std::memcpy(Ti.m, T_init_row, 64);
    k_make_T0<<<1, 64, 0, h->stream>>>(h->s_misc.as<unsigned long long>(), n,
                                       make_float3(h->c_ref[0], h->c_ref[1], h->c_ref[2]), Ti, p2pl ? 1 : 0,
                                       c_override ? 1 : 0,
                                       c_override ? make_float3(c_override[0], c_override[1], c_override[2])
                                                  : make_float3(0.f, 0.f, 0.f),
                                       h->s_prep.as<PrepState>(), h->d_prep_host);
    h->prep_pending = true;
    pmark("T0+copy");
    const PrepState* ps = h->s_prep.as<PrepState>();
    k_prepare_source<<<grid_for(n), 256, 0, h->stream>>>(
        h->s_raw.as<float>(), h->s_stride, (p2pl && h->has_snrm) ? h->s_nrm_raw.as<float>() : nullptr, h->s_nstride, n, ps,
        p2pl ? 1 : 0, h->perm, h->s_xyz.as<float4>(), (p2pl && h->has_snrm) ? h->s_nrm.as<float4>() : nullptr,
        h->i_hint.as<uint8_t>(), h->i_hist.as<uint32_t>(), h->i_acc.as<double>(), kAccRows * kSums);
    if (!p2pl)
        k_pack_cov<<<grid_for(n), 256, 0, h->stream>>>(h->s_cov_raw.as<float>(), n, h->perm, h->s_cov.as<float4>());
    pmark("prepare_source");
    HIPCHK(h, hipGetLastError());
    h->prepared = true;
    h->have_match = false;
    return REG_OK;
}

// ---- iteration state ---------------------------------------------------------------------------------

// (Re)initialise the device-side iteration state: pose T (row-major), mode and checker configuration.
static reg_status init_iter_state(reg_handle* h, const float* T_row, int update) {
    IterState* st = h->h_iter;
    // the pinned staging copy may still be in flight from the previous call: wait for THAT copy only (an event
    // recorded right behind it), not for everything else enqueued on the stream
    if (h->iter_copy_pending) HIPCHK(h, hipEventSynchronize(h->ev_iter));
    std::memset(st, 0, sizeof(IterState));
    for (int i = 0; i < 16; ++i) {
        st->T[i] = T_row[i];
        st->Td[i] = (double)T_row[i];
    }
    st->chk = Checkers();
    st->chk.max_iter = h->prm.max_iter;
    st->chk.min_diff_rot = h->prm.min_diff_rot;
    st->chk.min_diff_trans = h->prm.min_diff_trans;
    st->chk.smooth_len = h->prm.smooth_len;
    st->chk.init(T_row);
    st->cost = h->prm.cost;
    st->fixed_iters = h->prm.fixed_iters;
    st->max_iter = h->prm.max_iter;
    st->update = update;
    st->gicp_rot_eps = h->prm.gicp_rot_eps;
    st->gicp_trans_eps = h->prm.gicp_trans_eps;
    st->band_lo = st->band_hi = INFINITY;
    st->limit_last = st->limit_prev = INFINITY;
    st->use_trim = (h->prm.cost == REG_COST_P2PL && h->prm.use_trimmed) ? 1 : 0;
    st->trim_ratio = h->prm.trim_ratio;
    st->band_cap = kBandCap;
    st->debug_narrow_band = (h->prm.debug_flags & 8) ? 1 : 0;
    h->xicp_pending = false;
    for (int k = 0; k < 6; ++k) st->xicp_flags[k] = 1;
    if (h->prm.use_xicp && h->prm.cost == REG_COST_P2PL && update) {
        HIPCHK(h, h->i_xicp.reserve(sizeof(XicpState)));
        st->xicp_stage = 1;
        st->xicp_enough = h->prm.xicp_enough;
        st->xicp_insufficient = h->prm.xicp_insufficient;
        st->xicp_cos_min = (float)std::cos((double)h->prm.xicp_min_angle_deg * 3.14159265358979323846 / 180.0);
        st->xicp_cos_strong = (float)std::cos((double)h->prm.xicp_strong_angle_deg * 3.14159265358979323846 / 180.0);
        // T_refMean_dataIn = T_refIn_refMean^-1 * T_init (ICP.cpp:1067): the frame change of the analysis
        float A[16], Trd[16];
        m4_identity(A);
        for (int k = 0; k < 3; ++k) A[4 * k + 3] = -h->c_ref[k];
        m4_mul(A, h->T_init, Trd);
        for (int k = 0; k < 12; ++k) st->xicp_Trd[k] = Trd[k];
        h->xicp_pending = true;
    }
    HIPCHK(h, hipMemcpyAsync(h->i_iter.p, st, sizeof(IterState), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipEventRecord(h->ev_iter, h->stream));
    h->iter_copy_pending = true;
    return REG_OK;
}

// R3+R4.  Buffer hygiene of the trimmed-quantile histograms needs no memset launches: the match kernel
// zeroes hist2, the level-2 select kernel zeroes hist0, the linearize kernel zeroes hist1.
static void prof_mark(reg_handle* h, int kind, bool start) {
    if (!h->profiling) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, h->stream);
    h->prof_ev.push_back(e);
    if (start) h->prof_kind.push_back(kind);
}

static reg_status enqueue_match(reg_handle* h, bool zero_hist = false) {
    const bool trim = h->prm.cost == REG_COST_P2PL && h->prm.use_trimmed;
    if (trim && zero_hist) HIPCHK(h, hipMemsetAsync(h->i_hist.p, 0, 3 * 2048 * 4, h->stream));
    const bool fused_hist = h->prm.match_variant == 3;
    uint32_t* hist0 = (trim && fused_hist) ? h->i_hist.as<uint32_t>() : nullptr;
    uint32_t* hist2 = trim ? h->i_hist.as<uint32_t>() + 4096 : nullptr;
    const IterState* it = h->i_iter.as<IterState>();
    prof_mark(h, 0, true);
    if (h->prm.match_variant == 1) {
        k_match<<<h->n_blocks, 256, 0, h->stream>>>(h->s_xyz.as<float4>(), h->n, it, h->grid, h->i_pos.as<int>(),
                                                    h->i_d2.as<float>(), hist0, hist2, h->shift0);
    } else {
        uint8_t* hint = h->prm.match_variant == 2 ? nullptr : h->i_hint.as<uint8_t>();
        if (h->prm.lanes_per_point == 4) {
            const int blocks = grid_for(h->n * 4);
            k_match_g8<4><<<8 * ((blocks + 7) / 8), 256, 0, h->stream>>>(h->s_xyz.as<float4>(), h->n, it, h->grid,
                                                                          h->i_pos.as<int>(), h->i_d2.as<float>(), hist0,
                                                                          hist2, hint, h->shift0, h->prm.debug_flags, blocks);
        } else if (h->prm.lanes_per_point == 2) {
            const int blocks = grid_for(h->n * 2);
            k_match_g8<2><<<8 * ((blocks + 7) / 8), 256, 0, h->stream>>>(h->s_xyz.as<float4>(), h->n, it, h->grid,
                                                                          h->i_pos.as<int>(), h->i_d2.as<float>(), hist0,
                                                                          hist2, hint, h->shift0, h->prm.debug_flags, blocks);
        } else {
            const int blocks = grid_for(h->n * 8);
            k_match_g8<8><<<8 * ((blocks + 7) / 8), 256, 0, h->stream>>>(h->s_xyz.as<float4>(), h->n, it, h->grid,
                                                                          h->i_pos.as<int>(), h->i_d2.as<float>(), hist0,
                                                                          hist2, hint, h->shift0, h->prm.debug_flags, blocks);
        }
    }
    prof_mark(h, 0, false);
    h->have_match = true;
    return REG_OK;
}

// exact k-th smallest finite d2: level 0 and 1 histograms here, the last level inside the linearize kernel
static reg_status enqueue_select(reg_handle* h) {
    uint32_t* hist0 = h->i_hist.as<uint32_t>();
    SelectState* st = h->i_state.as<SelectState>();
    const IterState* it = h->i_iter.as<IterState>();
    const int hb = std::min(h->n_blocks, 128);
    const float ratio = h->prm.trim_ratio;
    if (h->prm.match_variant != 3)
        k_hist_level0<<<hb, 256, 0, h->stream>>>(h->i_d2.as<float>(), h->n, h->shift0, hist0, it);
    k_select_level<<<hb, 256, 0, h->stream>>>(h->i_d2.as<float>(), h->n, 1, h->shift0, ratio, hist0, hist0 + 2048,
                                              nullptr, st, it);
    k_select_level<<<hb, 256, 0, h->stream>>>(h->i_d2.as<float>(), h->n, 2, h->shift0, ratio, hist0 + 2048,
                                              hist0 + 4096, hist0, st, it);
    return REG_OK;
}

static FilterCfg make_filter_cfg(const reg_handle* h, int trim_mode);

static reg_status enqueue_linearize(reg_handle* h, bool want_w, bool limit_from_state = false) {
    float* w = want_w ? h->i_w.as<float>() : nullptr;
    const IterState* it = h->i_iter.as<IterState>();
    if (h->prm.cost == REG_COST_P2PL) {
        const FilterCfg f = make_filter_cfg(h, h->prm.use_trimmed ? (limit_from_state ? 1 : 2) : 0);
        k_linearize_p2pl<<<h->n_blocks, 256, 0, h->stream>>>(
            h->s_xyz.as<float4>(), h->has_snrm ? h->s_nrm.as<float4>() : nullptr, h->n, it, h->i_pos.as<int>(),
            h->i_d2.as<float>(), h->t_pts.as<float4>(), h->t_nrm.as<float4>(), f, h->i_state.as<SelectState>(),
            h->i_hist.as<uint32_t>() + 4096, h->i_hist.as<uint32_t>() + 2048, h->shift0, w,
            h->i_partials.as<double>());
    } else {
        k_linearize_gicp<<<h->n_blocks, 256, 0, h->stream>>>(h->s_xyz.as<float4>(), h->s_cov.as<float4>(), h->n, it,
                                                             h->i_pos.as<int>(), h->i_d2.as<float>(),
                                                             h->t_pts.as<float4>(), h->t_cov.as<float4>(), w,
                                                             h->i_partials.as<double>());
    }
    ++h->seq;
    k_reduce_update<<<1, 1024, 0, h->stream>>>(h->i_partials.as<double>(), h->n_blocks, h->i_iter.as<IterState>(),
                                               h->d_mirror, h->seq, 0, nullptr, nullptr,
                                               h->prm.cost == REG_COST_P2PL ? h->i_state.as<SelectState>() : nullptr,
                                               nullptr, 0, 0, h->i_xicp.as<XicpState>());
    if (h->xicp_pending) {
        // R8x, first iteration: collect the information sums on the matched pairs, then decide + solve + update
        h->xicp_pending = false;
        const int blocks = (int)std::min<int64_t>(512, (h->n + 255) / 256);
        k_xicp_center<<<blocks, 256, 0, h->stream>>>(h->s_xyz.as<float4>(), h->n, h->i_iter.as<IterState>(),
                                                     h->i_pos.as<int>(), h->i_w.as<float>(), h->i_xicp.as<XicpState>());
        k_xicp_detect<<<blocks, 256, 0, h->stream>>>(h->s_xyz.as<float4>(), h->n, h->i_iter.as<IterState>(),
                                                     h->i_pos.as<int>(), h->i_w.as<float>(), h->t_nrm.as<float4>(),
                                                     h->i_xicp.as<XicpState>());
        k_reduce_update<<<1, 1024, 0, h->stream>>>(nullptr, 0, h->i_iter.as<IterState>(), h->d_mirror, h->seq, 2, nullptr,
                                                   nullptr, nullptr, nullptr, 0, 0, h->i_xicp.as<XicpState>());
    }
    return REG_OK;
}

static FilterCfg make_filter_cfg(const reg_handle* h, int trim_mode) {
    FilterCfg f;
    f.use_trim = trim_mode;
    f.use_normal = h->prm.use_surface_normal;
    f.use_maxdist = h->prm.use_max_dist_filter;
    f.debug = h->prm.debug_flags;
    f.cos_max_angle = std::cos(h->prm.max_normal_angle);  // cosf in T=float (OutlierFiltersImpl.cpp:229)
    const float md = h->prm.outlier_max_dist;
    f.outlier_max_d2 = md * md;
    return f;
}

// Fused iteration (point-to-plane): search + weights + normal equations in one kernel, band resolution +
// solve + update in the second.  Two launches per Gauss-Newton iteration.
template <int G>
static void launch_fused(reg_handle* h, const FilterCfg& f, float* w, uint8_t* hint) {
    const int blocks = grid_for(h->n * G);
    prof_mark(h, 1, true);
    k_iter_fused<G><<<8 * ((blocks + 7) / 8), 256, 0, h->stream>>>(
        h->s_xyz.as<float4>(), h->has_snrm ? h->s_nrm.as<float4>() : nullptr, h->n, h->i_iter.as<IterState>(), h->grid,
        h->t_nrm.as<float4>(), f, h->i_pos.as<int>(), h->i_d2.as<float>(), w, hint, h->i_band.as<float>(), kBandCap,
        h->i_acc.as<double>(), blocks);
    prof_mark(h, 1, false);
    ++h->seq;
    k_reduce_update<<<1, 1024, 0, h->stream>>>(h->i_acc.as<double>(), blocks, h->i_iter.as<IterState>(), h->d_mirror,
                                               h->seq, 1, h->i_band.as<float>(), w, nullptr, nullptr, 0, 0, nullptr);
}

static reg_status enqueue_fused(reg_handle* h, bool want_w) {
    const FilterCfg f = make_filter_cfg(h, 0);
    float* w = want_w ? h->i_w.as<float>() : nullptr;
    uint8_t* hint = h->prm.match_variant == 2 ? nullptr : h->i_hint.as<uint8_t>();
    if (h->prm.lanes_per_point == 4)
        launch_fused<4>(h, f, w, hint);
    else
        launch_fused<8>(h, f, w, hint);
    h->have_match = true;
    return REG_OK;
}

// One Gauss-Newton iteration worth of kernels (R3-R9), nothing waits on the host.
static reg_status enqueue_iteration(reg_handle* h, bool want_w) {
    reg_status s = enqueue_match(h);
    if (s != REG_OK) return s;
    if (h->prm.cost == REG_COST_P2PL && h->prm.use_trimmed) {
        s = enqueue_select(h);
        if (s != REG_OK) return s;
    }
    return enqueue_linearize(h, want_w);
}

static inline unsigned long long mirror_seq(const reg_handle* h) {
    return __atomic_load_n(&h->h_mirror->seq, __ATOMIC_ACQUIRE);
}

// Wait until the update kernel with sequence number `seq` (or a later one) has published its mirror.
// Polling the mapped word is ~10 us cheaper per wait than hipStreamSynchronize; a stream query every few
// thousand spins turns a device fault (or an early `done`) into a return instead of a hang.
static reg_status wait_seq(reg_handle* h, unsigned long long seq) {
    for (unsigned spins = 0;; ++spins) {
        if (mirror_seq(h) >= seq) return REG_OK;
        if ((spins & 0x3fff) == 0x3fff) {
            hipError_t e = hipStreamQuery(h->stream);
            if (e == hipSuccess) return REG_OK;  // everything enqueued has run (later iterations were no-ops)
            if (e != hipErrorNotReady) {
                h->err = std::string("device fault while waiting for the iteration: ") + hipGetErrorString(e);
                return REG_DEVICE_ERROR;
            }
        }
    }
}

// one full pass R3-R7 at T (row-major) without pose update; sums -> h->h_mirror->sums
static reg_status iterate_once(reg_handle* h, const float* T_row, bool want_w) {
    reg_status s = init_iter_state(h, T_row, 0);
    if (s != REG_OK) return s;
    s = enqueue_iteration(h, want_w);
    if (s != REG_OK) return s;
    s = wait_seq(h, h->seq);
    if (s != REG_OK) return s;
    HIPCHK(h, hipGetLastError());
    return REG_OK;
}

static void sums_to_system(const double* sums, bool p2pl, float* H, float* b) {
    int k = 0;
    for (int a = 0; a < 6; ++a)
        for (int c = a; c < 6; ++c) {
            const float v = (float)sums[k++];
            H[6 * a + c] = v;
            H[6 * c + a] = v;
        }
    for (int a = 0; a < 6; ++a) b[a] = p2pl ? -(float)sums[21 + a] : (float)sums[21 + a];
}

// R10: T = T_refIn_refMean * T_iter * T_refMean_readMean * T_readIn_readMean^-1 (ICP.cpp:1345); GICP: T_iter itself
static void compose_rowmajor(reg_handle* h, const float* T_iter, float* Tout_row, bool later_kernel_reported = false) {
    if (h->prep_pending) {
        // k_make_T0 wrote PrepState into mapped host memory (system-scope fence); it is visible once that kernel has
        // completed: either a later kernel of the same stream has already reported through the mirror, or wait here
        if (!later_kernel_reported) (void)hipStreamSynchronize(h->stream);
        std::memcpy(h->c_read, h->h_prep->c_read, 12);
        std::memcpy(h->T0, h->h_prep->T0, 64);
        h->prep_pending = false;
    }
    if (h->prm.cost == REG_COST_P2PL) {
        float A[16], B[16], t1[16], t2[16];
        m4_identity(A);
        m4_identity(B);
        for (int k = 0; k < 3; ++k) {
            A[4 * k + 3] = h->c_ref[k];
            B[4 * k + 3] = -h->c_read[k];
        }
        m4_mul(A, T_iter, t1);
        m4_mul(t1, h->T0, t2);
        m4_mul(t2, B, Tout_row);
    } else {
        std::memcpy(Tout_row, T_iter, 64);
    }
}

static void fill_result(reg_handle* h, const double* sums, reg_result* res) {
    res->n_inliers = (int64_t)llround(sums[28]);
    res->n_matched = (int64_t)llround(sums[29]);
    res->error = sums[27];
    res->fitness = h->n > 0 ? sums[28] / (double)h->n : 0.0;
    res->inlier_rmse = sums[28] > 0 ? std::sqrt(sums[30] / sums[28]) : 0.0;
    sums_to_system(sums, h->prm.cost == REG_COST_P2PL, res->H_last, res->b_last);
    res->target_build_ms = h->target_build_ms;
}

extern "C" {

reg_status reg_prepare(reg_handle* h, const float T_init[16]) {
    if (!h || !T_init) return REG_BAD_ARGUMENT;
    float Tr[16];
    col_to_row(T_init, Tr);
    return prepare_rowmajor(h, Tr);
}

reg_status reg_linearize(reg_handle* h, const float T_iter[16], float H[36], float b[6], double* err,
                         int64_t* n_inliers) {
    reg_status s = check_ready(h, true);
    if (s != REG_OK) return s;
    if (!T_iter) return REG_BAD_ARGUMENT;
    float Tr[16];
    col_to_row(T_iter, Tr);
    if (!m4_is_finite(Tr)) return REG_BAD_TRANSFORM;
    HIPCHK(h, hipSetDevice(h->prm.device));
    s = iterate_once(h, Tr, true);
    if (s != REG_OK) return s;
    const double* sums = h->h_mirror->sums;
    if (H && b) sums_to_system(sums, h->prm.cost == REG_COST_P2PL, H, b);
    if (err) *err = sums[27];
    if (n_inliers) *n_inliers = (int64_t)llround(sums[28]);
    if (sums[28] == 0.0) {
        h->err = "ErrorMinimizer: no point to minimize";
        return REG_NO_CORRESPONDENCES;
    }
    return REG_OK;
}

// == ICP::compute on the prepared reading.  The whole while(iterate) loop (ICP.cpp:1027-1311) runs on the
// device; the host only keeps the queue fed.  fixed_iters > 0: every iteration is enqueued at once.
// Checker mode: the host stays at most kAhead iterations ahead of what it has seen complete, so at most
// kAhead enqueued iterations turn into no-ops after convergence.
reg_status reg_register(reg_handle* h, const float T_init[16], float T_out[16], reg_result* res) {
    if (!h || !T_init || !T_out) return REG_BAD_ARGUMENT;
    reg_result local;
    if (!res) res = &local;
    std::memset(res, 0, sizeof(*res));
    std::memcpy(T_out, T_init, 64);
    float Ti[16];
    col_to_row(T_init, Ti);
    const auto t_reg0 = std::chrono::steady_clock::now();
    auto rmark = [&](const char* what) {
        if (getenv("O3D_TRACE"))
            fprintf(stderr, "[o3dreg] register %-14s t=%.1fus\n", what,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_reg0).count());
    };
    reg_status s = prepare_rowmajor(h, Ti);
    if (s != REG_OK) return s;
    rmark("prepared");
    const bool p2pl = h->prm.cost == REG_COST_P2PL;
    float T_start[16];
    if (p2pl)
        m4_identity(T_start);
    else
        std::memcpy(T_start, Ti, 64);
    s = init_iter_state(h, T_start, 1);
    if (s != REG_OK) return s;
    rmark("iter state");
    h->profiling = h->prm.profile_loop != 0;
    // loop_ms: HIP events only when profiling (record + synchronise cost ~20 us of host time per registration);
    // otherwise the host clock around the loop -- the loop ends when the last update kernel's mirror has arrived
    const bool event_timing = h->profiling || getenv("O3D_EVENT_TIMING") != nullptr;
    if (event_timing) HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    const auto t_loop_begin = std::chrono::steady_clock::now();
    rmark("ev0");
    const unsigned long long seq0 = h->seq;
    const int fixed = h->prm.fixed_iters;
    const int limit = fixed > 0 ? fixed : h->prm.max_iter;
    // Iterations 0..kGenericFirst-1 run on the generic (select-based) path: the trimmed limit still moves too
    // much to be predicted.  Afterwards the fused two-kernel iteration is used; if its band prediction fails the
    // device stalls the queue and the host repairs that iteration on the generic path.
    const bool can_fuse = p2pl && h->prm.disable_fused != 1;
    const bool trimming = p2pl && h->prm.use_trimmed && h->prm.trim_ratio != 1.0f;
    const int kGenericFirst = trimming ? 2 : 1;
    const int kAhead = getenv("O3D_KAHEAD") ? atoi(getenv("O3D_KAHEAD")) : 2;
    const HostMirror* mir = h->h_mirror;
    int generic_left = kGenericFirst;
    const bool trace = getenv("O3D_TRACE") != nullptr;
    const float settle_tol = getenv("O3D_SETTLE") ? (float)atof(getenv("O3D_SETTLE")) : 0.05f;
    unsigned long long last_traced = 0;
    const auto t_loop0 = std::chrono::steady_clock::now();
    unsigned long long acked = seq0;   // every sequence <= acked has either reported or been a no-op
    int stalls = 0;
    for (;;) {
        const unsigned long long m_seq = std::max(mirror_seq(h), seq0);
        const bool any = m_seq > seq0;
        if (any && mir->done) break;
        if (any && mir->stall && m_seq > acked) {
            // band prediction failed at sequence m_seq: everything enqueued behind it is a no-op; drain, repair
            HIPCHK(h, hipStreamSynchronize(h->stream));
            acked = h->seq;
            generic_left = 2;
            ++stalls;
            continue;
        }
        if (trace && m_seq != last_traced) {
            last_traced = m_seq;
            fprintf(stderr, "[o3dreg] seq %llu iter %d stall %d band_n %d limit %.6g prev %.6g band [%.6g, %.6g)\n",
                    m_seq - seq0, mir->iterations, mir->stall, mir->stall ? mir->band_count : mir->pad_nband, mir->limit_last, mir->limit_prev,
                    mir->band_lo, mir->band_hi);
        }
        acked = std::max(acked, m_seq);
        const int completed = any ? mir->iterations : 0;
        const int inflight = (int)(h->seq - acked);
        if (completed + inflight < limit && inflight < kAhead) {
            // fuse only once the trimmed limit has settled (last two seen limits within 5 %): the band is then
            // narrow (few hundred records) and the prediction practically never fails
            bool settled = true;
            if (trimming) {
                settled = any && mir->limit_prev < INFINITY && mir->limit_last < INFINITY &&
                          std::fabs(mir->limit_last - mir->limit_prev) <= settle_tol * mir->limit_last;
            }
            const auto tq0 = std::chrono::steady_clock::now();
            const bool go_generic = !can_fuse || generic_left > 0 || !settled;
            if (go_generic) {
                s = enqueue_iteration(h, true);   // weights are always written: reg_get_correspondences reports them
                if (generic_left > 0) --generic_left;
            } else {
                // Fixed iteration count: nothing the host could learn changes what has to run, so the whole rest of
                // the registration is submitted in one go (a failed band prediction turns what follows into no-ops
                // and is repaired above).  Submitting while the device crosses a kernel boundary costs about 6 us per
                // iteration (measured: rocprofv3 timeline, profiles/), hence no trickle-feeding here.
                int burst = fixed > 0 && !getenv("O3D_NO_BURST") ? limit - (completed + inflight) : 1;
                for (; burst > 0 && s == REG_OK; --burst) s = enqueue_fused(h, true);
            }
            if (trace) {
                const auto tq1 = std::chrono::steady_clock::now();
                fprintf(stderr, "[o3dreg] t=%.1fus enqueue seq %llu (%s) took %.1fus; mirror at %llu\n",
                        std::chrono::duration<double, std::micro>(tq0 - t_loop0).count(), h->seq - seq0,
                        go_generic ? "generic" : "fused", std::chrono::duration<double, std::micro>(tq1 - tq0).count(),
                        std::max(mirror_seq(h), seq0) - seq0);
            }
            if (s != REG_OK) return s;
            continue;
        }
        if (inflight == 0) break;  // nothing in flight and nothing left to enqueue
        s = wait_seq(h, acked + 1);
        if (s != REG_OK) return s;
        if (mirror_seq(h) <= acked) {
            // the stream drained without a report: the remaining sequences were no-ops (done or stalled earlier)
            if (hipStreamQuery(h->stream) == hipSuccess && mirror_seq(h) <= acked) acked = h->seq;
        }
    }
    h->last_stalls = stalls;
    rmark("loop done");
    if (event_timing) {
        HIPCHK(h, hipEventRecord(h->ev1, h->stream));
        HIPCHK(h, hipEventSynchronize(h->ev1));
        (void)hipEventElapsedTime(&res->loop_ms, h->ev0, h->ev1);
    } else {
        res->loop_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_loop_begin).count();
    }
    HIPCHK(h, hipGetLastError());
    rmark("loop timed");
    if (getenv("O3D_STAMPS")) {
        fprintf(stderr, "update kernel stamps (cycles): reduce %llu [rows %llu select %llu band-add %llu] solve %llu update+check %llu mirror %llu\n", mir->stamps[0],
                mir->stamps[4], mir->stamps[5], mir->stamps[6], mir->stamps[1], mir->stamps[2], mir->stamps[3]);
        fprintf(stderr, "   select detail: verify+stage-issue %llu, zero+hist+barriers %llu, wave0 pick/rank %llu\n", mir->stamps[3], mir->stamps[7] >> 32, mir->stamps[7] & 0xffffffffull);
    }
    if (h->profiling) {
        for (size_t i = 0; i + 1 < h->prof_ev.size(); i += 2) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]) == hipSuccess) {
                const int kind = h->prof_kind[i / 2];
                res->prof_ms[kind] += ms;
                res->prof_launches[kind] += 1;
            }
        }
        for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
        h->prof_ev.clear();
        h->prof_kind.clear();
        h->profiling = false;
    }
    res->iterations = mir->iterations;
    for (int k = 0; k < 6; ++k) {
        res->localizable[k] = h->prm.use_xicp ? mir->localizable[k] : 1;
        res->xicp_combined[k] = mir->xicp_comb[k];
        res->xicp_high[k] = mir->xicp_high[k];
    }
    res->n_constraints = h->prm.use_xicp ? mir->n_constraints : 0;
    res->converged = mir->converged;
    res->max_iter_reached = mir->max_iter_reached;
    res->rank_last = mir->rank_last;
    fill_result(h, mir->sums, res);
    if (mir->status != REG_OK) {
        h->err = mir->sums[29] == 0.0 ? "No matches available for computing distance quantiles"
                                      : "ErrorMinimizer: no point to minimize";
        return (reg_status)mir->status;
    }
    float T_iter[16], Tout_row[16];
    std::memcpy(T_iter, mir->T, 64);
    compose_rowmajor(h, T_iter, Tout_row, /*later_kernel_reported=*/true);
    row_to_col(T_iter, res->T_iter_last);
    row_to_col(Tout_row, T_out);
    res->n_band_stalls = h->last_stalls;
    rmark("return");
    return REG_OK;
}

reg_status reg_compute(reg_handle* h, const float* xyz, int64_t xyz_stride, const float* nrm, int64_t nrm_stride,
                       const float* cov, int64_t n, int on_device, const float T_init[16], float T_out[16],
                       reg_result* res) {
    reg_status s = reg_set_source(h, xyz, xyz_stride, nrm, nrm_stride, cov, n, on_device);
    if (s != REG_OK) return s;
    return reg_register(h, T_init, T_out, res);
}

reg_status reg_get_correspondences(reg_handle* h, int32_t* ids, float* d2, float* w) {
    reg_status s = check_ready(h, true);
    if (s != REG_OK) return s;
    if (!h->have_match) {
        h->err = "no iteration has run yet";
        return REG_NOT_CONFIGURED;
    }
    HIPCHK(h, hipSetDevice(h->prm.device));
    const int64_t n = h->n;
    if (ids) {
        HIPCHK(h, h->i_ids.reserve((size_t)n * 4));
        k_ids_from_pos<<<grid_for(n), 256, 0, h->stream>>>(h->i_pos.as<int>(), h->t_pts.as<float4>(), n, h->perm,
                                                           h->i_ids.as<int32_t>());
        HIPCHK(h, hipMemcpyAsync(ids, h->i_ids.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, h->i_tmpf.reserve((size_t)n * 8));
    if (d2) {
        k_unpermute_f32<<<grid_for(n), 256, 0, h->stream>>>(h->i_d2.as<float>(), n, h->perm, h->i_tmpf.as<float>());
        HIPCHK(h, hipMemcpyAsync(d2, h->i_tmpf.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    }
    if (w) {
        k_unpermute_f32<<<grid_for(n), 256, 0, h->stream>>>(h->i_w.as<float>(), n, h->perm, h->i_tmpf.as<float>() + n);
        HIPCHK(h, hipMemcpyAsync(w, h->i_tmpf.as<float>() + n, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return REG_OK;
}

// ---- distributed halves ----

reg_status reg_source_centroid_sums(reg_handle* h, int64_t sums[3]) {
    reg_status s = check_ready(h, false);
    if (s != REG_OK) return s;
    if (!sums) return REG_BAD_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->prm.device));
    long long t[3];
    s = device_centroid_sums(h, h->s_raw.as<float>(), h->s_stride, h->n, h->s_misc, t);
    if (s != REG_OK) return s;
    for (int k = 0; k < 3; ++k) sums[k] = t[k];
    return REG_OK;
}

reg_status reg_prepare_centroid(reg_handle* h, const float T_init[16], const float c_read[3]) {
    if (!h || !T_init || !c_read) return REG_BAD_ARGUMENT;
    float Tr[16];
    col_to_row(T_init, Tr);
    return prepare_rowmajor(h, Tr, c_read);
}

reg_status reg_compose(reg_handle* h, const float T_iter[16], float T_out[16]) {
    reg_status s = check_ready(h, true);
    if (s != REG_OK) return s;
    if (!T_iter || !T_out) return REG_BAD_ARGUMENT;
    float Tr[16], To[16];
    col_to_row(T_iter, Tr);
    compose_rowmajor(h, Tr, To);
    row_to_col(To, T_out);
    return REG_OK;
}

reg_status reg_match_local(reg_handle* h, const float T_iter[16]) {
    reg_status s = check_ready(h, true);
    if (s != REG_OK) return s;
    float Tr[16];
    col_to_row(T_iter, Tr);
    HIPCHK(h, hipSetDevice(h->prm.device));
    s = init_iter_state(h, Tr, 0);
    if (s != REG_OK) return s;
    s = enqueue_match(h, true);
    if (s != REG_OK) return s;
    HIPCHK(h, hipMemsetAsync(h->i_state.p, 0, sizeof(SelectState), h->stream));
    return REG_OK;
}

reg_status reg_trim_histogram(reg_handle* h, int level, uint32_t prefix, uint32_t hist[2048]) {
    reg_status s = check_ready(h, true);
    if (s != REG_OK) return s;
    if (level < 0 || level > 2 || !hist || !h->have_match) return REG_BAD_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->prm.device));
    // generic 11/11/10-bit split, independent of the single-GPU pipeline's histograms
    uint32_t* d_hist = h->i_hist.as<uint32_t>() + 2048 * level;
    HIPCHK(h, hipMemsetAsync(d_hist, 0, 2048 * 4, h->stream));
    k_hist_prefix<<<std::min(h->n_blocks, 256), 256, 0, h->stream>>>(h->i_d2.as<float>(), h->n, level, prefix, d_hist);
    HIPCHK(h, hipMemcpyAsync(hist, d_hist, 2048 * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return REG_OK;
}

// R5-R7 on this rank's slice for the pose given to the preceding reg_match_local
reg_status reg_reduce_local(reg_handle* h, const float T_iter[16], float trim_limit, double sums[32]) {
    reg_status s = check_ready(h, true);
    if (s != REG_OK) return s;
    if (!h->have_match || !sums || !T_iter) return REG_BAD_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->prm.device));
    SelectState st;
    std::memset(&st, 0, sizeof(st));
    st.limit = trim_limit;
    st.done = 1;
    HIPCHK(h, hipMemcpyAsync(h->i_state.p, &st, sizeof(st), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // `st` is a stack variable
    s = enqueue_linearize(h, true, true);
    if (s != REG_OK) return s;
    s = wait_seq(h, h->seq);
    if (s != REG_OK) return s;
    std::memcpy(sums, h->h_mirror->sums, kSums * 8);
    return REG_OK;
}

// R8 (+ T_iter update) from globally summed sums; pure host code, identical on every rank.
reg_status reg_solve_update(const reg_params* p, const double sums[32], const float T_iter[16], float T_next[16],
                            int32_t* rank) {
    if (!p || !sums || !T_iter || !T_next) return REG_BAD_ARGUMENT;
    float Tr[16], H[36], b[6];
    col_to_row(T_iter, Tr);
    if (sums[28] == 0.0) return REG_NO_CORRESPONDENCES;
    if (p->cost == REG_COST_P2PL) {
        sums_to_system(sums, true, H, b);
        float x[6], dT[16];
        const int r = solve6_p2pl_fast(H, b, x);
        if (rank) *rank = r;
        x_to_T(x, dT);
        m4_mul(dT, Tr, Tr);
    } else {
        double Hd[36], g[6], dl[6], E[16], Tn[16];
        int k = 0;
        for (int a = 0; a < 6; ++a)
            for (int c = a; c < 6; ++c) Hd[6 * a + c] = Hd[6 * c + a] = sums[k++];
        for (int a = 0; a < 6; ++a) g[a] = -sums[21 + a];
        int r = 6;
        if (!solve_ldlt6(Hd, g, dl, 1e-10)) r = solve_sym6(Hd, g, dl, 1e-12);
        if (rank) *rank = r;
        se3_exp(dl, E);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double t = 0;
                for (int kk = 0; kk < 4; ++kk) t += (double)Tr[4 * i + kk] * E[4 * kk + j];
                Tn[4 * i + j] = t;
            }
        for (int i = 0; i < 16; ++i) Tr[i] = (float)Tn[i];
    }
    row_to_col(Tr, T_next);
    return REG_OK;
}

// ---- stream-ordered distributed path --------------------------------------------------------------
// The multi-GPU loop without a host round trip per iteration: every phase only ENQUEUES kernels on the handle's
// stream; between the phases the caller all-reduces (RCCL, same stream) the buffers returned by reg_dist_buffers.
//   phase 0: R3+R4 on this rank's slice, level-0 histogram of d2          -> all-reduce hist[0..2048)
//   phase 1: pick level 0 from the GLOBAL histogram, build level 1        -> all-reduce hist[2048..4096)
//   phase 2: pick level 1, build level 2                                  -> all-reduce hist[4096..6144)
//   phase 3: weights (exact global trimmed limit) + normal equations      -> all-reduce sums[0..32)
//   phase 4: R8+R9 on the device from the global sums (identical on every rank)
// Without TrimmedDist (or GICP) phases 1 and 2 are no-ops and no histogram needs reducing.
reg_status reg_dist_begin(reg_handle* h, const float T_start[16]) {
    reg_status s = check_ready(h, true);
    if (s != REG_OK) return s;
    if (h->prm.use_xicp) {
        h->err = "use_xicp: the localizability analysis is not available on the distributed path (its information sums "
                 "are not exchanged between ranks yet)";
        return REG_BAD_ARGUMENT;
    }
    float Tr[16];
    if (T_start) {
        col_to_row(T_start, Tr);
    } else if (h->prm.cost == REG_COST_P2PL) {
        m4_identity(Tr);
    } else {
        std::memcpy(Tr, h->T_init, 64);
    }
    HIPCHK(h, hipSetDevice(h->prm.device));
    s = init_iter_state(h, Tr, 1);
    if (s != REG_OK) return s;
    HIPCHK(h, hipMemsetAsync(h->i_hist.p, 0, 3 * 2048 * 4, h->stream));
    h->dist_seq0 = h->seq;
    return REG_OK;
}

reg_status reg_dist_buffers(reg_handle* h, void** hist, void** sums) {
    if (!h || !hist || !sums) return REG_BAD_ARGUMENT;
    if (h->n == 0) return REG_NOT_CONFIGURED;
    *hist = h->i_hist.p;
    *sums = h->i_sums.p;
    return REG_OK;
}

// Buffers of the fused multi-GPU iteration (phases 5 and 6): `contrib` is this rank's block (contrib_bytes), `gathered`
// receives the blocks of all `n_ranks` ranks in rank order (one all-gather between phase 5 and phase 6).
reg_status reg_dist_fused_buffers(reg_handle* h, int n_ranks, int rank, void** contrib, void** gathered,
                                  int64_t* contrib_bytes) {
    if (!h || !contrib || !gathered || !contrib_bytes || n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks)
        return REG_BAD_ARGUMENT;
    if (h->n == 0) return REG_NOT_CONFIGURED;
    HIPCHK(h, hipSetDevice(h->prm.device));
    HIPCHK(h, h->d_contrib.reserve((size_t)kContribFloats * 4));
    HIPCHK(h, h->d_gathered.reserve((size_t)n_ranks * kContribFloats * 4));
    HIPCHK(h, hipMemsetAsync(h->d_contrib.p, 0, (size_t)kContribFloats * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_gathered.p, 0, (size_t)n_ranks * kContribFloats * 4, h->stream));
    h->dist_ranks = n_ranks;
    h->dist_rank = rank;
    *contrib = h->d_contrib.p;
    *gathered = h->d_gathered.p;
    *contrib_bytes = (int64_t)kContribFloats * 4;
    return REG_OK;
}

// Non-blocking view of the mirror the update kernel writes (the stream-ordered drivers steer by it).
reg_status reg_dist_poll(reg_handle* h, reg_dist_status* out) {
    if (!h || !out) return REG_BAD_ARGUMENT;
    const HostMirror* mir = h->h_mirror;
    const unsigned long long s = mirror_seq(h);
    out->sequences_done = s > h->dist_seq0 ? (int64_t)(s - h->dist_seq0) : 0;
    out->sequences_enqueued = (int64_t)(h->seq - h->dist_seq0);
    const bool any = s > h->dist_seq0;
    out->iterations = any ? mir->iterations : 0;
    out->done = any ? mir->done : 0;
    out->stall = any ? mir->stall : 0;
    out->limit_last = any ? mir->limit_last : INFINITY;
    out->limit_prev = any ? mir->limit_prev : INFINITY;
    out->stream_idle = hipStreamQuery(h->stream) == hipSuccess ? 1 : 0;
    return REG_OK;
}

reg_status reg_dist_phase(reg_handle* h, int phase) {
    reg_status s = check_ready(h, true);
    if (s != REG_OK) return s;
    const bool trim = h->prm.cost == REG_COST_P2PL && h->prm.use_trimmed;
    uint32_t* hist0 = h->i_hist.as<uint32_t>();
    SelectState* st = h->i_state.as<SelectState>();
    const IterState* it = h->i_iter.as<IterState>();
    const int hb = std::min(h->n_blocks, 128);
    switch (phase) {
        case 0:
            s = enqueue_match(h);
            if (s != REG_OK) return s;
            if (trim) k_hist_level0<<<hb, 256, 0, h->stream>>>(h->i_d2.as<float>(), h->n, h->shift0, hist0, it);
            break;
        case 1:
            if (trim)
                k_select_level<<<hb, 256, 0, h->stream>>>(h->i_d2.as<float>(), h->n, 1, h->shift0, h->prm.trim_ratio,
                                                          hist0, hist0 + 2048, nullptr, st, it);
            break;
        case 2:
            if (trim)
                k_select_level<<<hb, 256, 0, h->stream>>>(h->i_d2.as<float>(), h->n, 2, h->shift0, h->prm.trim_ratio,
                                                          hist0 + 2048, hist0 + 4096, hist0, st, it);
            break;
        case 3: {
            if (h->prm.cost == REG_COST_P2PL) {
                const FilterCfg f = make_filter_cfg(h, h->prm.use_trimmed ? 2 : 0);
                k_linearize_p2pl<<<h->n_blocks, 256, 0, h->stream>>>(
                    h->s_xyz.as<float4>(), h->has_snrm ? h->s_nrm.as<float4>() : nullptr, h->n, it, h->i_pos.as<int>(),
                    h->i_d2.as<float>(), h->t_pts.as<float4>(), h->t_nrm.as<float4>(), f, st, hist0 + 4096, hist0 + 2048,
                    h->shift0, nullptr, h->i_partials.as<double>());
            } else {
                k_linearize_gicp<<<h->n_blocks, 256, 0, h->stream>>>(h->s_xyz.as<float4>(), h->s_cov.as<float4>(), h->n, it,
                                                                     h->i_pos.as<int>(), h->i_d2.as<float>(),
                                                                     h->t_pts.as<float4>(), h->t_cov.as<float4>(), nullptr,
                                                                     h->i_partials.as<double>());
            }
            k_partials_sum<<<1, 1024, 0, h->stream>>>(h->i_partials.as<double>(), h->n_blocks, h->i_sums.as<double>(), it);
            break;
        }
        case 4:
            ++h->seq;
            k_reduce_update<<<1, 1024, 0, h->stream>>>(h->i_sums.as<double>(), 1, h->i_iter.as<IterState>(), h->d_mirror,
                                                       h->seq, 0, nullptr, nullptr,
                                                       h->prm.cost == REG_COST_P2PL ? st : nullptr, nullptr, 0, 0, nullptr);
            break;
        case 5: {
            // fused iteration, local half: search + weights + normal equations + band records (into this rank's
            // contribution block), then the block header.  Followed by the caller's ONE all-gather.
            if (h->prm.cost != REG_COST_P2PL || h->dist_ranks <= 0) return REG_BAD_ARGUMENT;
            const FilterCfg f = make_filter_cfg(h, 0);
            uint8_t* hint = h->prm.match_variant == 2 ? nullptr : h->i_hint.as<uint8_t>();
            const int blocks = grid_for(h->n * 8);
            float* contrib = h->d_contrib.as<float>();
            k_iter_fused<8><<<8 * ((blocks + 7) / 8), 256, 0, h->stream>>>(
                h->s_xyz.as<float4>(), h->has_snrm ? h->s_nrm.as<float4>() : nullptr, h->n, h->i_iter.as<IterState>(),
                h->grid, h->t_nrm.as<float4>(), f, h->i_pos.as<int>(), h->i_d2.as<float>(), h->i_w.as<float>(), hint,
                contrib + kContribHdr, kContribCap, h->i_acc.as<double>(), blocks);
            k_pack_contrib<<<1, 64, 0, h->stream>>>(h->i_acc.as<double>(), it, contrib);
            h->have_match = true;
            break;
        }
        case 6:
            // fused iteration, global half: every rank reduces the SAME gathered blocks in the same order
            if (h->dist_ranks <= 0) return REG_BAD_ARGUMENT;
            ++h->seq;
            k_reduce_update<<<1, 1024, 0, h->stream>>>(nullptr, 0, h->i_iter.as<IterState>(), h->d_mirror, h->seq, 1,
                                                       nullptr, h->i_w.as<float>(), nullptr, h->d_gathered.as<float>(),
                                                       h->dist_ranks, h->dist_rank, nullptr);
            break;
        default:
            return REG_BAD_ARGUMENT;
    }
    return REG_OK;
}

// Waits for everything enqueued on the stream, then reports like reg_register (T_out composed with R10).
reg_status reg_dist_finish(reg_handle* h, float T_out[16], reg_result* res) {
    reg_status s = check_ready(h, true);
    if (s != REG_OK) return s;
    if (!T_out) return REG_BAD_ARGUMENT;
    reg_result local;
    if (!res) res = &local;
    std::memset(res, 0, sizeof(*res));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    const HostMirror* mir = h->h_mirror;
    if (mirror_seq(h) <= h->dist_seq0) {
        h->err = "reg_dist_finish: no iteration has completed since reg_dist_begin";
        return REG_NOT_CONFIGURED;
    }
    res->iterations = mir->iterations;
    res->converged = mir->converged;
    res->max_iter_reached = mir->max_iter_reached;
    res->rank_last = mir->rank_last;
    fill_result(h, mir->sums, res);
    if (mir->status != REG_OK) {
        h->err = "ErrorMinimizer: no point to minimize";
        return (reg_status)mir->status;
    }
    float T_iter[16], Tout_row[16];
    std::memcpy(T_iter, mir->T, 64);
    compose_rowmajor(h, T_iter, Tout_row);
    row_to_col(T_iter, res->T_iter_last);
    row_to_col(Tout_row, T_out);
    return REG_OK;
}

// ---- measurement hook -------------------------------------------------------------------------------
// Runs `reps` iterations' worth of kernels at T_iter (no pose update) and reports the average device time
// (ms, HIP events on the handle's stream) of: [0] the match kernel, [1] the trimmed-quantile select passes,
// [2] linearize + reduce.  Used by bench.py for the roofline object; not part of the registration semantics.
reg_status reg_profile_kernels(reg_handle* h, const float T_iter[16], int reps, float ms[3]) {
    reg_status s = check_ready(h, true);
    if (s != REG_OK) return s;
    if (!T_iter || !ms || reps <= 0) return REG_BAD_ARGUMENT;
    float Tr[16];
    col_to_row(T_iter, Tr);
    HIPCHK(h, hipSetDevice(h->prm.device));
    s = init_iter_state(h, Tr, 0);
    if (s != REG_OK) return s;
    hipEvent_t e[4];
    for (int i = 0; i < 4; ++i) HIPCHK(h, hipEventCreate(&e[i]));
    double acc[3] = {0, 0, 0};
    const bool trim = h->prm.cost == REG_COST_P2PL && h->prm.use_trimmed;
    for (int r = -1; r < reps; ++r) {  // r == -1: warm-up, not counted
        HIPCHK(h, hipEventRecord(e[0], h->stream));
        s = enqueue_match(h);
        if (s != REG_OK) return s;
        HIPCHK(h, hipEventRecord(e[1], h->stream));
        if (trim) {
            s = enqueue_select(h);
            if (s != REG_OK) return s;
        }
        HIPCHK(h, hipEventRecord(e[2], h->stream));
        s = enqueue_linearize(h, false);
        if (s != REG_OK) return s;
        HIPCHK(h, hipEventRecord(e[3], h->stream));
        HIPCHK(h, hipEventSynchronize(e[3]));
        for (int i = 0; i < 3 && r >= 0; ++i) {
            float t = 0;
            (void)hipEventElapsedTime(&t, e[i], e[i + 1]);
            acc[i] += t;
        }
    }
    for (int i = 0; i < 3; ++i) ms[i] = (float)(acc[i] / reps);
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(e[i]);
    HIPCHK(h, hipGetLastError());
    return REG_OK;
}

// ---- host-only exports ----------------------------------------------------------------------------

int reg_host_solve6(const float A[36], const float b[6], float x[6]) { return solve6_p2pl(A, b, x); }

int reg_host_solve6_xicp(const float A[36], const float b[6], const int32_t flags[6], float x[6]) {
    int f[6];
    for (int k = 0; k < 6; ++k) f[k] = flags[k];
    return solve6_xicp(A, b, f, x);
}

void reg_host_x_to_T(const float x[6], float T[16]) {
    float Tr[16];
    x_to_T(x, Tr);
    row_to_col(Tr, T);
}

void reg_host_centroid(const float* xyz, int64_t stride, int64_t n, float out[3]) {
    long long s[3] = {0, 0, 0};
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) s[k] += llrint((double)xyz[i * stride + k] * 65536.0);
    for (int k = 0; k < 3; ++k) out[k] = n > 0 ? (float)((double)s[k] / (65536.0 * (double)n)) : 0.f;
}

}  // extern "C"
