// kernels_build.hpp -- target preparation (R1): centroid, centring + bbox, sort keys, brick table, halo bins
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once

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
    const float4 pt = centred[src];
    pts_sorted[i] = pt;
    if (nrm_sorted) {
        // point and normal INTERLEAVED (32 bytes, one 128-byte line holds four pairs): whoever needs the matched point's
        // normal needs the point too, and a random 16-byte gather costs a whole line -- two arrays meant two lines per match
        const float* q = nrm + (int64_t)src * nrm_stride;
        nrm_sorted[2 * i] = pt;
        nrm_sorted[2 * i + 1] = make_float4(q[0], q[1], q[2], 0.f);
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
                              int32_t* __restrict__ dir, int bdx, int bdy, unsigned long long* __restrict__ rows) {
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
    if (rows && old == 0) {   // first point of its bin: mark the bin's x-row (z_local * 8 + y_local) as occupied
        const uint64_t bk = key >> (3 * kBrickLog2);
        const uint32_t m18 = (1u << kBrickBits) - 1u;
        const uint32_t bx = (uint32_t)bk & m18, by = (uint32_t)(bk >> kBrickBits) & m18, bz = (uint32_t)(bk >> (2 * kBrickBits)) & m18;
        atomicOr(&rows[((size_t)bz * bdy + by) * bdx + bx], 1ull << (local >> kBrickLog2));
    }
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
