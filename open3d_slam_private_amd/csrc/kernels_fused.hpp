// kernels_fused.hpp -- the fused iteration kernel (search + weights + normal equations in one launch)
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once

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
        const Best b = nearest_group<G>(g, p, sub, -1, &lvl, seg_lds + (threadIdx.x / G) * kSegWords<G>,
                                        hv >= 2 ? hv - 2 : -1);
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
