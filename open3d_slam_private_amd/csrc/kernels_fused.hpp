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
// `cap` band records}; ONE all-gather per iteration hands every rank all blocks.  The capacity depends on the group
// size: a settled band holds 500-4000 records of a 200 k-point reading IN TOTAL (C4, measured), so a rank of a small group
// needs room for more of them; the all-gather moves ~1 MB per iteration whatever the group size (<= 8 ranks).
constexpr int kContribHdr = 128;                       // floats: [0..63] = 32 doubles, [64] = band count (uint32 bits)
__host__ __device__ __forceinline__ int contrib_cap_for(int n_ranks) {
    return n_ranks <= 1 ? 8192 : (n_ranks == 2 ? 4096 : (n_ranks <= 4 ? 2048 : (n_ranks <= 8 ? 1024 : 512)));
}
__host__ __device__ __forceinline__ size_t contrib_floats(int cap) { return (size_t)kContribHdr + (size_t)cap * 32; }
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

// Component k of the 32 sums as a product E[a_k] * E[c_k] of two entries of the per-point factor table
// E = {F0..F5, r, 1}: k < 21 upper triangle F_a F_c (a <= c), 21..26 F_a r, 27 r r; 28..31 are counts (set directly).
// Entries are BYTE offsets into E.  One lane of a group publishes E in LDS; every lane forms only its own components.
struct alignas(16) ProdCode {
    uint8_t a[32];
    uint8_t c[32];
};
constexpr ProdCode make_prod_code() {
    ProdCode t{};
    int k = 0;
    for (int a6 = 0; a6 < 6; ++a6)
        for (int c6 = a6; c6 < 6; ++c6) {
            t.a[k] = (uint8_t)(4 * a6);
            t.c[k] = (uint8_t)(4 * c6);
            ++k;
        }
    for (int a6 = 0; a6 < 6; ++a6) {
        t.a[21 + a6] = (uint8_t)(4 * a6);
        t.c[21 + a6] = 4 * 6;
    }
    t.a[27] = 4 * 6;
    t.c[27] = 4 * 6;
    for (int s = 28; s < 32; ++s) {
        t.a[s] = 4 * 7;
        t.c[s] = 4 * 7;
    }
    return t;
}
__device__ const ProdCode kProdCode = make_prod_code();

template <int G>
__global__ void __launch_bounds__(256, O3D_MATCH_WAVES)
k_iter_fused(const float4* __restrict__ src, const float4* __restrict__ src_nrm, int64_t n, IterState* __restrict__ it,
             Grid g, const float4* __restrict__ tgt_nrm, FilterCfg f, int* __restrict__ pos_out,
             float* __restrict__ d2_out, float* __restrict__ w_out, uint8_t* __restrict__ hint,
             float* __restrict__ band, int band_cap, double* __restrict__ partials, int n_blocks) {
    constexpr int CP = kSums / G;   // components owned by each lane of a group
    static_assert(CP % 4 == 0 && CP >= 4, "a lane owns whole code words (4 components each)");
    __shared__ double sh[4][kSums];
    __shared__ __attribute__((aligned(16))) uint32_t seg_lds[(256 / G) * kSegWords<G>];
    const int lb = xcd_block(n_blocks);
    const int64_t tid = lb * (int64_t)blockDim.x + threadIdx.x;
    const int64_t q = lb < n_blocks ? (tid / G) : n;
    const int sub = (int)(tid & (G - 1));
    // What does not depend on the iteration state is requested BEFORE the state is looked at: the reading point and
    // its hint travel together with the state's scalar loads (the kernel is bound by its chain of dependent round
    // trips, not by instruction issue: see DESIGN.md).  Unconditional loads from clamped addresses: a load under a
    // branch makes the compiler wait for it at the join.
    const int64_t qc = q < n ? q : n - 1;
    const float4 s = src[qc];
    const uint8_t hraw = *(hint ? hint + qc : reinterpret_cast<const uint8_t*>(src));
    const int hv = hint ? (int)hraw : 0;
    // one batch of scalar loads for every state field (tested one after the other they cost a round trip each; the
    // empty asm keeps the compiler from sinking the pose loads below the early exit)
    const int st_done = it->done, st_stall = it->stall;
    const Xf T = load_xf(it);
    const float band_lo = it->band_lo, band_hi = it->band_hi;
#pragma unroll
    for (int k = 0; k < 12; ++k) asm volatile("" ::"s"(T.m[k]));
    asm volatile("" ::"s"(band_lo), "s"(band_hi));
    asm volatile("" ::"v"(s.x), "v"(s.y), "v"(s.z));   // ... nor the reading point's
    if ((st_done | st_stall) != 0) return;
    double mine[CP];
#pragma unroll
    for (int j = 0; j < CP; ++j) mine[j] = 0.0;
    if (q < n) {
        const float3 p = xf_point(T, s.x, s.y, s.z);
        int lvl;
        const Best b = nearest_group<G>(g, p, sub, -1, &lvl, seg_lds + (threadIdx.x / G) * kSegWords<G>,
                                        hv >= 2 ? hv - 2 : -1);
        float w = 0.f, dd = INFINITY;
        uint32_t code_a[CP / 4] = {}, code_c[CP / 4] = {};
        int cls = 2;  // 0: certainly kept, 1: band, 2: dropped / unmatched
        bool keep = false;   // certainly kept with a non-zero weight: contributes to the sums in this launch
        // the group's segment list is free again: its first 8 words become the factor table E (see ProdCode)
        uint32_t* const etab = seg_lds + (threadIdx.x / G) * kSegWords<G>;
        if (b.pos >= 0) {
            dd = b.d2;
            w = 1.f;
            if (f.use_maxdist && !(dd <= f.outlier_max_d2)) w = 0.f;
            const float4 nn = tgt_nrm[2 * (size_t)b.pos + 1];
            const float4 tq = tgt_nrm[2 * (size_t)b.pos];   // {point, normal} pair: requested together, one line
            // factor codes of the CP components this lane owns (same batch; 64 bytes shared by every wave)
#pragma unroll
            for (int wi = 0; wi < CP / 4; ++wi) {
                code_a[wi] = reinterpret_cast<const uint32_t*>(kProdCode.a)[sub * (CP / 4) + wi];
                code_c[wi] = reinterpret_cast<const uint32_t*>(kProdCode.c)[sub * (CP / 4) + wi];
            }
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
            keep = w != 0.f && cls == 0;
            if (keep) {
                // F = [p x n ; n], r = (p - q) . n: lane 0 of the group publishes them, every lane then forms only the
                // CP products it owns (the weight is 0 or 1 here, so w * F_a * F_c == F_a * F_c exactly)
                float a = p.y * nn.z, bq = p.z * nn.y;
                const float F0 = a - bq;
                a = p.z * nn.x; bq = p.x * nn.z;
                const float F1 = a - bq;
                a = p.x * nn.y; bq = p.y * nn.x;
                const float F2 = a - bq;
                const float dx = p.x - tq.x, dy = p.y - tq.y, dz = p.z - tq.z;
                float r = dx * nn.x;
                float t2 = dy * nn.y;
                r = r + t2;
                t2 = dz * nn.z;
                r = r + t2;
                if (sub == 0) {
                    *reinterpret_cast<float4*>(etab) = make_float4(F0, F1, F2, nn.x);
                    *reinterpret_cast<float4*>(etab + 4) = make_float4(nn.y, nn.z, r, 1.f);
                }
            }
            if (cls == 1 && sub == 0) {
                // band record: decided by the update kernel
                float vals[kSums];
#pragma unroll
                for (int k = 0; k < kSums; ++k) vals[k] = 0.f;
                if (w != 0.f) {
                    p2pl_products(p, tq, nn, w, vals);
                    vals[28] = 1.f;
                    vals[30] = dd;
                }
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
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // E written by lane 0, read by the whole group (same wave)
        __builtin_amdgcn_wave_barrier();
        // certainly-kept contributions: lane `sub` owns components sub*CP .. sub*CP+CP-1
#pragma unroll
        for (int j = 0; j < CP; ++j) {
            float v = 0.f;
            if (keep) {
                const uint32_t oa = (code_a[j / 4] >> (8 * (j & 3))) & 0xffu, oc = (code_c[j / 4] >> (8 * (j & 3))) & 0xffu;
                const float U = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(etab) + oa);
                const float V = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(etab) + oc);
                v = U * V;
            }
            if (j >= CP - 4 && sub == G - 1) {   // components 28..31: kept, matched, kept d2, below-band counts
                const int jj = j - (CP - 4);
                v = jj == 0 ? (keep ? 1.f : 0.f)
                  : jj == 1 ? (b.pos >= 0 ? 1.f : 0.f)
                  : jj == 2 ? (keep ? dd : 0.f)
                            : ((b.pos >= 0 && cls == 0) ? 1.f : 0.f);   // rank bookkeeping is independent of w
            }
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

// =================================================================================================
// Fused iteration with temporal coherence (the settled tail of a registration)
// =================================================================================================
// Once the pose only moves by fractions of a millimetre per iteration, almost every reading point keeps its neighbour.
// The searches of earlier iterations leave, per reading point, an ANCHOR (the transformed position the search ran at) and
// a BOUND G2 = min(runner-up d2, covered radius^2): every reference point other than the winner was at least sqrt(G2)
// away from the anchor.  At the new position p' (|p' - anchor| = delta) the triangle inequality gives, for every other
// point x, dist(p', x) >= sqrt(G2) - delta; so if the old winner b satisfies dist(p', b) + delta < sqrt(G2) (with 1e-5
// relative slack on each term, 30x the rounding of the fp32 expressions involved), b is still the UNIQUE nearest
// neighbour under the fp32 comparison the full search would make, and its d2 -- evaluated with the same NC5 expression
// -- is the value the full search would report.  Such a point costs ~100 bytes (reading point, anchor, previous match,
// matched point and normal) instead of a halo run of ~35 candidate records; the others (a few per cent: runner-up nearly
// as close as the winner, or the winner beyond the halo radius) are queued for k_coh_search, the full search, which
// refreshes anchor and bound.  Results are identical to k_iter_fused by construction.
//
// Two launches, so that the searches -- which cluster in space (clutter, thin structures) and would pile up in a few
// workgroups -- are spread evenly over the chip by the second one:
//   k_coh_check   256 reading points per workgroup, one lane per point: shortcut test; weights, class, factor row
//                 E = {F0..F5, r, 1 | kept, matched, kept d2, below-band} in LDS for the points that pass (zero row
//                 otherwise); the 32 sums from the LDS table (thread t owns component t & 31 for the 32 rows of part
//                 t >> 5); failed points are appended to a queue (one atomic per workgroup).
//   k_coh_search  fixed grid, grid-stride over the queue in chunks of 32 points: nearest_group with G = 8 lanes per point,
//                 then the same epilogue and a 32-row LDS table.
#ifndef O3D_COH_PRUNE
#define O3D_COH_PRUNE 0
#endif
constexpr int kCohRow = 13;   // 12 floats per factor row + 1 pad (bank spread)
struct CohStats {
    unsigned long long n_points, n_searched;
};

// A reading point whose d2 falls into the predicted band: its products go into a record the update kernel decides on
// (out of line: a few hundred points per launch take this path, and its 32 live floats would otherwise set the register
// budget of the whole kernel).
__device__ __noinline__ void write_band_record(IterState* it, float* __restrict__ band, int band_cap, float3 p, float4 tq,
                                               float4 nn, float w, float d2, int q) {
    float vals[kSums];
#pragma unroll
    for (int k = 0; k < kSums; ++k) vals[k] = 0.f;
    if (w != 0.f) {
        p2pl_products(p, tq, nn, w, vals);
        vals[28] = 1.f;
        vals[30] = d2;
    }
    // one returning atomic per wave (the lanes that reach this point share it): same-address atomics serialise
    const unsigned long long act = __ballot(true);
    const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)act) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(&it->band_count, (unsigned)__popcll(act));
    base = (unsigned)__shfl((int)base, leader);
    const unsigned slot = base + (unsigned)__popcll(act & ((1ull << lane) - 1ull));
    if (slot < (unsigned)band_cap) {
#pragma unroll
        for (int k = 0; k < 31; ++k)
            if (k != 29) band[band_at(k, slot)] = vals[k];
        band[band_at(29, slot)] = d2;   // [29] = d2 (the "matched" count is added from the class)
        band[band_at(31, slot)] = __int_as_float(q);
    }
}

// Weights, class and factor row of ONE matched reading point (one lane); writes the outputs of the point.
// row[0..5] = F, [6] = r, [7] = 1 (kept rows only), [8] kept, [9] matched, [10] kept d2, [11] below the band.
__device__ __forceinline__ void coh_epilogue(const Xf& T, const FilterCfg& f, float band_lo, float band_hi, const float3 p,
                                             int64_t q, int mpos, float md2, const float4 tq, const float4 nn,
                                             const float4* __restrict__ src_nrm,
                                             IterState* it, float* __restrict__ band, int band_cap, int* __restrict__ pos_io,
                                             float* __restrict__ d2_out, float* __restrict__ w_out, float (&row)[12]) {
#pragma unroll
    for (int k = 0; k < 12; ++k) row[k] = 0.f;
    float w = 0.f;
    int cls = 2;   // 0: certainly kept, 1: band, 2: dropped / unmatched
    if (mpos >= 0) {
        w = 1.f;
        if (f.use_maxdist && !(md2 <= f.outlier_max_d2)) w = 0.f;
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
        cls = md2 < band_lo ? 0 : (md2 < band_hi ? 1 : 2);
        row[9] = 1.f;                               // matched
        if (cls == 0) row[11] = 1.f;                // below the band (rank bookkeeping is independent of w)
        if (w != 0.f && cls == 0) {
            // F = [p x n ; n], r = (p - q) . n  (the weight is 1: w * F_a * F_c == F_a * F_c exactly)
            float a = p.y * nn.z, bq = p.z * nn.y;
            row[0] = a - bq;
            a = p.z * nn.x; bq = p.x * nn.z;
            row[1] = a - bq;
            a = p.x * nn.y; bq = p.y * nn.x;
            row[2] = a - bq;
            row[3] = nn.x; row[4] = nn.y; row[5] = nn.z;
            const float dx = p.x - tq.x, dy = p.y - tq.y, dz = p.z - tq.z;
            float r = dx * nn.x;
            float t2 = dy * nn.y;
            r = r + t2;
            t2 = dz * nn.z;
            r = r + t2;
            row[6] = r;
            row[7] = 1.f;
            row[8] = 1.f;                           // kept
            row[10] = md2;                          // kept d2
        }
        if (cls == 1) write_band_record(it, band, band_cap, p, tq, nn, w, md2, (int)q);   // decided by the update kernel
    }
    pos_io[q] = mpos;
    d2_out[q] = mpos >= 0 ? md2 : INFINITY;
    if (w_out) w_out[q] = (cls == 2) ? 0.f : w;   // band points: provisional, patched by the update kernel
}

// Component `comp` of the 32 sums over `rows` consecutive factor rows starting at E (fp32 product, fp64 sum).
__device__ __forceinline__ double coh_component(const float* E, int rows, int comp) {
    double acc = 0.0;
    if (comp < 28) {
        const int ia = kProdCode.a[comp] >> 2, ic = kProdCode.c[comp] >> 2;
        for (int i = 0; i < rows; ++i) {
            const float v = E[i * kCohRow + ia] * E[i * kCohRow + ic];
            acc += (double)v;
        }
    } else {
        const int ia = 8 + (comp - 28);
        for (int i = 0; i < rows; ++i) acc += (double)E[i * kCohRow + ia];
    }
    return acc;
}

__global__ void __launch_bounds__(256)
k_coh_check(const float4* __restrict__ src, const float4* __restrict__ src_nrm, int64_t n, IterState* __restrict__ it,
            Grid g, const float4* __restrict__ tgt_nrm, FilterCfg f, int* __restrict__ pos_io, float* __restrict__ d2_out,
            float* __restrict__ w_out, const float4* __restrict__ cache /* 4 x n: anchor | matched point | normal | runner-up */,
            uint32_t* __restrict__ queue, int queue_cap, float* __restrict__ band, int band_cap, double* __restrict__ partials,
            int n_blocks) {
    __shared__ __attribute__((aligned(16))) float E[256 * kCohRow];
    __shared__ int s_wave_cnt[4];
    __shared__ unsigned s_base;
    __shared__ double sh[4][kSums];
    const int t = threadIdx.x;
    const int lb = xcd_block(n_blocks);
    const int64_t q = lb < n_blocks ? (int64_t)lb * 256 + t : n;
    const int64_t qc = q < n ? q : n - 1;
    // everything that does not depend on the state travels with the state's scalar loads (see k_iter_fused)
    // The matched reference point and its normal travel with the reading point (the match persists, so they are kept in
    // the per-point cache instead of being gathered from the 5 M-point arrays every iteration: a random 16-byte gather
    // costs a whole 128-byte line, FETCH_SIZE calibration in profiles/): the shortcut path only STREAMS.
    const float4 s = src[qc];
    const float4 c = cache[qc];
    const float4 tq = cache[(size_t)n + qc];
    const float4 nn = cache[2 * (size_t)n + qc];
    const float4 ru = cache[3 * (size_t)n + qc];   // the runner-up of the last search (w = 1: valid, the bound excludes it)
    const int pprev = pos_io[qc];
    const int st_done = it->done, st_stall = it->stall;
    const Xf T = load_xf(it);
    const float band_lo = it->band_lo, band_hi = it->band_hi;
    unsigned int* const qcount = it->qcount;
#pragma unroll
    for (int k = 0; k < 12; ++k) asm volatile("" ::"s"(T.m[k]));
    asm volatile("" ::"s"(band_lo), "s"(band_hi));
    asm volatile("" ::"v"(s.x), "v"(c.w), "v"(pprev), "v"(tq.x), "v"(nn.x), "v"(ru.x));
    if ((st_done | st_stall) != 0) return;
    const float3 p = xf_point(T, s.x, s.y, s.z);
    // ---- does the previous neighbour provably remain the nearest one?
    bool pass = false;
    float md2 = INFINITY;
    if (q < n && c.w > 0.f && pprev >= 0 && nn.w == 1.f) {   // nn.w: matched point + normal of `pprev` are cached
        const float dx = p.x - tq.x, dy = p.y - tq.y, dz = p.z - tq.z;
        float a = dx * dx;
        float b = dy * dy;
        float d2 = a + b;
        a = dz * dz;
        d2 = d2 + a;                                   // NC5: the value the full search would compute for this pair
        const float ex = p.x - c.x, ey = p.y - c.y, ez = p.z - c.z;
        const float delta = __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez);
        const float lhs = (__builtin_amdgcn_sqrtf(d2) + delta) * 1.00001f + 1e-30f;
        const float rhs = __builtin_amdgcn_sqrtf(c.w) * 0.99999f;
        pass = d2 <= g.max_d2 && lhs < rhs;
        if (ru.w == 1.f) {
            // Two-candidate form: the bound c.w covers every point EXCEPT the winner and the runner-up of the last search, so
            // the runner-up is tested by itself -- with the comparison the full search would make (fp32 d2, strict: an exact
            // tie is decided by the original indices, which only the full search knows).  Reading points that sit between
            // two nearly equidistant reference points (8 % of C3, 25 % of C4 with the one-candidate bound) keep their match.
            const float fx = p.x - ru.x, fy = p.y - ru.y, fz = p.z - ru.z;
            float a2 = fx * fx;
            float b2 = fy * fy;
            float r2 = a2 + b2;
            a2 = fz * fz;
            r2 = r2 + a2;
            pass = pass && d2 < r2;
        }
        md2 = d2;
    }
    const bool need = q < n && !pass;
    // queue the points that need a search: wave ballot, one atomic per workgroup
    const unsigned long long bal = __ballot(need);
    const int lane = t & 63, wave = t >> 6;
    if (lane == 0) s_wave_cnt[wave] = (int)__popcll(bal);
    __syncthreads();
    if (t == 0) {
        const int tot = s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
        s_base = tot ? atomicAdd(&qcount[(lb & (kQueues - 1)) * kQueueStride], (unsigned)tot) : 0u;
    }
    __syncthreads();
    if (need) {
        unsigned at = s_base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        for (int w2 = 0; w2 < wave; ++w2) at += (unsigned)s_wave_cnt[w2];
        queue[(size_t)(lb & (kQueues - 1)) * queue_cap + at] = (uint32_t)q;   // at < queue_cap: host_loop.hpp sizes it
    }
    // ---- weights, class, factor row of the points that passed (zero row otherwise: k_coh_search accounts for those)
    float row[12];
    if (pass) {
        coh_epilogue(T, f, band_lo, band_hi, p, q, pprev, md2, tq, nn, src_nrm, it, band, band_cap, pos_io, d2_out, w_out, row);
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) row[k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) E[t * kCohRow + k] = row[k];
    __syncthreads();
    // ---- the 32 sums of this workgroup's rows
    const int comp = t & (kSums - 1), part = t >> 5;
    double acc = coh_component(E + (part * 32) * kCohRow, 32, comp);
    acc += __shfl_xor(acc, 32);                          // the two parts of a wave
    if (lane < kSums) sh[wave][lane] = acc;
    __syncthreads();
    if (t < kSums && lb < n_blocks) {
        const double tot = (sh[0][t] + sh[1][t]) + (sh[2][t] + sh[3][t]);
        unsafeAtomicAdd(&partials[(size_t)(lb & (kAccRows - 1)) * kSums + t], tot);
    }
}

template <int G>
__global__ void __launch_bounds__(256, O3D_SEARCH_WAVES)
k_coh_search(const float4* __restrict__ src, const float4* __restrict__ src_nrm, int64_t n, IterState* __restrict__ it,
             Grid g, const float4* __restrict__ tgt_nrm, FilterCfg f, int* __restrict__ pos_io, float* __restrict__ d2_out,
             float* __restrict__ w_out, uint8_t* __restrict__ hint, float4* __restrict__ cache,
             const uint32_t* __restrict__ queue, int queue_cap, float* __restrict__ band, int band_cap,
             double* __restrict__ partials, float slack, CohStats* __restrict__ stats) {
    constexpr int kQ = 256 / G;   // reading points per chunk
    __shared__ __attribute__((aligned(16))) uint32_t seg_lds[kQ * kSegWords<G>];
    __shared__ __attribute__((aligned(16))) float E[kQ * kCohRow];
    __shared__ double sh[4][kSums];
    __shared__ unsigned s_pre[kQueues + 1];   // exclusive prefix of the sub-queue counts
    const int t = threadIdx.x;
    const int st_done = it->done, st_stall = it->stall;
    const Xf T = load_xf(it);
    const float band_lo = it->band_lo, band_hi = it->band_hi;
    const unsigned int* const qcount = it->qcount;
    if ((st_done | st_stall) != 0) return;
    if (t < 64) {
        static_assert(kQueues == 64, "one lane per sub-queue");
        const unsigned cnt = qcount[t * kQueueStride];
        unsigned incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = (unsigned)__shfl_up((int)incl, o);
            if (t >= o) incl += v;
        }
        s_pre[t + 1] = incl;
        if (t == 0) s_pre[0] = 0u;
    }
    __syncthreads();
    const unsigned count = s_pre[kQueues];
    if ((unsigned)blockIdx.x * kQ >= count) return;      // workgroup-uniform: nothing queued for this workgroup
    const int sub = t & (G - 1), grp = t / G;
    const int comp = t & (kSums - 1), part = t >> 5;     // 8 parts x (kQ / 8) rows
    double acc = 0.0;
    for (unsigned base = (unsigned)blockIdx.x * kQ; base < count; base += gridDim.x * kQ) {
        const unsigned slot = base + (unsigned)grp;
        float row[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) row[k] = 0.f;
        if (slot < count) {
            int sq = 0;   // sub-queue holding global slot `slot`: largest sq with s_pre[sq] <= slot
#pragma unroll
            for (int step = kQueues / 2; step >= 1; step >>= 1)
                if (s_pre[sq + step] <= slot) sq += step;
            const int64_t qq = (int64_t)queue[(size_t)sq * queue_cap + (slot - s_pre[sq])];
            const float4 s2 = src[qq];
            const int hv = hint ? (int)hint[qq] : 0;
            const float3 p2 = xf_point(T, s2.x, s2.y, s2.z);
            int lvl;
            float cov2;
            const Best bb = nearest_group<G, O3D_COH_PRUNE != 0, true>(g, p2, sub, -1, &lvl, seg_lds + grp * kSegWords<G>,
                                                          hv >= 2 ? hv - 2 : -1, &cov2, slack);
            if (sub == 0) {
                if (hint) hint[qq] = (uint8_t)(lvl + 1);
                const int pc = bb.pos >= 0 ? bb.pos : 0;
                const float4 tq = tgt_nrm[2 * (size_t)pc];       // {point, normal} pair: one line
                float4 nn = tgt_nrm[2 * (size_t)pc + 1];
                nn.w = 1.f;
                float4 ru = make_float4(INFINITY, INFINITY, INFINITY, 1.f);   // no runner-up seen: infinitely far
                if (bb.pos2 >= 0) {
                    const float4 t2 = g.pts[bb.pos2];
                    ru = make_float4(t2.x, t2.y, t2.z, 1.f);
                }
                // bound: every point other than the winner and the runner-up is at least sqrt(min(third, cov2)) away
                cache[qq] = make_float4(p2.x, p2.y, p2.z, bb.pos >= 0 ? fminf(bb.third, cov2) : -1.f);
                cache[(size_t)n + qq] = tq;
                cache[2 * (size_t)n + qq] = nn;
                cache[3 * (size_t)n + qq] = ru;
                coh_epilogue(T, f, band_lo, band_hi, p2, qq, bb.pos, bb.d2, tq, nn, src_nrm, it, band, band_cap, pos_io, d2_out, w_out,
                             row);
            }
        }
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < 12; ++k) E[grp * kCohRow + k] = row[k];
        }
        __syncthreads();
        acc += coh_component(E + (part * (kQ / 8)) * kCohRow, kQ / 8, comp);
        __syncthreads();   // the table is rewritten by the next chunk
    }
    acc += __shfl_xor(acc, 32);
    const int lane = t & 63, wave = t >> 6;
    if (lane < kSums) sh[wave][lane] = acc;
    __syncthreads();
    if (t < kSums) {
        const double tot = (sh[0][t] + sh[1][t]) + (sh[2][t] + sh[3][t]);
        unsafeAtomicAdd(&partials[(size_t)(blockIdx.x & (kAccRows - 1)) * kSums + t], tot);
    }
    if (stats && t == 0 && blockIdx.x == 0) {
        atomicAdd(&stats->n_points, (unsigned long long)n);
        atomicAdd(&stats->n_searched, (unsigned long long)count);
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
