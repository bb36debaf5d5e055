// kernels_match.hpp -- the select-based iteration (R3-R7): match, trimmed-quantile select, linearize
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once

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
#ifndef O3D_MATCH_REGROUP
#define O3D_MATCH_REGROUP 1
#endif
template <int G>
constexpr int kSegWords = (2 * G * kSegPerLane + 2 + 3) & ~3;   // a multiple of 4 words: 16-byte aligned rows (the fused kernel reuses a row as float4s)

// Cooperative variant: G (8 or 4) lanes per reading point (256/G points per 256-thread workgroup).
// `hint` (one byte per point, may be null) carries the terminating level of the previous iteration.
template <int G>
__global__ void __launch_bounds__(256, O3D_MATCH_WAVES)
k_match_g8(const float4* __restrict__ src, int64_t n, const IterState* __restrict__ it, Grid g, int* __restrict__ pos,
           float* __restrict__ d2, uint32_t* __restrict__ hist0 /* 2048 or null */,
           uint32_t* __restrict__ hist2_to_zero, uint8_t* __restrict__ hint, int shift0, int debug, int n_blocks,
           float4* __restrict__ cache /* 4 x n, or null: anchor + bound (row 0; row 3 = "no runner-up on record") for the
                                         temporal-coherence shortcut of the fused iterations (k_coh_check) */,
           int xcd_tile /* workgroups per XCD tile (xcd_block_tiled) */) {
    __shared__ uint32_t sh[2048];
    __shared__ __attribute__((aligned(16))) uint32_t seg_lds[(256 / G) * kSegWords<G>];
    const int lb = xcd_block_tiled(n_blocks, xcd_tile);
    const int64_t tid = lb * (int64_t)blockDim.x + threadIdx.x;
    const int64_t q = lb < n_blocks ? (tid / G) : n;
    const int sub = (int)(tid & (G - 1));
    // reading point and hint are requested before the state is looked at (one batch of loads instead of two dependent
    // round trips: see k_iter_fused); unconditional loads from clamped addresses
    const int64_t qc = q < n ? q : n - 1;
    const float4 s = src[qc];
    const uint8_t hraw = *(hint ? hint + qc : reinterpret_cast<const uint8_t*>(src));
    const int st_done = it->done;
    const Xf T = load_xf(it);
#pragma unroll
    for (int k = 0; k < 12; ++k) asm volatile("" ::"s"(T.m[k]));
    asm volatile("" ::"v"(s.x), "v"(s.y), "v"(s.z));
    if (st_done) return;
    if (hist2_to_zero && blockIdx.x == 0)
        for (int k = threadIdx.x; k < 2048; k += blockDim.x) hist2_to_zero[k] = 0;
    if (hist0) {
        for (int k = threadIdx.x; k < 2048; k += blockDim.x) sh[k] = 0;
        __syncthreads();
    }
    if (debug & 4) {  // timing experiment: fixed cost of the launch + reading load only
        if (q < n && sub == 0) {
            pos[q] = -1;
            d2[q] = xf_point(T, s.x, s.y, s.z).x;
        }
        return;
    }
    auto emit = [&](int64_t qq, const float3 pp, const Best& b, int lvl, float cov2) {
        pos[qq] = b.pos;
        d2[qq] = b.pos >= 0 ? b.d2 : INFINITY;
        if (hint) hint[qq] = (uint8_t)(lvl + 1);
        if (cache) {
            cache[qq] = make_float4(pp.x, pp.y, pp.z, b.pos >= 0 ? fminf(b.second, cov2) : -1.f);
            cache[3 * (size_t)n + qq] = make_float4(0.f, 0.f, 0.f, 0.f);   // no runner-up on record: one-candidate bound
        }
        if (hist0 && b.pos >= 0) atomicAdd(&sh[__float_as_uint(b.d2) >> shift0], 1u);
    };
    // Two parts.  (1) Every query tries the halo level (one lookup, one short run: after a pose update most neighbours are back
    // within its radius even when the previous search needed a large box; hint h >= 2: the last search ended at regular level
    // h - 1 -> if the halo cannot answer, continue one regular level below that).  (2) The queries the halo could not answer
    // are REGROUPED through LDS, ordered by the level they continue at (those that hold a halo candidate -- exactly one more
    // level scan -- in front of those that do not): a wave pays for the level scans of its slowest group, so unanswered queries
    // spread one or two per wave kept every wave of the workgroup in the level code (iteration 1 at C3: 27 % of the queries,
    // 92 % of the waves); packed, the waves behind them retire after the halo part and groups of a wave run alike.
    // (one-candidate bound here: tracking the runner-up's identity in this kernel -- whose first launches scan hundreds of
    // candidates per point -- cost it 20 %, more than the fused iterations gain from the tighter bound; the fused iterations'
    // own searches, k_coh_search, do track it)
    constexpr int kP = 256 / G;               // queries of this workgroup
    constexpr int kKeys = 2 * kMaxLevels + 2;
    __shared__ float4 un_p[kP], un_b[kP];     // {p, cov}, {d2, idx, pos, second} of the unanswered queries, in their new order
    __shared__ uint32_t un_l[kP];             // level | local query << 8
    __shared__ uint32_t key_cnt[kKeys + 1];
    const int grp = (int)(threadIdx.x / G);
    const int gbase = (int)(threadIdx.x & 63) & ~(G - 1);
    for (int k = threadIdx.x; k <= kKeys; k += blockDim.x) key_cnt[k] = 0u;
    __syncthreads();
    const float3 p = xf_point(T, s.x, s.y, s.z);
    Best b;
    float cov = 0.f, cov2 = 0.f;
    int l = 0, lvl = 0, key = -1;
    uint32_t rnk = 0;
    if (q < n) {
        const int hv = hint ? (int)hraw : 0;
        if (nearest_halo<G, false>(g, p, sub, gbase, -1, hv >= 2 ? hv - 2 : -1, b, cov, l, &lvl, &cov2)) {
            if (sub == 0) emit(q, p, b, lvl, cov2);
#if !O3D_MATCH_REGROUP
        } else if (true) {   // A/B: the levels at once, by the same lanes
            b = nearest_levels<G, true, false>(g, p, sub, gbase, l, b, cov, seg_lds + grp * kSegWords<G>, 0.f, &lvl, &cov2);
            if (sub == 0) emit(q, p, b, lvl, cov2);
#endif
        } else {
            key = min(l, kMaxLevels) + (b.pos >= 0 ? 0 : kMaxLevels + 1);
            if (sub == 0) rnk = atomicAdd(&key_cnt[key], 1u);
            rnk = (uint32_t)__shfl((int)rnk, gbase);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {   // exclusive prefix of the key counts (wave 0)
        const uint32_t c = (int)threadIdx.x < kKeys ? key_cnt[threadIdx.x] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
            if ((int)threadIdx.x >= o) incl += v;
        }
        if ((int)threadIdx.x < kKeys) key_cnt[threadIdx.x] = incl - c;
        if (threadIdx.x == kKeys - 1) key_cnt[kKeys] = incl;
    }
    __syncthreads();
    if (key >= 0 && sub == 0) {
        const uint32_t at = key_cnt[key] + rnk;
        un_p[at] = make_float4(p.x, p.y, p.z, cov);
        un_b[at] = make_float4(b.d2, __uint_as_float(b.idx), __int_as_float(b.pos), b.second);
        un_l[at] = (uint32_t)l | ((uint32_t)grp << 8);
    }
    __syncthreads();
    if (grp < (int)key_cnt[kKeys]) {
        const float4 ep = un_p[grp], eb = un_b[grp];
        const uint32_t el = un_l[grp];
        const int64_t q2 = (int64_t)lb * kP + (int64_t)(el >> 8);
        const float3 p2 = make_float3(ep.x, ep.y, ep.z);
        Best b2;
        b2.d2 = eb.x;
        b2.idx = __float_as_uint(eb.y);
        b2.pos = __float_as_int(eb.z);
        b2.second = eb.w;
        b2.pos2 = -1;
        b2.third = INFINITY;
        b2 = nearest_levels<G, true, false>(g, p2, sub, gbase, (int)(el & 255u), b2, ep.w, seg_lds + grp * kSegWords<G>, 0.f, &lvl, &cov2);
        if (sub == 0) emit(q2, p2, b2, lvl, cov2);
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
__device__ __forceinline__ void block_pick256_regs(const uint32_t (&loc)[8], uint32_t rank, uint32_t* wave_tot /*[4]*/,
                                                   uint32_t* out /*[3]*/) {
    const int t = threadIdx.x;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += loc[k];
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

// This thread's 8 bins of a 2048-entry histogram (two 16-byte loads; every select histogram is allocated with 2048 entries)
__device__ __forceinline__ void load_hist8(const uint32_t* __restrict__ hist, uint32_t (&loc)[8]) {
    const uint4 a = reinterpret_cast<const uint4*>(hist)[threadIdx.x * 2];
    const uint4 b = reinterpret_cast<const uint4*>(hist)[threadIdx.x * 2 + 1];
    loc[0] = a.x; loc[1] = a.y; loc[2] = a.z; loc[3] = a.w;
    loc[4] = b.x; loc[5] = b.y; loc[6] = b.z; loc[7] = b.w;
}

__device__ __forceinline__ void block_pick256(const uint32_t* __restrict__ hist, int nb, uint32_t rank,
                                              uint32_t* wave_tot /*[4]*/, uint32_t* out /*[3]*/) {
    uint32_t loc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int bin = threadIdx.x * 8 + k;
        loc[k] = bin < nb ? hist[bin] : 0u;
    }
    block_pick256_regs(loc, rank, wave_tot, out);
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
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t pick[3];
    // Everything the kernel reads is requested in ONE batch before the state is tested: this thread's first kPre
    // distances, its 8 bins of the previous level's histogram and the select state (the kernel is a chain of
    // dependent round trips otherwise: done -> histogram -> state -> distances).
    constexpr int kPre = 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    uint32_t pre[kPre];
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        const int64_t i = i0 + u * stride;
        pre[u] = __float_as_uint(d2[i < n ? i : n - 1]);
    }
    uint32_t loc[8];
    load_hist8(hist_prev, loc);
    const int st_done = it->done;
    const uint32_t st_rank = st->rank, st_prefix = st->prefix;
    asm volatile("" ::"v"(pre[0]), "v"(pre[1]), "v"(pre[2]), "v"(pre[3]));
    asm volatile("" ::"v"(loc[0]), "v"(loc[4]), "s"(st_rank), "s"(st_prefix));
    if (st_done) return;
    for (int k = threadIdx.x; k < 2048; k += blockDim.x) sh[k] = 0;
    uint32_t prefix, rank_in;
    if (level == 1) {
        // total count of finite distances = sum of hist0
        block_pick256_regs(loc, 0xffffffffu, wave_tot, pick);
        const uint32_t total = pick[2];
        __syncthreads();
        block_pick256_regs(loc, trim_rank(total, ratio), wave_tot, pick);
        prefix = pick[0] << shift0;
        rank_in = pick[1];
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->n_finite = total;
            st->prefix = prefix;
            st->rank = rank_in;
            if (total == 0) st->limit = INFINITY;
        }
    } else {
        block_pick256_regs(loc, st_rank, wave_tot, pick);
        prefix = st_prefix | (pick[0] << (shift0 - 11));
        rank_in = pick[1];
    }
    __syncthreads();
    const int s1 = shift0 - 11;  // low bit of the level-1 digit; level 2 = the s1 lowest bits
    const uint32_t mask = level == 1 ? ~((1u << shift0) - 1u) : ~((1u << s1) - 1u);
    auto count = [&](uint32_t u) {
        if (u != 0x7f800000u && (u & mask) == prefix) {
            const uint32_t b = level == 1 ? ((u >> s1) & 2047u) : (u & ((1u << s1) - 1u));
            atomicAdd(&sh[b], 1u);
        }
    };
#pragma unroll
    for (int u = 0; u < kPre; ++u)
        if (i0 + u * stride < n) count(pre[u]);
    for (int64_t i = i0 + kPre * stride; i < n; i += stride) count(__float_as_uint(d2[i]));
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
    // this thread's first kPre distances travel with the state (one round trip instead of two)
    constexpr int kPre = 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    uint32_t pre[kPre];
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        const int64_t i = i0 + u * stride;
        pre[u] = __float_as_uint(d2[i < n ? i : n - 1]);
    }
    const int st_done = it->done;
    asm volatile("" ::"v"(pre[0]), "v"(pre[1]), "v"(pre[2]), "v"(pre[3]));
    if (st_done) return;
    for (int k = threadIdx.x; k < 2048; k += blockDim.x) sh[k] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPre; ++u)
        if (i0 + u * stride < n && pre[u] != 0x7f800000u) atomicAdd(&sh[pre[u] >> shift0], 1u);
    for (int64_t i = i0 + kPre * stride; i < n; i += stride) {
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
                 float* __restrict__ w_out, double* __restrict__ partials,
                 float4* __restrict__ cache /* 4 x n (k_coh_check) or null: rows 1, 2 = matched point, its normal */) {
    // One batch for everything that does not depend on the match: match position, distance, reading point and normal
    // of this thread's point, the pose, the select state and this thread's 8 bins of the last radix level; then ONE
    // more batch for the matched point and its normal.  (It was a chain of six dependent round trips: done -> state ->
    // histogram -> position -> normal -> target point.)
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t ic = i < n ? i : n - 1;
    const int ps_raw = pos[ic];
    const float dd = d2[ic];
    const float4 s = src[ic];
    const float4 sn = *(f.use_normal ? src_nrm + ic : src + ic);
    uint32_t loc[8];
    load_hist8(hist2, loc);
    const int st_done = it->done;
    const Xf T = load_xf(it);
    const uint32_t st_nfin = st->n_finite, st_pad0 = st->pad[0], st_pad1 = st->pad[1];
    const float st_limit = st->limit;
#pragma unroll
    for (int k = 0; k < 12; ++k) asm volatile("" ::"s"(T.m[k]));
    asm volatile("" ::"s"(st_nfin), "s"(st_pad0), "s"(st_pad1), "s"(st_limit));
    asm volatile("" ::"v"(ps_raw), "v"(dd), "v"(s.x), "v"(sn.x), "v"(loc[0]), "v"(loc[4]));
    if (st_done) return;
    const int ps = i < n ? ps_raw : -1;
    const int psc = ps >= 0 ? ps : 0;
    const float4 nn_ld = tgt_nrm[2 * (size_t)psc + 1];   // {point, normal} pair: one line
    const float4 q_ld = tgt_nrm[2 * (size_t)psc];
    // trimmed-quantile limit: last radix level, re-derived by every workgroup (f.use_trim == 2),
    // or taken from the state as given by the caller (f.use_trim == 1: distributed path)
    float limit = INFINITY;
    if (f.use_trim == 2) {
        __shared__ uint32_t wave_tot[4];
        __shared__ uint32_t pick[3];
        if (st_nfin != 0) {
            const int nb = 1 << (shift0 - 11);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if ((int)threadIdx.x * 8 + k >= nb) loc[k] = 0u;
            block_pick256_regs(loc, st_pad1, wave_tot, pick);
            limit = __uint_as_float(st_pad0 | pick[0]);
        }
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) st->limit = limit;
            for (int k = threadIdx.x; k < 2048; k += blockDim.x) hist1_to_zero[k] = 0;
        }
    } else if (f.use_trim == 1) {
        limit = st_limit;
    }
    double v[kSums];
#pragma unroll
    for (int k = 0; k < kSums; ++k) v[k] = 0.0;
    if (i < n) {
        float w = 0.f;
        if (ps >= 0) {
            v[29] = 1.0;
            w = 1.f;
            if (f.use_maxdist && !(dd <= f.outlier_max_d2)) w = 0.f;
            if (f.use_trim && !(dd <= limit)) w = 0.f;
            const float3 p = xf_point(T, s.x, s.y, s.z);
            const float4 nn = (f.debug & 1) ? make_float4(0.f, 0.f, 1.f, 0.f) : nn_ld;
            if (f.use_normal) {
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
                const float4 q = (f.debug & 1) ? make_float4(s.x, s.y, s.z, 0.f) : q_ld;
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
        if (cache && ps >= 0) {   // the pair this kernel has just gathered rides along with the reading point from now on
            cache[(size_t)n + i] = q_ld;
            cache[2 * (size_t)n + i] = make_float4(nn_ld.x, nn_ld.y, nn_ld.z, 1.f);
        }
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
    // one batch for what does not depend on the match (position, reading point, its covariance, pose), one for the
    // matched point and its covariance (see k_linearize_p2pl)
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t ic = i < n ? i : n - 1;
    const int ps_raw = pos[ic];
    const float4 s = src[ic];
    const float4 a0 = src_cov[2 * ic], a1 = src_cov[2 * ic + 1];
    const int st_done = it->done;
    const Xf T = load_xf(it);
#pragma unroll
    for (int k = 0; k < 12; ++k) asm volatile("" ::"s"(T.m[k]));
    asm volatile("" ::"v"(ps_raw), "v"(s.x), "v"(a0.x), "v"(a1.x));
    if (st_done) return;
    double v[kSums];
#pragma unroll
    for (int k = 0; k < kSums; ++k) v[k] = 0.0;
    if (i < n) {
        const int ps = ps_raw;
        float w = 0.f;
        if (ps >= 0) {
            w = 1.f;
            const float3 tp = xf_point(T, s.x, s.y, s.z);
            const float4 q = tgt[ps];
            const float4 b0 = tgt_cov[2 * (int64_t)ps], b1 = tgt_cov[2 * (int64_t)ps + 1];
            const double r[3] = {(double)q.x - (double)tp.x, (double)q.y - (double)tp.y, (double)q.z - (double)tp.z};
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
