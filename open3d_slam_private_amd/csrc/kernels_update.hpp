// kernels_update.hpp -- last kernel of an iteration: reduce, exact band select, solve, pose update, checkers, mirror
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once

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

// Rare, register-hungry branches of the update kernel's serial section, kept out of line: inlined, their fully
// unrolled fp64 eigen-solves pushed the kernel (capped at 128 VGPRs by its 1024-thread workgroup) to ~700 spilled
// registers, and the spill traffic landed on the common path as scratch round trips.  Results go through LDS so
// that the caller's arrays stay in registers.
__device__ __noinline__ void upd_load_sym6(const double* tot, float* H, float* b6) {
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) {
            const float v = (float)tot[k++];
            H[6 * i + j] = v;
            H[6 * j + i] = v;
        }
    for (int i = 0; i < 6; ++i) b6[i] = -(float)tot[21 + i];
}
__device__ __noinline__ int upd_solve6_xicp(const double* tot, const int* flags, float* x_out) {
    float H[36], b6[6], x[6];
    upd_load_sym6(tot, H, b6);
    const int rank = solve6_xicp(H, b6, flags, x);
    for (int i = 0; i < 6; ++i) x_out[i] = x[i];
    return rank;
}
__device__ __noinline__ int upd_solve6_p2pl(const double* tot, float* x_out) {
    float H[36], b6[6], x[6];
    upd_load_sym6(tot, H, b6);
    const int rank = solve6_p2pl(H, b6, x);
    for (int i = 0; i < 6; ++i) x_out[i] = x[i];
    return rank;
}
__device__ __noinline__ int upd_solve_sym6(const double* tot, double* dl_out) {
    double Hd[36], g[6], dl[6];
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) Hd[6 * i + j] = Hd[6 * j + i] = tot[k++];
    for (int i = 0; i < 6; ++i) g[i] = -tot[21 + i];
    const int rank = solve_sym6(Hd, g, dl, 1e-12);
    for (int i = 0; i < 6; ++i) dl_out[i] = dl[i];
    return rank;
}
// R8x stage A for one 3x3 block (o = 0: rotation, 3: translation) of the system
__device__ __noinline__ void upd_xicp_stage_a(const double* tot, const float* Trd, float* dst, int o) {
    double S[9], V[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const int lo = (i < j ? i : j) + o, hi = (i < j ? j : i) + o;
            S[3 * i + j] = (double)(float)tot[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
        }
    eig3_desc(S, V);
    for (int kk = 0; kk < 3; ++kk)
        for (int rr = 0; rr < 3; ++rr) {
            const float a0 = Trd[rr] * (float)V[kk], a1 = Trd[4 + rr] * (float)V[3 + kk];
            const float a2 = Trd[8 + rr] * (float)V[6 + kk];
            const float sacc = a0 + a1;
            dst[3 * kk + rr] = sacc + a2;
        }
}

__global__ void __launch_bounds__(1024)
k_reduce_update(const double* __restrict__ partials, int n_blocks, IterState* it, HostMirror* host,
                unsigned long long seq, int fused, const float* __restrict__ band, float* __restrict__ w_out,
                const SelectState* __restrict__ sel, const float* __restrict__ gathered, int n_ranks, int my_rank,
                XicpState* __restrict__ xs) {
    const int contrib_cap = contrib_cap_for(n_ranks);              // gathered blocks: per-rank record capacity
    const size_t contrib_stride = contrib_floats(contrib_cap);     // floats per rank block
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
    constexpr int kDirectCap = 256;   // bands up to this many records are ranked directly (4 threads per record)
    __shared__ __attribute__((aligned(16))) uint32_t bdk[kDirectCap];
    __shared__ uint32_t dcnt[kDirectCap];
    __shared__ __attribute__((aligned(16))) uint32_t mir_w[(sizeof(HostMirror) + 3) / 4];
    __shared__ uint32_t small[64];
    __shared__ uint32_t s_cnt, s_csel, s_need_radix;
    __shared__ uint32_t rk_off[65];   // multi-GPU: first global band index of every rank's records (+ total)
    __shared__ uint32_t rk_bad;
    __shared__ int s_skip_mirror;
    __shared__ float s_x[6];     // results of the out-of-line solvers (rare branches of the serial section)
    __shared__ double s_dl[6];
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
        // (eight loads in flight per thread: one after the other, the ~25 rows a part sums at C3 were a chain of 25 L2 / Infinity
        //  Cache round trips -- most of this kernel's 15 us on the select-based path)
        const int n_rows = fused ? kAccRows : n_blocks;
        for (int b0 = part; b0 < n_rows; b0 += 32 * 8) {
            double v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 32 * u;
                v8[u] = partials[(size_t)(b < n_rows ? b : part) * kSums + comp];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) t += (b0 + 32 * u < n_rows) ? v8[u] : 0.0;
        }
    }
    // Single-GPU fused path: the first kSpec * 32 band records (the usual band holds fewer) are loaded speculatively in
    // the same batch as the state -- their addresses do not depend on the record count, only their validity does; this
    // takes one dependent round trip (state -> band) off the kernel's critical path.  Slots beyond the count hold stale
    // records of earlier iterations and are masked at every use.
    constexpr int kSpec = 6;
    const bool spec = fused == 1 && !gathered && band != nullptr;
    uint32_t spec_d2 = 0;
    int spec_idx = 0;               // the record's point index (component 31): only read when the record is trimmed away
    float spec_pre[kSpec];
#pragma unroll
    for (int u = 0; u < kSpec; ++u) spec_pre[u] = 0.f;
    if (spec) {
        spec_d2 = __float_as_uint(band[band_at(29, threadIdx.x)]);
        if (w_out) spec_idx = __float_as_int(band[band_at(31, threadIdx.x)]);
        if (comp != 29 && comp != 31) {
#pragma unroll
            for (int u = 0; u < kSpec; ++u) spec_pre[u] = band[band_at(comp, (uint32_t)part + 32u * u)];
        }
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
                const uint32_t cnt = reinterpret_cast<const uint32_t*>(gathered + (size_t)r * contrib_stride)[64];
                if (cnt > (uint32_t)contrib_cap) bad = 1;
                off += min(cnt, (uint32_t)contrib_cap);
            }
            rk_off[n_ranks] = off;
            if (off > (uint32_t)kBandCap) bad = 1;   // more records than the LDS staging holds (large groups): stall + generic repair
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
        return gathered[(size_t)r * contrib_stride + kContribHdr + (size_t)(i - rk_off[r]) * kRec + c];
    };
    // fused path: issue this thread's band-record loads right away (they only depend on the record count); the
    // barriers below are LDS-only, so the loads stay in flight behind the partial sums
    const bool trim = s_use_trim && s_ratio != 1.0f;
    const uint32_t n_band = (fused && trim && !band_bad) ? s_band_count : 0u;
    const bool add_comp = comp != 29 && comp != 31;
    // Registers hold what the USUAL band needs (<= 1024 records' d2, the first kPre * 32 records' components): the first
    // fused iterations' larger bands take the loops further down (the kernel sits on the 128-register cap of its 1024-thread
    // workgroup; every array kept live across it ends up in scratch on the common path).
    constexpr int kPre = 8;
    float pre[kPre];
    uint32_t my_d2[1];
    if (n_band) {
        my_d2[0] = 0u;
        if (spec)
            my_d2[0] = spec_d2;
        else if (threadIdx.x < n_band)
            my_d2[0] = __float_as_uint(rec(threadIdx.x, 29));
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            pre[u] = 0.f;
            if (u < kSpec && spec) {
                pre[u] = spec_pre[u < kSpec ? u : 0];
            } else if (32u * u < n_band) {     // workgroup-uniform
                const uint32_t i = min((uint32_t)part + 32u * u, n_band - 1);
                if (add_comp) pre[u] = rec(i, comp);
            }
        }
    }
    if (gathered) {
        for (int r = part; r < n_ranks; r += 32)
            t += reinterpret_cast<const double*>(gathered + (size_t)r * contrib_stride)[comp];
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
                HostMirror::SeqRecord* rec = &host->ring[seq % kSeqRing];
                rec->iterations = sit->iterations;
                rec->done = 0;
                rec->stall = 1;
                rec->limit_last = sit->limit_last;
                rec->limit_prev = sit->limit_prev;
                __threadfence_system();
                __hip_atomic_store(&rec->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        if (n_finite != 0) {
            // stage the band's d2 bit patterns in LDS (the first 1024 were loaded at kernel start)
            if (threadIdx.x < n_band) bd2[threadIdx.x] = my_d2[0];
            for (uint32_t i = threadIdx.x + 1024u; i < n_band; i += 1024u) bd2[i] = __float_as_uint(rec(i, 29));
            // One-level select: the band's values lie in [band_lo, band_hi), so the order-preserving key
            // (u - u_lo) * 2048 / (u_hi - u_lo) spreads them over 2048 bins (about one value per bin); the bin that
            // holds rank r is then resolved by direct ranking.  Crowded bin (> 64 equal-ish values): radix levels.
            const uint32_t u_lo = __float_as_uint(s_band_lo), u_hi = __float_as_uint(s_band_hi);
            uint32_t rank = k - n_below, prefix = 0;
            // Small band (the usual case once the limit has settled: ~200 records): direct ranking on unique keys
            // ((u - u_lo) << 8 | record), four threads per record, each counting a quarter of the band with one
            // compare per pair -- three LDS barriers instead of the six of the histogram path below.
            const bool direct = n_band <= (uint32_t)kDirectCap && u_hi > u_lo && (u_hi - u_lo) < (1u << 24) &&
                                !(sit->debug_narrow_band & 2);   // debug_flags & 64: histogram path for every band (tests)
            if (direct) {
                sx1 = __builtin_amdgcn_s_memtime();
                const uint32_t n_pad = (n_band + 15u) & ~15u;   // whole uint4 per quarter; padding keys compare as "not below"
                if (threadIdx.x < n_pad)
                    bdk[threadIdx.x] = threadIdx.x < n_band ? (((my_d2[0] - u_lo) << 8) | threadIdx.x) : 0xffffffffu;
                if (threadIdx.x < (uint32_t)kDirectCap) dcnt[threadIdx.x] = 0u;
                lds_barrier();
                sx2 = __builtin_amdgcn_s_memtime();
                {
                    const uint32_t i = threadIdx.x & (kDirectCap - 1), qd = threadIdx.x / kDirectCap;   // qd is wave-uniform
                    if (i < n_band) {
                        const uint32_t e = bdk[i];
                        const uint32_t qlen = n_pad / 4u, j0 = qd * qlen;
                        uint32_t rr = 0;
                        for (uint32_t j = j0; j < j0 + qlen; j += 4) {
                            const uint4 o = *reinterpret_cast<const uint4*>(&bdk[j]);
                            rr += (o.x < e ? 1u : 0u) + (o.y < e ? 1u : 0u) + (o.z < e ? 1u : 0u) + (o.w < e ? 1u : 0u);
                        }
                        atomicAdd(&dcnt[i], rr);
                    }
                }
                lds_barrier();
                if (threadIdx.x < n_band && dcnt[threadIdx.x] == rank) s_limit = __uint_as_float(my_d2[0]);
                lds_barrier();
                sx3 = __builtin_amdgcn_s_memtime();
            } else {
                // order-preserving key without integer division: trunc(double(u - u_lo) * 2048 / span) (monotone in u)
                const double kscale = 2048.0 / (double)(u_hi > u_lo ? u_hi - u_lo : 1u);
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
            }
            stB = __builtin_amdgcn_s_memtime();
            const float limit = s_limit;
            // ---- add the band records that survive the trim (component-wise, 32 parts)
            double acc = 0;
            if (add_comp) {
#pragma unroll
                for (int u = 0; u < kPre; ++u) {
                    const uint32_t i = (uint32_t)part + 32u * u;
                    if (32u * u < n_band && i < n_band && __uint_as_float(bd2[i]) <= limit) acc += (double)pre[u];
                }
                for (uint32_t i0 = part + 32u * kPre; i0 < n_band; i0 += 32 * 16) {   // only when n_band > 32 * kPre
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
                            w_out[(spec && i < 1024u) ? spec_idx : __float_as_int(rec(i, 31))] = 0.f;
                        } else if (i >= rk_off[my_rank] && i < rk_off[my_rank + 1]) {
                            w_out[__float_as_int(rec(i, 31))] = 0.f;   // only this rank's own points
                        }
                    }
                }
            lds_barrier();   // LDS-only: the weight stores above drain behind the solve (the kernel's end orders them)
            if (threadIdx.x < kSums) {
                double s2 = 0;
#pragma unroll
                for (int p = 0; p < 16; ++p) s2 += sh[p][threadIdx.x];
                tot[threadIdx.x] += s2;
            }
            lds_barrier();
            stC = __builtin_amdgcn_s_memtime();
        }
    } else if (!fused && trim && sel) {
        if (threadIdx.x == 0) s_limit = sel->limit;
        __syncthreads();
    }
    if (threadIdx.x < kSums) sit->sums[threadIdx.x] = tot[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + kQueues && sit->qcount) sit->qcount[(threadIdx.x - 64) * kQueueStride] = 0u;
    if (threadIdx.x >= 64) return;   // the rest is wave 0 only (wave-synchronous: no workgroup barriers below)
    const int lane = threadIdx.x;
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
    unsigned long long st2 = st1, st3 = st1;
    const bool p2pl = sit->cost == REG_COST_P2PL;
    // Everything lane 0 will read from the LDS-resident state is fetched here in one batch, behind the solve below: read
    // one by one inside the serial section, every access cost its own ~100-cycle LDS round trip (3.4 k cycles measured
    // before the pose update even started).
    const float r_limit_last = sit->limit_last, r_limit_sel = s_limit;
    const int r_dbg_narrow = sit->debug_narrow_band & 1, r_update = sit->update, r_xstage = sit->xicp_stage;
    const int r_fixed = sit->fixed_iters, r_iters = sit->iterations;
    int r_xnc = sit->xicp_nc;
    const unsigned r_band_count = sit->band_count;
    const double r_tot28 = tot[28];
    float r_T[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r_T[i] = sit->T[i];
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
    // R8x, first iteration: eigen-directions of the rotation (lane 0) and translation (lane 1) blocks of A, expressed in
    // the frame the data came from (two fp64 Jacobi iterations side by side instead of one after the other on lane 0);
    // the analysis kernels that follow collect the information sums, then this kernel runs again (finish) to decide,
    // solve and update.  Nothing is reported to the host yet.
    const bool stage_a = !finish && r_update != 0 && p2pl && r_xstage == 1 && r_tot28 != 0.0 && xs != nullptr;
    if (stage_a && lane < 2) upd_xicp_stage_a(tot, sit->xicp_Trd, lane == 0 ? xs->vr : xs->vt, lane == 0 ? 0 : 3);
    if (lane == 0) {
        // band for the next iteration from the limits seen so far
        const float limit = finish ? r_limit_last : r_limit_sel;
        if (!finish) {
            sit->limit_prev = r_limit_last;
            sit->limit_last = limit;
        }
        if (finish) {
            // keep the band computed when the sums were reduced
        } else if (!trim || !(limit < INFINITY)) {
            sit->band_lo = INFINITY;   // no trimming / nothing to predict from: every finite match is "certainly kept"
            sit->band_hi = INFINITY;
        } else {
            const float prev = r_limit_last;   // == the new limit_prev
            float m = 0.3f;
            if (prev < INFINITY && prev > 0.f) m = fminf(fmaxf(2.0f * fabsf(limit - prev) / limit + 0.003f, 0.003f), 0.6f);
            if (r_dbg_narrow) m = 1e-7f;   // test hook: forces band mispredictions (stall + repair path)
            sit->band_lo = limit * (1.0f - m);
            sit->band_hi = limit * (1.0f + m);
        }
        const int nband_report = (int)r_band_count;
        sit->band_count = 0;
        sit->stall = 0;
        bool do_update = r_update != 0;
        if (stage_a) {
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
            r_xnc = nc;
            sit->xicp_stage = 0;
        }
        if (do_update) {
            if (r_tot28 == 0.0) {
                sit->status = REG_NO_CORRESPONDENCES;
                sit->done = 1;
            } else if (p2pl) {
                float x[6], dT[16], Tn[16];
                int rank = 6;
                if (r_xnc > 0) {
                    // R8x: no update along the non-localizable eigen-directions of the CURRENT A (PointToPlane.cpp:459-505)
                    rank = upd_solve6_xicp(tot, sit->xicp_flags, s_x);
                    for (int i = 0; i < 6; ++i) x[i] = s_x[i];
                } else if (well) {
                    for (int i = 0; i < 6; ++i) x[i] = (float)xsol[i];
                } else {
                    // ill-conditioned / rank deficient: eigen-solve with the fp32 rank threshold (minimum norm)
                    rank = upd_solve6_p2pl(tot, s_x);
                    for (int i = 0; i < 6; ++i) x[i] = s_x[i];
                }
                sit->rank_last = rank;
                x_to_T(x, dT);
                m4_mul(dT, r_T, Tn);  // T_iter = real * T_iter (ICP.cpp:1213-1215)
                for (int i = 0; i < 16; ++i) sit->T_prev[i] = r_T[i];
                for (int i = 0; i < 16; ++i) sit->T[i] = Tn[i];
                sit->iterations = r_iters + 1;
                bool iterate;
                if (r_fixed > 0)
                    iterate = r_iters + 1 < r_fixed;
                else
                    iterate = sit->chk.check(Tn);
                if (!iterate) sit->done = 1;
            } else if (sit->gicp_stop_rule == 1 && sit->fixed_iters <= 0 &&
                       ((sit->iterations >= 1 &&
                         fabs(r_tot28 / (double)sit->n_total - sit->fit_prev) < (double)sit->gicp_rel_fitness &&
                         fabs(sqrt(tot[30] / r_tot28) - sit->rmse_prev) < (double)sit->gicp_rel_rmse) ||
                        sit->iterations >= sit->max_iter)) {
                // Open3D ICPConvergenceCriteria: this evaluation's fitness / rmse against the previous one's; no further update.
                // The sums reported with this sequence belong to the final pose (T_prev == T).
                if (sit->iterations >= sit->max_iter &&
                    !(sit->iterations >= 1 && fabs(r_tot28 / (double)sit->n_total - sit->fit_prev) < (double)sit->gicp_rel_fitness &&
                      fabs(sqrt(tot[30] / r_tot28) - sit->rmse_prev) < (double)sit->gicp_rel_rmse))
                    sit->chk.max_iter_reached = true;
                else
                    sit->chk.converged = true;
                for (int i = 0; i < 16; ++i) sit->T_prev[i] = r_T[i];
                sit->done = 1;
            } else {
                double dl[6], E[16], Tn[16];
                int rank = 6;
                sit->fit_prev = r_tot28 / (double)sit->n_total;
                sit->rmse_prev = sqrt(tot[30] / r_tot28);
                if (well) {
                    for (int i = 0; i < 6; ++i) dl[i] = xsol[i];
                } else {
                    rank = upd_solve_sym6(tot, s_dl);
                    for (int i = 0; i < 6; ++i) dl[i] = s_dl[i];
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
                    sit->T_prev[i] = r_T[i];
                    sit->Td[i] = Tn[i];
                    sit->T[i] = (float)Tn[i];
                }
                sit->iterations += 1;
                if (sit->fixed_iters > 0) {
                    if (sit->iterations >= sit->fixed_iters) sit->done = 1;
                } else if (sit->gicp_stop_rule == 1) {
                    // decided by the next evaluation (above)
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
        for (int i = 0; i < 16; ++i) m->T_prev[i] = sit->T_prev[i];
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
    HostMirror::SeqRecord* srec = &host->ring[seq % kSeqRing];
    {
        // the record's six payload words in ONE coalesced store (lanes 2..7 of the record's word layout)
        const HostMirror* m = reinterpret_cast<const HostMirror*>(mir_w);
        static_assert(sizeof(HostMirror::SeqRecord) == 32, "record layout: seq(8) iterations done stall pad limit_last limit_prev");
        uint32_t word = 0;
        if (lane == 2) word = (uint32_t)m->iterations;
        if (lane == 3) word = (uint32_t)m->done;
        if (lane == 6) word = __float_as_uint(m->limit_last);
        if (lane == 7) word = __float_as_uint(m->limit_prev);
        if (lane >= 2 && lane < 8) reinterpret_cast<uint32_t*>(srec)[lane] = word;   // stall = pad = 0
    }
    __threadfence_system();   // one fence for the mirror words and the record; then the two sequence words
    if (lane == 0) {
        __hip_atomic_store(&srec->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
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

// Sums behind GetInformationMatrixFromPointClouds: over the matches within max_d2, the moments of the matched reference
// point in the reference's own frame (sorted point + centroid): count, x, y, z, xx, yy, zz, xy, xz, yz.
__global__ void __launch_bounds__(256)
k_info_sums(const int* __restrict__ pos, const float* __restrict__ d2, int64_t n, const float4* __restrict__ tgt,
            float cx, float cy, float cz, float max_d2, double* __restrict__ out /* [10], zeroed */) {
    double v[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = pos[i];
        if (j < 0 || !(d2[i] <= max_d2)) continue;
        const float4 t = tgt[j];
        const double x = (double)(t.x + cx), y = (double)(t.y + cy), z = (double)(t.z + cz);
        v[0] += 1.0;
        v[1] += x;
        v[2] += y;
        v[3] += z;
        v[4] += x * x;
        v[5] += y * y;
        v[6] += z * z;
        v[7] += x * y;
        v[8] += x * z;
        v[9] += y * z;
    }
    xicp_block_add<10>(v, out);
}
