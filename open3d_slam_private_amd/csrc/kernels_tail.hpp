// kernels_tail.hpp -- the settled tail of a registration as ONE persistent launch (round 3)
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once

// =================================================================================================
// Persistent settled tail (replaces, per Gauss-Newton iteration, k_coh_check + k_coh_search + k_reduce_update)
// =================================================================================================
// What the three-launch iteration paid for: three kernel boundaries (~2.8 us each), a launch ramp and cold L2s per
// kernel, a 23 MB re-stream of the per-point state (reading point, anchor, matched point, normal, runner-up) from
// HBM / Infinity Cache, and a one-workgroup update kernel on a 256-CU chip -- 41 us for 1.2 us of HBM work at C3.
// Here ONE launch runs every remaining iteration of the registration (reference loop: ICP.cpp:1027-1311):
//   * one 512-thread workgroup per CU (1024 reading-point slots, two per thread), co-resident by construction (grid <= CU
//     count, checked against the occupancy query on the host);
//   * the per-point state (the four cache rows of k_coh_check) lives in LDS (64 KB per workgroup) and never leaves the
//     chip between iterations; only the reading point / normal (constant) are re-read, coalesced, from L2;
//   * slots are dealt to workgroups in octets of 8 consecutive (Morton-ordered) reading points, round-robin inside one
//     XCD class (blockIdx % 8 keeps one contiguous eighth of the reading, as everywhere else): the points whose shortcut
//     fails cluster in space (clutter, thin structures), the interleave spreads a cluster over all CUs of the XCD, so
//     every workgroup searches ITS OWN few failures (8 lanes per point, nearest_group as in k_coh_search) -- no queue,
//     no stealing, no second grid-wide rendezvous;
//   * per iteration ONE all-to-all exchange.  Every wave adds its 32 partial sums (fp64, reduced over the wave by a halving
//     butterfly: no LDS table) to the accumulator row of its XCD class with ONE 256-byte memory-side fp64 atomic instruction;
//     band records go into one global list (slot = a returning atomic per wave that has any, issued before the sums are
//     formed so that its latency hides behind them).  The workgroup then arrives on its class counter.  After the rendezvous
//     EVERY workgroup reads the same 8 x 32 sums, the record count and -- speculatively, in the same batch of loads -- one
//     record per thread: one dependent memory hop instead of three (rows -> offsets -> records), 2 KB + the records instead
//     of 67 KB per workgroup.  Then every workgroup redundantly verifies the predicted band with the exact counts, selects
//     the exact quantile inside it, adds the surviving records, solves the 6x6 system, updates the pose and runs the
//     checkers -- identical arithmetic on identical inputs, hence identical poses on every CU without a broadcast and
//     without a one-workgroup kernel.  (The accumulation order of the atomics is not fixed: A, b are equal up to the fp64
//     summation order from run to run, as with the accumulator replicas of the three-launch iteration; every workgroup
//     reads the SAME accumulated values, which is what the redundant solve needs.)  Accumulators and counters live in a
//     ring of four epochs; workgroup 0 clears the set two epochs ahead.
// Hand-off form (MI355X guide, "Valid forms"): every handed-off byte is stored with an agent-scope relaxed atomic store
// (write-through, sc1), every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup barrier, ONE lane then adds
// to the arrival counter; the consumer polls with relaxed agent-scope loads from one wave, (optionally) fences, joins the
// workgroup barrier, and loads every handed-off byte with agent-scope relaxed atomic loads (sc1: L1 bypassed).
// Every spin is bounded (s_memrealtime): a lost workgroup turns into REG_DEVICE_ERROR on the host, not a hang.
#ifndef O3D_TAIL_ACQ
#define O3D_TAIL_ACQ 0   // 1: agent-scope acquire fence after the poll in addition to the sc1 loads (+1.5 us per exchange, measured;
                         // the hand-off form below needs none: every handed-off byte is loaded sc1)
#endif
#ifndef O3D_TAIL_THREADS
#define O3D_TAIL_THREADS 512
#endif
constexpr int kTailThreads = O3D_TAIL_THREADS;   // 8 waves = 2 per SIMD: 256 VGPRs per lane (a 1024-thread workgroup, capped at 128, spilled ~90
                                        // registers around the search and paid a scratch round trip in every phase)
#ifndef O3D_TAIL_SLOTS
#define O3D_TAIL_SLOTS 1024
#endif
#ifndef O3D_TAIL_WGS_PER_CU
#define O3D_TAIL_WGS_PER_CU 1
#endif
constexpr int kTailSlots = O3D_TAIL_SLOTS;   // reading-point slots per workgroup: thread t owns slots t and t + 512
constexpr int kTailWgsPerCu = O3D_TAIL_WGS_PER_CU;   // co-resident workgroups per CU the plan may use (launch bounds follow)
constexpr int kTailPts = kTailSlots / kTailThreads;
constexpr int kTailBandCap = 1024;     // band records one iteration may hold in all (more: stall, select-based repair)
constexpr int kTailRec = 12;           // floats per band record: F0..F5, r, d2, kept, 0, kept d2, 0  (= a factor row)
constexpr int kTailHistRow = 32;       // doubles per workgroup in the coarse-histogram rows (wide bands): 256 one-byte counts
constexpr int kTailCoarse = 256;       // coarse bins of a wide band (second exchange of the iteration: see k_tail)
constexpr float kTailWideRel = 0.02f;  // a band wider than this fraction of its lower edge takes the two-exchange form
constexpr int kTailMaxIters = 64;      // iterations per launch
constexpr int kTailSyncWords = 256;    // zero when a launch starts: arrival counters, record counters, error word, statistics
constexpr int kTailRing = 4;           // epochs an accumulator set / record counter lives before it is reused
constexpr int kTailBandCntWord = 128;  // record counter of epoch e: word [128 + 16 * (e % 4)]
constexpr int kTailAccRows = 16;       // accumulator rows per epoch: same-address fp64 atomics serialise (~40 ns each, measured:
                                       // 256 waves on 8 rows cost the exchange 4 us), so one atomic per WORKGROUP on 16 rows
constexpr int kTailAccDoubles = kTailRing * kTailAccRows * kSums;   // accumulators behind the sync words: [epoch % 4][row][32]
constexpr int kTailSyncBytes = kTailSyncWords * 4 + kTailAccDoubles * 8;   // the block a launch finds zeroed (two copies: workgroup 0 of a launch zeroes the other one on its way out)
constexpr int kTailArriveStride = 16;  // arrival counters 64 bytes apart: word [x * 16], x = XCD class 0..7
constexpr int kTailErrWord = 192;      // != 0: a grid barrier timed out
constexpr int kTailSearchedWord = 193; // statistics: points searched (summed over iterations and workgroups)
constexpr int kTailItersWord = 194;    // statistics: iterations run by this launch (workgroup 0)
constexpr int kTailCauseWord = 195;    // statistics: why the launch stalled (workgroup 0): 1 a coarse count saturated its byte, 2 more band records than
                                       // kTailBandCap, 4 the rank lies outside the predicted band (8: outside the wide band)
constexpr int kTailStampWord = 200;    // O3D_TAIL_STAMPS builds: 12 x uint64 per-phase ticks (10 ns) of workgroup 0, then of the last workgroup
constexpr int kTailGroups = kTailThreads / 8;   // searches per round
constexpr int kTailBins = 1024;        // bins of the one-level select inside the band (two per thread)
#ifndef O3D_TAIL_STAMPS
#define O3D_TAIL_STAMPS 0
#endif
#ifndef O3D_TAIL_G4
#define O3D_TAIL_G4 0   // 1: 4 lanes per point in the tail kernel's searches when a workgroup has more failures than 8-lane groups
                        // (measured round 3: no gain at C2 / C3 / C4 -- the early iterations are not bound by the number of rounds)
#endif
#if O3D_TAIL_STAMPS
#define TAIL_STAMP(i)                                                      \
    do {                                                                   \
        const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();  \
        stamp_acc[i] += now_ - stamp_last;                                 \
        stamp_last = now_;                                                 \
    } while (0)
#else
#define TAIL_STAMP(i) do { } while (0)
#endif

struct TailCfg {
    int64_t n;
    int chunk8;          // reading points per XCD class (a multiple of 8)
    int wpc;             // workgroups per XCD class (grid = 8 * wpc)
    int tile;            // octets per XCD tile (0: one contiguous eighth of the reading per XCD class); chunk8 is a multiple of 8 * tile
    int max_iters;       // iterations this launch may run (<= kTailMaxIters)
    float slack;         // candidate-bounded boxes of the fallback searches (see nearest_group)
    unsigned long long seq;
    unsigned long long timeout_ticks;   // s_memrealtime ticks (100 MHz) a barrier may take before the launch gives up
};

// LDS carve-up (dynamic region only; every offset a multiple of 16)
struct TailLds {
    static constexpr int kRows = 0;                                   // float4 [5][1024] (row 4: GICP only)
    static constexpr int kUnion = kRows + 5 * kTailSlots * 16;        // segment lists of the searches | ranking keys of the select
    static constexpr int kUnionSeg = (kTailThreads / 8) * kSegWords<8> * 4, kUnionSeg4 = O3D_TAIL_G4 ? (kTailThreads / 4) * kSegWords<4> * 4 : 0;
    static constexpr int kUnionMax = kUnionSeg > kUnionSeg4 ? (kUnionSeg > kTailBandCap * 8 ? kUnionSeg : kTailBandCap * 8)
                                                            : (kUnionSeg4 > kTailBandCap * 8 ? kUnionSeg4 : kTailBandCap * 8);
    static constexpr int kUnionBytes = (kUnionMax + 255) & ~255;
    static constexpr int kD2 = kUnion + kUnionBytes;                  // float [1024]
    static constexpr int kFail = kD2 + kTailSlots * 4;              // uint16 [1024]
    static constexpr int kWcls = kFail + kTailSlots * 2;            // uint8 [1024]
    static constexpr int kHint = kWcls + kTailSlots;                // uint8 [1024]
    static constexpr int kSh = kHint + kTailSlots;                  // double [16][32]
    static constexpr int kTot = kSh + 16 * kSums * 8;                 // double [32]
    static constexpr int kState = kTot + kSums * 8;                   // IterState (words)
    static constexpr int kStateBytes = (int)((sizeof(IterState) + 15) & ~size_t(15));
    static constexpr int kMisc = kState + kStateBytes;                // 16 words
    static constexpr int kWtot = kMisc + 64;                          // uint32 [16] per-wave totals (band prefix, scans)
    static constexpr int kOff = kWtot + 64;                           // uint32 [257 -> 272] band offsets of the workgroups
    static constexpr int kX = kOff + 272 * 4;                         // float s_x[8] + pad: scratch of the out-of-line solvers
    static constexpr int kMirror = kX + 96;                           // HostMirror staging (workgroup 0)
    static constexpr int kMirrorBytes = (int)((sizeof(HostMirror) + 15) & ~size_t(15));
    static constexpr int kHist = kMirror + kMirrorBytes;              // uint32 [kTailBins]
    static constexpr int kTotal = kHist + kTailBins * 4;
    static constexpr int kBk = kUnion;                                // uint64 [kTailBandCap] keys of the picked bin's members
};
static_assert(TailLds::kBk + kTailBandCap * 8 - TailLds::kUnion <= TailLds::kUnionBytes, "phase-C scratch exceeds the union");
static_assert(kTailGroups * kSegWords<8> * 4 <= TailLds::kUnionBytes, "segment lists exceed the union");
static_assert(TailLds::kTotal * kTailWgsPerCu <= 160 * 1024, "LDS budget of one CU");
constexpr int kTailLdsBytes = TailLds::kTotal;

__device__ __forceinline__ double tail_ld_f64(const double* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<unsigned long long*>(const_cast<double*>(p)), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)v);
}
__device__ __forceinline__ void tail_st_f64(double* p, double x) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long tail_ld_u64(const void* p) {
    return __hip_atomic_load(reinterpret_cast<unsigned long long*>(const_cast<void*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tail_st_u64(void* p, unsigned long long x) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned tail_ld_u32(const unsigned* p) {
    return __hip_atomic_load(const_cast<unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// R8 + R9 of the tail kernel, wave 0 of every workgroup (identical inputs -> identical poses everywhere): Gauss-Jordan on the
// augmented 6x7 system, one entry per lane, fp64 (as in k_reduce_update), then x -> dT, T_iter <- dT * T_iter, the next
// band and the checkers on lane 0.  Out of line: its register needs must not add to the search loop's.
__device__ __noinline__ void tail_solve_update(IterState* sit, const double* tot, const uint32_t* misc, float* s_x, bool trim,
                                               unsigned n_band_raw) {
    const int lane = (int)(threadIdx.x & 63);
    const float r_limit_last = sit->limit_last, r_limit_sel = __uint_as_float(misc[2]);
    const int r_dbg_narrow = sit->debug_narrow_band & 1;
    const int r_fixed = sit->fixed_iters, r_iters = sit->iterations;
    const int r_xnc = sit->xicp_nc;
    const double r_tot28 = tot[28];
    float r_T[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r_T[i] = sit->T[i];
    const int r = lane >> 3, c = lane & 7;
    double a = 0.0;
    if (r < 6 && c < 7) {
        if (c < 6) {
            const int lo = r < c ? r : c, hi = r < c ? c : r;
            const int k = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
            a = (double)(float)tot[k];
        } else {
            a = (double)(-(float)tot[21 + r]);
        }
    }
    const double a_orig = a;
    double dmax = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) dmax = fmax(dmax, fabs(__shfl(a_orig, j * 8 + j)));
    bool well = dmax > 0.0;
    const double piv_thr = 1e-4 * dmax;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double pj = __shfl(a, j * 8 + j);
        well = well && (pj > piv_thr);
        const double ajc = __shfl(a, j * 8 + c);
        const double arj = __shfl(a, r * 8 + j);
        const double qd = ajc / pj;
        a = (r == j) ? qd : a - arj * qd;
    }
    double xsol[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) xsol[i] = __shfl(a, i * 8 + 6);
    if (lane != 0) return;
    const float limit = r_limit_sel;
    sit->limit_prev = r_limit_last;
    sit->limit_last = limit;
    if (!trim || !(limit < INFINITY)) {
        sit->band_lo = INFINITY;
        sit->band_hi = INFINITY;
    } else {
        const float prev = r_limit_last;
        float m = 0.3f;
        if (prev < INFINITY && prev > 0.f) m = fminf(fmaxf(2.0f * fabsf(limit - prev) / limit + 0.003f, 0.003f), 0.6f);
        if (r_dbg_narrow) m = 1e-7f;   // test hook: forces band mispredictions (stall + repair path)
        sit->band_lo = limit * (1.0f - m);
        sit->band_hi = limit * (1.0f + m);
    }
    sit->band_count = n_band_raw;   // reported with the mirror (pad_nband)
    sit->stall = 0;
    if (r_tot28 == 0.0) {
        sit->status = REG_NO_CORRESPONDENCES;
        sit->done = 1;
        return;
    }
    float x[6], dT[16], Tn[16];
    int rank = 6;
    if (r_xnc > 0) {
        // R8x: no update along the non-localizable eigen-directions of the CURRENT A (PointToPlane.cpp:459-505)
        rank = upd_solve6_xicp(tot, sit->xicp_flags, s_x);
        for (int i = 0; i < 6; ++i) x[i] = s_x[i];
    } else if (well) {
        for (int i = 0; i < 6; ++i) x[i] = (float)xsol[i];
    } else {
        rank = upd_solve6_p2pl(tot, s_x);   // ill-conditioned / rank deficient: eigen-solve, minimum norm
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
}

// GICP factor of ONE pair, added into the 32 running sums (the arithmetic of k_linearize_gicp, kernels_match.hpp:
// r = q - T p, M = (Cq + R Cp R^T)^-1, J = [R skew(p), -R]; H += J^T M J, b += J^T M r, e += 0.5 r^T M r; fp64 per pair).
__device__ __forceinline__ void tail_gicp_factor(const Xf& T, const float4 s, const float4 a0, const float4 a1, const float4 q,
                                                 const float4 b0, const float4 b1, float d2, double (&v)[kSums]) {
    const float3 tp = xf_point(T, s.x, s.y, s.z);
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
            v[k++] += t;
        }
#pragma unroll
    for (int a = 0; a < 6; ++a) v[21 + a] += J[a] * Mr[0] + J[6 + a] * Mr[1] + J[12 + a] * Mr[2];
    v[27] += 0.5 * (r[0] * Mr[0] + r[1] * Mr[1] + r[2] * Mr[2]);
    v[28] += 1.0;
    v[29] += 1.0;
    v[30] += (double)d2;
}

// R8 + R9 of the tail kernel for the GICP cost: the arithmetic of k_reduce_update's GICP branch (fp64 Gauss-Jordan, se(3)
// exponential, right-multiplied update, the two stop rules), wave 0 of every workgroup.
__device__ __noinline__ void tail_solve_update_gicp(IterState* sit, const double* tot, double* s_dl) {
    const int lane = (int)(threadIdx.x & 63);
    const int r = lane >> 3, c = lane & 7;
    double a = 0.0;
    if (r < 6 && c < 7) {
        if (c < 6) {
            const int lo = r < c ? r : c, hi = r < c ? c : r;
            a = tot[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
        } else {
            a = -tot[21 + r];
        }
    }
    const double a_orig = a;
    double dmax = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) dmax = fmax(dmax, fabs(__shfl(a_orig, j * 8 + j)));
    bool well = dmax > 0.0;
    const double piv_thr = 1e-10 * dmax;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double pj = __shfl(a, j * 8 + j);
        well = well && (pj > piv_thr);
        const double ajc = __shfl(a, j * 8 + c);
        const double arj = __shfl(a, r * 8 + j);
        const double qd = ajc / pj;
        a = (r == j) ? qd : a - arj * qd;
    }
    double xsol[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) xsol[i] = __shfl(a, i * 8 + 6);
    if (lane != 0) return;
    const double cnt = tot[28];
    sit->limit_prev = sit->limit_last;
    sit->limit_last = INFINITY;
    sit->band_lo = INFINITY;
    sit->band_hi = INFINITY;
    sit->band_count = 0;
    sit->stall = 0;
    float r_T[16];
    for (int i = 0; i < 16; ++i) r_T[i] = sit->T[i];
    if (cnt == 0.0) {
        sit->status = REG_NO_CORRESPONDENCES;
        sit->done = 1;
        return;
    }
    const double fit = cnt / (double)sit->n_total, rmse = sqrt(tot[30] / cnt);
    if (sit->gicp_stop_rule == 1 && sit->fixed_iters <= 0) {
        const bool conv = sit->iterations >= 1 && fabs(fit - sit->fit_prev) < (double)sit->gicp_rel_fitness &&
                          fabs(rmse - sit->rmse_prev) < (double)sit->gicp_rel_rmse;
        if (conv || sit->iterations >= sit->max_iter) {   // Open3D ICPConvergenceCriteria: no further update (see k_reduce_update)
            if (conv)
                sit->chk.converged = true;
            else
                sit->chk.max_iter_reached = true;
            for (int i = 0; i < 16; ++i) sit->T_prev[i] = r_T[i];
            sit->done = 1;
            return;
        }
    }
    sit->fit_prev = fit;
    sit->rmse_prev = rmse;
    double dl[6], E[16], Tn[16];
    int rank = 6;
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

template <bool kGicp>
__global__ void __launch_bounds__(kTailThreads, (kTailThreads / 256) * kTailWgsPerCu)   // (threads, waves per SIMD)
k_tail(const float4* __restrict__ src, const float4* __restrict__ src_nrm /* GICP: the reading's covariances, 2 x float4 */,
       IterState* __restrict__ it_g, Grid g, const float4* __restrict__ tgt_nrm /* GICP: the reference's covariances */,
       FilterCfg f, int* __restrict__ pos_io, float* __restrict__ d2_out,
       float* __restrict__ w_out, uint8_t* __restrict__ hint_g, float4* __restrict__ cache, unsigned* __restrict__ sync,
       double* __restrict__ hist_g /* [2][grid][kTailHistRow] */, float* __restrict__ band_g /* [2][kTailBandCap][12] */,
       HostMirror* host, TailCfg cfg, unsigned* __restrict__ sync_next /* the counter block of the NEXT launch: zeroed on the way out */) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float4* const st_rows = reinterpret_cast<float4*>(lds + TailLds::kRows);            // [row * kTailSlots + slot]
    uint32_t* const seg_lds = reinterpret_cast<uint32_t*>(lds + TailLds::kUnion);
    unsigned long long* const bk = reinterpret_cast<unsigned long long*>(lds + TailLds::kBk);
    float* const d2s = reinterpret_cast<float*>(lds + TailLds::kD2);
    uint16_t* const fail = reinterpret_cast<uint16_t*>(lds + TailLds::kFail);
    uint8_t* const wcls = reinterpret_cast<uint8_t*>(lds + TailLds::kWcls);
    uint8_t* const hnt = reinterpret_cast<uint8_t*>(lds + TailLds::kHint);
    double (*const sh)[kSums] = reinterpret_cast<double (*)[kSums]>(lds + TailLds::kSh);
    double* const tot = reinterpret_cast<double*>(lds + TailLds::kTot);
    uint32_t* const s_state = reinterpret_cast<uint32_t*>(lds + TailLds::kState);
    IterState* const sit = reinterpret_cast<IterState*>(s_state);
    // misc: [0] failures of this iteration [2] limit bits [3] poll ok [5] band overflow [6] picked bin [7] rank inside it
    //       [8] its count [9] members gathered [10] coarse bin [12] band points (wide)
    uint32_t* const misc = reinterpret_cast<uint32_t*>(lds + TailLds::kMisc);
    uint32_t* const wtot = reinterpret_cast<uint32_t*>(lds + TailLds::kWtot);
    uint32_t* const off = reinterpret_cast<uint32_t*>(lds + TailLds::kOff);
    float* const s_x = reinterpret_cast<float*>(lds + TailLds::kX);
    uint32_t* const mir_w = reinterpret_cast<uint32_t*>(lds + TailLds::kMirror);
    uint32_t* const hist = reinterpret_cast<uint32_t*>(lds + TailLds::kHist);
    double* const acc_g = reinterpret_cast<double*>(sync + kTailSyncWords);   // [epoch % 4][XCD class][32], zeroed with the sync words
    constexpr int kStateWords = (int)(sizeof(IterState) / 4);
    constexpr int kWaves = kTailThreads / 64;
    static_assert(kStateWords <= kTailThreads && kTailThreads >= 256, "state staging / count scan by the first threads");

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nwg = (int)gridDim.x;
    const int xcls = (int)(blockIdx.x & 7), wq = (int)(blockIdx.x >> 3);
    const int64_t n = cfg.n;
    // slot sl -> reading point: octet o = (sl >> 3) * wpc + wq of XCD class xcls
    int64_t q[kTailPts];
    bool valid[kTailPts];
    float4 s[kTailPts], sn[kTailPts];   // the reading point and its normal: constant, in registers for the whole launch
    float4 sc1[kTailPts];               // (GICP: sn = covariance xx xy xz yy, sc1 = yz zz)
#pragma unroll
    for (int u = 0; u < kTailPts; ++u) {
        const int sl = t + u * kTailThreads;
        const int64_t oc = (int64_t)(sl >> 3) * cfg.wpc + wq;   // octet inside the XCD class
        const int64_t r_in = oc * 8 + (sl & 7);
        if (cfg.tile > 0) {
            // XCD class x owns tiles x, x + 8, ... of `tile` consecutive octets (see xcd_block_tiled: a contiguous eighth ties every
            // exchange to the eighth whose shortcut fails most)
            const int64_t tl = oc / cfg.tile, rr = oc - tl * cfg.tile;
            q[u] = ((tl * 8 + xcls) * cfg.tile + rr) * 8 + (sl & 7);
        } else {
            q[u] = (int64_t)xcls * cfg.chunk8 + r_in;
        }
        valid[u] = r_in < cfg.chunk8 && q[u] < n;
        if (!valid[u]) q[u] = 0;
    }
    // ---- entry: state, per-point rows, previous match
    if (t < kStateWords) s_state[t] = reinterpret_cast<const uint32_t*>(it_g)[t];
#pragma unroll
    for (int u = 0; u < kTailPts; ++u) {
        const int sl = t + u * kTailThreads;
        float4 r0 = cache[q[u]], r1 = cache[(size_t)n + q[u]];
        const float4 r2 = cache[2 * (size_t)n + q[u]], r3 = cache[3 * (size_t)n + q[u]];
        const int pprev = pos_io[q[u]];
        s[u] = src[q[u]];
        sc1[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kGicp) {
            // the select-based GICP iteration leaves no {matched point, covariance} rows in the cache: gathered once, here
            sn[u] = src_nrm[2 * q[u]];
            sc1[u] = src_nrm[2 * q[u] + 1];
            const int pc = pprev >= 0 ? pprev : 0;
            r1 = g.pts[pc];
            float4 c0 = tgt_nrm[2 * (size_t)pc], c1 = tgt_nrm[2 * (size_t)pc + 1];
            c1.w = (valid[u] && pprev >= 0) ? 1.f : 0.f;
            st_rows[2 * kTailSlots + sl] = c0;
            st_rows[4 * kTailSlots + sl] = c1;
        } else {
            sn[u] = f.use_normal ? src_nrm[q[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
            st_rows[2 * kTailSlots + sl] = r2;
        }
        if (!valid[u]) r0.w = -1.f;
        r1.w = __int_as_float(valid[u] ? pprev : -1);
        st_rows[sl] = r0;
        st_rows[kTailSlots + sl] = r1;
        st_rows[3 * kTailSlots + sl] = r3;
        hnt[sl] = hint_g ? hint_g[q[u]] : (uint8_t)0;
        d2s[sl] = INFINITY;
        wcls[sl] = 0;
    }
    for (int k = t; k < kTailBins; k += kTailThreads) hist[k] = 0u;
    if (t < 16) misc[t] = 0u;
    __syncthreads();
    if (sit->done || sit->stall) {   // workgroup-uniform: an earlier sequence ended or stalled the loop
        if (blockIdx.x == 0)
            for (int w2 = t; w2 < kTailSyncBytes / 4; w2 += kTailThreads) sync_next[w2] = 0u;   // (as on the regular way out)
        return;
    }
    const unsigned cnt_lane = lane < 8 ? (unsigned)((nwg - lane + 7) >> 3) : 0u;   // workgroups of XCD class `lane`
    unsigned n_searched = 0;
    int exit_reason = 0;   // 1 done, 2 stall, 3 iteration budget of this launch, 4 barrier timeout
    int k_local = 0;
    unsigned epoch = 0;    // exchanges this workgroup has completed (grid-uniform)
#if O3D_TAIL_STAMPS
    unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memrealtime();
#endif
    // (set at the top of every iteration from a loop-VARIANT copy of the thread index: computed once in front of the loop, the
    //  addresses derived from them stayed live across the whole loop -- the compiler spilled them around the search and
    //  reloaded them, one scratch round trip each, in every phase)
    int tv = t;

    // ---- one all-to-all rendezvous.  Before: every wave has ISSUED its atomics / write-through stores of this epoch.  After
    //      (true): they are visible to agent-scope (sc1) loads of every workgroup.  false: timed out.
    auto exchange = [&](unsigned set) -> bool {
        if (blockIdx.x == 0) {   // the accumulator set / record counter two epochs ahead (nobody reads or adds to them now)
            const unsigned nxt = (epoch + 2u) & (kTailRing - 1);
            if (t < kTailAccRows * kSums) tail_st_f64(acc_g + (size_t)nxt * kTailAccRows * kSums + tv, 0.0);
            if (t == 0) __hip_atomic_store(&sync[kTailBandCntWord + 16 * nxt], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave drains its write-through stores (records, histograms)
        __syncthreads();                                    // ... and the waves' partial sums are in LDS (publish_sums)
        ++epoch;
        if (wave == 0) {
            // the workgroup's 32 sums: ONE 256-byte fp64 atomic instruction into one of kTailAccRows rows of set `set`
            if (lane < kSums) {
                double v = 0.0;
#pragma unroll
                for (int w2 = 0; w2 < kWaves; ++w2) v += sh[w2][lane];
                const int row = (int)(blockIdx.x & 7) * (kTailAccRows / 8) + (int)((blockIdx.x >> 3) & (kTailAccRows / 8 - 1));
                unsafeAtomicAdd(acc_g + ((size_t)set * kTailAccRows + row) * kSums + lane, v);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&sync[xcls * kTailArriveStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * cnt_lane;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool okp = false;
            for (;;) {
                const unsigned v = lane < 8 ? tail_ld_u32(&sync[lane * kTailArriveStride]) : 0u;
                okp = __all(lane >= 8 || v >= target) != 0;
                if (okp) break;
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > cfg.timeout_ticks) break;
            }
#if O3D_TAIL_ACQ
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            if (lane == 0) misc[3] = okp ? 1u : 0u;
        }
        __syncthreads();
        return misc[3] != 0u;
    };
    // ---- this wave's 32 partial sums -> LDS (halving butterfly over the wave, 32 shuffles: no factor table); exchange() adds
    //      the workgroup's total to the accumulators
    auto publish_sums = [&](double (&v)[kSums]) {
        wave_reduce32(v);
        if ((lane & 1) == 0) sh[wave][lane >> 1] = v[0];
    };
    // ---- the kTailAccRows accumulator rows of set `set`, summed in row order -> tot (LDS)
    auto reduce_sums = [&](unsigned set, bool add) {
        static_assert(kTailAccRows * kSums == kTailThreads, "one accumulator entry per thread");
        double a = tail_ld_f64(acc_g + (size_t)set * kTailAccRows * kSums + tv);   // thread t: row t >> 5, component t & 31
        a += __shfl_xor(a, 32);
        if (lane < kSums) sh[wave][lane] = a;   // (wave 0 consumed the partial sums in here before exchange() let anybody through)
        __syncthreads();
        if (t < kSums) {
            double v = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < kWaves; ++w2) v += sh[w2][t];
            tot[t] = add ? tot[t] + v : v;
        }
        __syncthreads();
    };
    // ---- one factor row (F, r; fp32 products, fp64 sums) into the 32 running sums
    auto accumulate = [&](double (&v)[kSums], const float (&F)[7], float md2) {
        int k = 0;
#pragma unroll
        for (int a6 = 0; a6 < 6; ++a6)
#pragma unroll
            for (int c6 = a6; c6 < 6; ++c6) v[k++] += (double)(F[a6] * F[c6]);
#pragma unroll
        for (int a6 = 0; a6 < 6; ++a6) v[21 + a6] += (double)(F[a6] * F[6]);
        v[27] += (double)(F[6] * F[6]);
        v[28] += 1.0;            // kept
        v[30] += (double)md2;    // kept d2
    };
    // ---- band records of this thread's points (those with `mine[u]`) into the global list of buffer `buf`: one returning
    //      atomic per wave that has any (the slot order is the atomics' order: the same for every reader)
    auto reserve_records = [&](unsigned set, const bool (&mine)[kTailPts], unsigned (&slot)[kTailPts]) {
        unsigned long long bal[kTailPts];
        unsigned nw = 0;
#pragma unroll
        for (int u = 0; u < kTailPts; ++u) {
            bal[u] = __ballot(mine[u]);
            nw += (unsigned)__popcll(bal[u]);
        }
        unsigned base = 0;
        if (nw != 0u) {   // wave-uniform
            if (lane == 0) base = atomicAdd(&sync[kTailBandCntWord + 16 * set], nw);
            base = (unsigned)__shfl((int)base, 0);
        }
        unsigned before = 0;
#pragma unroll
        for (int u = 0; u < kTailPts; ++u) {
            slot[u] = base + before + (unsigned)__popcll(bal[u] & ((1ull << lane) - 1ull));
            before += (unsigned)__popcll(bal[u]);
        }
    };
    auto store_record = [&](int buf, unsigned slotb, const float (&F)[7], float md2, float w) {
        if (slotb >= (unsigned)kTailBandCap) return;   // (the count tells every reader that the list overflowed)
        float* rec = band_g + ((size_t)buf * kTailBandCap + slotb) * kTailRec;
        const float kept = w != 0.f ? 1.f : 0.f;
        const float v[12] = {F[0], F[1], F[2], F[3], F[4], F[5], F[6], md2, kept, 0.f, w != 0.f ? md2 : 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 6; ++k)
            tail_st_u64(rec + 2 * k, ((unsigned long long)__float_as_uint(v[2 * k + 1]) << 32) | __float_as_uint(v[2 * k]));
    };

    for (;; ++k_local) {
        // ================= phase A: shortcut test, own searches, factor rows, 32 sums, band records =================
        tv = t;
        asm volatile("" : "+v"(tv));
        Xf T;
#pragma unroll
        for (int k = 0; k < 12; ++k) T.m[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sit->T[k])));   // uniform: SGPRs
        const float band_lo = sit->band_lo, band_hi = sit->band_hi;
        const bool trim = !kGicp && sit->use_trim && sit->trim_ratio != 1.0f;
        // A WIDE band (the trimmed limit still moves by per cents: first iterations after the large corrections) holds
        // thousands of points: too many to hand every workgroup as records.  Two exchanges then: first the certain sums and,
        // per workgroup, the counts of its band points in kTailCoarse coarse bins; every workgroup finds the coarse bin b*
        // that holds the rank; the band shrinks to that bin -- points in lower bins become certain (a second, delta sum),
        // points of b* become the records -- and the iteration ends as a narrow one.  Exact for the same reason: bins are
        // an order-preserving function of the fp32 bit pattern of d2.
        const bool wide = trim && band_hi < INFINITY && (band_hi - band_lo) > kTailWideRel * band_lo;
        const uint32_t u_lo = __float_as_uint(band_lo), u_hi = __float_as_uint(band_hi);
        const double f_scale = (wide ? (double)kTailCoarse : 1.0) / (double)(u_hi > u_lo ? u_hi - u_lo : 1u);
        auto coarse_bin = [&](float d2) -> uint32_t {   // wide bands only; d2 in [band_lo, band_hi)
            return min((uint32_t)((double)(__float_as_uint(d2) - u_lo) * f_scale), (uint32_t)(kTailCoarse - 1));
        };
        float3 p[kTailPts];
#pragma unroll
        for (int u = 0; u < kTailPts; ++u) {
            const int sl = t + u * kTailThreads;
            p[u] = xf_point(T, s[u].x, s[u].y, s[u].z);
            const float4 c = st_rows[sl], tq = st_rows[kTailSlots + sl], nn = st_rows[2 * kTailSlots + sl],
                         ru = st_rows[3 * kTailSlots + sl];
            const int pprev = __float_as_int(tq.w);
            bool pass = false;
            const float attr_ok = kGicp ? st_rows[4 * kTailSlots + sl].w : nn.w;   // matched point + attribute rows are cached
            if (valid[u] && c.w > 0.f && pprev >= 0 && attr_ok == 1.f) {
                const float dx = p[u].x - tq.x, dy = p[u].y - tq.y, dz = p[u].z - tq.z;
                float a = dx * dx;
                float b2 = dy * dy;
                float d2 = a + b2;
                a = dz * dz;
                d2 = d2 + a;                                   // NC5: the value the full search would compute for this pair
                const float ex = p[u].x - c.x, ey = p[u].y - c.y, ez = p[u].z - c.z;
                const float delta = __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez);
                const float lhs = (__builtin_amdgcn_sqrtf(d2) + delta) * 1.00001f + 1e-30f;
                const float rhs = __builtin_amdgcn_sqrtf(c.w) * 0.99999f;
                pass = d2 <= g.max_d2 && lhs < rhs;
                if (ru.w == 1.f) {   // two-candidate form (see k_coh_check)
                    const float fx = p[u].x - ru.x, fy = p[u].y - ru.y, fz = p[u].z - ru.z;
                    float a2 = fx * fx;
                    float b3 = fy * fy;
                    float r2 = a2 + b3;
                    a2 = fz * fz;
                    r2 = r2 + a2;
                    pass = pass && d2 < r2;
                }
                if (pass) d2s[sl] = d2;
            }
            if (valid[u] && !pass) {
                const unsigned at = atomicAdd(&misc[0], 1u);
                fail[at] = (uint16_t)sl;
                st_rows[sl] = make_float4(p[u].x, p[u].y, p[u].z, -1.f);   // where the search runs (it rewrites the row)
            }
        }
        __syncthreads();
        TAIL_STAMP(0);   // shortcut test
        const int nf = (int)misc[0];
        n_searched += (t == 0) ? (unsigned)nf : 0u;
        // ---- this workgroup's own searches: G lanes per point.  Settled iterations have a few dozen failures per workgroup:
        //      8 lanes per point, one round.
        auto search_rounds = [&](auto gtag) {
            constexpr int G = decltype(gtag)::value;
            constexpr int kGroups = kTailThreads / G;
            const int sub = t & (G - 1), grp = t / G;
            for (int base = 0; base < nf; base += kGroups) {
                const int fi = base + grp;
                if (fi < nf) {   // uniform inside a group
                    const int slot = (int)fail[fi];
                    const float4 a0 = st_rows[slot];
                    const float3 p2 = make_float3(a0.x, a0.y, a0.z);
                    const int hv = (int)hnt[slot];
                    int lvl;
                    float cov2;
                    const Best bb = nearest_group<G, O3D_COH_PRUNE != 0, true>(g, p2, sub, -1, &lvl, seg_lds + grp * kSegWords<G>,
                                                                               hv >= 2 ? hv - 2 : -1, &cov2, cfg.slack);
                    if (sub == 0) {
                        hnt[slot] = (uint8_t)(lvl + 1);
                        const int pc = bb.pos >= 0 ? bb.pos : 0;
                        float4 tq, nn;
                        if (kGicp) {
                            tq = g.pts[pc];
                            nn = tgt_nrm[2 * (size_t)pc];                 // covariance xx xy xz yy
                            float4 c1 = tgt_nrm[2 * (size_t)pc + 1];     // yz zz
                            c1.w = 1.f;
                            st_rows[4 * kTailSlots + slot] = c1;
                        } else {
                            tq = tgt_nrm[2 * (size_t)pc];       // {point, normal} pair: one line
                            nn = tgt_nrm[2 * (size_t)pc + 1];
                            nn.w = 1.f;
                        }
                        float4 ru = make_float4(INFINITY, INFINITY, INFINITY, 1.f);   // no runner-up seen: infinitely far
                        if (bb.pos2 >= 0) {
                            const float4 t2 = g.pts[bb.pos2];
                            ru = make_float4(t2.x, t2.y, t2.z, 1.f);
                        }
                        tq.w = __int_as_float(bb.pos);
                        st_rows[slot] = make_float4(p2.x, p2.y, p2.z, bb.pos >= 0 ? fminf(bb.third, cov2) : -1.f);
                        st_rows[kTailSlots + slot] = tq;
                        st_rows[2 * kTailSlots + slot] = nn;
                        st_rows[3 * kTailSlots + slot] = ru;
                        d2s[slot] = bb.pos >= 0 ? bb.d2 : INFINITY;
                    }
                }
            }
        };
#if O3D_TAIL_G4
        if (nf > kTailThreads / 8)
            search_rounds(std::integral_constant<int, 4>());
        else
#endif
            search_rounds(std::integral_constant<int, 8>());
        __syncthreads();   // search results visible to the owners
        TAIL_STAMP(1);   // own searches
        // ---- weights, class, factor row of this thread's points (coh_epilogue without the global writes), accumulated in
        //      registers: certain rows into the 32 sums, band rows kept for the records
        float rF[kTailPts][7], md2v[kTailPts], wv[kTailPts];
        int clsv[kTailPts];
        const unsigned set = epoch & (kTailRing - 1);
        int buf = (int)(epoch & 1u);
        {
            double v[kSums];
#pragma unroll
            for (int k = 0; k < kSums; ++k) v[k] = 0.0;
#pragma unroll
            for (int u = 0; u < kTailPts; ++u) {
                const int sl = t + u * kTailThreads;
#pragma unroll
                for (int k = 0; k < 7; ++k) rF[u][k] = 0.f;
                const float4 tq = st_rows[kTailSlots + sl], nn = st_rows[2 * kTailSlots + sl];
                // passed: the previous match with the distance the check computed; searched: the search's result (both in LDS)
                const int mpos = valid[u] ? __float_as_int(tq.w) : -1;
                const float md2 = d2s[sl];
                float w = 0.f;
                int cls = 2;   // 0: certainly kept, 1: band, 2: dropped / unmatched
                if (kGicp) {
                    if (valid[u] && mpos >= 0) {
                        w = 1.f;
                        cls = 0;
                        tail_gicp_factor(T, s[u], sn[u], sc1[u], tq, nn, st_rows[4 * kTailSlots + sl], md2, v);
                    }
                } else if (valid[u] && mpos >= 0) {
                    w = 1.f;
                    if (f.use_maxdist && !(md2 <= f.outlier_max_d2)) w = 0.f;
                    if (f.use_normal) {
                        const float3 nr = normalize3(xf_rot(T, sn[u].x, sn[u].y, sn[u].z));
                        const float3 nt = normalize3(make_float3(nn.x, nn.y, nn.z));
                        float a = nr.x * nt.x;
                        float bb2 = nr.y * nt.y;
                        float val = a + bb2;
                        a = nr.z * nt.z;
                        val = val + a;
                        if (val < f.cos_max_angle) w = 0.f;
                    }
                    cls = md2 < band_lo ? 0 : (md2 < band_hi ? 1 : 2);
                    if (w != 0.f && cls != 2) {
                        // F = [p x n ; n], r = (p - q) . n  (the weight is 1: w * F_a * F_c == F_a * F_c exactly)
                        float a = p[u].y * nn.z, bq = p[u].z * nn.y;
                        rF[u][0] = a - bq;
                        a = p[u].z * nn.x; bq = p[u].x * nn.z;
                        rF[u][1] = a - bq;
                        a = p[u].x * nn.y; bq = p[u].y * nn.x;
                        rF[u][2] = a - bq;
                        rF[u][3] = nn.x; rF[u][4] = nn.y; rF[u][5] = nn.z;
                        const float dx = p[u].x - tq.x, dy = p[u].y - tq.y, dz = p[u].z - tq.z;
                        float r = dx * nn.x;
                        float t2 = dy * nn.y;
                        r = r + t2;
                        t2 = dz * nn.z;
                        r = r + t2;
                        rF[u][6] = r;
                    }
                    v[29] += 1.0;                               // matched
                    if (cls == 0) v[31] += 1.0;                 // below the band (rank bookkeeping is independent of w)
                    if (w != 0.f && cls == 0) accumulate(v, rF[u], md2);
                }
                if (!(valid[u] && mpos >= 0)) d2s[sl] = INFINITY;
                wcls[sl] = (uint8_t)((w != 0.f && cls != 2 ? 1 : 0) | (cls << 1));
                md2v[u] = md2;
                wv[u] = w;
                clsv[u] = cls;
                if (wide && cls == 1) atomicAdd(&hist[coarse_bin(md2)], 1u);
            }
            // the record slots first (a returning atomic: its round trip hides behind the butterfly), then the sums
            unsigned slot[kTailPts];
            if (!wide) {
                bool mine[kTailPts];
#pragma unroll
                for (int u = 0; u < kTailPts; ++u) mine[u] = clsv[u] == 1;
                reserve_records(set, mine, slot);
            }
            publish_sums(v);
            if (!wide) {
#pragma unroll
                for (int u = 0; u < kTailPts; ++u)
                    if (clsv[u] == 1) store_record(buf, slot[u], rF[u], md2v[u], wv[u]);
            }
        }
        TAIL_STAMP(2);   // weights, classes, sums, records
        if (wide) {
            __syncthreads();   // the workgroup's coarse counts are complete
            if (t < kTailCoarse / 8) {
                // this workgroup's coarse counts, one byte each (saturated: 255 means "too many", the iteration then stalls)
                unsigned long long pk = 0ull;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t c2 = hist[8 * t + j];
                    hist[8 * t + j] = 0u;
                    pk |= (unsigned long long)min(c2, 255u) << (8 * j);
                }
                tail_st_u64(hist_g + ((size_t)buf * nwg + blockIdx.x) * kTailHistRow + tv, pk);
            }
        }
        if (t == 0) {
            misc[0] = 0u;   // next iteration's failure count
            misc[5] = 0u;
            misc[9] = 0u;
            misc[2] = __float_as_uint(INFINITY);
        }
        TAIL_STAMP(3);
        // ================= exchange =================
        if (!exchange(set)) {
            exit_reason = 4;
            if (t == 0) __hip_atomic_fetch_add(&sync[kTailErrWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        TAIL_STAMP(5);   // arrive + wait for every workgroup
        // ================= phase C (every workgroup, identical arithmetic): reduce, verify, select, add, solve, update =====
        uint32_t sel_bin = 0;   // coarse bin the band has shrunk to (wide bands)
        bool stall = false;
        unsigned rset = set;    // accumulator set / record counter holding this iteration's records
        // narrow bands: the record count and, speculatively, one record per thread travel with the sums (one hop)
        unsigned long long rv[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
        unsigned n_band_raw = 0;
        auto load_records_spec = [&]() {
            n_band_raw = tail_ld_u32(&sync[kTailBandCntWord + 16 * rset]);
            const float* rec = band_g + ((size_t)buf * kTailBandCap + (unsigned)tv) * kTailRec;
#pragma unroll
            for (int k = 0; k < 6; ++k) rv[k] = tail_ld_u64(rec + 2 * k);
        };
        if (!wide) load_records_spec();
        reduce_sums(set, false);
        TAIL_STAMP(6);   // sums
        if (trim && wide && (uint32_t)llround(tot[29]) != 0u) {
            // ---- coarse counts of all workgroups -> hist[0..255]; thread t sums word t & 63 (4 bins) of rows t >> 6, + 8, ...
            {
                const unsigned* wbase = reinterpret_cast<const unsigned*>(hist_g + (size_t)buf * nwg * kTailHistRow) + (tv & 63);
                constexpr int kG = kTailThreads / 64, kL = (256 * kTailWgsPerCu) / kG;   // (at most 256 CUs x workgroups per CU)
                unsigned c4[4] = {0, 0, 0, 0};
                bool sat = false;
                // (eight loads in flight at a time: a register array of all kL words cost 32 - 64 registers)
                for (int u0 = 0; u0 < kL; u0 += 8) {
                    unsigned wv2[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int b2 = (tv >> 6) + kG * (u0 + u);
                        wv2[u] = b2 < nwg ? tail_ld_u32(wbase + (size_t)b2 * (kTailHistRow * 2)) : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned bv = (wv2[u] >> (8 * j)) & 255u;
                            sat = sat || bv == 255u;
                            c4[j] += bv;
                        }
                }
                if (sat) misc[5] = 1u;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c4[j]) atomicAdd(&hist[4 * (t & 63) + j], c4[j]);
            }
            __syncthreads();
            const uint32_t n_finite = (uint32_t)llround(tot[29]), n_below = (uint32_t)llround(tot[31]);
            const uint32_t kq = trim_rank(n_finite, sit->trim_ratio);
            if (t < 256) {   // exclusive prefix of the 256 coarse counts (waves 0-3)
                const uint32_t hv2 = hist[t];
                hist[t] = 0u;
                uint32_t incl = hv2;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
                    if (lane >= o) incl += v;
                }
                if (lane == 63) wtot[wave] = incl;
                off[t] = incl - hv2;
            }
            __syncthreads();
            if (t < 256) {
                uint32_t base = 0;
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) base += w2 < wave ? wtot[w2] : 0u;
                off[t] += base;
                if (t == 0) misc[12] = wtot[0] + wtot[1] + wtot[2] + wtot[3];   // every band point
            }
            __syncthreads();
            {
                const uint32_t total_band = misc[12];
                if (!(n_below <= kq && kq < n_below + total_band) || misc[5] != 0u) stall = true;
                if (!stall && t < 256) {
                    const uint32_t excl = off[t], nxt = t == 255 ? total_band : off[t + 1];
                    const uint32_t rank = kq - n_below;
                    if (nxt > excl && rank >= excl && rank < nxt) misc[10] = (uint32_t)t;   // exactly one thread
                }
            }
            __syncthreads();
            if (!stall) {
                sel_bin = misc[10];
                // ---- second exchange: band points below the bin become certain (delta sums), those inside it the records
                buf = (int)(epoch & 1u);
                rset = epoch & (kTailRing - 1);
                double v[kSums];
#pragma unroll
                for (int k = 0; k < kSums; ++k) v[k] = 0.0;
                bool mine[kTailPts];
                unsigned slot[kTailPts];
#pragma unroll
                for (int u = 0; u < kTailPts; ++u) {
                    mine[u] = false;
                    if (clsv[u] == 1) {
                        const uint32_t cb = coarse_bin(md2v[u]);
                        mine[u] = cb == sel_bin;
                        if (cb < sel_bin) {
                            v[31] += 1.0;   // below the (shrunken) band
                            if (wv[u] != 0.f) accumulate(v, rF[u], md2v[u]);
                        }
                    }
                }
                reserve_records(rset, mine, slot);
                publish_sums(v);
#pragma unroll
                for (int u = 0; u < kTailPts; ++u)
                    if (mine[u]) store_record(buf, slot[u], rF[u], md2v[u], wv[u]);
                if (!exchange(rset)) {
                    exit_reason = 4;
                    if (t == 0) __hip_atomic_fetch_add(&sync[kTailErrWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                load_records_spec();
                reduce_sums(rset, true);
            }
        }
        const bool band_bad = misc[5] != 0u || n_band_raw > (unsigned)kTailBandCap;
        const unsigned n_band = (trim && !band_bad && !stall) ? n_band_raw : 0u;
        if (trim) {
            const uint32_t n_finite = (uint32_t)llround(tot[29]), n_below = (uint32_t)llround(tot[31]);
            const uint32_t kq = trim_rank(n_finite, sit->trim_ratio);
            const bool ok = !stall && (n_finite == 0 || (!band_bad && n_below <= kq && kq < n_below + n_band));
            if (!ok) {   // workgroup- and grid-uniform
                if (blockIdx.x == 0 && t == 0)
                    sync[kTailCauseWord] = (misc[5] != 0u ? 1u : 0u) | (n_band_raw > (unsigned)kTailBandCap ? 2u : 0u) |
                                           ((!stall && !band_bad) ? 4u : 0u) | ((stall && misc[5] == 0u) ? 8u : 0u);
                exit_reason = 2;
                break;
            }
            if (n_finite != 0) {
                // ---- the band's records stay in registers (record t came with the sums, records beyond the first kTailThreads take
                //      a second batch); only their d2 keys go through LDS.  One-level select: an order-preserving key spreads the
                //      band's values (inside [band_lo, band_hi), or inside coarse bin sel_bin of it) over kTailBins bins -- about
                //      one value per bin; the bin holding the rank is resolved by direct ranking
                unsigned long long rv2[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
                uint32_t my_bin[kTailPts], my_u[kTailPts];
#pragma unroll
                for (int u = 0; u < kTailPts; ++u) {
                    const unsigned i = (unsigned)(t + u * kTailThreads);
                    my_bin[u] = 0xffffffffu;
                    my_u[u] = 0u;
                    if (i < n_band) {
                        if (u > 0) {
                            const float* rec = band_g + ((size_t)buf * kTailBandCap + i) * kTailRec;
#pragma unroll
                            for (int k = 0; k < 6; ++k) rv2[k] = tail_ld_u64(rec + 2 * k);
                        }
                        my_u[u] = (unsigned)((u == 0 ? rv[3] : rv2[3]) >> 32);   // component 7: d2
                        const double fpos = (double)(my_u[u] - u_lo) * f_scale - (double)sel_bin;   // in [0, 1) up to rounding
                        my_bin[u] = fpos > 0.0 ? min((uint32_t)(fpos * (double)kTailBins), (uint32_t)(kTailBins - 1)) : 0u;
                        atomicAdd(&hist[my_bin[u]], 1u);
                    }
                }
                __syncthreads();
                TAIL_STAMP(7);   // band: keys
                {
                    // exclusive scan of the kTailBins counts, kBpt adjacent bins per thread
                    constexpr int kBpt = kTailBins / kTailThreads;
                    static_assert(kBpt == 1 || kBpt == 2, "one or two bins per thread");
                    const uint32_t h0 = hist[kBpt * t], h1 = kBpt == 2 ? hist[kBpt * t + kBpt - 1] : 0u;
                    hist[kBpt * t] = 0u;   // ready for the next iteration
                    if (kBpt == 2) hist[kBpt * t + 1] = 0u;
                    const uint32_t loc = h0 + h1;
                    uint32_t incl = loc;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
                        if (lane >= o) incl += v;
                    }
                    if (lane == 63) wtot[wave] = incl;
                    __syncthreads();
                    uint32_t base = 0;
#pragma unroll
                    for (int w2 = 0; w2 < kWaves; ++w2) base += w2 < wave ? wtot[w2] : 0u;
                    const uint32_t excl = base + incl - loc, rank = kq - n_below;
                    if (loc && rank >= excl && rank < excl + loc) {   // exactly one thread
                        const bool first = rank < excl + h0;
                        misc[6] = (uint32_t)(kBpt * t + (first ? 0 : 1));
                        misc[7] = first ? rank - excl : rank - excl - h0;
                        misc[8] = first ? h0 : h1;
                    }
                }
                __syncthreads();
                TAIL_STAMP(8);   // band: scan + pick
#pragma unroll
                for (int u = 0; u < kTailPts; ++u)
                    if (my_bin[u] == misc[6]) bk[atomicAdd(&misc[9], 1u)] = ((unsigned long long)my_u[u] << 32) | (unsigned)(t + u * kTailThreads);
                __syncthreads();
                {
                    const uint32_t csel = misc[8], rsel = misc[7];
                    for (uint32_t i = (uint32_t)t; i < csel; i += kTailThreads) {
                        const unsigned long long e = bk[i];
                        uint32_t rr = 0;
                        for (uint32_t j = 0; j < csel; ++j) rr += bk[j] < e ? 1u : 0u;
                        if (rr == rsel) misc[2] = (uint32_t)(e >> 32);
                    }
                }
                __syncthreads();
                TAIL_STAMP(9);   // band: members + rank
                const float limit = __uint_as_float(misc[2]);
                // ---- add the band records that survive the trim: each thread its own (list order x butterfly: the same sum on
                //      every workgroup), straight from the registers the records arrived in
                double v[kSums];
#pragma unroll
                for (int k = 0; k < kSums; ++k) v[k] = 0.0;
#pragma unroll
                for (int u = 0; u < kTailPts; ++u) {
                    const unsigned i = (unsigned)(t + u * kTailThreads);
                    const unsigned long long* r6 = u == 0 ? rv : rv2;
                    const float d2r = __uint_as_float((unsigned)(r6[3] >> 32)), keptr = __uint_as_float((unsigned)r6[4]);
                    if (i < n_band && d2r <= limit && keptr != 0.f) {
                        const float F[7] = {__uint_as_float((unsigned)r6[0]), __uint_as_float((unsigned)(r6[0] >> 32)),
                                            __uint_as_float((unsigned)r6[1]), __uint_as_float((unsigned)(r6[1] >> 32)),
                                            __uint_as_float((unsigned)r6[2]), __uint_as_float((unsigned)(r6[2] >> 32)),
                                            __uint_as_float((unsigned)r6[3])};
                        accumulate(v, F, d2r);
                    }
                }
                wave_reduce32(v);
                if ((lane & 1) == 0) sh[wave][lane >> 1] = v[0];
                __syncthreads();
                if (t < kSums) {
                    double v2 = 0.0;
#pragma unroll
                    for (int w2 = 0; w2 < kWaves; ++w2) v2 += sh[w2][t];
                    tot[t] += v2;
                }
                __syncthreads();
            }
        }
        TAIL_STAMP(10);   // band add
        if (t < kSums) sit->sums[t] = tot[t];
        if (wave == 0) {
            if (kGicp)
                tail_solve_update_gicp(sit, tot, reinterpret_cast<double*>(s_x));
            else
                tail_solve_update(sit, tot, misc, s_x, trim, n_band_raw);
        }
        __syncthreads();
        TAIL_STAMP(11);   // solve, pose update, checkers
        if (sit->done) {
            exit_reason = 1;
            ++k_local;
            break;
        }
        if (k_local + 1 >= cfg.max_iters) {
            exit_reason = 3;
            ++k_local;
            break;
        }
    }
    // ================= exit: this workgroup's outputs; workgroup 0 reports =================
    if (exit_reason == 2 || exit_reason == 4) {
        if (t == 0) {
            sit->stall = 1;
            sit->band_count = 0;
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < kTailPts; ++u) {
        if (!valid[u]) continue;
        // (a stalled iteration is re-run on the select-based path, which rewrites all of this)
        const int sl = t + u * kTailThreads;
        const float4 tq = st_rows[kTailSlots + sl];
        const int cls = (int)(wcls[sl] >> 1);
        float w = (wcls[sl] & 1) ? 1.f : 0.f;
        if (cls == 1 && !(d2s[sl] <= __uint_as_float(misc[2]))) w = 0.f;   // band points: decided by the exact limit
        pos_io[q[u]] = __float_as_int(tq.w);
        d2_out[q[u]] = d2s[sl];
        if (w_out) w_out[q[u]] = w;
        if (hint_g) hint_g[q[u]] = hnt[sl];
    }
    if (exit_reason == 3) {
        // Out of this launch's iteration budget: the next launch continues from the rows as they stand (anchors, bounds, matched
        // points and runner-ups of the kernel's own searches), not from what the last select-based iteration left -- a point
        // re-matched in here would otherwise re-enter with its new position next to its old point's row.
        for (int sl = t; sl < kTailSlots; sl += kTailThreads) {
            const int64_t oc = (int64_t)(sl >> 3) * cfg.wpc + wq;
            const int64_t r_in = oc * 8 + (sl & 7);
            int64_t qq;
            if (cfg.tile > 0) {
                const int64_t tl = oc / cfg.tile, rr = oc - tl * cfg.tile;
                qq = ((tl * 8 + xcls) * cfg.tile + rr) * 8 + (sl & 7);
            } else {
                qq = (int64_t)xcls * cfg.chunk8 + r_in;
            }
            if (r_in < cfg.chunk8 && qq < n) {
#pragma unroll
                for (int r = 0; r < 4; ++r) cache[(size_t)r * (size_t)n + qq] = st_rows[r * kTailSlots + sl];
            }
        }
    }
    if (t == 0 && n_searched) atomicAdd(&sync[kTailSearchedWord], n_searched);
#if O3D_TAIL_STAMPS
    if (t == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
        unsigned long long* out = reinterpret_cast<unsigned long long*>(sync + kTailStampWord) + (blockIdx.x == 0 ? 0 : 12);
        for (int i = 0; i < 12; ++i) out[i] = stamp_acc[i];
    }
#endif
    if (blockIdx.x != 0) return;
    for (int w2 = t; w2 < kTailSyncBytes / 4; w2 += kTailThreads) sync_next[w2] = 0u;
    if (t == 0) sync[kTailItersWord] = (unsigned)k_local;
    if (wave != 0) return;
    if (lane == 0) {
        HostMirror* m = reinterpret_cast<HostMirror*>(mir_w);
        for (int i = 0; i < 16; ++i) m->T[i] = sit->T[i];
        for (int i = 0; i < 16; ++i) m->T_prev[i] = sit->T_prev[i];
        m->iterations = sit->iterations;
        m->done = sit->done;
        m->status = exit_reason == 4 ? (int)REG_DEVICE_ERROR : sit->status;
        m->rank_last = sit->rank_last;
        m->converged = sit->chk.converged ? 1 : 0;
        m->max_iter_reached = sit->chk.max_iter_reached ? 1 : 0;
        m->stall = sit->stall;
        m->band_count = (int)sit->band_count;
        m->limit_last = sit->limit_last;
        m->limit_prev = sit->limit_prev;
        m->band_lo = sit->band_lo;
        m->band_hi = sit->band_hi;
        m->pad_nband = (int)sit->band_count;
        m->pad2 = exit_reason;
        for (int i = 0; i < 6; ++i) {
            m->localizable[i] = sit->xicp_flags[i];
            m->xicp_comb[i] = sit->xicp_comb[i];
            m->xicp_high[i] = sit->xicp_high[i];
        }
        m->n_constraints = sit->xicp_nc;
        m->pad3 = k_local;
        for (int i = 0; i < 8; ++i) m->stamps[i] = 0ull;
        if (sit->stall == 0) sit->band_count = 0;
    }
    if (lane < kSums) reinterpret_cast<HostMirror*>(mir_w)->sums[lane] = sit->sums[lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int w2 = lane; w2 < kStateWords; w2 += 64) reinterpret_cast<uint32_t*>(it_g)[w2] = s_state[w2];
    constexpr int kMirrorWords = (int)(offsetof(HostMirror, seq) / 4);
    uint32_t* hw = reinterpret_cast<uint32_t*>(host);
    for (int w2 = lane; w2 < kMirrorWords; w2 += 64) hw[w2] = mir_w[w2];
    HostMirror::SeqRecord* srec = &host->ring[cfg.seq % kSeqRing];
    {
        const HostMirror* m = reinterpret_cast<const HostMirror*>(mir_w);
        uint32_t word = 0;
        if (lane == 2) word = (uint32_t)m->iterations;
        if (lane == 3) word = (uint32_t)m->done;
        if (lane == 4) word = (uint32_t)m->stall;
        if (lane == 6) word = __float_as_uint(m->limit_last);
        if (lane == 7) word = __float_as_uint(m->limit_prev);
        if (lane >= 2 && lane < 8) reinterpret_cast<uint32_t*>(srec)[lane] = word;
    }
    __threadfence_system();
    if (lane == 0) {
        __hip_atomic_store(&srec->seq, cfg.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host->seq, cfg.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
