// host_loop.hpp -- reading preparation, iteration state, kernel enqueue helpers, reg_register and the other single-GPU entry points
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once

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
// n_global > 0: the centroid sums in s_misc have already been produced (reg_dist_centroid_sums) and reduced over all
// ranks by the caller; they describe n_global points.
static reg_status prepare_rowmajor(reg_handle* h, const float* T_init_row, const float* c_override = nullptr,
                                   int64_t n_global = 0, const IterState* init_state = nullptr) {
    reg_status s = check_ready(h, false);
    if (s != REG_OK) return s;
    if (!m4_is_finite(T_init_row)) {
        h->err = "initial transformation contains non-finite values";
        return REG_BAD_TRANSFORM;
    }
    const bool ptrace = h->env.trace;
    const auto pt0 = std::chrono::steady_clock::now();
    auto pmark = [&](const char* what) {
        if (ptrace) fprintf(stderr, "[o3dreg] prepare %-18s t=%.1fus\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - pt0).count());
    };
    HIPCHK(h, hipSetDevice(h->prm.device));
    std::memcpy(h->T_init, T_init_row, 64);
    const int64_t n = h->n;
    const bool p2pl = h->prm.cost == REG_COST_P2PL;
    {
        // R3: the rotation block of the pre-transform T0 is that of T_init bit for bit (the two centring matrices are
        // pure translations); the kernel applies the re-orthogonalised copy to the points, this only reports it
        float Tc[16];
        h->rotation_corrected = (p2pl && rigid_correct(T_init_row, Tc)) ? 1 : 0;
    }
    // centroid sums -> (device) centroid + T0; the host copy arrives later through the pinned staging buffer and is
    // only needed for the final composition (R10), so nothing here waits for the device
    HIPCHK(h, h->s_misc.reserve(256));   // allocated and cleared when the handle was created
    HIPCHK(h, h->s_prep.reserve(sizeof(PrepState)));
    // Centroid sums of a single-GPU registration alternate between two slots of s_misc (words 8.. and 16..): the
    // prepare kernel of THIS registration clears the slot the NEXT one accumulates into, so no memset launch is needed
    // (slot 0 belongs to reg_dist_centroid_sums, which clears it itself).
    unsigned long long* sums = h->s_misc.as<unsigned long long>();
    unsigned long long* sums_next = nullptr;
    if (p2pl && !c_override && n_global <= 0) {
        sums = h->s_misc.as<unsigned long long>() + (h->cent_slot ? 16 : 8);
        sums_next = h->s_misc.as<unsigned long long>() + (h->cent_slot ? 8 : 16);
        h->cent_slot ^= 1;
        // (one workgroup per CU: every workgroup ends in three atomics on ONE cache line, ~15 ns each, one after the other)
        const int blocks = (int)std::min<int64_t>(256, (n + 255) / 256);
        k_centroid_sums<<<blocks, 256, 0, h->stream>>>(h->s_raw.as<float>(), h->s_stride, n, sums);
    }
    pmark("centroid");
    PrepArgs pa;
    pa.sums = sums;
    pa.sums_to_clear = sums_next;
    pa.n = n_global > 0 ? n_global : n;
    pa.c_ref = make_float3(h->c_ref[0], h->c_ref[1], h->c_ref[2]);
    std::memcpy(pa.T_init.m, T_init_row, 64);
    pa.centre = p2pl ? 1 : 0;
    pa.use_override = c_override ? 1 : 0;
    pa.c_override = c_override ? make_float3(c_override[0], c_override[1], c_override[2]) : make_float3(0.f, 0.f, 0.f);
    pa.out = h->s_prep.as<PrepState>();
    pa.host_out = h->d_prep_host;
    h->prep_pending = true;
    StateInit si;
    si.dst = nullptr;
    if (init_state) {
        si.init = *init_state;
        si.dst = h->i_iter.as<IterState>();
    } else {
        std::memset(&si.init, 0, sizeof(si.init));
    }
    k_prepare_source<<<grid_for(n), 256, 0, h->stream>>>(
        h->s_raw.as<float>(), h->s_stride, (p2pl && h->has_snrm) ? h->s_nrm_raw.as<float>() : nullptr, h->s_nstride, n, pa,
        p2pl ? 1 : 0, h->perm, h->s_xyz.as<float4>(), (p2pl && h->has_snrm) ? h->s_nrm.as<float4>() : nullptr,
        h->i_hint.as<uint8_t>(), h->i_hist.as<uint32_t>(), h->i_acc.as<double>(), kAccRows * kSums, si,
        h->i_cache.as<float4>());
    if (!p2pl)
        k_pack_cov<<<grid_for(n), 256, 0, h->stream>>>(h->s_cov_raw.as<float>(), n, h->perm, h->s_cov.as<float4>());
    pmark("prepare_source");
    HIPCHK(h, hipGetLastError());
    h->prepared = true;
    h->match_launches = 0;
    h->have_match = false;
    return REG_OK;
}

// ---- iteration state ---------------------------------------------------------------------------------

// (Re)initialise the device-side iteration state: pose T (row-major), mode and checker configuration.
// Host part: fills *st (plain memory).  reg_register hands the result to k_prepare_source as a kernel argument (the
// state then reaches the device with the launch that is enqueued anyway: a pinned-staging hipMemcpyAsync + event
// cost ~20 us of host time in front of the first search kernel); the other entry points copy it (init_iter_state).
static reg_status build_iter_state(reg_handle* h, const float* T_row, int update, IterState* st) {
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
    st->gicp_stop_rule = h->prm.gicp_stop_rule;
    st->gicp_rel_fitness = h->prm.gicp_rel_fitness;
    st->gicp_rel_rmse = h->prm.gicp_rel_rmse;
    st->n_total = (float)(h->n_total_hint > 0 ? h->n_total_hint : h->n);
    st->band_lo = st->band_hi = INFINITY;
    st->limit_last = st->limit_prev = INFINITY;
    st->use_trim = (h->prm.cost == REG_COST_P2PL && h->prm.use_trimmed) ? 1 : 0;
    st->trim_ratio = h->prm.trim_ratio;
    st->band_cap = kBandCap;
    st->qcount = h->i_qcount.as<unsigned int>();
    st->debug_narrow_band = ((h->dbg.debug_flags & 8) ? 1 : 0) | ((h->dbg.debug_flags & 64) ? 2 : 0);   // bit 1: no direct band ranking
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
    return REG_OK;
}

static reg_status init_iter_state(reg_handle* h, const float* T_row, int update) {
    IterState* st = h->h_iter;
    // the pinned staging copy may still be in flight from the previous call: wait for THAT copy only (an event
    // recorded right behind it), not for everything else enqueued on the stream
    if (h->iter_copy_pending) HIPCHK(h, hipEventSynchronize(h->ev_iter));
    reg_status s = build_iter_state(h, T_row, update, st);
    if (s != REG_OK) return s;
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

// Launch with the kernel's own begin / end timestamps when the loop is being profiled (hipExtLaunchKernelGGL attaches
// the two events to the dispatch packet itself: the same interval rocprofv3 --kernel-trace reports, without the
// gaps that events recorded around a launch include).
template <class K, class... A>
static void launch_timed(reg_handle* h, int kind, K kernel, dim3 grid, dim3 block, A... args) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!h->profiling || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        hipLaunchKernelGGL(kernel, grid, block, 0, h->stream, args...);
        return;
    }
    hipExtLaunchKernelGGL(kernel, grid, block, 0, h->stream, e0, e1, 0, args...);
    h->prof_ev.push_back(e0);
    h->prof_ev.push_back(e1);
    h->prof_kind.push_back(kind);
}

static reg_status enqueue_match(reg_handle* h, bool zero_hist = false) {
    const bool trim = h->prm.cost == REG_COST_P2PL && h->prm.use_trimmed;
    if (trim && zero_hist) HIPCHK(h, hipMemsetAsync(h->i_hist.p, 0, 3 * 2048 * 4, h->stream));
    const bool fused_hist = h->dbg.match_variant == 3;
    uint32_t* hist0 = (trim && fused_hist) ? h->i_hist.as<uint32_t>() : nullptr;
    uint32_t* hist2 = trim ? h->i_hist.as<uint32_t>() + 4096 : nullptr;
    const IterState* it = h->i_iter.as<IterState>();
    if (h->dbg.match_variant == 1) {
        prof_mark(h, 0, true);
        k_match<<<h->n_blocks, 256, 0, h->stream>>>(h->s_xyz.as<float4>(), h->n, it, h->grid, h->i_pos.as<int>(),
                                                    h->i_d2.as<float>(), hist0, hist2, h->shift0);
        prof_mark(h, 0, false);
    } else {
        uint8_t* hint = h->dbg.match_variant == 2 ? nullptr : h->i_hint.as<uint8_t>();
        const int lanes = h->dbg.lanes_per_point == 4 ? 4 : (h->dbg.lanes_per_point == 2 ? 2 : 8);
        const int blocks = grid_for(h->n * lanes);
        // first searches of a registration: tiles large enough to keep an XCD's L2 on one region of the reference; later ones (halo
        // runs): finer
        const int tile = h->match_launches < 2 ? h->env.xcd_tile_first : h->env.xcd_tile_later;
        ++h->match_launches;
        const dim3 grid(xcd_tiled_grid(blocks, tile)), block(256);
        auto go = [&](auto kernel) {
            launch_timed(h, 0, kernel, grid, block, (const float4*)h->s_xyz.as<float4>(), h->n, it, h->grid,
                         h->i_pos.as<int>(), h->i_d2.as<float>(), hist0, hist2, hint, h->shift0, h->dbg.debug_flags, blocks,
                         h->i_cache.as<float4>(), tile);
        };
        if (lanes == 4)
            go(k_match_g8<4>);
        else if (lanes == 2)
            go(k_match_g8<2>);
        else
            go(k_match_g8<8>);
    }
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
    if (h->dbg.match_variant != 3)
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
            h->i_partials.as<double>(), h->i_cache.as<float4>());
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
    f.debug = h->dbg.debug_flags;
    f.cos_max_angle = std::cos(h->prm.max_normal_angle);  // cosf in T=float (OutlierFiltersImpl.cpp:229)
    const float md = h->prm.outlier_max_dist;
    f.outlier_max_d2 = md * md;
    return f;
}

// Fused iteration (point-to-plane): search + weights + normal equations in one kernel, band resolution +
// solve + update in the second.  Two launches per Gauss-Newton iteration.
// slack of the candidate-bounded boxes in the coherent kernel's fallback searches: a quarter bin (see nearest_group)
static inline float coherent_slack(const reg_handle* h) { return 0.25f * h->info.cell_size; }
// k_coh_search: a fixed grid that strides over the queue (32 points per workgroup and pass): enough workgroups for the
// usual few per cent of the reading in one pass, never more than the reading needs
static inline int coherent_search_grid(const reg_handle* h) { return (int)std::min<int64_t>(1024, (h->n + 31) / 32); }
// capacity of one of the kQueues sub-queues: workgroup lb (256 points) appends to sub-queue lb % kQueues
static inline int coherent_queue_cap(int64_t n) { return (int)(((n + 255) / 256 + kQueues - 1) / kQueues) * 256; }

template <int G>
static void launch_fused(reg_handle* h, const FilterCfg& f, float* w, uint8_t* hint) {
    int blocks;
    if (h->dbg.debug_flags & 16) {   // A/B switch: the fused kernel without the temporal-coherence shortcut
        blocks = grid_for(h->n * G);
        launch_timed(h, 1, k_iter_fused<G>, dim3(8 * ((blocks + 7) / 8)), dim3(256), (const float4*)h->s_xyz.as<float4>(),
                     (const float4*)(h->has_snrm ? h->s_nrm.as<float4>() : nullptr), h->n, h->i_iter.as<IterState>(), h->grid,
                     (const float4*)h->t_nrm.as<float4>(), f, h->i_pos.as<int>(), h->i_d2.as<float>(), w, hint,
                     h->i_band.as<float>(), (int)kBandCap, h->i_acc.as<double>(), blocks);
    } else {
        blocks = grid_for(h->n);
        launch_timed(h, 1, k_coh_check, dim3(8 * ((blocks + 7) / 8)), dim3(256), (const float4*)h->s_xyz.as<float4>(),
                     (const float4*)(h->has_snrm ? h->s_nrm.as<float4>() : nullptr), h->n, h->i_iter.as<IterState>(), h->grid,
                     (const float4*)h->t_nrm.as<float4>(), f, h->i_pos.as<int>(), h->i_d2.as<float>(), w,
                     (const float4*)h->i_cache.as<float4>(), h->i_queue.as<uint32_t>(), coherent_queue_cap(h->n),
                     h->i_band.as<float>(), (int)kBandCap, h->i_acc.as<double>(), blocks);
        launch_timed(h, 2, k_coh_search<G>, dim3(coherent_search_grid(h)), dim3(256), (const float4*)h->s_xyz.as<float4>(),
                     (const float4*)(h->has_snrm ? h->s_nrm.as<float4>() : nullptr), h->n, h->i_iter.as<IterState>(), h->grid,
                     (const float4*)h->t_nrm.as<float4>(), f, h->i_pos.as<int>(), h->i_d2.as<float>(), w, hint,
                     h->i_cache.as<float4>(), (const uint32_t*)h->i_queue.as<uint32_t>(), coherent_queue_cap(h->n),
                     h->i_band.as<float>(), (int)kBandCap, h->i_acc.as<double>(), coherent_slack(h),
                     h->env.coh_stats ? h->i_stats.as<CohStats>() : (CohStats*)nullptr);
    }
    ++h->seq;
    k_reduce_update<<<1, 1024, 0, h->stream>>>(h->i_acc.as<double>(), blocks, h->i_iter.as<IterState>(), h->d_mirror,
                                               h->seq, 1, h->i_band.as<float>(), w, nullptr, nullptr, 0, 0, nullptr);
}

static reg_status enqueue_fused(reg_handle* h, bool want_w) {
    const FilterCfg f = make_filter_cfg(h, 0);
    float* w = want_w ? h->i_w.as<float>() : nullptr;
    uint8_t* hint = h->dbg.match_variant == 2 ? nullptr : h->i_hint.as<uint8_t>();
    if (h->dbg.lanes_per_point == 4)
        launch_fused<4>(h, f, w, hint);
    else
        launch_fused<8>(h, f, w, hint);
    h->have_match = true;
    return REG_OK;
}

// ---- persistent settled tail (kernels_tail.hpp): one launch for every remaining iteration -----------------------------
// Two persistent kernels on one device could each hold a part of the CUs and wait for the other's rest forever, so at most
// ONE registration per device and process runs its tail this way at a time (try_lock at the start of reg_register: the
// others -- BASELINE config C5, one registration per HIP stream -- keep the three-launch iteration, which fills the gaps).
// Across processes every grid barrier of the kernel is bounded (TailCfg::timeout_ticks) and ends in REG_DEVICE_ERROR.
constexpr int kTailMaxDevices = 64;
static std::mutex g_tail_mutex[kTailMaxDevices];
// Registrations of this process currently inside reg_register, per device.  The persistent kernel takes every CU for the
// whole tail (512 threads x 256 VGPRs + 117 KB of LDS per CU: nothing else is resident beside it), which is what ONE
// registration wants and what a batch of concurrent ones does not: 64 C2 registrations on 8 streams run 57 k iterations/s
// on the three-launch iteration and 50 k when one of them at a time holds the chip (bench.py --mode replicas, round 3).
// So the tail is only taken when no other registration is in flight on the device.
static std::atomic<int> g_reg_active[kTailMaxDevices];
struct RegActiveGuard {
    int dev;
    explicit RegActiveGuard(int d) : dev(d >= 0 && d < kTailMaxDevices ? d : -1) {
        if (dev >= 0) g_reg_active[dev].fetch_add(1, std::memory_order_relaxed);
    }
    ~RegActiveGuard() {
        if (dev >= 0) g_reg_active[dev].fetch_sub(1, std::memory_order_relaxed);
    }
    bool alone() const { return dev >= 0 && g_reg_active[dev].load(std::memory_order_relaxed) == 1; }
};
static int g_tail_cus[kTailMaxDevices];          // 0: not probed yet, < 0: the kernel cannot be co-resident on this device
static std::mutex g_tail_probe_mutex;

struct TailPlan {
    bool ok = false;
    int grid = 0, wpc = 0, chunk8 = 0, tile = 0;
};

// CUs of the device if one k_tail workgroup fits on each (occupancy query), probed once per process and device.
static int tail_device_cus(int dev) {
    if (dev < 0 || dev >= kTailMaxDevices) return -1;
    std::lock_guard<std::mutex> lk(g_tail_probe_mutex);
    if (g_tail_cus[dev] != 0) return g_tail_cus[dev];
    int cus = 0, nb = 0;
    hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int nb2 = 0;
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kTailLdsBytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kTailLdsBytes);
    if (e == hipSuccess)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k_tail<false>), kTailThreads, kTailLdsBytes);
    if (e == hipSuccess)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, reinterpret_cast<const void*>(k_tail<true>), kTailThreads, kTailLdsBytes);
    g_tail_cus[dev] = (e == hipSuccess && nb >= kTailWgsPerCu && nb2 >= kTailWgsPerCu && cus >= 8) ? cus : -1;
    (void)hipGetLastError();
    return g_tail_cus[dev];
}

static TailPlan tail_plan_for(int64_t n, int cus, int wpc_cap, int tile) {
    TailPlan pl;
    if (cus < 8 || n <= 0) return pl;
    int wpc_max = cus * kTailWgsPerCu / 8;
    if (wpc_cap > 0) wpc_max = std::min(wpc_max, wpc_cap);
    pl.wpc = (int)std::max<int64_t>(1, std::min<int64_t>(wpc_max, (n + 2047) / 2048));
    pl.tile = std::max(0, tile);
    const int64_t unit = 8 * (int64_t)std::max(1, pl.tile);   // points per XCD class: whole octets, whole tiles
    pl.chunk8 = (int)((((n + 7) / 8) + unit - 1) / unit * unit);
    const int64_t octets = pl.chunk8 / 8, per_wg = (octets + pl.wpc - 1) / pl.wpc * 8;
    pl.grid = 8 * pl.wpc;
    pl.ok = per_wg <= kTailSlots;
    return pl;
}
static TailPlan tail_plan(const reg_handle* h) {
    return tail_plan_for(h->n, tail_device_cus(h->prm.device), h->env.tail_wpc, h->env.tail_tile);
}

static bool tail_eligible(const reg_handle* h) {
    if (h->prm.cost != REG_COST_P2PL && h->env.no_gicp_tail) return false;
    return h->dbg.disable_fused != 1 && !(h->dbg.debug_flags & (16 | 64 | 128)) && h->dbg.lanes_per_point != 4 && !h->env.no_tail &&
           tail_plan(h).ok;
}

// Size of the last pose update as the host mirror shows it: translation of (T T_prev^-1) and its rotation angle (small-angle: the
// norm of the skew part).  Row-major 4x4, rigid.
static void last_step_motion(const HostMirror* m, float* trans, float* rot) {
    const float* A = m->T;
    const float* B = m->T_prev;
    // D = A * inv(B), inv(B) = [Rb^T, -Rb^T tb]
    float R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = A[4 * i + 0] * B[4 * j + 0] + A[4 * i + 1] * B[4 * j + 1] + A[4 * i + 2] * B[4 * j + 2];
    float t[3];
    for (int i = 0; i < 3; ++i) t[i] = A[4 * i + 3] - (R[3 * i + 0] * B[3] + R[3 * i + 1] * B[7] + R[3 * i + 2] * B[11]);
    *trans = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    const float sx = 0.5f * (R[7] - R[5]), sy = 0.5f * (R[2] - R[6]), sz = 0.5f * (R[3] - R[1]);
    *rot = std::sqrt(sx * sx + sy * sy + sz * sz);
}

// Enqueue the tail for at most `max_iters` iterations; the kernel reports ONCE (sequence h->seq) when it leaves.
static reg_status enqueue_tail(reg_handle* h, const TailPlan& pl, int max_iters, bool want_w) {
    // Two copies of the counter / accumulator block: a launch works on one and its workgroup 0 zeroes the OTHER one when it
    // leaves (nobody uses that one then: the launch before it, in stream order, is over) -- no memset launch (5 us) in front of
    // every tail launch.  Zeroed by the host when first allocated and after any launch that ended in an error.
    if (!h->i_tail_sync.p) h->tail_sync_dirty = true;
    HIPCHK(h, h->i_tail_sync.reserve(2 * (size_t)kTailSyncBytes));
    if (h->tail_sync_dirty) {
        HIPCHK(h, hipMemsetAsync(h->i_tail_sync.p, 0, 2 * (size_t)kTailSyncBytes, h->stream));
        h->tail_sync_dirty = false;
    }
    const int sync_slot = h->tail_sync_slot;
    h->tail_sync_slot ^= 1;
    h->tail_sync_last = sync_slot;
    unsigned* const sync_cur = h->i_tail_sync.as<unsigned>() + (size_t)sync_slot * (kTailSyncBytes / 4);
    unsigned* const sync_next = h->i_tail_sync.as<unsigned>() + (size_t)(sync_slot ^ 1) * (kTailSyncBytes / 4);
    HIPCHK(h, h->i_tail_rows.reserve((size_t)2 * pl.grid * kTailHistRow * 8));
    HIPCHK(h, h->i_tail_band.reserve((size_t)2 * kTailBandCap * kTailRec * 4));
    const FilterCfg f = make_filter_cfg(h, 0);
    TailCfg cfg;
    cfg.n = h->n;
    cfg.chunk8 = pl.chunk8;
    cfg.wpc = pl.wpc;
    cfg.tile = pl.tile;
    cfg.max_iters = std::max(1, std::min(max_iters, kTailMaxIters));
    cfg.slack = coherent_slack(h);
    cfg.seq = ++h->seq;
    cfg.timeout_ticks = (unsigned long long)(h->env.tail_timeout_s * 1e8);
    float* w = want_w ? h->i_w.as<float>() : nullptr;
    uint8_t* hint = h->dbg.match_variant == 2 ? nullptr : h->i_hint.as<uint8_t>();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = h->profiling && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    const bool gicp = h->prm.cost != REG_COST_P2PL;
    auto args = [&](auto launch) {
        // GICP: the covariances of the reading / the reference travel in the two attribute arguments
        const float4* s_attr = gicp ? (const float4*)h->s_cov.as<float4>() : (const float4*)(h->has_snrm ? h->s_nrm.as<float4>() : nullptr);
        const float4* t_attr = gicp ? (const float4*)h->t_cov.as<float4>() : (const float4*)h->t_nrm.as<float4>();
        auto go = [&](auto kernel) {
            launch(kernel, dim3(pl.grid), dim3(kTailThreads), (const float4*)h->s_xyz.as<float4>(), s_attr, h->i_iter.as<IterState>(),
                   h->grid, t_attr, f, h->i_pos.as<int>(), h->i_d2.as<float>(), w, hint, h->i_cache.as<float4>(),
                   sync_cur, h->i_tail_rows.as<double>(), h->i_tail_band.as<float>(), h->d_mirror, cfg, sync_next);
        };
        if (gicp)
            go(k_tail<true>);
        else
            go(k_tail<false>);
    };
    if (timed) {
        args([&](auto k, dim3 g, dim3 b, auto... a) { hipExtLaunchKernelGGL(k, g, b, kTailLdsBytes, h->stream, e0, e1, 0, a...); });
        h->prof_ev.push_back(e0);
        h->prof_ev.push_back(e1);
        h->prof_kind.push_back(3);
    } else {
        if (e0) (void)hipEventDestroy(e0);
        args([&](auto k, dim3 g, dim3 b, auto... a) { hipLaunchKernelGGL(k, g, b, kTailLdsBytes, h->stream, a...); });
    }
    HIPCHK(h, hipGetLastError());
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
        __builtin_ia32_pause();          // the sibling hyper-thread may be another handle's enqueueing thread
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
        // k_prepare_source wrote PrepState into mapped host memory (system-scope fence); it is visible once that kernel has
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
    if (h->src_prep_pending && hipEventQuery(h->ev_s1) == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_s0, h->ev_s1) == hipSuccess) h->source_prep_ms = ms;
        h->src_prep_pending = false;
    }
    res->source_prep_ms = h->source_prep_ms;
    res->rotation_corrected = h->rotation_corrected;
}

extern "C" {

void reg_host_tail_plan(int64_t n, int32_t cus, int32_t tile, int32_t plan[4]) {
    const TailPlan pl = tail_plan_for(n, cus, 0, tile);
    plan[0] = pl.ok ? 1 : 0;
    plan[1] = pl.grid;
    plan[2] = pl.wpc;
    plan[3] = pl.chunk8;
}

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
        if (h->env.trace)
            fprintf(stderr, "[o3dreg] register %-14s t=%.1fus\n", what,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_reg0).count());
    };
    const bool p2pl = h->prm.cost == REG_COST_P2PL;
    float T_start[16];
    if (p2pl)
        m4_identity(T_start);
    else
        std::memcpy(T_start, Ti, 64);
    std::memcpy(h->T_init, Ti, 64);   // the R8x frame change in the state is derived from it (prepare sets it again)
    IterState st0;
    reg_status s = build_iter_state(h, T_start, 1, &st0);
    if (s != REG_OK) return s;
    rmark("iter state");
    s = prepare_rowmajor(h, Ti, nullptr, 0, &st0);   // the state travels as an argument of the prepare kernel
    if (s != REG_OK) return s;
    rmark("prepared");
    h->profiling = h->dbg.profile_loop != 0;
    // loop_ms: HIP events only when profiling (record + synchronise cost ~20 us of host time per registration);
    // otherwise the host clock around the loop -- the loop ends when the last update kernel's mirror has arrived
    const bool event_timing = h->profiling || h->env.event_timing;
    if (event_timing) HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    const auto t_loop_begin = std::chrono::steady_clock::now();
    rmark("ev0");
    const unsigned long long seq0 = h->seq;
    const int fixed = h->prm.fixed_iters;
    // (GICP stop rule 1 re-evaluates the correspondences once more after the last update: one more sequence)
    const int limit = fixed > 0 ? fixed : h->prm.max_iter + ((!p2pl && h->prm.gicp_stop_rule == 1) ? 1 : 0);
    // Iterations 0..kGenericFirst-1 run on the generic (select-based) path: the trimmed limit still moves too
    // much to be predicted.  Afterwards the fused two-kernel iteration is used; if its band prediction fails the
    // device stalls the queue and the host repairs that iteration on the generic path.
    const bool can_fuse = p2pl && h->dbg.disable_fused != 1;
    const bool trimming = p2pl && h->prm.use_trimmed && h->prm.trim_ratio != 1.0f;
    const int kGenericFirst = trimming ? 2 : (p2pl ? 1 : std::max(1, h->env.gicp_tail_after));
    const int kAhead = h->env.lookahead;
    const HostMirror* mir = h->h_mirror;
    int generic_left = kGenericFirst;
    const bool trace = h->env.trace;
    float settle_tol = h->env.settle_tol;
    unsigned long long last_traced = 0;
    const auto t_loop0 = std::chrono::steady_clock::now();
    unsigned long long acked = seq0;   // every sequence <= acked has either reported or been a no-op
    int stalls = 0;
    // The persistent tail (kernels_tail.hpp) replaces the burst of three-launch fused iterations when this registration
    // holds the device's tail lock (one persistent kernel per device at a time: see g_tail_mutex).
    // (GICP has no three-launch fused form: without the tail lock its iterations stay select-based)
    const TailPlan tail_pl = ((can_fuse || !p2pl) && tail_eligible(h)) ? tail_plan(h) : TailPlan();
    const RegActiveGuard active(h->prm.device);
    std::unique_lock<std::mutex> tail_lock;
    if (tail_pl.ok && (active.alone() || h->env.tail_always)) {
        tail_lock = std::unique_lock<std::mutex>(g_tail_mutex[h->prm.device], std::try_to_lock);
    }
    const bool use_tail = tail_pl.ok && tail_lock.owns_lock();
    // The tail kernel copes with a limit that still moves (wide bands cost it a second exchange, not a stall): it takes over as
    // soon as the predicted band (at most +-60 % around the last limit) can be expected to hold the next one
    // ... but only where it can pay: with a fixed iteration count always; with the checkers deciding (the mapper's registrations:
    // 4 - 7 iterations) not before tail_min_iters iterations have run without convergence -- a launch behind the iteration that
    // converges, or for one or two early iterations, costs more than the three-launch iteration it replaces
    // (tools/tools_checker_priors.py: 60 k -> 600 k, 3.6 / 4.1 / 6.9 iterations: +8 / +6 / +13 % with an unconditional tail)
    // (GICP the same: from 3 / 4-iteration registrations a launch after one or three iterations measures 0.108 / 0.135 and
    //  0.115 / 0.139 ms against 0.100 / 0.149 without -- a speculative launch behind the converging iteration costs what it saves)
    const int tail_min_iters = fixed > 0 ? 0 : h->env.tail_min_iters;
    const float fused_settle_tol = settle_tol;
    unsigned long long tail_seq = 0;   // != 0: a tail launch is in flight; nothing is enqueued behind it
    bool tail_off = false;
    h->last_tail_launches = 0;
    h->last_tail_iters = 0;
    for (;;) {
        if (tail_seq) {
            // The tail reports once, when it leaves.  It leaves at once, WITHOUT a report, when a sequence enqueued in front of
            // it ended the loop (checker mode: the last select-based iteration converged while the tail was already queued):
            // that sequence's own report says so, no need to wait for the stream to drain.
            for (unsigned spins = 0;; ++spins) {
                const unsigned long long m = mirror_seq(h);
                if (m >= tail_seq) break;
                if (m + 1 == tail_seq) {
                    // (the RECORD of that sequence, not the mirror's loose fields: the tail's own mirror words -- done = 1 --
                    //  become visible before its sequence word does)
                    const HostMirror::SeqRecord* rec = &mir->ring[(tail_seq - 1) % kSeqRing];
                    if (__atomic_load_n(&rec->seq, __ATOMIC_ACQUIRE) == tail_seq - 1 && (rec->done || rec->stall)) break;
                }
                __builtin_ia32_pause();
                if ((spins & 0xfff) == 0xfff) {
                    const hipError_t e = hipStreamQuery(h->stream);
                    if (e == hipSuccess) break;
                    if (e != hipErrorNotReady) {
                        h->tail_sync_dirty = true;
                        h->err = std::string("device fault while waiting for the tail kernel: ") + hipGetErrorString(e);
                        return REG_DEVICE_ERROR;
                    }
                }
            }
            const bool reported = mirror_seq(h) >= tail_seq;
            unsigned words[4] = {0, 0, 0, 0};
            if (!reported || h->env.coh_stats || mir->status == REG_DEVICE_ERROR)
                (void)hipMemcpy(words, h->i_tail_sync.as<unsigned>() + (size_t)h->tail_sync_last * (kTailSyncBytes / 4) + kTailErrWord, sizeof(words),
                                 hipMemcpyDeviceToHost);
            if (words[0] != 0 || (reported && mir->status == REG_DEVICE_ERROR)) {
                h->tail_sync_dirty = true;
                h->err = "persistent tail kernel: a grid barrier timed out (workgroups not co-resident? another process "
                         "running a persistent kernel on this GPU?); set O3D_NO_TAIL=1 to use the three-launch iteration";
                return REG_DEVICE_ERROR;
            }
            // The stream drained without a report: the tail found the loop done -- or STALLED by a three-launch iteration in front of
            // it (checker mode runs those until tail_min_iters): that stall is still to be repaired by the branch below, so it must not
            // be acknowledged here (the next tail launch would leave at once again, for ever).
            if (!reported && !(mir->stall && std::max(mirror_seq(h), seq0) > acked)) acked = h->seq;
            if (h->env.coh_stats)
                fprintf(stderr, "[o3dreg] tail launch: %u iterations, %u point searches (%.2f %% of the point-iterations), stall cause %u, band records of its last iteration %d\n", words[2],
                        words[1], words[2] ? 100.0 * words[1] / ((double)words[2] * (double)h->n) : 0.0, words[3], reported ? mir->pad_nband : -1);
            if (reported) h->last_tail_iters += mir->pad3;
            // checker mode: a tail launch that stalled (the trimmed limit left even the +-60 % band: the registration is still in
            // its fast phase) is not tried again in this registration -- the repair and the three-launch iterations carry on
            // (far priors at C3 size: 1.15 launches and 0.53 stalls per registration otherwise, tools/tools_checker_priors.py)
            if (reported && mir->stall && fixed <= 0) tail_off = true;
#if O3D_TAIL_STAMPS
            {
                unsigned long long st[24];
                (void)hipMemcpy(st, h->i_tail_sync.as<unsigned>() + (size_t)h->tail_sync_last * (kTailSyncBytes / 4) + kTailStampWord, sizeof(st),
                                 hipMemcpyDeviceToHost);
                const double it_n = std::max(1, reported ? mir->pad3 : 1);
                static const char* names[12] = {"check", "search", "epilogue", "bandrec+comps", "sum+publish+drain", "arrive+wait", "row sums",
                                                "band-stage", "band-scan", "band-rank", "band-add", "solve+update"};
                for (int wg = 0; wg < 2; ++wg) {
                    fprintf(stderr, "[o3dreg] tail stamps (us per iteration, %s workgroup):", wg ? "last" : "first");
                    for (int i = 0; i < 12; ++i) fprintf(stderr, " %s %.2f", names[i], st[12 * wg + i] * 0.01 / it_n);
                    fprintf(stderr, "\n");
                }
            }
#endif
            tail_seq = 0;
            continue;
        }
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
            float mt = 0.f, mr = 0.f;
            last_step_motion(mir, &mt, &mr);
            fprintf(stderr, "[o3dreg] seq %llu iter %d stall %d band_n %d limit %.6g prev %.6g band [%.6g, %.6g) step %.2e m %.2e rad\n",
                    m_seq - seq0, mir->iterations, mir->stall, mir->stall ? mir->band_count : mir->pad_nband, mir->limit_last, mir->limit_prev,
                    mir->band_lo, mir->band_hi, mt, mr);
        }
        acked = std::max(acked, m_seq);
        const int completed = any ? mir->iterations : 0;
        const int inflight = (int)(h->seq - acked);
        if (completed + inflight < limit && inflight < kAhead) {
            // fuse once the trimmed limit has settled (last two limits the host has seen within settle_tol, 5 %): the first
            // fused iteration is the expensive one -- its band is as wide as the limit still moves (wide band -> histogram
            // select in the update kernel, many coherence failures) -- so starting too early costs more than another
            // select-based iteration (round-2 sweep, DESIGN.md 6.0: 25 % -> 5 %: C3 1.653 -> 1.574 ms)
            const bool tail_now = use_tail && !tail_off && completed + inflight >= tail_min_iters;
            settle_tol = tail_now ? h->env.tail_settle_tol : fused_settle_tol;
            bool settled = true;
            if (trimming) {
                settled = any && mir->limit_prev < INFINITY && mir->limit_last < INFINITY &&
                          std::fabs(mir->limit_last - mir->limit_prev) <= settle_tol * mir->limit_last;
            }
            // ... and the pose must have stopped jumping: a registration from a far prior sits on a plateau of the trimmed limit (most
            // pairs at the matching radius) while it still turns by degrees per iteration, then the limit collapses 10 - 20 x within
            // one iteration -- a band predicted on the plateau stalls there, and a tail launch entered there searches every point
            // first (O3D_TRACE: 2.5e-2 rad steps at a limit moving by 1 %; the benchmark's registration enters at 6e-4 rad / 4 mm)
            if (settled && any && (p2pl || fixed <= 0) && (can_fuse || tail_now)) {
                float mt = 0.f, mr = 0.f;
                last_step_motion(mir, &mt, &mr);
                if (mt > h->env.settle_trans || mr > h->env.settle_rot) settled = false;
            }
            const auto tq0 = std::chrono::steady_clock::now();
            const bool go_generic = !(can_fuse || tail_now) || generic_left > 0 || !settled;
            if (go_generic) {
                s = enqueue_iteration(h, true);   // weights are always written: reg_get_correspondences reports them
                if (generic_left > 0) --generic_left;
            } else if (tail_now) {
                // the rest of the registration in ONE launch (it leaves when done, stalled, or out of its iteration budget)
                s = enqueue_tail(h, tail_pl, limit - (completed + inflight), true);
                tail_seq = h->seq;
                ++h->last_tail_launches;
            } else {
                // Fixed iteration count: nothing the host could learn changes what has to run, so the whole rest of
                // the registration is submitted in one go (a failed band prediction turns what follows into no-ops
                // and is repaired above).  Submitting while the device crosses a kernel boundary costs about 6 us per
                // iteration (measured: rocprofv3 timeline, profiles/), hence no trickle-feeding here.
                int burst = fixed > 0 && !h->env.no_burst ? limit - (completed + inflight) : 1;
                for (; burst > 0 && s == REG_OK; --burst) s = enqueue_fused(h, true);
            }
            if (trace) {
                const auto tq1 = std::chrono::steady_clock::now();
                fprintf(stderr, "[o3dreg] t=%.1fus enqueue seq %llu (%s) took %.1fus; mirror at %llu\n",
                        std::chrono::duration<double, std::micro>(tq0 - t_loop0).count(), h->seq - seq0,
                        go_generic ? "generic" : (tail_now ? "tail" : "fused"), std::chrono::duration<double, std::micro>(tq1 - tq0).count(),
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
    if (h->env.hints) {
        // diagnostics: at which radius level did the searches of the LAST iteration end (0 = halo, l + 1 = level l)
        std::vector<uint8_t> hv((size_t)h->n);
        if (hipMemcpy(hv.data(), h->i_hint.p, (size_t)h->n, hipMemcpyDeviceToHost) == hipSuccess) {
            long long cnt[18] = {0};
            for (uint8_t v : hv) ++cnt[v < 17 ? v : 17];
            fprintf(stderr, "[o3dreg] terminating level of the last iteration (halo, then levels with rho =");
            for (int l = 0; l < h->grid.n_levels; ++l) fprintf(stderr, " %.3f", h->grid.rho[l]);
            fprintf(stderr, "):");
            for (int l = 0; l <= h->grid.n_levels; ++l) fprintf(stderr, " %lld", cnt[l]);
            fprintf(stderr, "\n");
        }
    }
    if (h->env.coh_stats) {
        CohStats cs;
        if (hipMemcpy(&cs, h->i_stats.p, sizeof(cs), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[o3dreg] coherent fused iterations: %llu point-iterations, %llu searched (%.2f %%)\n", cs.n_points,
                    cs.n_searched, cs.n_points ? 100.0 * (double)cs.n_searched / (double)cs.n_points : 0.0);
            (void)hipMemset(h->i_stats.p, 0, sizeof(cs));
        }
    }
    if (h->env.stamps) {
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
    row_to_col(mir->T_prev, res->T_iter_prev);
    row_to_col(Tout_row, T_out);
    res->n_band_stalls = h->last_stalls;
    res->n_tail_launches = h->last_tail_launches;
    res->n_tail_iterations = h->last_tail_iters;
    rmark("return");
    return REG_OK;
}

reg_status reg_information_matrix(reg_handle* h, const float T[16], float max_dist, double info[36], int64_t* n_pairs) {
    if (!h || !T || !info) return REG_BAD_ARGUMENT;
    if (n_pairs) *n_pairs = 0;
    if (!(max_dist > 0.f) || max_dist > h->prm.max_dist) {
        h->err = "reg_information_matrix: max_dist must be positive and within the handle's max_dist";
        return REG_BAD_ARGUMENT;
    }
    float Tr[16];
    col_to_row(T, Tr);
    reg_status s = prepare_rowmajor(h, Tr);
    if (s != REG_OK) return s;
    const bool p2pl = h->prm.cost == REG_COST_P2PL;
    float T_start[16];
    if (p2pl)
        m4_identity(T_start);
    else
        std::memcpy(T_start, Tr, 64);
    s = init_iter_state(h, T_start, 0);
    if (s != REG_OK) return s;
    s = enqueue_match(h, /*zero_hist=*/false);
    if (s != REG_OK) return s;
    h->have_match = true;
    HIPCHK(h, h->i_sums.reserve(kSums * 8));
    HIPCHK(h, hipMemsetAsync(h->i_sums.p, 0, 10 * sizeof(double), h->stream));
    const int blocks = (int)std::min<int64_t>(512, (h->n + 255) / 256);
    k_info_sums<<<blocks, 256, 0, h->stream>>>(h->i_pos.as<int>(), h->i_d2.as<float>(), h->n, h->t_pts.as<float4>(),
                                               h->c_ref[0], h->c_ref[1], h->c_ref[2], max_dist * max_dist,
                                               h->i_sums.as<double>());
    double m[10];
    HIPCHK(h, hipMemcpyAsync(m, h->i_sums.p, sizeof(m), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    // sum over pairs of G^T G with rows [0 z -y 1 0 0], [-z 0 x 0 1 0], [y -x 0 0 0 1]
    const double c = m[0], sx = m[1], sy = m[2], sz = m[3], xx = m[4], yy = m[5], zz = m[6], xy = m[7], xz = m[8], yz = m[9];
    const double I[36] = {yy + zz, -xy, -xz, 0, -sz, sy,
                          -xy, xx + zz, -yz, sz, 0, -sx,
                          -xz, -yz, xx + yy, -sy, sx, 0,
                          0, sz, -sy, c, 0, 0,
                          -sz, 0, sx, 0, c, 0,
                          sy, -sx, 0, 0, 0, c};
    std::memcpy(info, I, sizeof(I));
    if (n_pairs) *n_pairs = (int64_t)llround(c);
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

}  // extern "C"
