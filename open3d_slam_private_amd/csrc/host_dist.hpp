// host_dist.hpp -- multi-GPU entry points (host-synchronous halves, stream-ordered phases), measurement hook, host-only exports
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once
#include <thread>

// Deadline for every wait of the distributed loop: a peer that died, or a collective that never completes, keeps the
// stream busy without raising a fault -- the survivors must not spin forever.
struct Deadline {
    std::chrono::steady_clock::time_point t_end;
    explicit Deadline(double seconds) : t_end(std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(seconds))) {}
    bool expired() const { return std::chrono::steady_clock::now() > t_end; }
};

// hipStreamSynchronize with a deadline: a collective whose peer died never completes and raises no fault.
static reg_status dist_stream_wait(reg_handle* h, double timeout_s, const char* what) {
    const Deadline dl(timeout_s);
    for (unsigned spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(h->stream);
        if (e == hipSuccess) return REG_OK;
        if (e != hipErrorNotReady) {
            h->err = std::string(what) + ": " + hipGetErrorString(e);
            return REG_DEVICE_ERROR;
        }
        if ((spins & 63) == 63) {
            if (dl.expired()) {
                h->err = std::string(what) + ": the stream did not drain within the deadline (O3D_DIST_TIMEOUT_S): a peer died or a "
                                              "collective never completed";
                return REG_DEVICE_ERROR;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        } else {
            __builtin_ia32_pause();
        }
    }
}

extern "C" {

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

// Stream-ordered reading preparation for the multi-GPU path (no host round trip): enqueue this rank's integer centroid
// sums into a device buffer, let the caller all-reduce (sum) those 3 int64 on the same stream, then prepare from them.
reg_status reg_dist_centroid_sums(reg_handle* h, void** sums_dev) {
    reg_status s = check_ready(h, false);
    if (s != REG_OK) return s;
    if (!sums_dev) return REG_BAD_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->prm.device));
    HIPCHK(h, h->s_misc.reserve(256));
    HIPCHK(h, hipMemsetAsync(h->s_misc.p, 0, 3 * sizeof(unsigned long long), h->stream));
    const int blocks = (int)std::min<int64_t>(1024, (h->n + 255) / 256);
    k_centroid_sums<<<blocks, 256, 0, h->stream>>>(h->s_raw.as<float>(), h->s_stride, h->n, h->s_misc.as<unsigned long long>());
    *sums_dev = h->s_misc.p;   // 3 x int64 (fixed point, 2^16 per unit): sum over ranks = sums of the whole reading
    return REG_OK;
}

reg_status reg_dist_prepare(reg_handle* h, const float T_init[16], int64_t n_global) {
    if (!h || !T_init || n_global < 1) return REG_BAD_ARGUMENT;
    float Tr[16];
    col_to_row(T_init, Tr);
    return prepare_rowmajor(h, Tr, nullptr, n_global);
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
    h->xicp_pending = false;   // the distributed path drives the analysis through phases 7-9
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
    const size_t block_bytes = contrib_floats(contrib_cap_for(n_ranks)) * 4;
    HIPCHK(h, h->d_contrib.reserve(block_bytes));
    HIPCHK(h, h->d_gathered.reserve((size_t)n_ranks * block_bytes));
    HIPCHK(h, hipMemsetAsync(h->d_contrib.p, 0, block_bytes, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_gathered.p, 0, (size_t)n_ranks * block_bytes, h->stream));
    h->dist_ranks = n_ranks;
    h->dist_rank = rank;
    *contrib = h->d_contrib.p;
    *gathered = h->d_gathered.p;
    *contrib_bytes = (int64_t)block_bytes;
    return REG_OK;
}

// Select-by-gather variant of the trimmed iteration: instead of three dependent histogram all-reduces, every rank
// all-gathers the squared match distances (n_max floats per rank, +inf padded) and runs the exact 3-level select on
// the gathered array redundantly -- the same kernels on the same multiset of values as the single-GPU path, hence the
// same limit, with ONE collective in front of the sums all-reduce instead of three.
//   phase 10: match            -> all-gather d2_local (n_max floats) into d2_all (n_ranks * n_max floats)
//   phase 11: select on d2_all + linearize of the local slice + partial sums   -> all-reduce the 32 sums -> phase 4
reg_status reg_dist_gather_buffers(reg_handle* h, int n_ranks, int64_t n_max, void** d2_local, void** d2_all) {
    if (!h || !d2_local || !d2_all || n_ranks < 1 || n_ranks > 64) return REG_BAD_ARGUMENT;
    if (h->n == 0) return REG_NOT_CONFIGURED;
    if (n_max < h->n) {
        h->err = "reg_dist_gather_buffers: n_max is smaller than this rank's reading";
        return REG_BAD_ARGUMENT;
    }
    HIPCHK(h, hipSetDevice(h->prm.device));
    if (h->i_d2.cap < (size_t)n_max * 4) {
        if (h->have_match) {
            h->err = "reg_dist_gather_buffers: call it right after reg_set_source (the distance buffer cannot grow once "
                     "matches exist)";
            return REG_BAD_ARGUMENT;
        }
        HIPCHK(h, h->i_d2.reserve((size_t)n_max * 4));
    }
    if (n_max > h->n)   // padding: +inf is "no match" for the select kernels
        HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)(h->i_d2.as<float>() + h->n), 0x7f800000, (size_t)(n_max - h->n), h->stream));
    HIPCHK(h, h->d_d2all.reserve((size_t)n_ranks * n_max * 4));
    h->dist_nmax = n_max;
    h->dist_gather_ranks = n_ranks;
    *d2_local = h->i_d2.p;
    *d2_all = h->d_d2all.p;
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
    const hipError_t qe = hipStreamQuery(h->stream);
    if (qe != hipSuccess && qe != hipErrorNotReady) {
        h->err = std::string("device fault while waiting for an iteration: ") + hipGetErrorString(qe);
        return REG_DEVICE_ERROR;
    }
    out->stream_idle = qe == hipSuccess ? 1 : 0;
    return REG_OK;
}

// The record of ONE specific sequence (1 = the first iteration enqueued after reg_dist_begin).  out->sequences_done is
// set to seq_rel when that update kernel has reported, to 0 when it has not (yet, or ever: update kernels that find the
// loop done or stalled do not report -- out->stream_idle then tells "never").
reg_status reg_dist_record(reg_handle* h, int64_t seq_rel, reg_dist_status* out) {
    if (!h || !out || seq_rel < 1) return REG_BAD_ARGUMENT;
    const unsigned long long want = h->dist_seq0 + (unsigned long long)seq_rel;
    // idle must be sampled BEFORE the record: "idle and no record" then really means the kernel did not report
    const hipError_t qe = hipStreamQuery(h->stream);
    if (qe != hipSuccess && qe != hipErrorNotReady) {   // a device fault must end the caller's polling loop
        h->err = std::string("device fault while waiting for an iteration: ") + hipGetErrorString(qe);
        return REG_DEVICE_ERROR;
    }
    const int idle = qe == hipSuccess ? 1 : 0;
    const HostMirror::SeqRecord* rec = &h->h_mirror->ring[want % kSeqRing];
    const unsigned long long got = __atomic_load_n(&rec->seq, __ATOMIC_ACQUIRE);
    std::memset(out, 0, sizeof(*out));
    out->sequences_enqueued = (int64_t)(h->seq - h->dist_seq0);
    out->stream_idle = idle;
    out->limit_last = out->limit_prev = INFINITY;
    if (got == want) {
        out->sequences_done = seq_rel;
        out->iterations = rec->iterations;
        out->done = rec->done;
        out->stall = rec->stall;
        out->limit_last = rec->limit_last;
        out->limit_prev = rec->limit_prev;
    }
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
                    h->shift0, h->prm.use_xicp ? h->i_w.as<float>() : nullptr, h->i_partials.as<double>(), h->i_cache.as<float4>());
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
                                                       h->prm.cost == REG_COST_P2PL ? st : nullptr, nullptr, 0, 0,
                                                       h->prm.use_xicp ? h->i_xicp.as<XicpState>() : nullptr);
            break;
        case 5: {
            // fused iteration, local half: search + weights + normal equations + band records (into this rank's
            // contribution block), then the block header.  Followed by the caller's ONE all-gather.
            if (h->prm.cost != REG_COST_P2PL || h->dist_ranks <= 0) return REG_BAD_ARGUMENT;
            const FilterCfg f = make_filter_cfg(h, 0);
            uint8_t* hint = h->dbg.match_variant == 2 ? nullptr : h->i_hint.as<uint8_t>();
            float* contrib = h->d_contrib.as<float>();
            if (h->dbg.debug_flags & 16) {
                const int blocks = grid_for(h->n * 8);
                k_iter_fused<8><<<8 * ((blocks + 7) / 8), 256, 0, h->stream>>>(
                    h->s_xyz.as<float4>(), h->has_snrm ? h->s_nrm.as<float4>() : nullptr, h->n, h->i_iter.as<IterState>(),
                    h->grid, h->t_nrm.as<float4>(), f, h->i_pos.as<int>(), h->i_d2.as<float>(), h->i_w.as<float>(), hint,
                    contrib + kContribHdr, contrib_cap_for(h->dist_ranks), h->i_acc.as<double>(), blocks);
            } else {
                const int blocks = grid_for(h->n);
                k_coh_check<<<8 * ((blocks + 7) / 8), 256, 0, h->stream>>>(
                    h->s_xyz.as<float4>(), h->has_snrm ? h->s_nrm.as<float4>() : nullptr, h->n, h->i_iter.as<IterState>(),
                    h->grid, h->t_nrm.as<float4>(), f, h->i_pos.as<int>(), h->i_d2.as<float>(), h->i_w.as<float>(),
                    h->i_cache.as<float4>(), h->i_queue.as<uint32_t>(), coherent_queue_cap(h->n), contrib + kContribHdr,
                    contrib_cap_for(h->dist_ranks), h->i_acc.as<double>(), blocks);
                k_coh_search<8><<<coherent_search_grid(h), 256, 0, h->stream>>>(
                    h->s_xyz.as<float4>(), h->has_snrm ? h->s_nrm.as<float4>() : nullptr, h->n, h->i_iter.as<IterState>(),
                    h->grid, h->t_nrm.as<float4>(), f, h->i_pos.as<int>(), h->i_d2.as<float>(), h->i_w.as<float>(), hint,
                    h->i_cache.as<float4>(), h->i_queue.as<uint32_t>(), coherent_queue_cap(h->n), contrib + kContribHdr,
                    contrib_cap_for(h->dist_ranks), h->i_acc.as<double>(), coherent_slack(h), (CohStats*)nullptr);
            }
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
        case 10:
            s = enqueue_match(h);
            if (s != REG_OK) return s;
            break;
        case 11: {
            if (h->prm.cost != REG_COST_P2PL || h->dist_nmax <= 0) return REG_BAD_ARGUMENT;
            if (trim) {
                const int64_t n_all = (int64_t)h->dist_gather_ranks * h->dist_nmax;
                const float* d2_all = h->d_d2all.as<float>();
                const int gb = (int)std::min<int64_t>(128, (n_all + 255) / 256);
                k_hist_level0<<<gb, 256, 0, h->stream>>>(d2_all, n_all, h->shift0, hist0, it);
                k_select_level<<<gb, 256, 0, h->stream>>>(d2_all, n_all, 1, h->shift0, h->prm.trim_ratio, hist0, hist0 + 2048,
                                                          nullptr, st, it);
                k_select_level<<<gb, 256, 0, h->stream>>>(d2_all, n_all, 2, h->shift0, h->prm.trim_ratio, hist0 + 2048,
                                                          hist0 + 4096, hist0, st, it);
            }
            const FilterCfg f = make_filter_cfg(h, h->prm.use_trimmed ? 2 : 0);
            k_linearize_p2pl<<<h->n_blocks, 256, 0, h->stream>>>(
                h->s_xyz.as<float4>(), h->has_snrm ? h->s_nrm.as<float4>() : nullptr, h->n, it, h->i_pos.as<int>(),
                h->i_d2.as<float>(), h->t_pts.as<float4>(), h->t_nrm.as<float4>(), f, st, hist0 + 4096, hist0 + 2048,
                h->shift0, h->prm.use_xicp ? h->i_w.as<float>() : nullptr, h->i_partials.as<double>(), h->i_cache.as<float4>());
            k_partials_sum<<<1, 1024, 0, h->stream>>>(h->i_partials.as<double>(), h->n_blocks, h->i_sums.as<double>(), it);
            break;
        }
        // R8x on the distributed path, first iteration only: after phase 4 (which then only stashes the eigen-directions)
        //   7: this rank's share of the matched-point centre   -> caller all-reduces the 4 doubles of reg_dist_xicp_buffers
        //   8: this rank's share of the 12 information sums    -> caller all-reduces them
        //   9: decide, solve (constrained or not), update, report -- with the sequence number of phase 4
        case 7:
        case 8: {
            if (!h->prm.use_xicp) return REG_BAD_ARGUMENT;
            const int blocks = (int)std::min<int64_t>(512, (h->n + 255) / 256);
            if (phase == 7)
                k_xicp_center<<<blocks, 256, 0, h->stream>>>(h->s_xyz.as<float4>(), h->n, it, h->i_pos.as<int>(),
                                                             h->i_w.as<float>(), h->i_xicp.as<XicpState>());
            else
                k_xicp_detect<<<blocks, 256, 0, h->stream>>>(h->s_xyz.as<float4>(), h->n, it, h->i_pos.as<int>(),
                                                             h->i_w.as<float>(), h->t_nrm.as<float4>(),
                                                             h->i_xicp.as<XicpState>());
            break;
        }
        case 9:
            if (!h->prm.use_xicp) return REG_BAD_ARGUMENT;
            k_reduce_update<<<1, 1024, 0, h->stream>>>(nullptr, 0, h->i_iter.as<IterState>(), h->d_mirror, h->seq, 2, nullptr,
                                                       nullptr, nullptr, nullptr, 0, 0, h->i_xicp.as<XicpState>());
            break;
        default:
            return REG_BAD_ARGUMENT;
    }
    return REG_OK;
}

reg_status reg_dist_xicp_buffers(reg_handle* h, void** center, void** sums) {
    if (!h || !center || !sums) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    if (!h->prm.use_xicp) {
        h->err = "reg_dist_xicp_buffers: use_xicp is off";
        return REG_NOT_CONFIGURED;
    }
    HIPCHK(h, hipSetDevice(h->prm.device));
    HIPCHK(h, h->i_xicp.reserve(sizeof(XicpState)));
    XicpState* xs = h->i_xicp.as<XicpState>();
    *center = xs->center;   // 4 doubles
    *sums = xs->comb;       // 12 doubles (comb[6] directly followed by high[6])
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
    // (with a deadline when the stream carries collectives: a dead peer must not hang the survivors here either)
    if (h->dist) {
        s = dist_stream_wait(h, h->env.dist_timeout_s, "reg_dist_finish");
        if (s != REG_OK) return s;
    } else {
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
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
    for (int k = 0; k < 6; ++k) {
        res->localizable[k] = h->prm.use_xicp ? mir->localizable[k] : 1;
        res->xicp_combined[k] = mir->xicp_comb[k];
        res->xicp_high[k] = mir->xicp_high[k];
    }
    res->n_constraints = h->prm.use_xicp ? mir->n_constraints : 0;
    fill_result(h, mir->sums, res);
    if (mir->status != REG_OK) {
        h->err = "ErrorMinimizer: no point to minimize";
        return (reg_status)mir->status;
    }
    float T_iter[16], Tout_row[16];
    std::memcpy(T_iter, mir->T, 64);
    compose_rowmajor(h, T_iter, Tout_row);
    row_to_col(T_iter, res->T_iter_last);
    row_to_col(h->h_mirror->T_prev, res->T_iter_prev);
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
