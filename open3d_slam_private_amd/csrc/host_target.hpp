// host_target.hpp -- handle, parameters, reference-side entry points (table build, target preparation rows, normals), reading upload
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once

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

// Diagnostic / tuning switches from the environment, read once when the handle is created (tools/ scripts set them;
// production code leaves them unset).
struct EnvSwitches {
    bool trace = false;         // O3D_TRACE: host-side timeline of prepare / enqueue / reports on stderr
    bool event_timing = false;  // O3D_EVENT_TIMING: loop_ms from HIP events even when not profiling
    bool halo_occ = true;       // O3D_NO_HALO_OCC: halo-bin edge from the floored bin edge instead of the density-derived one (A/B)
    bool no_dynprune = false;   // O3D_NO_DYNPRUNE: level scans keep the ball they started with (A/B)
    bool no_burst = false;      // O3D_NO_BURST: trickle-feed the fused iterations (A/B of the burst submission)
    bool hints = false;         // O3D_HINTS: histogram of the terminating search level of the last iteration
    bool stamps = false;        // O3D_STAMPS: in-kernel cycle stamps of the update kernel
    bool coh_stats = false;     // O3D_COH_STATS: share of the reading points the coherent fused kernel had to search
    bool tail_always = false;   // O3D_TAIL_ALWAYS: take the persistent tail even while other registrations are in flight on the device (A/B)
    bool no_tail = false;       // O3D_NO_TAIL: three-launch fused iterations instead of the persistent tail kernel (A/B, escape hatch)
    int tail_wpc = 0;           // O3D_TAIL_WPC: cap on the tail kernel's workgroups per XCD class (0: CUs / 8)
    double dist_timeout_s = 30.0; // O3D_DIST_TIMEOUT_S: deadline of every wait of the distributed path
    float tail_timeout_s = 2.f; // O3D_TAIL_TIMEOUT_S: bound of every grid barrier of the tail kernel
    int xcd_tile_first = 32;      // O3D_XCD_TILE_FIRST / _LATER: workgroups per XCD tile of the search kernel, first two launches of a
    int xcd_tile_later = 16;      //   registration / later ones (0: one contiguous eighth of the reading per XCD)
    float settle_trans = 0.02f;   // O3D_SETTLE_TRANS / O3D_SETTLE_ROT: largest last pose step [m] / [rad] at which the band-predicting iterations
    float settle_rot = 4e-3f;     //   (three-launch, tail kernel) may take over from the select-based ones
    int tail_min_iters = 8;       // O3D_TAIL_MIN_ITERS: checker mode (no fixed count): iterations that must have run before the tail kernel may take over
    int tail_tile = 0;            // O3D_TAIL_TILE: octets per XCD tile of the tail kernel's slot mapping (0: contiguous eighths)
    bool no_gicp_tail = false;    // O3D_NO_GICP_TAIL=1: GICP stays on its select-based iteration
    int gicp_tail_after = 1;      // O3D_GICP_TAIL_AFTER: select-based GICP iterations before the tail kernel takes over
    float tail_settle_tol = 1.2f; // O3D_TAIL_SETTLE: relative change of the trimmed limit below which the tail kernel takes over
    int lookahead = 2;          // O3D_KAHEAD
    float settle_tol = 0.05f;   // O3D_SETTLE: relative change of the trimmed limit below which the fused iterations start (round 2 sweep:
                                // 0.05 beats 0.25 by 5 % on C3 -- an early fused iteration has a wide band and moves every point by millimetres)
    float halo_ratio = 1.5f;    // O3D_HALO_RATIO: halo-bin edge in units of the brick-table bin edge (tuning sweeps)
    float halo_rho = 0.4f;      // O3D_HALO_RHO: exactness radius of the halo level in units of the halo-bin edge (round 2 sweep,
                                // profiles/r02_table_sweep.txt: 0.4 beats 0.25 by 2-6 % on C2 / C3 / C4 -- fewer queries fall through
                                // to the level scans, whose latency chain bounds the search kernels)
    float reach_bins = 7.25f;   // O3D_REACH_BINS: the automatic bin edge is at least max_dist / this
    float bin_occupancy = 8.f;  // O3D_BIN_OCC: points per occupied bin the automatic bin edge aims at
    float level_ratio = 2.0f;   // O3D_LEVEL_RATIO: ratio of consecutive search radii (c/2, ... up to max_dist)
    void read() {
        trace = getenv("O3D_TRACE") != nullptr;
        event_timing = getenv("O3D_EVENT_TIMING") != nullptr;
        no_burst = getenv("O3D_NO_BURST") != nullptr;
        no_dynprune = getenv("O3D_NO_DYNPRUNE") != nullptr;
        halo_occ = getenv("O3D_NO_HALO_OCC") == nullptr;
        hints = getenv("O3D_HINTS") != nullptr;
        stamps = getenv("O3D_STAMPS") != nullptr;
        coh_stats = getenv("O3D_COH_STATS") != nullptr;
        tail_always = getenv("O3D_TAIL_ALWAYS") != nullptr;
        no_tail = getenv("O3D_NO_TAIL") != nullptr;
        if (const char* v = getenv("O3D_TAIL_WPC")) tail_wpc = std::max(0, atoi(v));
        if (const char* v = getenv("O3D_DIST_TIMEOUT_S")) dist_timeout_s = std::max(0.5, atof(v));
        if (const char* v = getenv("O3D_TAIL_SETTLE")) tail_settle_tol = (float)atof(v);
        if (const char* v = getenv("O3D_NO_GICP_TAIL")) no_gicp_tail = atoi(v) != 0;
        if (const char* v = getenv("O3D_TAIL_TILE")) tail_tile = std::max(0, atoi(v));
        if (const char* v = getenv("O3D_TAIL_MIN_ITERS")) tail_min_iters = std::max(0, atoi(v));
        if (const char* v = getenv("O3D_SETTLE_TRANS")) settle_trans = (float)atof(v);
        if (const char* v = getenv("O3D_SETTLE_ROT")) settle_rot = (float)atof(v);
        if (const char* v = getenv("O3D_XCD_TILE_FIRST")) xcd_tile_first = std::max(0, atoi(v));
        if (const char* v = getenv("O3D_XCD_TILE_LATER")) xcd_tile_later = std::max(0, atoi(v));
        if (const char* v = getenv("O3D_GICP_TAIL_AFTER")) gicp_tail_after = atoi(v);
        if (const char* v = getenv("O3D_TAIL_TIMEOUT_S")) tail_timeout_s = std::min(30.f, std::max(0.01f, (float)atof(v)));
        if (const char* v = getenv("O3D_KAHEAD")) lookahead = std::max(1, atoi(v));
        if (const char* v = getenv("O3D_SETTLE")) settle_tol = (float)atof(v);
        if (const char* v = getenv("O3D_HALO_RATIO")) halo_ratio = std::min(4.0f, std::max(0.5f, (float)atof(v)));
        if (const char* v = getenv("O3D_HALO_RHO")) halo_rho = std::min(1.0f, std::max(0.05f, (float)atof(v)));
        if (const char* v = getenv("O3D_REACH_BINS")) reach_bins = std::max(1.0f, (float)atof(v));
        if (const char* v = getenv("O3D_BIN_OCC")) bin_occupancy = std::min(64.0f, std::max(1.0f, (float)atof(v)));
        if (const char* v = getenv("O3D_LEVEL_RATIO")) level_ratio = std::min(4.0f, std::max(1.2f, (float)atof(v)));
    }
};

struct DistCtx;   // multi-GPU group state (host_rccl.hpp)

struct reg_handle {
    reg_params prm;
    reg_debug_params dbg;   // experiment switches (include/o3dslam_reg_debug.h); all zero in production
    EnvSwitches env;
    std::string err;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool device_ok = false;   // false: reg_create could not get a HIP device (every entry point then fails loudly)
    bool structure_only = false;   // workspace handle of reg_estimate_normals: bin table only, no attributes
    reg_handle* normals_ws = nullptr;
    DevBuf n_out, n_eig, n_cov, n_ids, n_extra, n_mom;
    DevBuf i_xicp;                 // XicpState (R8x first-iteration analysis)
    DevBuf c_in_xyz, c_in_nrm, c_in_cov, c_flags, c_offs, c_xyz, c_nrm, c_cov, c_idx;   // reg_set_target_f64
    DevBuf r_in_xyz, r_in_nrm, r_in_cov, r_xyz, r_nrm, r_cov;                            // reg_set_source_f64
    int64_t crop_kept = 0;
    DevBuf v_fout, v_oout, v_oxyz, v_onrm, v_ocov;   // reg_voxelize_within_volume
    DevBuf v_ukeys, v_ustart;                        // reg_carve_indices
    DevBuf d_d2all;                                  // select-by-gather (multi-GPU): all ranks' squared distances
    int64_t dist_nmax = 0;
    int dist_gather_ranks = 0;
    bool xicp_pending = false;     // the next generic iteration is followed by the analysis kernels
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_iter = nullptr, ev_s0 = nullptr, ev_s1 = nullptr;
    bool src_prep_pending = false;   // ev_s0 / ev_s1 bracket the last reg_set_source; folded into source_prep_ms lazily
    float source_prep_ms = 0.f;
    int rotation_corrected = 0;
    bool iter_copy_pending = false;

    // target
    int64_t m = 0;
    bool has_tnrm = false, has_tcov = false;
    float c_ref[3] = {0, 0, 0};
    DevBuf t_raw, t_nrm_raw, t_cov_raw, t_centred, t_keys, t_keys2, t_vals, t_vals2, t_pts, t_nrm, t_cov, t_flags,
        t_scan, t_hash, t_cells, t_tmp, t_misc, t_dir, t_rows;
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
    DevBuf t_halo_start, t_halo_cursor, t_halo_pts, i_band, i_acc, i_cache, i_stats, i_queue, i_qcount;
    DevBuf s_prep;
    PrepState* h_prep = nullptr;      // mapped pinned host copy of the device-side preparation state
    PrepState* d_prep_host = nullptr; // device view of h_prep
    int cent_slot = 0;                // which centroid-sum slot of s_misc the next single-GPU registration uses
    bool prep_pending = false;        // h_prep not yet folded into c_read / T0
    DevBuf i_hint, s_keys, s_keys2, s_perm, s_perm2, s_tmp, i_tmpf;
    const uint32_t* perm = nullptr;   // slot -> input index (null: identity)
    int last_stalls = 0;
    int64_t n_total_hint = 0;   // multi-GPU: points of the WHOLE reading (fitness of the GICP stop rule); 0: this handle's n
    int match_launches = 0;   // search launches since the reading was prepared (picks the XCD tile)
    int tail_sync_slot = 0, tail_sync_last = 0;   // which copy of the tail kernel's counter block the next / the last launch uses
    bool tail_sync_dirty = true;
    int last_tail_launches = 0, last_tail_iters = 0;   // persistent tail: launches / iterations of the last reg_register
    DevBuf i_tail_sync, i_tail_rows, i_tail_band;      // persistent tail: counters | per-workgroup sum rows | band records
    unsigned long long dist_seq0 = 0;
    DevBuf d_contrib, d_gathered;     // multi-GPU fused iteration: this rank's block / all ranks' blocks
    int dist_ranks = 0, dist_rank = 0;
    DistCtx* dist = nullptr;          // reg_dist_init: RCCL communicator (or custom transport) of this handle's group
    unsigned long long src_epoch = 0; // counts reg_set_source calls (the distributed path re-reads the slice sizes per reading)
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
    p->use_xicp = 0;            // ICPChainBase::setDefault has no degeneracy awareness
    p->xicp_enough = 250.f;             // icp.yaml:50-55
    p->xicp_insufficient = 180.f;
    p->xicp_min_angle_deg = 80.f;
    p->xicp_strong_angle_deg = 45.f;
    p->gicp_stop_rule = 0;
    p->gicp_rel_fitness = 1e-6f;        // open3d ICPConvergenceCriteria defaults
    p->gicp_rel_rmse = 1e-6f;
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
    p->use_xicp = 1;            // degeneracyAwareness: OptimizedEqualityConstraints (icp.yaml:50-55)
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
    if (p->gicp_stop_rule != 0 && p->gicp_stop_rule != 1) return REG_BAD_ARGUMENT;
    reg_handle* h = new reg_handle();
    h->prm = *p;
    std::memset(&h->dbg, 0, sizeof(h->dbg));
    h->env.read();
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
    (void)hipEventCreate(&h->ev_s0);
    (void)hipEventCreate(&h->ev_s1);
    if (hipHostMalloc((void**)&h->h_mirror, sizeof(HostMirror), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->d_mirror, h->h_mirror, 0) != hipSuccess ||
        hipHostMalloc((void**)&h->h_iter, sizeof(IterState), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_prep, sizeof(PrepState), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->d_prep_host, h->h_prep, 0) != hipSuccess ||
        h->i_iter.reserve(sizeof(IterState)) != hipSuccess || h->s_misc.reserve(256) != hipSuccess ||
        hipMemset(h->s_misc.p, 0, 256) != hipSuccess) {   // centroid-sum slots start cleared (see prepare_rowmajor)
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
    (void)reg_dist_shutdown(h);
    if (h->normals_ws) reg_destroy(h->normals_ws);
    h->n_out.release();
    h->i_xicp.release();
    for (DevBuf* b : {&h->c_in_xyz, &h->c_in_nrm, &h->c_in_cov, &h->c_flags, &h->c_offs, &h->c_xyz, &h->c_nrm, &h->c_cov, &h->c_idx, &h->v_fout, &h->v_oout, &h->v_oxyz, &h->v_onrm, &h->v_ocov, &h->v_ukeys, &h->v_ustart, &h->d_d2all, &h->r_in_xyz, &h->r_in_nrm, &h->r_in_cov, &h->r_xyz, &h->r_nrm, &h->r_cov}) b->release();
    h->n_eig.release();
    h->n_cov.release();
    h->n_ids.release();
    h->n_extra.release();
    h->n_mom.release();
    DevBuf* bufs[] = {&h->t_raw, &h->t_nrm_raw, &h->t_cov_raw, &h->t_centred, &h->t_keys, &h->t_keys2, &h->t_vals,
                      &h->t_vals2, &h->t_pts, &h->t_nrm, &h->t_cov, &h->t_flags, &h->t_scan, &h->t_hash, &h->t_cells,
                      &h->t_tmp, &h->t_misc, &h->t_dir, &h->t_rows, &h->s_raw, &h->s_nrm_raw, &h->s_cov_raw, &h->s_xyz, &h->s_nrm, &h->s_cov,
                      &h->s_misc, &h->i_pos, &h->i_d2, &h->i_w, &h->i_hist, &h->i_state, &h->i_partials, &h->i_sums,
                      &h->i_ids, &h->d_contrib, &h->d_gathered, &h->s_prep, &h->i_iter, &h->t_halo_start, &h->t_halo_cursor, &h->t_halo_pts, &h->i_band, &h->i_acc, &h->i_cache, &h->i_stats, &h->i_queue, &h->i_qcount, &h->i_hint, &h->s_keys, &h->s_keys2, &h->s_perm, &h->s_perm2, &h->s_tmp, &h->i_tmpf, &h->i_tail_sync, &h->i_tail_rows, &h->i_tail_band};
    for (DevBuf* b : bufs) b->release();
    if (h->h_mirror) (void)hipHostFree(h->h_mirror);
    if (h->h_iter) (void)hipHostFree(h->h_iter);
    if (h->h_prep) (void)hipHostFree(h->h_prep);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_iter) (void)hipEventDestroy(h->ev_iter);
    if (h->ev_s0) (void)hipEventDestroy(h->ev_s0);
    if (h->ev_s1) (void)hipEventDestroy(h->ev_s1);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* reg_last_error(const reg_handle* h) { return h ? h->err.c_str() : "null handle"; }

reg_status reg_debug_configure(reg_handle* h, const reg_debug_params* d) {
    if (!h || !d || d->struct_size != (int32_t)sizeof(reg_debug_params)) return REG_BAD_ARGUMENT;
    h->dbg = *d;
    return REG_OK;
}

reg_status reg_set_stream(reg_handle* h, void* hip_stream) {
    if (!h) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    if ((hipStream_t)hip_stream == h->stream) return REG_OK;
    HIPCHK(h, hipSetDevice(h->prm.device));
    // Work already queued on the old stream (the D2D upload and Morton sort of reg_set_source, a table build, an
    // iteration) must have finished before anything is enqueued on the new one: the buffers are shared and there is no
    // other ordering between the two streams.  Switching streams is rare; a full wait is the simple, safe form.
    if (h->stream) HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    h->iter_copy_pending = false;   // the staging copy has completed with the old stream
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
    // The group search needs the dense directory + row masks (12 bytes per brick coordinate): up to 256 M coordinates = 3 GB
    // of a 288 GB device.  Beyond that the bin edge is too small for the extent of the cloud (reg_set_target enlarges an
    // automatic edge and rejects an explicit one).
    if (n_dir > ((size_t)256 << 20)) {
        h->err = "cell_size too small for the target extent (more than 2^28 brick coordinates)";
        return REG_UNSUPPORTED;
    }
    const bool use_dir = true, use_rows = true;
    if (use_dir) {
        HIPCHK(h, h->t_dir.reserve(n_dir * 4));
        HIPCHK(h, hipMemsetAsync(h->t_dir.p, 0xff, n_dir * 4, h->stream));
        if (use_rows) {
            HIPCHK(h, h->t_rows.reserve(n_dir * 8));
            HIPCHK(h, hipMemsetAsync(h->t_rows.p, 0, n_dir * 8, h->stream));
        }
    }
    k_fill_tables<<<grid_for(m), 256, 0, h->stream>>>(h->t_keys2.as<uint64_t>(), h->t_flags.as<uint32_t>(),
                                                       h->t_scan.as<uint32_t>(), m, h->t_hash.as<HashEntry>(), cap - 1,
                                                       h->t_cells.as<uint32_t>(), h->t_misc.as<uint32_t>(),
                                                       use_dir ? h->t_dir.as<int32_t>() : nullptr, bdx, bdy,
                                                       use_rows ? h->t_rows.as<unsigned long long>() : nullptr);
    g.brick_dir = use_dir ? h->t_dir.as<int32_t>() : nullptr;
    g.brick_rows = use_rows ? h->t_rows.as<unsigned long long>() : nullptr;
    g.dyn_prune = h->env.no_dynprune ? 0 : 1;
    g.bdx = bdx;
    g.bdy = bdy;
    g.bdz = bdz;
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
    h->info.table_bytes = (int64_t)((size_t)cap * sizeof(HashEntry) + n_cells * 4 + (use_dir ? n_dir * 4 : 0) + (use_rows ? n_dir * 8 : 0));
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
        rho *= h->env.level_ratio;
        if (std::isinf(md) && rho > 32.f * c) break;  // unbounded search: beyond 32 bin edges fall through to a full scan
    }
    g.rho[n] = md;
    g.rho_box[n] = std::isinf(md) ? INFINITY : md + 1e-3f * md + abs_margin;
    ++n;
    g.n_levels = n;
}

// Level-0 accelerator: dense halo bins of edge c_h = 1.5 c with rho_h = 0.4 c_h (O3D_HALO_RHO; each point is listed in
// 1-2 bins per axis: ~4.5 copies).  Skipped when the dense grid would be too large or on request.
static reg_status build_halo(reg_handle* h, float c, const float bmin[3], const float bmax[3], float max_abs) {
    Grid& g = h->grid;
    g.use_halo = 0;
    g.level_after_halo = 0;
    if (h->dbg.disable_halo == 1) return REG_OK;  // A/B experiments
    const float ch = h->env.halo_ratio * c;
    const float abs_margin = 4e-7f * (1.0f + max_abs);
    const float rho_h = h->env.halo_rho * ch * (1.0f - 4e-3f) - 2.f * abs_margin;
    if (!(rho_h > 0.f)) return REG_OK;
    const float r_ins = rho_h + 1e-3f * rho_h + abs_margin;
    const float inv = 1.0f / ch;
    double dims[3];
    for (int k = 0; k < 3; ++k) dims[k] = std::floor((double)(bmax[k] - bmin[k]) * inv) + 1.0;
    const double nb = dims[0] * dims[1] * dims[2];
    if (nb > 512e6) return REG_OK;   // 2 GB of bin starts: nothing against 288 GB of HBM (a 20 M-point map needs ~30 M bins)
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

static reg_status set_target_impl(reg_handle* h, const float* xyz, int64_t xyz_stride, const float* nrm, int64_t nrm_stride,
                                  const float* cov, int64_t m, int on_device);

reg_status reg_set_target(reg_handle* h, const float* xyz, int64_t xyz_stride, const float* nrm, int64_t nrm_stride,
                          const float* cov, int64_t m, int on_device) {
    if (!h) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    // The reference re-derives everything in compute() after initReference(): a reading prepared against the previous
    // reference (centred on the old c_ref, pre-transformed with the old T0) must not be used with the new one.
    h->prepared = false;
    const reg_status s = set_target_impl(h, xyz, xyz_stride, nrm, nrm_stride, cov, m, on_device);
    if (s != REG_OK) h->m = 0;   // a failed build leaves NO reference (tables, origin and dims may be half-updated)
    return s;
}

static reg_status set_target_impl(reg_handle* h, const float* xyz, int64_t xyz_stride, const float* nrm, int64_t nrm_stride,
                                  const float* cov, int64_t m, int on_device) {
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
    float cs_occ = 0.f;   // automatic edge before the reach floor (0: explicit cell_size)
    if (cs > 0.f) {
        reg_status s = build_grid(h, cs, bmin, bmax, &occupied);
        if (s != REG_OK) return s;
    } else {
        const float ext = std::max(bmax[0] - bmin[0], std::max(bmax[1] - bmin[1], bmax[2] - bmin[2]));
        // start from the edge that would give `occ` points per bin if the cloud were a single ext x ext sheet, then correct
        // with the measured occupancy (points per occupied bin grow with the square of the edge on surface-like clouds)
        const float occ = h->env.bin_occupancy;
        const float cs_abs = std::max(ext / (float)(1u << 20), 1e-6f);
        cs = std::max(std::max(ext * std::sqrt(occ / (float)m), 1e-4f * std::max(ext, 1e-3f)), cs_abs);
        auto build = [&](float& edge) {
            reg_status s = build_grid(h, edge, bmin, bmax, &occupied);
            for (int grow = 0; s == REG_UNSUPPORTED && grow < 8; ++grow) {   // extent / edge^3 beyond the dense directory
                edge *= 1.3f;
                s = build_grid(h, edge, bmin, bmax, &occupied);
            }
            return s;
        };
        for (int pass = 0; pass < 3; ++pass) {
            const reg_status s = build(cs);
            if (s != REG_OK) return s;
            const float per = (float)m / (float)std::max(1u, occupied);
            if ((per <= 1.15f * occ && per >= 0.85f * occ) || pass == 2) break;
            const float next = std::max(cs * std::sqrt(occ / per), cs_abs);
            if (std::fabs(next - cs) < 0.05f * cs) break;
            cs = next;
        }
        // A bin box of the largest radius should not span more than ~15 bins per axis: the early iterations of a registration
        // search with radii up to max_dist, and their cost grows with the number of rows in the box (measured at 20 M points:
        // 0.050 m bins 3.48 ms per registration, 0.069 m bins 3.27 ms).
        const float cs_reach = (std::isfinite(h->prm.max_dist) && !h->structure_only) ? h->prm.max_dist / h->env.reach_bins : 0.f;
        cs_occ = cs;
        if (cs < cs_reach) {
            cs = cs_reach;
            const reg_status s = build(cs);
            if (s != REG_OK) return s;
        }
    }
    // sorted arrays
    HIPCHK(h, h->t_pts.reserve((size_t)m * 16));
    if (d_nrm) HIPCHK(h, h->t_nrm.reserve((size_t)m * 32));   // {point, normal} pairs in sorted order
    if (d_cov) HIPCHK(h, h->t_cov.reserve((size_t)m * 32));
    k_gather_target<<<grid_for(m), 256, 0, h->stream>>>(h->t_centred.as<float4>(), h->t_vals2.as<uint32_t>(), m, d_nrm,
                                                         nrm_stride, d_cov, h->t_pts.as<float4>(),
                                                         d_nrm ? h->t_nrm.as<float4>() : nullptr,
                                                         d_cov ? h->t_cov.as<float4>() : nullptr);
    h->grid.pts = h->t_pts.as<float4>();
    set_levels(h, cs, max_abs);
    {
        // The halo bins serve the settled searches (neighbours a few centimetres away): their edge follows the DENSITY of the
        // map, not the reach floor of the level grid (20 M points: runs of ~170 points with 0.10 m halo bins, ~90 with 0.075 m)
        const float c_halo = (h->env.halo_occ && cs_occ > 0.f) ? std::min(cs, cs_occ) : cs;
        reg_status hs = build_halo(h, c_halo, bmin, bmax, max_abs);
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
    h->m = 0;            // whatever happens below, the previous reference is gone (its staging buffers are reused)
    h->prepared = false;
    if (m <= 0) {
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

reg_status reg_carve_indices(reg_handle* h, const double* map_xyz, const double* map_normals, int64_t m,
                             const double* scan_xyz, int64_t n_scan, int on_device, const double sensor[3],
                             const reg_crop* subset, double voxel_size, double max_ray, double truncation, double min_dot,
                             int32_t* removed, int64_t* n_removed) {
    if (!h) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    if (n_removed) *n_removed = 0;
    if (m < 0 || n_scan < 0 || m > 0x7fffffffLL || n_scan > 0x7fffffffLL || !sensor || !(voxel_size > 0.0) ||
        (m > 0 && (!map_xyz || !removed)) || (n_scan > 0 && !scan_xyz)) {
        if (h) h->err = "reg_carve_indices: bad argument (voxel_size > 0, sensor, arrays)";
        return REG_BAD_ARGUMENT;
    }
    if (m == 0 || n_scan == 0) return REG_OK;
    CropCfg c;
    std::memset(&c, 0, sizeof(c));
    if (subset) {
        if (subset->type < REG_CROP_NONE || subset->type > REG_CROP_CYLINDER) return REG_BAD_ARGUMENT;
        c.type = subset->type;
        c.cx = subset->center[0];
        c.cy = subset->center[1];
        c.cz = subset->center[2];
        c.rmin = subset->radius_min;
        c.rmax = subset->radius_max;
        c.zmin = subset->min_z;
        c.zmax = subset->max_z;
    }
    HIPCHK(h, hipSetDevice(h->prm.device));
    const double *d_map = map_xyz, *d_nrm = map_normals, *d_scan = scan_xyz;
    if (!on_device) {
        HIPCHK(h, h->c_in_xyz.reserve((size_t)m * 24));
        HIPCHK(h, hipMemcpyAsync(h->c_in_xyz.p, map_xyz, (size_t)m * 24, hipMemcpyHostToDevice, h->stream));
        d_map = h->c_in_xyz.as<double>();
        if (map_normals) {
            HIPCHK(h, h->c_in_nrm.reserve((size_t)m * 24));
            HIPCHK(h, hipMemcpyAsync(h->c_in_nrm.p, map_normals, (size_t)m * 24, hipMemcpyHostToDevice, h->stream));
            d_nrm = h->c_in_nrm.as<double>();
        }
        HIPCHK(h, h->c_in_cov.reserve((size_t)n_scan * 24));
        HIPCHK(h, hipMemcpyAsync(h->c_in_cov.p, scan_xyz, (size_t)n_scan * 24, hipMemcpyHostToDevice, h->stream));
        d_scan = h->c_in_cov.as<double>();
    }
    const double inv = 1.0 / voxel_size;
    // 1. candidate map points (inside the subset volume), keyed by voxel, stably sorted
    HIPCHK(h, h->c_flags.reserve((size_t)m * 4));
    HIPCHK(h, h->c_offs.reserve((size_t)m * 4));
    HIPCHK(h, h->v_fout.reserve((size_t)m * 4));
    HIPCHK(h, h->v_oout.reserve((size_t)m * 4));
    HIPCHK(h, h->t_misc.reserve(256));
    HIPCHK(h, hipMemsetAsync(h->t_misc.p, 0, 4, h->stream));
    k_vox_classify<<<grid_for(m), 256, 0, h->stream>>>(d_map, m, c, inv, h->c_flags.as<uint32_t>(), h->v_fout.as<uint32_t>(),
                                                       h->t_misc.as<uint32_t>());
    size_t tb = 0;
    HIPCHK(h, rocprim::exclusive_scan(nullptr, tb, h->c_flags.as<uint32_t>(), h->c_offs.as<uint32_t>(), 0u, (size_t)m,
                                      rocprim::plus<uint32_t>(), h->stream));
    HIPCHK(h, h->t_tmp.reserve(tb));
    HIPCHK(h, rocprim::exclusive_scan(h->t_tmp.p, tb, h->c_flags.as<uint32_t>(), h->c_offs.as<uint32_t>(), 0u, (size_t)m,
                                      rocprim::plus<uint32_t>(), h->stream));
    uint32_t tail[3] = {0, 0, 0};
    HIPCHK(h, hipMemcpyAsync(&tail[0], h->c_offs.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&tail[1], h->c_flags.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&tail[2], h->t_misc.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (tail[2]) {
        h->err = "voxel_size too small for the extent of the map (voxel index exceeds 2^20)";
        return REG_BAD_ARGUMENT;
    }
    const int64_t n_in = (int64_t)tail[0] + tail[1];
    if (n_in == 0) return REG_OK;
    HIPCHK(h, h->t_keys.reserve((size_t)n_in * 8));
    HIPCHK(h, h->t_keys2.reserve((size_t)n_in * 8));
    HIPCHK(h, h->t_vals.reserve((size_t)n_in * 4));
    HIPCHK(h, h->t_vals2.reserve((size_t)n_in * 4));
    k_carve_keys<<<grid_for(m), 256, 0, h->stream>>>(d_map, m, inv, h->c_flags.as<uint32_t>(), h->c_offs.as<uint32_t>(),
                                                     h->t_keys.as<uint64_t>(), h->t_vals.as<uint32_t>());
    size_t sb = 0;
    HIPCHK(h, rocprim::radix_sort_pairs(nullptr, sb, h->t_keys.as<uint64_t>(), h->t_keys2.as<uint64_t>(),
                                        h->t_vals.as<uint32_t>(), h->t_vals2.as<uint32_t>(), (size_t)n_in, 0, 3 * kVoxBits,
                                        h->stream));
    HIPCHK(h, h->t_tmp.reserve(sb));
    HIPCHK(h, rocprim::radix_sort_pairs(h->t_tmp.p, sb, h->t_keys.as<uint64_t>(), h->t_keys2.as<uint64_t>(),
                                        h->t_vals.as<uint32_t>(), h->t_vals2.as<uint32_t>(), (size_t)n_in, 0, 3 * kVoxBits,
                                        h->stream));
    // 2. unique voxels
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
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int64_t nu = (int64_t)lv[0] + lv[1];
    HIPCHK(h, h->v_ukeys.reserve((size_t)nu * 8));
    HIPCHK(h, h->v_ustart.reserve((size_t)(nu + 1) * 4));
    k_carve_unique<<<grid_for(n_in), 256, 0, h->stream>>>(h->t_keys2.as<uint64_t>(), n_in, h->t_flags.as<uint32_t>(),
                                                          h->t_scan.as<uint32_t>(), h->v_ukeys.as<uint64_t>(),
                                                          h->v_ustart.as<uint32_t>());
    const uint32_t n_in32 = (uint32_t)n_in;
    HIPCHK(h, hipMemcpyAsync(h->v_ustart.as<uint32_t>() + nu, &n_in32, 4, hipMemcpyHostToDevice, h->stream));
    // 3. rays -> marks (v_fout / v_oout are reused as mark / offsets over the map)
    HIPCHK(h, hipMemsetAsync(h->v_fout.p, 0, (size_t)m * 4, h->stream));
    k_carve_rays<<<grid_for(n_scan), 256, 0, h->stream>>>(d_scan, n_scan, sensor[0], sensor[1], sensor[2], voxel_size, max_ray,
                                                          truncation, min_dot, inv, h->v_ukeys.as<uint64_t>(),
                                                          h->v_ustart.as<uint32_t>(), nu, h->t_vals2.as<uint32_t>(), d_nrm,
                                                          h->v_fout.as<uint32_t>());
    // 4. ascending list of marked map indices
    HIPCHK(h, rocprim::exclusive_scan(nullptr, tb, h->v_fout.as<uint32_t>(), h->v_oout.as<uint32_t>(), 0u, (size_t)m,
                                      rocprim::plus<uint32_t>(), h->stream));
    HIPCHK(h, h->t_tmp.reserve(tb));
    HIPCHK(h, rocprim::exclusive_scan(h->t_tmp.p, tb, h->v_fout.as<uint32_t>(), h->v_oout.as<uint32_t>(), 0u, (size_t)m,
                                      rocprim::plus<uint32_t>(), h->stream));
    uint32_t rt[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(&rt[0], h->v_oout.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&rt[1], h->v_fout.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int64_t nr = (int64_t)rt[0] + rt[1];
    int32_t* d_out = removed;
    if (!on_device) {
        HIPCHK(h, h->c_idx.reserve((size_t)std::max<int64_t>(nr, 1) * 4));
        d_out = h->c_idx.as<int32_t>();
        h->crop_kept = 0;   // c_idx no longer holds the crop map of reg_set_target_f64
    }
    if (nr > 0) {
        k_carve_collect<<<grid_for(m), 256, 0, h->stream>>>(h->v_fout.as<uint32_t>(), h->v_oout.as<uint32_t>(), m, d_out);
        if (!on_device) HIPCHK(h, hipMemcpyAsync(removed, d_out, (size_t)nr * 4, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (n_removed) *n_removed = nr;
    return REG_OK;
}

reg_status reg_smooth_normals(reg_handle* h, float* normals, const int32_t* ids, int64_t n, int k, int on_device,
                              int32_t* n_passes) {
    if (!h) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    if (!normals || !ids || k < 1 || k > kPcaMaxK || n > 0x7fffffffLL) {
        h->err = "reg_smooth_normals: bad argument (normals, ids != NULL, 1 <= k <= 32)";
        return REG_BAD_ARGUMENT;
    }
    if (n <= 0) {
        h->err = "The point cloud is empty";
        return REG_EMPTY_SOURCE;
    }
    HIPCHK(h, hipSetDevice(h->prm.device));
    float* d_n = normals;
    const int32_t* d_i = ids;
    if (!on_device) {
        HIPCHK(h, h->n_out.reserve((size_t)n * 12));
        HIPCHK(h, h->n_ids.reserve((size_t)n * k * 4));
        HIPCHK(h, hipMemcpyAsync(h->n_out.p, normals, (size_t)n * 12, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->n_ids.p, ids, (size_t)n * k * 4, hipMemcpyHostToDevice, h->stream));
        d_n = h->n_out.as<float>();
        d_i = h->n_ids.as<int32_t>();
    }
    // work arrays: the original normals, the pass in which a point was finished (-1: not yet), the finished count
    HIPCHK(h, h->n_extra.reserve((size_t)n * 12 + (size_t)n * 4 + 64));
    float* orig = h->n_extra.as<float>();
    int* level = reinterpret_cast<int*>(orig + (size_t)n * 3);
    unsigned int* n_done = reinterpret_cast<unsigned int*>(level + n);
    HIPCHK(h, hipMemcpyAsync(orig, d_n, (size_t)n * 12, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(level, 0xff, (size_t)n * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(n_done, 0, 8, h->stream));   // [0] finished points, [1] an id >= n was seen
    int pass = 0;
    unsigned int done = 0, done2[2] = {0, 0};
    const int batch = 32;   // passes between two looks at the count (finished sweeps are cheap no-ops)
    while (done < (unsigned int)n) {
        if ((int64_t)pass > n) {
            h->err = "reg_smooth_normals: the sweep did not terminate (neighbour ids out of range?)";
            return REG_BAD_ARGUMENT;
        }
        for (int b = 0; b < batch; ++b, ++pass)
            k_smooth_pass<<<grid_for(n), 256, 0, h->stream>>>(orig, d_n, d_i, n, k, pass, level, n_done);
        HIPCHK(h, hipMemcpyAsync(done2, n_done, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        done = done2[0];
        if (done2[1]) {
            h->err = "reg_smooth_normals: a neighbour id is >= n";
            return REG_BAD_ARGUMENT;
        }
    }
    if (n_passes) *n_passes = pass;
    if (!on_device) {
        HIPCHK(h, hipMemcpyAsync(normals, d_n, (size_t)n * 12, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    HIPCHK(h, hipGetLastError());
    return REG_OK;
}

reg_status reg_estimate_normals(reg_handle* h, const float* xyz, int64_t xyz_stride, int64_t n, int on_device, int k,
                                float max_dist, const float* viewpoint, int regularise, const reg_normals_out* out,
                                int64_t* n_rescanned) {
    if (!h) return REG_BAD_ARGUMENT;
    if (!out) return REG_BAD_ARGUMENT;
    float *normals = out->normals, *eigvals = out->eigvals, *covs = out->covs;
    int32_t* ids = out->ids;
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
        p.use_xicp = 0;
        reg_handle* w = nullptr;
        const reg_status cs = reg_create(&p, &w);
        if (w) w->dbg.disable_halo = 1;   // the k-NN search uses the brick table only
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
    float *d_n = normals, *d_e = eigvals, *d_c = covs, *d_v = out->eigvecs, *d_d = out->densities, *d_m = out->mean_dists;
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
        if (out->eigvecs || out->densities || out->mean_dists) {
            HIPCHK(h, h->n_extra.reserve((size_t)n * (9 + 1 + 1) * 4));
            float* base = h->n_extra.as<float>();
            if (out->eigvecs) d_v = base;
            if (out->densities) d_d = base + (size_t)n * 9;
            if (out->mean_dists) d_m = base + (size_t)n * 10;
        }
    }
    HIPCHK(h, w->t_misc.reserve(256));
    HIPCHK(h, hipMemsetAsync(w->t_misc.p, 0, 4, h->stream));
    const float vp[3] = {viewpoint ? viewpoint[0] : 0.f, viewpoint ? viewpoint[1] : 0.f, viewpoint ? viewpoint[2] : 0.f};
    const int64_t blocks = (n + (256 / kPcaGroup) - 1) / (256 / kPcaGroup);
    HIPCHK(h, h->n_mom.reserve((size_t)n * sizeof(PcaMoments)));
    if (k <= 12)
        k_knn_pca<128><<<(unsigned)blocks, 256, 0, h->stream>>>(w->grid, d_raw, xyz_stride, n, k, start, d_i,
                                                                w->t_misc.as<uint32_t>(), h->n_mom.as<PcaMoments>());
    else
        k_knn_pca<256><<<(unsigned)blocks, 256, 0, h->stream>>>(w->grid, d_raw, xyz_stride, n, k, start, d_i,
                                                                w->t_misc.as<uint32_t>(), h->n_mom.as<PcaMoments>());
    k_pca_finish<<<grid_for(n), 256, 0, h->stream>>>(h->n_mom.as<PcaMoments>(), n, vp[0], vp[1], vp[2], viewpoint ? 1 : 0,
                                                     regularise, d_n, d_e, d_c, d_v, d_d, d_m);
    uint32_t resc = 0;
    HIPCHK(h, hipMemcpyAsync(&resc, w->t_misc.p, 4, hipMemcpyDeviceToHost, h->stream));
    if (!on_device) {
        HIPCHK(h, hipMemcpyAsync(normals, d_n, (size_t)n * 12, hipMemcpyDeviceToHost, h->stream));
        if (eigvals) HIPCHK(h, hipMemcpyAsync(eigvals, d_e, (size_t)n * 12, hipMemcpyDeviceToHost, h->stream));
        if (covs) HIPCHK(h, hipMemcpyAsync(covs, d_c, (size_t)n * 24, hipMemcpyDeviceToHost, h->stream));
        if (ids) HIPCHK(h, hipMemcpyAsync(ids, d_i, (size_t)n * k * 4, hipMemcpyDeviceToHost, h->stream));
        if (out->eigvecs) HIPCHK(h, hipMemcpyAsync(out->eigvecs, d_v, (size_t)n * 36, hipMemcpyDeviceToHost, h->stream));
        if (out->densities) HIPCHK(h, hipMemcpyAsync(out->densities, d_d, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
        if (out->mean_dists) HIPCHK(h, hipMemcpyAsync(out->mean_dists, d_m, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
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
    HIPCHK(h, hipEventRecord(h->ev_s0, h->stream));
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
    h->n_total_hint = 0;
    ++h->src_epoch;
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
    HIPCHK(h, h->i_cache.reserve((size_t)n * 64));   // per reading point: anchor + bound | matched point | its normal | runner-up
    HIPCHK(h, h->i_queue.reserve(((size_t)((n + 255) / 256 + 63) / 64) * 256 * 64 * 4));   // kQueues sub-queues (coherent_queue_cap)
    if (!h->i_qcount.p) {
        HIPCHK(h, h->i_qcount.reserve(64 * 16 * 4));
        HIPCHK(h, hipMemsetAsync(h->i_qcount.p, 0, 64 * 16 * 4, h->stream));
    }
    if (!h->i_stats.p) {
        HIPCHK(h, h->i_stats.reserve(256));
        HIPCHK(h, hipMemsetAsync(h->i_stats.p, 0, 256, h->stream));
    }
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
    HIPCHK(h, hipEventRecord(h->ev_s1, h->stream));
    h->src_prep_pending = true;
    return REG_OK;
}

// R11, reading side: open3dToPointmatcher (open3d_conversions.cpp:57-118) converts every scan from Open3D's fp64 AoS
// (std::vector<Eigen::Vector3d> points_ / normals_, Matrix3d covariances_) to fp32 before icp_.compute
// (Mapper.cpp:288-289).  The cast runs on the device (round-to-nearest, as static_cast<float>), then reg_set_source.
reg_status reg_set_source_f64(reg_handle* h, const double* xyz, const double* normals, const double* covs, int64_t n,
                              int on_device) {
    if (!h) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    h->n = 0;
    h->prepared = false;
    h->have_match = false;
    if (n <= 0) {
        h->err = "The reading point cloud is empty.";
        return REG_EMPTY_SOURCE;
    }
    if (!xyz || n > 0x7fffffffLL) return REG_BAD_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->prm.device));
    const double *d_xyz = xyz, *d_nrm = normals, *d_cov = covs;
    if (!on_device) {
        HIPCHK(h, h->r_in_xyz.reserve((size_t)n * 24));
        HIPCHK(h, hipMemcpyAsync(h->r_in_xyz.p, xyz, (size_t)n * 24, hipMemcpyHostToDevice, h->stream));
        d_xyz = h->r_in_xyz.as<double>();
        if (normals) {
            HIPCHK(h, h->r_in_nrm.reserve((size_t)n * 24));
            HIPCHK(h, hipMemcpyAsync(h->r_in_nrm.p, normals, (size_t)n * 24, hipMemcpyHostToDevice, h->stream));
            d_nrm = h->r_in_nrm.as<double>();
        }
        if (covs) {
            HIPCHK(h, h->r_in_cov.reserve((size_t)n * 72));
            HIPCHK(h, hipMemcpyAsync(h->r_in_cov.p, covs, (size_t)n * 72, hipMemcpyHostToDevice, h->stream));
            d_cov = h->r_in_cov.as<double>();
        }
    }
    HIPCHK(h, h->r_xyz.reserve((size_t)n * 12));
    if (d_nrm) HIPCHK(h, h->r_nrm.reserve((size_t)n * 12));
    if (d_cov) HIPCHK(h, h->r_cov.reserve((size_t)n * 24));
    k_cast_cloud_f64<<<grid_for(n), 256, 0, h->stream>>>(d_xyz, d_nrm, d_cov, n, h->r_xyz.as<float>(),
                                                          d_nrm ? h->r_nrm.as<float>() : nullptr,
                                                          d_cov ? h->r_cov.as<float>() : nullptr);
    HIPCHK(h, hipGetLastError());
    return reg_set_source(h, h->r_xyz.as<float>(), 3, d_nrm ? h->r_nrm.as<float>() : nullptr, 3,
                          d_cov ? h->r_cov.as<float>() : nullptr, n, 1);
}

}  // extern "C"
