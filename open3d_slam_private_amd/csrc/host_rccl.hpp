// host_rccl.hpp -- multi-GPU registration behind the C ABI: record-driven steering (pure host state machine),
// RCCL transport (resolved lazily from librccl), reg_dist_register.
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
//
// One process per GPU.  The reading is point-partitioned (this rank's slice = what reg_set_source was given), the
// reference and its tables are replicated.  reg_dist_register is the C++ caller-facing form of ICP::compute for that
// layout (the consumer, Mapper::addRangeMeasurement, is C++: Mapper.cpp:343,372-373): it enqueues the library's kernels
// and the collectives on ONE stream, never synchronises inside an iteration, and takes every decision from the record of
// ONE specific sequence -- identical on every rank -- so that all ranks always enqueue the same collectives.
#pragma once

#include <dlfcn.h>
#include <thread>
#include <rccl/rccl.h>   // types and signatures; the functions themselves are resolved with dlopen at reg_dist_init

// =================================================================================================
// steering: a pure host-side state machine (no HIP, no collectives) -- exported for the CPU tests
// =================================================================================================
struct reg_dist_steer {
    // configuration
    int trimming = 1;        // TrimmedDist with ratio != 1: the trimmed limit has to settle before fused iterations
    int fixed = 0;           // fixed iteration count (checkers ignored): the fused tail is submitted in one burst
    int limit = 30;          // fixed_iters, or the Counter checker's maxIterationCount
    int can_fuse = 1;
    float settle_tol = 0.05f;
    // state (mirrors the loop variables of a straight-line driver)
    int64_t enq = 0;         // sequences enqueued since the loop began
    int64_t base_it = 0;     // iterations completed at the last repair point ...
    int64_t base_seq = 0;    // ... and the sequence count at that point
    int generic_left = 2;
    int state = 0;           // 0 top, 1 waiting for record enq-1, 2 decide, 3 waiting for the last record, 4 / 5 drained
    int have_rec = 0;
    reg_dist_reply rec;      // the record of sequence enq-1 of this round
    int n_generic = 0, n_fused = 0, n_stalls = 0;
};

static bool steer_settled(const reg_dist_steer* s) {
    if (!s->trimming) return true;
    if (!s->have_rec) return false;
    const float a = s->rec.limit_last, b = s->rec.limit_prev;
    return std::isfinite(a) && std::isfinite(b) && std::fabs(a - b) <= s->settle_tol * a;
}

// One step: consumes the reply to the previous action (ignored for the first call and after ENQUEUE actions) and
// returns the next action.  Every decision depends only on (configuration, replies) -- and the replies are records
// of specific sequences, identical on every rank.
static reg_dist_action steer_step(reg_dist_steer* s, const reg_dist_reply* reply) {
    reg_dist_action a;
    a.kind = REG_STEER_DONE;
    a.count = 0;
    a.seq = 0;
    for (;;) {
        switch (s->state) {
            case 0: {   // top of the loop
                s->have_rec = 0;
                if (s->enq - s->base_seq >= 2) {
                    s->state = 1;
                    a.kind = REG_STEER_RECORD;
                    a.seq = s->enq - 1;      // the second to last sequence enqueued
                    return a;
                }
                s->state = 2;
                break;
            }
            case 1: {   // reply: record of sequence enq-1
                if (!reply || !reply->available || reply->stall) {
                    // an earlier sequence stalled (band misprediction) or ended the loop: drain, look at the outcome
                    s->state = 4;
                    a.kind = REG_STEER_DRAIN;
                    return a;
                }
                if (reply->done) {
                    s->state = 6;
                    return a;   // DONE
                }
                s->rec = *reply;
                s->have_rec = 1;
                s->state = 2;
                break;
            }
            case 2: {   // decide what to enqueue
                const int64_t planned = s->base_it + (s->enq - s->base_seq);
                if (planned >= s->limit) {
                    if (s->enq > s->base_seq) {
                        s->state = 3;
                        a.kind = REG_STEER_RECORD;
                        a.seq = s->enq;      // everything is enqueued: the last report (or a stall in the burst)
                        return a;
                    }
                    s->state = 6;
                    return a;
                }
                if (!s->can_fuse || s->generic_left > 0 || !steer_settled(s)) {
                    if (s->generic_left > 0) --s->generic_left;
                    s->enq += 1;
                    s->n_generic += 1;
                    s->state = 0;
                    a.kind = REG_STEER_GENERIC;
                    a.count = 1;
                    return a;
                }
                const int64_t burst = s->fixed ? (s->limit - planned) : 1;
                a.kind = REG_STEER_FUSED;
                a.count = (int32_t)std::max<int64_t>(burst, 1);
                s->enq += a.count;
                s->n_fused += a.count;
                s->state = 0;
                return a;
            }
            case 3: {   // reply: record of the last sequence
                if (!reply || !reply->available || reply->stall) {
                    s->state = 5;
                    a.kind = REG_STEER_DRAIN;
                    return a;
                }
                s->state = 6;
                return a;
            }
            case 4:
            case 5: {   // reply: the latest state after the stream drained
                const bool all_done = reply && (reply->done || (s->state == 5 && reply->iterations >= s->limit));
                if (!reply || all_done) {
                    s->state = 6;
                    return a;
                }
                s->base_it = reply->iterations;
                s->base_seq = s->enq;
                s->generic_left = 2;   // repair through two select-based iterations
                s->n_stalls += 1;
                s->state = 0;
                break;
            }
            default:
                return a;   // DONE stays DONE
        }
    }
}

// =================================================================================================
// transport
// =================================================================================================
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        // a process that already carries an RCCL (PyTorch bundles one) keeps using it: RTLD_NOLOAD first
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!lib)
            for (const char* n : names)
                if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) {
            err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?");
            return false;
        }
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        auto GetVersion = (ncclResult_t (*)(int*))sym("ncclGetVersion");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !AllGather || !GetVersion) {
            err = "librccl lacks an expected entry point";
            lib = nullptr;
            return false;
        }
        // The API is compiled against <rccl/rccl.h> but resolved from whatever librccl the process carries (PyTorch bundles
        // its own): ncclUniqueId, the enums and the call signatures are only guaranteed within one major version.
        int ver = 0;
        if (GetVersion(&ver) != ncclSuccess || ver / 10000 != NCCL_MAJOR) {
            err = "librccl in this process reports version " + std::to_string(ver) + ", the library was built against major " +
                  std::to_string((int)NCCL_MAJOR);
            lib = nullptr;
            return false;
        }
        return true;
    }
};
static RcclApi g_rccl;

struct DistCtx {
    int n_ranks = 0, rank = 0;
    bool use_rccl = false;
    ncclComm_t comm = nullptr;
    reg_collectives custom;
    // per-reading bookkeeping
    unsigned long long src_epoch_seen = ~0ull;
    int64_t n_global = 0, n_max = 0;
    DevBuf counts;              // n_ranks x int64: every rank's slice size
    long long* h_count = nullptr;   // pinned: this rank's n (H2D) and the gathered counts (D2H)
    double timeout_s = 30.0;
    float settle_tol = 0.05f;
    reg_dist_steer last_steer;
};

static reg_status dist_fail(reg_handle* h, const std::string& what) {
    h->err = what;
    return REG_DEVICE_ERROR;
}

static reg_status dist_all_reduce(reg_handle* h, void* buf, int64_t count, int dtype) {
    DistCtx* d = h->dist;
    if (d->n_ranks == 1 && !d->use_rccl) return REG_OK;
    if (d->use_rccl) {   // also with ONE rank: the call sequence of the N > 1 loop is then what a one-GPU box exercises
        const ncclDataType_t t = dtype == REG_DT_I32 ? ncclUint32 : (dtype == REG_DT_I64 ? ncclInt64 : ncclFloat64);
        const ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)count, t, ncclSum, d->comm, h->stream);
        if (r != ncclSuccess) return dist_fail(h, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
        return REG_OK;
    }
    if (d->custom.all_reduce_sum(d->custom.ctx, buf, count, dtype, (void*)h->stream) != 0)
        return dist_fail(h, "custom all_reduce_sum failed");
    return REG_OK;
}

static reg_status dist_all_gather(reg_handle* h, const void* send, void* recv, int64_t bytes_per_rank) {
    DistCtx* d = h->dist;
    if (d->n_ranks == 1 && !d->use_rccl) {
        HIPCHK(h, hipMemcpyAsync(recv, send, (size_t)bytes_per_rank, hipMemcpyDeviceToDevice, h->stream));
        return REG_OK;
    }
    if (d->use_rccl) {
        const ncclResult_t r = g_rccl.AllGather(send, recv, (size_t)bytes_per_rank, ncclChar, d->comm, h->stream);
        if (r != ncclSuccess) return dist_fail(h, std::string("ncclAllGather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
        return REG_OK;
    }
    if (d->custom.all_gather(d->custom.ctx, send, recv, bytes_per_rank, (void*)h->stream) != 0)
        return dist_fail(h, "custom all_gather failed");
    return REG_OK;
}

// Blocks until sequence `seq_rel` has reported (reply.available = 1) or can no longer report because the stream drained
// without it (available = 0).
static reg_status dist_wait_record(reg_handle* h, int64_t seq_rel, reg_dist_reply* out) {
    const Deadline dl(h->dist->timeout_s);
    reg_dist_status st;
    for (unsigned spins = 0;; ++spins) {
        reg_status s = reg_dist_record(h, seq_rel, &st);
        if (s != REG_OK) return s;
        if (st.sequences_done != seq_rel && st.stream_idle) {
            s = reg_dist_record(h, seq_rel, &st);   // the report may have landed between the two reads
            if (s != REG_OK) return s;
            if (st.sequences_done != seq_rel) {
                std::memset(out, 0, sizeof(*out));
                out->limit_last = out->limit_prev = INFINITY;
                return REG_OK;
            }
        }
        if (st.sequences_done == seq_rel) {
            out->available = 1;
            out->iterations = st.iterations;
            out->done = st.done;
            out->stall = st.stall;
            out->limit_last = st.limit_last;
            out->limit_prev = st.limit_prev;
            return REG_OK;
        }
        if ((spins & 0xff) == 0xff && dl.expired())
            return dist_fail(h, "timeout waiting for sequence " + std::to_string(seq_rel) +
                                    " of the distributed loop (a peer rank died or a collective never completed)");
    }
}

static reg_status dist_drain(reg_handle* h, reg_dist_reply* out) {
    const Deadline dl(h->dist->timeout_s);
    reg_dist_status st;
    for (unsigned spins = 0;; ++spins) {
        reg_status s = reg_dist_poll(h, &st);
        if (s != REG_OK) return s;
        if (st.stream_idle) break;
        if ((spins & 0xff) == 0xff && dl.expired())
            return dist_fail(h, "timeout draining the stream of the distributed loop (a peer rank died or a collective never completed)");
    }
    reg_status s = reg_dist_poll(h, &st);
    if (s != REG_OK) return s;
    out->available = 1;
    out->iterations = st.iterations;
    out->done = st.done;
    out->stall = st.stall;
    out->limit_last = st.limit_last;
    out->limit_prev = st.limit_prev;
    return REG_OK;
}

__global__ void k_store_i64(long long* dst, long long v) { *dst = v; }

// Slice sizes of all ranks for the current reading: n_global (centroid, fitness) and n_max (gather buffers).
// Collective: every rank calls it for the same registrations (a new reading on ALL ranks, or on none).
static reg_status dist_refresh_counts(reg_handle* h) {
    DistCtx* d = h->dist;
    if (d->src_epoch_seen == h->src_epoch) return REG_OK;
    HIPCHK(h, d->counts.reserve((size_t)(d->n_ranks + 1) * 8));
    long long* dev = d->counts.as<long long>();
    k_store_i64<<<1, 1, 0, h->stream>>>(dev + d->n_ranks, (long long)h->n);
    reg_status s = dist_all_gather(h, dev + d->n_ranks, dev, 8);
    if (s != REG_OK) return s;
    HIPCHK(h, hipMemcpyAsync(d->h_count, dev, (size_t)d->n_ranks * 8, hipMemcpyDeviceToHost, h->stream));
    s = dist_stream_wait(h, d->timeout_s, "slice sizes of the reading (all-gather)");
    if (s != REG_OK) return s;
    d->n_global = 0;
    d->n_max = 0;
    for (int r = 0; r < d->n_ranks; ++r) {
        if (d->h_count[r] <= 0) return dist_fail(h, "rank " + std::to_string(r) + " holds an empty slice of the reading");
        d->n_global += d->h_count[r];
        d->n_max = std::max<int64_t>(d->n_max, d->h_count[r]);
    }
    d->src_epoch_seen = h->src_epoch;
    h->n_total_hint = d->n_global;
    void *a = nullptr, *b = nullptr;
    int64_t nbytes = 0;
    if (h->prm.cost == REG_COST_P2PL) {
        s = reg_dist_fused_buffers(h, d->n_ranks, d->rank, &a, &b, &nbytes);
        if (s != REG_OK) return s;
        if (h->prm.use_trimmed) {
            s = reg_dist_gather_buffers(h, d->n_ranks, d->n_max, &a, &b);
            if (s != REG_OK) return s;
        }
    }
    return REG_OK;
}

extern "C" {

reg_dist_steer* reg_dist_steer_create(int trimming, int fixed_iters, int max_iter, float settle_tol, int can_fuse) {
    reg_dist_steer* s = new reg_dist_steer();
    s->trimming = trimming ? 1 : 0;
    s->fixed = fixed_iters > 0 ? 1 : 0;
    s->limit = fixed_iters > 0 ? fixed_iters : max_iter;
    s->settle_tol = settle_tol;
    s->can_fuse = can_fuse ? 1 : 0;
    s->generic_left = s->trimming ? 2 : 1;
    return s;
}
void reg_dist_steer_destroy(reg_dist_steer* s) { delete s; }
reg_dist_action reg_dist_steer_step(reg_dist_steer* s, const reg_dist_reply* reply) { return steer_step(s, reply); }
void reg_dist_steer_counts(const reg_dist_steer* s, int32_t* n_generic, int32_t* n_fused, int32_t* n_stalls) {
    if (n_generic) *n_generic = s->n_generic;
    if (n_fused) *n_fused = s->n_fused;
    if (n_stalls) *n_stalls = s->n_stalls;
}

reg_status reg_dist_get_unique_id(char id[REG_DIST_ID_BYTES]) {
    if (!id) return REG_BAD_ARGUMENT;
    static_assert(sizeof(ncclUniqueId) <= REG_DIST_ID_BYTES, "id buffer");
    if (!g_rccl.load()) return REG_DEVICE_ERROR;
    ncclUniqueId u;
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return REG_DEVICE_ERROR;
    std::memset(id, 0, REG_DIST_ID_BYTES);
    std::memcpy(id, &u, sizeof(u));
    return REG_OK;
}

static reg_status dist_init_common(reg_handle* h, int rank, int n_ranks, DistCtx** out) {
    if (!h || n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks) return REG_BAD_ARGUMENT;
    if (!h->device_ok) return REG_DEVICE_ERROR;
    if (h->dist) {
        h->err = "reg_dist_init: the handle already belongs to a group (reg_dist_shutdown first)";
        return REG_BAD_ARGUMENT;
    }
    HIPCHK(h, hipSetDevice(h->prm.device));
    DistCtx* d = new DistCtx();
    d->n_ranks = n_ranks;
    d->rank = rank;
    std::memset(&d->custom, 0, sizeof(d->custom));
    d->timeout_s = h->env.dist_timeout_s;
    if (const char* v = getenv("O3D_DIST_SETTLE")) d->settle_tol = (float)atof(v);
    if (hipHostMalloc((void**)&d->h_count, (size_t)(n_ranks + 1) * 8, hipHostMallocDefault) != hipSuccess) {
        delete d;
        return dist_fail(h, "hipHostMalloc failed");
    }
    *out = d;
    return REG_OK;
}

reg_status reg_dist_init(reg_handle* h, const char id[REG_DIST_ID_BYTES], int rank, int n_ranks) {
    if (!id) return REG_BAD_ARGUMENT;
    DistCtx* d = nullptr;
    reg_status s = dist_init_common(h, rank, n_ranks, &d);
    if (s != REG_OK) return s;
    if (!g_rccl.load()) {
        (void)hipHostFree(d->h_count);
        delete d;
        return dist_fail(h, g_rccl.err);
    }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    const ncclResult_t r = g_rccl.CommInitRank(&d->comm, n_ranks, u, rank);
    if (r != ncclSuccess) {
        (void)hipHostFree(d->h_count);
        delete d;
        return dist_fail(h, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
    }
    d->use_rccl = true;
    h->dist = d;
    return REG_OK;
}

reg_status reg_dist_init_custom(reg_handle* h, const reg_collectives* c, int rank, int n_ranks) {
    if (!c || (n_ranks > 1 && (!c->all_reduce_sum || !c->all_gather))) return REG_BAD_ARGUMENT;
    DistCtx* d = nullptr;
    reg_status s = dist_init_common(h, rank, n_ranks, &d);
    if (s != REG_OK) return s;
    d->custom = *c;
    h->dist = d;
    return REG_OK;
}

reg_status reg_dist_shutdown(reg_handle* h) {
    if (!h) return REG_BAD_ARGUMENT;
    DistCtx* d = h->dist;
    if (!d) return REG_OK;
    // (a deadline here too: collectives still enqueued behind a dead peer never complete; the communicator is then abandoned
    //  rather than destroyed -- ncclCommDestroy would wait for them)
    const bool drained = !h->stream || dist_stream_wait(h, d->timeout_s, "reg_dist_shutdown") == REG_OK;
    if (d->use_rccl && d->comm && drained) (void)g_rccl.CommDestroy(d->comm);
    if (drained) {
        d->counts.release();
        if (d->h_count) (void)hipHostFree(d->h_count);
    }   // else: hipFree / hipHostFree wait for the device, i.e. for the stuck stream -- the few bytes are abandoned with it
    delete d;
    h->dist = nullptr;
    h->dist_ranks = 0;
    return REG_OK;
}

// One Gauss-Newton iteration on the select-based path, all ranks: the kernels of reg_dist_phase with the collectives
// in between, all on the handle's stream.
static reg_status dist_enqueue_generic(reg_handle* h, bool* xicp_first) {
    const bool p2pl = h->prm.cost == REG_COST_P2PL;
    const bool trim = p2pl && h->prm.use_trimmed;
    reg_status s;
    if (trim) {
        // select-by-gather: ONE all-gather of the squared match distances, the exact 3-level select redundantly on all
        if ((s = reg_dist_phase(h, 10)) != REG_OK) return s;
        if ((s = dist_all_gather(h, h->i_d2.p, h->d_d2all.p, h->dist->n_max * 4)) != REG_OK) return s;
        if ((s = reg_dist_phase(h, 11)) != REG_OK) return s;
    } else {
        if ((s = reg_dist_phase(h, 0)) != REG_OK) return s;
        if ((s = reg_dist_phase(h, 3)) != REG_OK) return s;
    }
    if ((s = dist_all_reduce(h, h->i_sums.p, kSums, REG_DT_F64)) != REG_OK) return s;
    if ((s = reg_dist_phase(h, 4)) != REG_OK) return s;
    if (*xicp_first) {
        // R8x, first iteration: phase 4 only stashed the eigen-directions; collect the analysis sums over all ranks, then
        // every rank decides, solves and updates identically (phase 9, same sequence number)
        XicpState* xs = h->i_xicp.as<XicpState>();
        if ((s = reg_dist_phase(h, 7)) != REG_OK) return s;
        if ((s = dist_all_reduce(h, xs->center, 4, REG_DT_F64)) != REG_OK) return s;
        if ((s = reg_dist_phase(h, 8)) != REG_OK) return s;
        if ((s = dist_all_reduce(h, xs->comb, 12, REG_DT_F64)) != REG_OK) return s;
        if ((s = reg_dist_phase(h, 9)) != REG_OK) return s;
        *xicp_first = false;
    }
    return REG_OK;
}

static reg_status dist_enqueue_fused(reg_handle* h) {
    reg_status s;
    if ((s = reg_dist_phase(h, 5)) != REG_OK) return s;
    if ((s = dist_all_gather(h, h->d_contrib.p, h->d_gathered.p, (int64_t)(contrib_floats(contrib_cap_for(h->dist_ranks)) * 4))) != REG_OK) return s;
    return reg_dist_phase(h, 6);
}

// == ICP::compute for a reading that is point-partitioned over the ranks of the group.  Collective: every rank calls it
// with the same T_init; every rank returns the same T_out and the GLOBAL result figures.
reg_status reg_dist_register(reg_handle* h, const float T_init[16], float T_out[16], reg_result* res) {
    if (!h || !T_init || !T_out) return REG_BAD_ARGUMENT;
    if (!h->dist) {
        h->err = "reg_dist_register: reg_dist_init has not been called";
        return REG_NOT_CONFIGURED;
    }
    reg_status s = check_ready(h, false);
    if (s != REG_OK) return s;
    DistCtx* d = h->dist;
    HIPCHK(h, hipSetDevice(h->prm.device));
    std::memcpy(T_out, T_init, 64);
    float Ti[16];
    col_to_row(T_init, Ti);
    if ((s = dist_refresh_counts(h)) != REG_OK) return s;
    const bool p2pl = h->prm.cost == REG_COST_P2PL;
    // R2 with the centroid of the WHOLE reading, stream-ordered: integer sums (NC1) -> all-reduce -> prepare
    if (p2pl) {
        void* sums_dev = nullptr;
        if ((s = reg_dist_centroid_sums(h, &sums_dev)) != REG_OK) return s;
        if ((s = dist_all_reduce(h, sums_dev, 3, REG_DT_I64)) != REG_OK) return s;
    }
    if ((s = prepare_rowmajor(h, Ti, nullptr, p2pl ? d->n_global : 0)) != REG_OK) return s;
    const auto t_loop_begin = std::chrono::steady_clock::now();
    if ((s = reg_dist_begin(h, nullptr)) != REG_OK) return s;
    bool xicp_first = h->prm.use_xicp && p2pl;
    const bool trimming = p2pl && h->prm.use_trimmed && h->prm.trim_ratio != 1.0f;
    reg_dist_steer* st = reg_dist_steer_create(trimming ? 1 : 0, h->prm.fixed_iters, h->prm.max_iter, d->settle_tol,
                                               (p2pl && h->dbg.disable_fused != 1) ? 1 : 0);
    reg_dist_reply reply;
    std::memset(&reply, 0, sizeof(reply));
    const reg_dist_reply* rp = nullptr;
    for (;;) {
        const reg_dist_action a = steer_step(st, rp);
        rp = nullptr;
        if (a.kind == REG_STEER_DONE) break;
        if (a.kind == REG_STEER_RECORD) {
            s = dist_wait_record(h, a.seq, &reply);
            rp = &reply;
        } else if (a.kind == REG_STEER_DRAIN) {
            s = dist_drain(h, &reply);
            rp = &reply;
        } else if (a.kind == REG_STEER_GENERIC) {
            s = dist_enqueue_generic(h, &xicp_first);
        } else {
            for (int k = 0; k < a.count && s == REG_OK; ++k) s = dist_enqueue_fused(h);
        }
        if (s != REG_OK) {
            reg_dist_steer_destroy(st);
            return s;
        }
    }
    d->last_steer = *st;
    reg_dist_steer_destroy(st);
    s = reg_dist_finish(h, T_out, res);
    if (res) {
        res->loop_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_loop_begin).count();
        res->fitness = d->n_global > 0 ? (double)res->n_inliers / (double)d->n_global : 0.0;
        res->n_band_stalls = d->last_steer.n_stalls;
    }
    return s;
}

reg_status reg_dist_info(reg_handle* h, int64_t* n_global, int32_t* n_generic, int32_t* n_fused, int32_t* n_stalls) {
    if (!h || !h->dist) return REG_NOT_CONFIGURED;
    if (n_global) *n_global = h->dist->n_global;
    reg_dist_steer_counts(&h->dist->last_steer, n_generic, n_fused, n_stalls);
    return REG_OK;
}

}  // extern "C"
