// reg_state.hpp -- device-resident iteration state, host mirror, small shared helpers
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once

// Iteration state living in device memory: the pose the kernels read, the checker history and the
// termination flags.  The update kernel (last kernel of an iteration) is its only writer, so a whole
// registration can be enqueued without a host round trip per Gauss-Newton iteration.
struct IterState {
    float T[16];          // T_iter, row-major (P2PL: centred frames; GICP: reading -> reference)
    float T_prev[16];     // the pose the last completed iteration RAN at (its matches, weights and sums belong to it)
    double Td[16];        // GICP: the same in double
    Checkers chk;         // DifferentialTransformationChecker / CounterTransformationChecker state
    int iterations;
    int done;             // 1: the remaining enqueued kernels return immediately
    int status;           // reg_status of the loop (REG_OK / REG_NO_CORRESPONDENCES)
    int rank_last;
    int cost;
    int fixed_iters;
    int max_iter;
    int update;           // 0: reduce only (reg_linearize / distributed halves), 1: solve + update + check
    float gicp_rot_eps, gicp_trans_eps;
    int gicp_stop_rule;       // 1: Open3D ICPConvergenceCriteria (relative fitness / rmse between consecutive evaluations)
    float gicp_rel_fitness, gicp_rel_rmse;
    float n_total;            // reading points (all ranks): fitness = matched / n_total
    double fit_prev, rmse_prev;
    double sums[kSums];
    // fused path (k_iter_fused): predicted band [band_lo, band_hi) around the trimmed-quantile limit
    float band_lo, band_hi;   // +inf / +inf: no trimming (every finite match is inside)
    float trim_ratio;
    int use_trim;             // 1: TrimmedDistOutlierFilter active
    int stall;                // 1: the band prediction failed; enqueued fused kernels return until the host repairs
    float limit_last;         // trimmed limit of the last completed iteration (+inf: none)
    float limit_prev;         // ... and of the one before
    unsigned int band_count;  // records appended to the band buffer in this iteration
    // coherent fused iteration: kQueues counters (one per 64 bytes: same-address returning atomics serialise at ~11 ns
    // each, 800 workgroups on ONE word cost ~9 us) of the reading points whose shortcut failed, queued for k_coh_search;
    // the update kernel clears them
    unsigned int* qcount;
    unsigned int band_cap;
    int debug_narrow_band;    // test hooks: bit 0 forces band mispredictions, bit 1 disables the direct band ranking
    // R8x (X-ICP localizability, OptimizedEqualityConstraints)
    int xicp_stage;           // 0: off / analysed, 1: analysis pending (first iteration), 2: sums being collected
    int xicp_nc;              // number of non-localizable directions (constraints)
    int xicp_flags[6];        // 1 = localizable; rotation eigen-directions 0-2, translation 3-5
    float xicp_enough, xicp_insufficient, xicp_cos_min, xicp_cos_strong;
    float xicp_Trd[12];       // T_refMean_dataIn (row-major 3x4): its inverse takes the matched data to the frame it came from
    double xicp_comb[6], xicp_high[6];   // the information sums of the analysis (reported with every mirror)
};

// Scratch of the first-iteration localizability analysis.
struct XicpState {
    float vr[9], vt[9];       // eigenvectors in the data frame, [k*3 + r]
    int pad[2];
    double center[4];         // sum of the matched reading points (data frame) + their count
    double comb[6], high[6];  // information sums: rotation 0-2, translation 3-5
};

// What the update kernel mirrors into mapped host memory (the host polls `seq`).
struct HostMirror {
    double sums[kSums];
    float T[16];
    float T_prev[16];
    int iterations, done, status, rank_last, converged, max_iter_reached, stall, band_count;
    float limit_last, limit_prev, band_lo, band_hi;
    int pad_nband, pad2;
    int localizable[6];
    int n_constraints, pad3;
    double xicp_comb[6], xicp_high[6];
    unsigned long long stamps[8];   // s_memtime stamps of the update kernel (diagnostics only; nothing reads them)
    unsigned long long seq;
    // Per-sequence records (ring of kSeqRing): what the update kernel with sequence number s reported.  The multi-GPU
    // drivers steer ONLY by the record of a specific sequence -- identical on every rank -- never by "whatever has
    // arrived so far", which depends on timing and would let ranks enqueue different collectives.
    struct SeqRecord {
        unsigned long long seq;
        int iterations, done, stall, pad;
        float limit_last, limit_prev;
    } ring[16];
};
constexpr int kSeqRing = 16;
constexpr int kQueues = 64;        // sub-queues of the coherent iteration's search queue (workgroup lb appends to lb % kQueues)
constexpr int kQueueStride = 16;   // counters are 16 words (64 bytes) apart

// XCD-aware workgroup order: the dispatcher deals workgroups round-robin over the 8 XCDs (blockIdx % 8
// shares an XCD).  With a Morton-ordered reading, giving each XCD ONE contiguous eighth of the reading means
// its private 4 MB L2 only has to hold that region's slice of the reference cloud and tables.
// Launch with gridDim.x = 8 * ceil(n_blocks / 8); returns the logical block (>= n_blocks: nothing to do).
__device__ __forceinline__ int xcd_block(int n_blocks) {
    const int chunk = (n_blocks + 7) >> 3;
    return (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
}

// ... in tiles: XCD x takes tiles x, x + 8, x + 16, ... of `tile` consecutive workgroups each.  One contiguous eighth per XCD
// ties the time of a launch to its slowest eighth: the first searches of a registration cost most where the initial guess is
// farthest off (rotation error x range), and that is ONE corner of the Morton-ordered reading (C3, second search launch:
// 160 us with eighths, 117 us in tiles of 16).  Launch with xcd_tiled_grid(n_blocks, tile) workgroups; tile <= 0: eighths.
__host__ __device__ __forceinline__ int xcd_tiled_grid(int n_blocks, int tile) {
    if (tile <= 0) return 8 * ((n_blocks + 7) / 8);
    const int per = 8 * tile;
    return per * ((n_blocks + per - 1) / per);
}
__device__ __forceinline__ int xcd_block_tiled(int n_blocks, int tile) {
    if (tile <= 0) return xcd_block(n_blocks);
    const int xcd = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
    const int t = j / tile, r = j - t * tile;
    return (t * 8 + xcd) * tile + r;
}

struct Xf4 {
    float m[16];
};

__device__ __forceinline__ Xf load_xf(const IterState* it) {
    Xf x;
#pragma unroll
    for (int k = 0; k < 12; ++k) x.m[k] = it->T[k];
    return x;
}
