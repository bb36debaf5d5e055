/*
 * o3dslam_reg.h -- C ABI of the MI355X-native scan-to-map registration path.
 *
 * Drop-in boundary for the ONE hot path of leggedrobotics/open3d_slam_private:
 *   B2  PointMatcher<float>::ICP::initReference / compute
 *         libpointmatcher/pointmatcher/PointMatcher.h:1036-1042, ICP.cpp:813-898,902-1349
 *         called from open3d_slam/src/Mapper.cpp:343,372-373
 *   B1  o3d_slam::CloudRegistration::registerClouds (GICP operator)
 *         open3d_slam/include/open3d_slam/CloudRegistration.hpp:19-73, src/CloudRegistration.cpp:16-21
 * (paths relative to the reference tree).  Plain pointers and sizes only; no C++,
 * torch or Eigen types cross this boundary; no exception crosses it -- every throw
 * site of the reference maps to a reg_status code.
 *
 * LAYOUT CONTRACT
 *  - points:  `xyz` + `xyz_stride` (in floats).  stride 4 == libpointmatcher's
 *    DataPoints::features (Eigen column-major (dim+1) x N float => {x,y,z,1} per point,
 *    PointMatcher.h:176,375); stride 3 == packed xyz.  The 4th component is ignored
 *    (assumed 1).
 *  - normals: `nrm` + `nrm_stride` (3 == the `normals` descriptor rows, packed).
 *  - covariances (GICP): 6 floats per point, (xx, xy, xz, yy, yz, zz).
 *  - 4x4 transforms are COLUMN-major float[16] (T[c*4+r]) == Eigen::Matrix4f::data()
 *    of PointMatcher<float>::TransformationParameters; reading -> reference.
 *  - `on_device` != 0: the pointers are HIP device pointers on the handle's device
 *    (no PCIe copy); == 0: host pointers (copied with hipMemcpyAsync).
 *
 * THREADING: one handle == one non-re-entrant registration context (like the single
 * `icp_` member of o3d_slam::Mapper, Mapper.hpp:72, serialised by mapManipulationMutex_).
 * Independent handles may be used concurrently, each on its own HIP stream.
 */
#ifndef O3DSLAM_REG_H
#define O3DSLAM_REG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REG_API __attribute__((visibility("default")))

typedef struct reg_handle reg_handle;

/* Status codes; each names the reference behaviour it replaces. */
typedef enum {
    REG_OK = 0,
    REG_EMPTY_TARGET = 1,        /* initReference() returns false            ICP.cpp:850-855 */
    REG_EMPTY_SOURCE = 2,        /* runtime_error "reading point cloud is empty" ICP.cpp:958-960 */
    REG_NO_CORRESPONDENCES = 3,  /* ConvergenceError  ErrorMinimizer.cpp:75-77, Matches.cpp:76-80 */
    REG_BAD_TRANSFORM = 4,       /* runtime_error / TransformationError      ICP.cpp:910-918 */
    REG_NOT_CONFIGURED = 5,      /* "You must setup a matcher ..."           ICP.cpp:819-824; no target/source set */
    REG_BAD_ARGUMENT = 6,        /* InvalidParameter                         Registrar.h:103-109 */
    REG_MISSING_FIELD = 7,       /* InvalidField: no `normals` descriptor    DataPoints.cpp:1112 */
    REG_DEVICE_ERROR = 8,        /* HIP runtime failure (no reference analogue); the product never falls back to CPU */
    REG_UNSUPPORTED = 9
} reg_status;

typedef enum {
    REG_COST_P2PL = 0, /* PointToPlaneErrorMinimizer   ICP.cpp:1512-1566 + ErrorMinimizers/PointToPlane.cpp:274-400 */
    REG_COST_GICP = 1  /* plane-to-plane GICP factor (north star; arithmetic not in the reference tree) */
} reg_cost;

/* Configuration == the hot-path subset of param/icp.yaml (open3d_slam_ros/param/icp.yaml:11-27,86-92). */
typedef struct {
    int32_t struct_size;        /* = sizeof(reg_params); checked */
    int32_t cost;               /* reg_cost */
    /* matcher: KDTreeMatcher (MatchersImpl.h:80-88) */
    int32_t knn;                /* must be 1 */
    float   max_dist;           /* maxDist; INFINITY allowed (default of the reference) */
    float   epsilon;            /* accepted for config compatibility; the search is always exact (epsilon = 0) */
    /* outlierFilters (OutlierFiltersImpl.cpp); chain = product of 0/1 weights */
    int32_t use_trimmed;        /* TrimmedDistOutlierFilter */
    float   trim_ratio;
    int32_t use_surface_normal; /* SurfaceNormalOutlierFilter */
    float   max_normal_angle;   /* rad */
    int32_t use_max_dist_filter;/* MaxDistOutlierFilter */
    float   outlier_max_dist;
    /* transformationCheckers (TransformationCheckersImpl.cpp:57-158) */
    int32_t max_iter;           /* CounterTransformationChecker.maxIterationCount */
    float   min_diff_rot;       /* DifferentialTransformationChecker */
    float   min_diff_trans;
    int32_t smooth_len;
    int32_t fixed_iters;        /* >0: run exactly this many iterations, checkers ignored (throughput runs) */
    /* GICP termination, rule 0 (Gauss-Newton, rotation-first tangent); rule 1: gicp_stop_rule below */
    float   gicp_rot_eps;       /* rad */
    float   gicp_trans_eps;     /* m */
    /* device-side search structure */
    float   cell_size;          /* voxel-bin edge in metres; 0 = choose from target density */
    int32_t device;             /* HIP device ordinal */
    int32_t sort_source;        /* 1: Morton-order the reading on upload (speed only; results in input order) */
    /* degeneracyAwareness: OptimizedEqualityConstraints (icp.yaml:50-55; ICP.cpp:629-672, 2187-2444;
       PointToPlane.cpp:459-505): on the FIRST iteration the eigen-directions of the rotation / translation blocks of A
       are tested for information content (sums of |alignment| over the matched pairs above two cosine thresholds);
       every iteration then solves with the update constrained to zero along the non-localizable directions.
       Point-to-plane only.  Inert when every direction is localizable. */
    int32_t use_xicp;
    float   xicp_enough;        /* enoughInformationThreshold (250 shipped): sum over alignment > cos(min angle) */
    float   xicp_insufficient;  /* insufficientInformationThreshold (180 shipped): sum over alignment > cos(strong angle) */
    float   xicp_min_angle_deg; /* point2NormalMinimalAlignmentAngleThreshold (80 shipped) */
    float   xicp_strong_angle_deg; /* point2NormalStrongAlignmentAngleThreshold (45 shipped) */
    /* GICP stop rule.  0: |d_rot| < gicp_rot_eps && |d_trans| < gicp_trans_eps after an update (small_gicp's
       TerminationCriteria), or max_iter updates.  1: open3d::pipelines::registration::ICPConvergenceCriteria as
       RegistrationIcpGeneralized uses it (open3d_slam/src/CloudRegistration.cpp:16-21,45-52: only max_iteration_ is
       configured, relative_fitness_ = relative_rmse_ = 1e-6 by default): after every update the correspondences are
       re-evaluated at the new pose and the loop stops when |fitness - previous fitness| < gicp_rel_fitness AND
       |inlier_rmse - previous inlier_rmse| < gicp_rel_rmse (fitness = matched / N, inlier_rmse = sqrt(sum d2 / matched)),
       else after max_iter updates (+ the final evaluation, whose fitness / rmse are reported).  Open3D 0.15.1 is an
       un-vendored dependency: restated from its published RegistrationICP loop, PARITY UNPINNED. */
    int32_t gicp_stop_rule;
    float   gicp_rel_fitness;   /* relative_fitness_ (1e-6) */
    float   gicp_rel_rmse;      /* relative_rmse_ (1e-6) */
    int32_t reserved;
} reg_params;

typedef struct {
    int32_t iterations;
    int32_t converged;          /* DifferentialTransformationChecker said stop */
    int32_t max_iter_reached;   /* CounterTransformationChecker threw MaxNumIterationsReached (ICP.cpp:1294-1298) */
    int32_t rank_last;          /* numerical rank of the last 6x6 system (6 = invertible) */
    int64_t n_inliers;          /* pairs with non-zero weight in the last iteration */
    int64_t n_matched;          /* pairs with a neighbour inside max_dist in the last iteration */
    double  error;              /* P2PL: sum w r^2;  GICP: sum 0.5 r^T M r  (last iteration) */
    double  fitness;            /* n_inliers / N   (open3d RegistrationResult::fitness_ analogue) */
    double  inlier_rmse;        /* sqrt(sum_inliers d^2 / n_inliers) */
    float   H_last[36];         /* last normal matrix, row-major (symmetric) */
    float   b_last[6];
    float   target_build_ms;    /* last reg_set_target: upload + voxel-bin build */
    float   loop_ms;            /* iteration loop of the last reg_register (device time, HIP events) */
    float   T_iter_last[16];    /* final T_iter (column-major): P2PL in the centred frames, GICP == T_out */
    int32_t n_band_stalls;      /* fused path: iterations whose trimmed-band prediction failed and were re-run on the generic path */
    int32_t n_constraints;      /* use_xicp: number of non-localizable directions (0 = plain solve) */
    float   prof_ms[4];         /* loop profiling (o3dslam_reg_debug.h: profile_loop): summed device time of [0] k_match, [1] k_iter_fused launches */
    int32_t prof_launches[4];   /* ... and how many launches that was */
    /* use_xicp: [0..2] rotation eigen-directions (descending eigenvalue), [3..5] translation; 1 = localizable
       (LocalizabilityCategory, PointMatcher.h:603-607); all 1 when the analysis is off */
    int32_t localizable[6];
    double  xicp_combined[6];   /* the two information sums per direction (ICP.cpp:2128-2155) */
    double  xicp_high[6];
    float   source_prep_ms;     /* last reg_set_source: upload + Morton order of the reading (device time, HIP events) */
    int32_t rotation_corrected; /* 1: |1 - det R| > 1e-3 in the pre-transform -> points moved with the re-orthogonalised copy
                                   (RigidTransformation::correctParameters, TransformationsImpl.cpp:73-76,105-166) */
    float   T_iter_prev[16];    /* the T_iter the LAST iteration ran at (column-major, same frames as T_iter_last): the
                                   correspondences / weights reg_get_correspondences reports belong to this pose */
    int32_t n_tail_launches;    /* launches of the persistent settled-tail kernel in this registration (0: three-launch path) */
    int32_t n_tail_iterations;  /* iterations those launches ran */
} reg_result;

/* ICPChainBase::setDefault (ICP.cpp:100-113): knn 1, eps 0, maxDist inf, Trimmed 0.85,
   Counter 40, Differential 0.001/0.001/3, point-to-plane. */
REG_API void reg_default_params(reg_params* p);
/* open3d_slam_ros/param/icp.yaml as shipped (maxDist 0.5, Trimmed 0.9, SurfaceNormal 1.57,
   Differential 0.001/0.008/3, Counter 30, degeneracyAwareness OptimizedEqualityConstraints 250 / 180 / 80 / 45 ->
   use_xicp = 1); epsilon is forced to 0 (exact search). */
REG_API void reg_shipped_params(reg_params* p);

REG_API reg_status reg_create(const reg_params* p, reg_handle** out);
REG_API void       reg_destroy(reg_handle* h);
REG_API const char* reg_last_error(const reg_handle* h);

/* All device work of the handle is enqueued on this hipStream_t (default: a stream the handle owns).  Work already
   queued on the previous stream is waited for before the switch (the handle's buffers are shared between them). */
REG_API reg_status reg_set_stream(reg_handle* h, void* hip_stream);

/* == ICP::initReference (ICP.cpp:847-898): copy, subtract centroid, build the search structure
   (voxel-bin table replaces KDTreeMatcher::init, MatchersImpl.cpp:78-83).
   P2PL needs `nrm`; GICP needs `cov`. */
REG_API reg_status reg_set_target(reg_handle* h, const float* xyz, int64_t xyz_stride, const float* nrm,
                                  int64_t nrm_stride, const float* cov, int64_t m, int on_device);

/* Reading cloud of the next reg_register / reg_prepare (ICP.cpp:952).  `nrm` is required when
   use_surface_normal is set (else REG_MISSING_FIELD); `cov` is required for GICP.
   BUFFER LIFETIME (reg_set_target, reg_set_source and their _f64 forms): host buffers are consumed before the call
   returns.  DEVICE buffers (on_device != 0) are read by copies / kernels enqueued on the handle's stream and the call
   may return before they have run: keep them valid and unmodified until a later blocking call on the handle
   (reg_register, reg_linearize, reg_get_correspondences, reg_dist_finish, ...) has returned or the stream has been
   synchronised.  reg_set_target itself blocks until the table is built.
   Time of the last reg_set_source (upload + Morton order) is reported as reg_result.source_prep_ms. */
REG_API reg_status reg_set_source(reg_handle* h, const float* xyz, int64_t xyz_stride, const float* nrm,
                                  int64_t nrm_stride, const float* cov, int64_t n, int on_device);

/* R11, reading side: Open3D's fp64 AoS arrays as Mapper::addRangeMeasurement holds them (points_ / normals_ n x 3
   doubles, covariances_ n x 9 doubles or NULL), cast to fp32 on the device exactly as open3dToPointmatcher does
   (open3d_conversions.cpp:57-118, static_cast<float> per coordinate; Mapper.cpp:288-289), then reg_set_source. */
REG_API reg_status reg_set_source_f64(reg_handle* h, const double* xyz, const double* normals, const double* covs,
                                      int64_t n, int on_device);

/* == ICP::compute(reading, -, T_init, false) (ICP.cpp:813-844 -> 902-1349) on the reading given to
   reg_set_source: R2 reading prep, the while(iterate) loop (R3-R9) on the device, R10 composition. */
REG_API reg_status reg_register(reg_handle* h, const float T_init[16], float T_out[16], reg_result* res);

/* Convenience == reg_set_source + reg_register. */
REG_API reg_status reg_compute(reg_handle* h, const float* xyz, int64_t xyz_stride, const float* nrm,
                               int64_t nrm_stride, const float* cov, int64_t n, int on_device,
                               const float T_init[16], float T_out[16], reg_result* res);

/* Factor-level hook (one pass of R3-R7, "Factor::linearize + reduction").
   reg_prepare does R2 for T_init (P2PL: centre + pre-transform the reading; GICP: no-op besides upload).
   reg_linearize evaluates the cost at T_iter (P2PL: transform in the centred frames, identity at
   iteration 0; GICP: full reading->reference transform) and returns the reduced system:
   P2PL: H = A, b as in ICP.cpp:1543,1565 (x = [rx ry rz tx ty tz]);  GICP: H = sum J^T M J, b = sum J^T M r. */
REG_API reg_status reg_prepare(reg_handle* h, const float T_init[16]);
REG_API reg_status reg_linearize(reg_handle* h, const float T_iter[16], float H[36], float b[6], double* err,
                                 int64_t* n_inliers);

/* Correspondences of the most recent iteration / linearize, in the reading's input order:
   ids = index into the target as given to reg_set_target (-1 = none within max_dist), d2 = squared
   distance (+inf = none) (PointMatcher.h:416-436), w = outlier weight (0/1).  Any pointer may be NULL. */
REG_API reg_status reg_get_correspondences(reg_handle* h, int32_t* ids, float* d2, float* w);

/* Distributed (point-partitioned reading) support: the per-iteration reduction is exposed in two
   halves so the caller can sum the partial systems of all ranks (RCCL all-reduce over xGMI) in between.
   reg_match_local     : R3+R4 on this rank's slice; fills hist[2048] with the level-`level` radix histogram
                         of finite d2 (level 0..2; prefix = bits fixed by previous levels).
   reg_reduce_local    : R5-R7 given the global trim limit -> 32 doubles {21 upper-tri H, 6 b, err, n_in, n_matched, sum d2 inliers, pad}.
   reg_apply_update    : R8+R9 from the globally summed 32 doubles (identical on every rank). */
REG_API reg_status reg_source_centroid_sums(reg_handle* h, int64_t sums[3]);   /* sum llrint(x*2^16): exact, order-free */
REG_API reg_status reg_prepare_centroid(reg_handle* h, const float T_init[16], const float c_read[3]); /* R2 with the GLOBAL reading centroid */
REG_API reg_status reg_compose(reg_handle* h, const float T_iter[16], float T_out[16]);              /* R10 */
REG_API reg_status reg_match_local(reg_handle* h, const float T_iter[16]);
REG_API reg_status reg_trim_histogram(reg_handle* h, int level, uint32_t prefix, uint32_t hist[2048]);
REG_API reg_status reg_reduce_local(reg_handle* h, const float T_iter[16], float trim_limit, double sums[32]);
REG_API reg_status reg_solve_update(const reg_params* p, const double sums[32], const float T_iter[16],
                                    float T_next[16], int32_t* rank);

/* Stream-ordered variant of the same exchange (no host round trip per iteration): each phase only ENQUEUES
   kernels on the handle's stream; between phases the caller all-reduces -- on the SAME stream (RCCL) -- the device
   buffers returned by reg_dist_buffers (hist: 3 x 2048 int32 counts, sums: 32 doubles).  Phases:
   0 match + level-0 histogram | 1, 2 radix levels of the exact global trimmed quantile | 3 weights + normal
   equations (local sums) | 4 solve + pose update + checkers on the device from the GLOBAL sums. */
REG_API reg_status reg_dist_begin(reg_handle* h, const float T_start[16] /* NULL: identity (P2PL) / T_init (GICP) */);
REG_API reg_status reg_dist_buffers(reg_handle* h, void** hist, void** sums);
REG_API reg_status reg_dist_phase(reg_handle* h, int phase);
REG_API reg_status reg_dist_finish(reg_handle* h, float T_out[16], reg_result* res);

/* Fused multi-GPU iteration (2 launches + ONE all-gather per iteration instead of 6 launches + 4 all-reduces), usable
   once the trimmed limit has settled -- same prediction / exact verification as the single-GPU fused iteration:
   phase 5: fused search + weights + normal equations on this rank's slice; its certain sums, band count and band
            records (<= 512) go into a fixed-size contribution block            -> all-gather(contrib -> gathered)
   phase 6: every rank reduces the SAME gathered blocks in rank order (identical results, no broadcast), verifies the
            band with the global counts, selects the exact global quantile, solves and updates on the device.
   A failed verification (or an overflowing block) sets `stall` on every rank alike; the caller then runs that
   iteration through phases 0-4. */
typedef struct {
    int64_t sequences_done;      /* update kernels that have reported since reg_dist_begin */
    int64_t sequences_enqueued;
    int32_t iterations;          /* completed Gauss-Newton iterations */
    int32_t done, stall, stream_idle;
    float   limit_last, limit_prev;
} reg_dist_status;
REG_API reg_status reg_dist_fused_buffers(reg_handle* h, int n_ranks, int rank, void** contrib, void** gathered,
                                          int64_t* contrib_bytes);
REG_API reg_status reg_dist_poll(reg_handle* h, reg_dist_status* out);
/* What the update kernel of ONE specific sequence reported (seq_rel = 1 for the first iteration enqueued after
   reg_dist_begin): sequences_done == seq_rel when available, 0 otherwise; stream_idle (sampled first) tells that it will
   never come (update kernels that find the loop done or stalled do not report).  Multi-GPU drivers must steer by these
   records only: they are identical on every rank, whereas "the latest state seen" depends on timing. */
REG_API reg_status reg_dist_record(reg_handle* h, int64_t seq_rel, reg_dist_status* out);
/* Reading preparation without a host round trip: reg_dist_centroid_sums enqueues this rank's integer centroid sums
   (numeric contract NC1) and returns their device address (3 x int64); the caller all-reduces (sum) them on the handle's
   stream; reg_dist_prepare then centres and pre-transforms the slice with the centroid of all n_global reading points. */
REG_API reg_status reg_dist_centroid_sums(reg_handle* h, void** sums_dev);
REG_API reg_status reg_dist_prepare(reg_handle* h, const float T_init[16], int64_t n_global);
/* Select-by-gather form of the trimmed iteration (fewer dependent collectives): phase 10 (match), all-gather the
   n_max floats at *d2_local of every rank into *d2_all (n_ranks x n_max), phase 11 (exact 3-level select on the
   gathered distances, redundantly on every rank; linearize of the local slice; partial sums), all-reduce the 32 sums,
   phase 4.  n_max >= every rank's reading size (the tail is padded with +inf).  Call after reg_set_source. */
REG_API reg_status reg_dist_gather_buffers(reg_handle* h, int n_ranks, int64_t n_max, void** d2_local, void** d2_all);
/* use_xicp on the distributed path: on the FIRST iteration, after phase 4, run phase 7, all-reduce (sum) the 4 doubles at
   *center, phase 8, all-reduce the 12 doubles at *sums, phase 9 (reports with the sequence number of phase 4). */
REG_API reg_status reg_dist_xicp_buffers(reg_handle* h, void** center, void** sums);

/* ---- multi-GPU registration behind the C ABI (BASELINE config C4) ---------------------------------
   One process per GPU.  The reading is point-partitioned: every rank gives ITS slice to reg_set_source; the reference
   (reg_set_target) and its tables are replicated.  reg_dist_register is ICP::compute (ICP.cpp:813-844, called from
   the C++ mapper, Mapper.cpp:343,372-373) for that layout: all kernels and collectives are enqueued on the handle's
   stream, nothing synchronises inside an iteration, and every rank returns the same T_out.  Per iteration the ranks
   exchange, over RCCL (xGMI): unsettled iterations -- one all-gather of the squared match distances (exact global
   trimmed quantile, selected redundantly on every rank) + one all-reduce of the 32-double (H, b, e, counts) record;
   settled iterations -- ONE all-gather of a fixed-size block {32 sums, band records} per rank.
   Group set-up: rank 0 calls reg_dist_get_unique_id and hands the 128 bytes to the other ranks by any means (MPI,
   a file, torch.distributed ...); every rank then calls reg_dist_init (ncclCommInitRank on the handle's device).
   COLLECTIVE CONTRACT: all ranks call reg_dist_init / reg_dist_register / reg_dist_shutdown in the same order, and a
   new reading is set on ALL ranks or on none between two registrations.  Every wait inside reg_dist_register has a
   deadline (O3D_DIST_TIMEOUT_S, default 30 s): a dead peer or a collective that never completes returns
   REG_DEVICE_ERROR on the survivors instead of hanging them -- the caller should then exit non-zero. */
#define REG_DIST_ID_BYTES 128
REG_API reg_status reg_dist_get_unique_id(char id[REG_DIST_ID_BYTES]);
REG_API reg_status reg_dist_init(reg_handle* h, const char id[REG_DIST_ID_BYTES], int rank, int n_ranks);
REG_API reg_status reg_dist_register(reg_handle* h, const float T_init[16], float T_out[16], reg_result* res);
REG_API reg_status reg_dist_shutdown(reg_handle* h);
/* after reg_dist_register: size of the whole reading and how the loop went (any pointer may be NULL) */
REG_API reg_status reg_dist_info(reg_handle* h, int64_t* n_global, int32_t* n_generic, int32_t* n_fused, int32_t* n_stalls);
/* A transport other than RCCL (tests on one GPU, other launchers): both callbacks ENQUEUE on `stream` (or complete
   before returning), operate in place on DEVICE memory and return 0 on success. */
enum { REG_DT_I32 = 0, REG_DT_I64 = 1, REG_DT_F64 = 2 };
typedef struct {
    void* ctx;
    int (*all_reduce_sum)(void* ctx, void* buf, int64_t count, int dtype, void* stream);
    int (*all_gather)(void* ctx, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
} reg_collectives;
REG_API reg_status reg_dist_init_custom(reg_handle* h, const reg_collectives* c, int rank, int n_ranks);

/* The steering of reg_dist_register as a pure host-side state machine (no device, no collectives), exported so that
   the decision logic every rank must agree on can be exercised without a GPU (world_size-2 gloo tests on CPU):
   step() consumes the reply to its previous action and returns the next one.
     REG_STEER_RECORD  wait for the record of sequence `seq` (reg_dist_record); reply.available = 0 when the stream
                       drained without it (an earlier sequence ended or stalled the loop)
     REG_STEER_GENERIC enqueue one select-based iteration (no reply)
     REG_STEER_FUSED   enqueue `count` fused iterations (no reply)
     REG_STEER_DRAIN   wait until the stream is idle, reply with the latest state (reg_dist_poll)
     REG_STEER_DONE    the registration is complete */
enum { REG_STEER_RECORD = 0, REG_STEER_GENERIC = 1, REG_STEER_FUSED = 2, REG_STEER_DRAIN = 3, REG_STEER_DONE = 4 };
typedef struct { int32_t kind, count; int64_t seq; } reg_dist_action;
typedef struct { int32_t available, iterations, done, stall; float limit_last, limit_prev; } reg_dist_reply;
typedef struct reg_dist_steer reg_dist_steer;
REG_API reg_dist_steer* reg_dist_steer_create(int trimming, int fixed_iters, int max_iter, float settle_tol, int can_fuse);
REG_API void reg_dist_steer_destroy(reg_dist_steer* s);
REG_API reg_dist_action reg_dist_steer_step(reg_dist_steer* s, const reg_dist_reply* reply /* NULL on the first call */);
REG_API void reg_dist_steer_counts(const reg_dist_steer* s, int32_t* n_generic, int32_t* n_fused, int32_t* n_stalls);

/* Host-side pieces of the path, exported so they can be checked without a GPU
   (PointToPlane.cpp:112-265 solve, :327-381 x -> 4x4; column-major 4x4). */
REG_API int  reg_host_solve6(const float A[36], const float b[6], float x[6]);
REG_API void reg_host_x_to_T(const float x[6], float T[16]);
/* R8x equality-constrained solve (PointToPlane.cpp:459-505, null-space form): flags[k] = 1 keeps eigen-direction k
   (0-2 rotation block, 3-5 translation block, descending eigenvalue), 0 forbids any update along it.
   Returns the rank of the reduced system. */
REG_API int  reg_host_solve6_xicp(const float A[36], const float b[6], const int32_t flags[6], float x[6]);
REG_API void reg_host_centroid(const float* xyz, int64_t stride, int64_t n, float out[3]);
/* Launch plan of the persistent tail kernel (csrc/kernels_tail.hpp) for a reading of n points on a device with `cus`
   compute units: plan = {usable (0/1), workgroups, workgroups per XCD class, reading points per XCD class}.  Octet
   oc = (s >> 3) * plan[2] + (b >> 3) of XCD class x = b & 7 is, with tile == 0, octet oc of the class's contiguous share
   (reading point x * plan[3] + 8 * oc + (s & 7)); with tile > 0 the classes take turns in tiles of `tile` octets: reading point
   8 * (((oc / tile) * 8 + x) * tile + oc % tile) + (s & 7).  Valid when 8 * oc + (s & 7) < plan[3] and the point lies below n
   (1024 slots per workgroup).  Host-only: lets the slot <-> point mapping be checked on CPU. */
REG_API void reg_host_tail_plan(int64_t n, int32_t cus, int32_t tile, int32_t plan[4]);

/* Measurement hook (bench.py roofline object): average device time in ms, by HIP events on the handle's
   stream, of [0] the match kernel, [1] the trimmed-quantile select passes, [2] linearize + final reduce. */
REG_API reg_status reg_profile_kernels(reg_handle* h, const float T_iter[16], int reps, float ms[3]);

/* ---- next row: surface normals / covariances by exact k-NN + PCA ------------------------------
   Replaces the CPU producers of the attributes the registration consumes:
     libpointmatcher/pointmatcher/DataPointsFilters/SurfaceNormal.cpp:152-252  (SurfaceNormalDataPointsFilter:
       self k-NN including the point itself, mean, C = NN NN^T, eigenvector of the smallest eigenvalue, clamped to
       [-1,1]; a neighbourhood of rank < 2 yields the zero vector),
     open3d_slam/src/CloudRegistration.cpp:25-43  (estimateNormals(KDTreeSearchParamKNN) +
       OrientNormalsTowardsCameraLocation before every point-to-plane registration),
     open3d_slam/src/helpers.cpp:153-165           (the same with a radius cap).
   xyz: n points, stride in floats.  k in [1,32] neighbours (the point itself counts), max_dist > 0 (may be +inf).
   viewpoint: NULL -> sign such that the largest component is positive; else normals face the viewpoint.
   Outputs (host pointers, or device pointers when on_device != 0), indexed like the input; every member but `normals`
   may be NULL (the names in brackets are the filter's keep* switches / descriptor names, SurfaceNormal.h:71-77):
     normals    n x 3                                                   [keepNormals, "normals"]
     eigvals    n x 3 ascending                                          [keepEigenValues + sortEigen, "eigValues"]
     eigvecs    n x 9: eigenvector k (ascending eigenvalue) at [9 i + 3 k + r]   [keepEigenVectors, "eigVectors"]
     covs       n x 6 {xx xy xz yy yz zz}: C/m, or with regularise != 0 the plane-like GICP covariance
                V diag(1e-3,1,1) V^T (small_gicp / Open3D GICP convention)
     densities  n: m / (4/3 pi r^3), r = largest distance of a neighbour from the neighbourhood mean (utils.h:106-128);
                0 for a degenerate neighbourhood                        [keepDensities, "densities"]
     mean_dists n: |p - mean| (SurfaceNormal.cpp:243-252); (float)SIZE_MAX when degenerate   [keepMeanDist, "meanDists"]
     ids        n x k neighbour indices ascending by (d2, index), -1 padded   [keepMatchedIds, "matchedIds"]
   n_rescanned (may be NULL): points whose candidate list exceeded the on-chip list (statistics; results are exact). */
typedef struct {
    float*   normals;
    float*   eigvals;
    float*   eigvecs;
    float*   covs;
    float*   densities;
    float*   mean_dists;
    int32_t* ids;
} reg_normals_out;
REG_API reg_status reg_estimate_normals(reg_handle* h, const float* xyz, int64_t xyz_stride, int64_t n, int on_device,
                                        int k, float max_dist, const float viewpoint[3], int regularise,
                                        const reg_normals_out* out, int64_t* n_rescanned);


/* SurfaceNormalDataPointsFilter's `smoothNormals` option (SurfaceNormal.cpp:259-283): every normal becomes the mean of its
   neighbours' normals (those pointing away from it flipped), IN PLACE in index order as the reference does it -- point i
   reads the already smoothed normals of its lower-indexed neighbours.  normals: n x 3 in / out; ids: n x k as
   reg_estimate_normals reports them (-1 = no neighbour).  Evaluated on the device as a level-synchronous sweep over the
   dependency DAG, bit-identical to the sequential loop; n_passes (may be NULL): sweeps launched. */
REG_API reg_status reg_smooth_normals(reg_handle* h, float* normals, const int32_t* ids, int64_t n, int k, int on_device,
                                      int32_t* n_passes);

/* ---- next row: target-side preparation on the device ------------------------------------------
   Replaces, in front of reg_set_target, what the mapper does on the host every referenceCloudSettingPeriod_:
     open3d_slam/src/ScanToMapRegistration.cpp:90-96   cropSubmap: scanMatcherCropper_->setPose(mapToRangeSensor); crop(map)
     open3d_slam/src/croppers.cpp:76-106               CroppingVolume::crop: order-preserving copy of the points (+ normals,
                                                       covariances) with isWithinVolume(p)
     open3d_slam/src/croppers.cpp:118-170              the volumes: MaxRadius |p-t| <= r; MinRadius |p-t| >= r; MinMaxRadius;
                                                       Cylinder z in [minZ,maxZ] (absolute) and |(p-t).xy| <= r; all in double
     open3d_utils/open3d_conversions/src/open3d_conversions.cpp:57-118  open3dToPointmatcher: fp64 AoS -> fp32 features / normals
       ("This is time consuming", Mapper.cpp:336)
   xyz / normals: m x 3 doubles (std::vector<Eigen::Vector3d>::data()); covs: m x 9 doubles (Matrix3d, symmetric) or NULL.
   The kept points keep their order; correspondence ids reported later index the CROPPED cloud (as in the reference,
   whose matcher only ever sees the patch); reg_get_target_source_indices maps them back. */
typedef enum {
    REG_CROP_NONE = 0, REG_CROP_MAX_RADIUS = 1, REG_CROP_MIN_RADIUS = 2, REG_CROP_MIN_MAX_RADIUS = 3, REG_CROP_CYLINDER = 4
} reg_crop_type;
typedef struct {
    int32_t type;          /* reg_crop_type */
    int32_t reserved;
    double  center[3];     /* pose_.translation() of the cropper (mapToRangeSensor) */
    double  radius_min;    /* MinRadius / MinMaxRadius */
    double  radius_max;    /* MaxRadius / MinMaxRadius / Cylinder radius */
    double  min_z, max_z;  /* Cylinder */
} reg_crop;
REG_API reg_status reg_set_target_f64(reg_handle* h, const double* xyz, const double* normals, const double* covs,
                                      int64_t m, int on_device, const reg_crop* crop, int64_t* n_kept);
/* idx[n_kept]: position of every kept point in the cloud given to reg_set_target_f64 */
REG_API reg_status reg_get_target_source_indices(reg_handle* h, int32_t* idx);

/* Map maintenance half of the same row: voxelizeWithinCroppingVolume (open3d_slam/src/helpers.cpp:117-192), which
   Submap::insertScan runs on the whole map after every inserted scan (Submap.cpp:39-96).  Points outside `volume` are
   copied through in their order; points inside are bucketed by voxel index floor(p * (1/voxel_size)) per axis
   (VoxelHashMap.hpp:48-51) and replaced by one averaged point per occupied voxel: sums in double in index order
   (AccumulatedPoint, helpers.cpp:30-72: NaN normals are skipped, the averaged normal is re-normalised, covariances are
   averaged).  The reference emits the voxels in std::unordered_map order (unspecified); here they follow the outside
   points in ascending (z, y, x) voxel index.  Inputs / outputs: m x 3 (x 9 for covs) doubles, host or device
   (on_device); the output arrays must hold m points.  voxel_size <= 0 copies the cloud through. */
REG_API reg_status reg_voxelize_within_volume(reg_handle* h, const double* xyz, const double* normals, const double* covs,
                                              int64_t m, int on_device, const reg_crop* volume, double voxel_size,
                                              double* out_xyz, double* out_normals, double* out_covs, int64_t* n_out,
                                              int64_t* n_outside);

/* Space carving (open3d_slam/src/Submap.cpp:130-143 -> helpers.cpp:238-283, getIdxsOfCarvedPoints): every scan point
   (already in the map frame) casts a ray from the sensor; the ray is sampled every `voxel_size` up to
   max(voxel_size, min(|p - sensor| - truncation, max_ray)); map points that share a voxel (floor(p * (1/voxel_size)),
   VoxelHashMap.hpp:43-51) with a sample are removed when |direction . normalized(normal)| > min_dot (always, without
   normals).  Only map points inside `subset` (the map builder's cropper, may be NULL = all) take part.  All in double,
   one rounding per operation.  removed[]: ascending map indices (the reference returns them in std::unordered_set
   order), capacity m.  Rays of zero length are skipped (the reference divides by zero there). */
REG_API reg_status reg_carve_indices(reg_handle* h, const double* map_xyz, const double* map_normals, int64_t m,
                                     const double* scan_xyz, int64_t n_scan, int on_device, const double sensor[3],
                                     const reg_crop* subset, double voxel_size, double max_ray, double truncation,
                                     double min_dot, int32_t* removed, int64_t* n_removed);

/* B1 result extra: the 6x6 information matrix of a registered pair, as the loop-closure and odometry
   constraint builders obtain it from open3d::pipelines::registration::GetInformationMatrixFromPointClouds
   (open3d_slam/src/constraint_builders.cpp:69-73, PlaceRecognition.cpp:148; arithmetic in un-vendored Open3D 0.15.1:
   PARITY UNPINNED, restated): for every reading point whose nearest reference point q (reference frame) lies within
   max_dist after applying T, G = [[0, z, -y, 1, 0, 0], [-z, 0, x, 0, 1, 0], [y, -x, 0, 0, 0, 1]] with (x, y, z) = q and
   info = sum G^T G (rotation-first ordering, row-major 6x6, float64).  max_dist must not exceed the handle's max_dist
   (the reach of the search structure).  Uses the reference and reading currently set; a prepared reading is
   re-prepared at T. */
REG_API reg_status reg_information_matrix(reg_handle* h, const float T[16], float max_dist, double info[36],
                                         int64_t* n_pairs);

/* Introspection of the search structure (tests, DESIGN.md numbers). */
typedef struct {
    int64_t n_points;
    int64_t n_bricks;
    int64_t n_cells_occupied;
    int64_t table_bytes;
    float   cell_size;
    float   origin[3];
    float   centroid[3];
    int32_t dims[3];
} reg_target_info;
REG_API reg_status reg_get_target_info(const reg_handle* h, reg_target_info* info);

#ifdef __cplusplus
}
#endif
#endif /* O3DSLAM_REG_H */
