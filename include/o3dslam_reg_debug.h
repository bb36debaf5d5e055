/*
 * o3dslam_reg_debug.h -- experiment / measurement switches of the registration library.
 *
 * NOT part of the drop-in boundary (include/o3dslam_reg.h): nothing a maintainer of the reference binds to.  Used by
 * the parity tests (forcing the rare branches: band misprediction, histogram select, generic-only path), the A/B
 * scripts under tools/ and bench.py's kernel timing.  All zero in production.
 */
#ifndef O3DSLAM_REG_DEBUG_H
#define O3DSLAM_REG_DEBUG_H

#include "o3dslam_reg.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t struct_size;        /* = sizeof(reg_debug_params); checked */
    int32_t profile_loop;       /* 1: every search kernel of reg_register carries begin/end HIP events (reg_result.prof_ms) */
    int32_t match_variant;      /* 0: 8 lanes per reading point + level hints (default); 1: one lane per point; 2: 8 lanes, no hints; 3: as 0 with the level-0 histogram fused into the match kernel */
    int32_t debug_flags;        /* ablation bits: 4, 8 change results (tests only: 8 forces band mispredictions); 32 = hash instead of the dense brick directory, 64 = histogram select for every trimmed band (results unchanged) */
    int32_t disable_halo;       /* 1: no halo-bin level */
    int32_t lanes_per_point;    /* 0 = default (8); 4 */
    int32_t disable_fused;      /* 1: every iteration on the generic (select-based) path */
    int32_t reserved;
} reg_debug_params;

/* Call right after reg_create, before reg_set_target (disable_halo and debug_flags & 32 act on the table build). */
REG_API reg_status reg_debug_configure(reg_handle* h, const reg_debug_params* d);

#ifdef __cplusplus
}
#endif
#endif /* O3DSLAM_REG_DEBUG_H */
