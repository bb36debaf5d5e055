// reg_core.hip -- C ABI (include/o3dslam_reg.h) + HIP kernels of the MI355X registration path.
// gfx950 only.  The product never falls back to a CPU path: every entry point that needs the
// device returns REG_DEVICE_ERROR when HIP fails.
#include "../../include/o3dslam_reg.h"
#include "../../include/o3dslam_reg_debug.h"
#include "host_math.hpp"
#include "reg_kernels.hpp"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

using namespace o3dreg;

// minimum waves per SIMD the search kernels are compiled for (2nd __launch_bounds__ argument): caps their VGPRs
#ifndef O3D_MATCH_WAVES
#define O3D_MATCH_WAVES 1
#endif
#ifndef O3D_SEARCH_WAVES
#define O3D_SEARCH_WAVES O3D_MATCH_WAVES
#endif

#include "reg_state.hpp"
#include "kernels_build.hpp"
#include "kernels_reading.hpp"
#include "kernels_match.hpp"
#include "kernels_fused.hpp"
#include "kernels_mapprep.hpp"
#include "kernels_xicp.hpp"
#include "kernels_update.hpp"
#include "kernels_tail.hpp"
#include "kernels_normals.hpp"

// host side: one handle = one non-re-entrant registration context (include/o3dslam_reg.h)
#include "host_target.hpp"
#include "host_loop.hpp"
#include "host_dist.hpp"
#include "host_rccl.hpp"

#if O3D_SEARCH_STATS
// diagnostic builds only: read (and clear) the search counters of reg_kernels.hpp
extern "C" __attribute__((visibility("default"))) int o3d_debug_search_stats(unsigned long long out[64], int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(o3dreg::g_search_stats), 512) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(o3dreg::g_search_stats), z, 512) != hipSuccess) return -1;
    }
    return 0;
}
#endif
