// kernels_mapprep.hpp -- target-side preparation rows: cropping volume, fp64 -> fp32, voxelize-within-volume
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once

// ---- target-side preparation: crop (croppers.cpp:76-170) + fp64 -> fp32 (open3d_conversions.cpp:57-118)
struct CropCfg {
    int type;
    double cx, cy, cz, rmin, rmax, zmin, zmax;
};
__device__ __forceinline__ bool crop_inside(const CropCfg& c, double x, double y, double z) {
    if (c.type == REG_CROP_NONE) return true;
    const double dx = x - c.cx, dy = y - c.cy, dz = z - c.cz;
    if (c.type == REG_CROP_CYLINDER) {
        double a = dx * dx;
        double b = dy * dy;
        const double d = sqrt(a + b);
        return z >= c.zmin && z <= c.zmax && d <= c.rmax;
    }
    double a = dx * dx;
    double b = dy * dy;
    double s2 = a + b;
    a = dz * dz;
    s2 = s2 + a;
    const double d = sqrt(s2);
    if (c.type == REG_CROP_MAX_RADIUS) return d <= c.rmax;
    if (c.type == REG_CROP_MIN_RADIUS) return d >= c.rmin;
    return d <= c.rmax && d >= c.rmin;
}
__global__ void k_crop_flags(const double* __restrict__ xyz, int64_t m, CropCfg c, uint32_t* __restrict__ flags) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    flags[i] = crop_inside(c, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]) ? 1u : 0u;
}
// offs = exclusive scan of flags: order-preserving compaction + conversion
__global__ void k_crop_gather(const double* __restrict__ xyz, const double* __restrict__ nrm, const double* __restrict__ cov,
                              int64_t m, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ offs,
                              float* __restrict__ oxyz, float* __restrict__ onrm, float* __restrict__ ocov,
                              int32_t* __restrict__ oidx) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= m || !flags[i]) return;
    const size_t o = offs[i];
    oxyz[3 * o + 0] = (float)xyz[3 * i + 0];
    oxyz[3 * o + 1] = (float)xyz[3 * i + 1];
    oxyz[3 * o + 2] = (float)xyz[3 * i + 2];
    if (nrm) {
        onrm[3 * o + 0] = (float)nrm[3 * i + 0];
        onrm[3 * o + 1] = (float)nrm[3 * i + 1];
        onrm[3 * o + 2] = (float)nrm[3 * i + 2];
    }
    if (cov) {
        const double* c = cov + 9 * i;   // Matrix3d, symmetric: xx xy xz / . yy yz / . . zz
        ocov[6 * o + 0] = (float)c[0];
        ocov[6 * o + 1] = (float)c[1];
        ocov[6 * o + 2] = (float)c[2];
        ocov[6 * o + 3] = (float)c[4];
        ocov[6 * o + 4] = (float)c[5];
        ocov[6 * o + 5] = (float)c[8];
    }
    oidx[o] = (int32_t)i;
}

// ---- voxelizeWithinCroppingVolume (helpers.cpp:117-192) ----
constexpr int kVoxBits = 21;                       // voxel index bits per axis in the sort key (offset binary)
constexpr long long kVoxOff = 1ll << (kVoxBits - 1);
__global__ void k_vox_classify(const double* __restrict__ xyz, int64_t m, CropCfg c, double inv, uint32_t* __restrict__ f_in,
                               uint32_t* __restrict__ f_out, uint32_t* __restrict__ overflow) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const bool in = crop_inside(c, x, y, z);
    f_in[i] = in ? 1u : 0u;
    f_out[i] = in ? 0u : 1u;
    if (in) {
        const double vx = floor(x * inv), vy = floor(y * inv), vz = floor(z * inv);
        if (!(fabs(vx) < (double)kVoxOff && fabs(vy) < (double)kVoxOff && fabs(vz) < (double)kVoxOff)) atomicOr(overflow, 1u);
    }
}
__global__ void k_vox_scatter(const double* __restrict__ xyz, const double* __restrict__ nrm, const double* __restrict__ cov,
                              int64_t m, double inv, const uint32_t* __restrict__ f_in, const uint32_t* __restrict__ o_in,
                              const uint32_t* __restrict__ o_out, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                              double* __restrict__ oxyz, double* __restrict__ onrm, double* __restrict__ ocov) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (f_in[i]) {
        const long long vx = (long long)floor(xyz[3 * i] * inv) + kVoxOff;
        const long long vy = (long long)floor(xyz[3 * i + 1] * inv) + kVoxOff;
        const long long vz = (long long)floor(xyz[3 * i + 2] * inv) + kVoxOff;
        keys[o_in[i]] = ((uint64_t)vz << (2 * kVoxBits)) | ((uint64_t)vy << kVoxBits) | (uint64_t)vx;
        vals[o_in[i]] = (uint32_t)i;
    } else {
        const size_t o = o_out[i];
        for (int k = 0; k < 3; ++k) oxyz[3 * o + k] = xyz[3 * i + k];
        if (nrm)
            for (int k = 0; k < 3; ++k) onrm[3 * o + k] = nrm[3 * i + k];
        if (cov)
            for (int k = 0; k < 9; ++k) ocov[9 * o + k] = cov[9 * i + k];
    }
}
__global__ void k_vox_heads(const uint64_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ flags) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// one thread per voxel head: sequential sums in double over the voxel's points, which the stable sort left in
// ascending index order (= the insertion order of the reference's accumulator)
__global__ void k_vox_reduce(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t n,
                             const uint32_t* __restrict__ heads, const uint32_t* __restrict__ vox_id,
                             const double* __restrict__ xyz, const double* __restrict__ nrm, const double* __restrict__ cov,
                             int64_t base, double* __restrict__ oxyz, double* __restrict__ onrm, double* __restrict__ ocov) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n || !heads[i]) return;
    const uint64_t key = keys[i];
    double p[3] = {0, 0, 0}, nn[3] = {0, 0, 0}, cc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int cnt = 0;
    for (int64_t j = i; j < n && keys[j] == key; ++j) {
        const size_t s = vals[j];
        p[0] += xyz[3 * s];
        p[1] += xyz[3 * s + 1];
        p[2] += xyz[3 * s + 2];
        if (nrm) {
            const double a = nrm[3 * s], b = nrm[3 * s + 1], c = nrm[3 * s + 2];
            if (!(a != a) && !(b != b) && !(c != c)) {
                nn[0] += a;
                nn[1] += b;
                nn[2] += c;
            }
        }
        if (cov)
            for (int k = 0; k < 9; ++k) cc[k] += cov[9 * s + k];
        ++cnt;
    }
    const size_t o = (size_t)base + vox_id[i];
    const double dc = (double)cnt;
    for (int k = 0; k < 3; ++k) oxyz[3 * o + k] = p[k] / dc;
    if (nrm) {
        double a[3] = {nn[0] / dc, nn[1] / dc, nn[2] / dc};
        double u = a[0] * a[0];
        double v = a[1] * a[1];
        double z2 = u + v;
        u = a[2] * a[2];
        z2 = z2 + u;
        if (z2 > 0.0) {   // Eigen normalized(): the zero vector stays zero
            const double r = sqrt(z2);
            a[0] = a[0] / r;
            a[1] = a[1] / r;
            a[2] = a[2] / r;
        }
        for (int k = 0; k < 3; ++k) onrm[3 * o + k] = a[k];
    }
    if (cov)
        for (int k = 0; k < 9; ++k) ocov[9 * o + k] = cc[k] / dc;
}
