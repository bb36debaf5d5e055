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

// R11, reading side (open3d_conversions.cpp:57-118): fp64 AoS -> fp32, same order, nothing dropped
__global__ void k_cast_cloud_f64(const double* __restrict__ xyz, const double* __restrict__ nrm, const double* __restrict__ cov,
                                 int64_t n, float* __restrict__ oxyz, float* __restrict__ onrm, float* __restrict__ ocov) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 3; ++k) oxyz[3 * i + k] = (float)xyz[3 * i + k];
    if (nrm)
        for (int k = 0; k < 3; ++k) onrm[3 * i + k] = (float)nrm[3 * i + k];
    if (cov) {
        const double* c = cov + 9 * i;   // Matrix3d, symmetric: xx xy xz / . yy yz / . . zz
        ocov[6 * i + 0] = (float)c[0];
        ocov[6 * i + 1] = (float)c[1];
        ocov[6 * i + 2] = (float)c[2];
        ocov[6 * i + 3] = (float)c[4];
        ocov[6 * i + 4] = (float)c[5];
        ocov[6 * i + 5] = (float)c[8];
    }
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

// ---- space carving (helpers.cpp:238-283) ----
// keys of the map points inside the subset volume (compacted by o_in), value = map index
__global__ void k_carve_keys(const double* __restrict__ xyz, int64_t m, double inv, const uint32_t* __restrict__ f_in,
                             const uint32_t* __restrict__ o_in, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= m || !f_in[i]) return;
    const long long vx = (long long)floor(xyz[3 * i] * inv) + kVoxOff;
    const long long vy = (long long)floor(xyz[3 * i + 1] * inv) + kVoxOff;
    const long long vz = (long long)floor(xyz[3 * i + 2] * inv) + kVoxOff;
    keys[o_in[i]] = ((uint64_t)vz << (2 * kVoxBits)) | ((uint64_t)vy << kVoxBits) | (uint64_t)vx;
    vals[o_in[i]] = (uint32_t)i;
}
// unique voxel keys and the first sorted position of each (ustart[nu] = n is written by the host)
__global__ void k_carve_unique(const uint64_t* __restrict__ keys, int64_t n, const uint32_t* __restrict__ heads,
                               const uint32_t* __restrict__ uid, uint64_t* __restrict__ ukeys, uint32_t* __restrict__ ustart) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n || !heads[i]) return;
    ukeys[uid[i]] = keys[i];
    ustart[uid[i]] = (uint32_t)i;
}
__global__ void k_carve_rays(const double* __restrict__ scan, int64_t n_scan, double sx, double sy, double sz, double step,
                             double max_ray, double trunc, double min_dot, double inv, const uint64_t* __restrict__ ukeys,
                             const uint32_t* __restrict__ ustart, int64_t nu, const uint32_t* __restrict__ vals,
                             const double* __restrict__ nrm, uint32_t* __restrict__ mark) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_scan) return;
    const double dx = scan[3 * i] - sx, dy = scan[3 * i + 1] - sy, dz = scan[3 * i + 2] - sz;
    double a = dx * dx;
    double b = dy * dy;
    double s2 = a + b;
    a = dz * dz;
    s2 = s2 + a;
    const double length = sqrt(s2);
    if (!(length > 0.0) || !(length < INFINITY)) return;
    const double ux = dx / length, uy = dy / length, uz = dz / length;
    const double reach = fmax(step, fmin(length - trunc, max_ray));
    double distance = 0.0;
    while (distance < reach) {
        double t = distance * ux;
        const double px = t + sx;
        t = distance * uy;
        const double py = t + sy;
        t = distance * uz;
        const double pz = t + sz;
        const double fx = floor(px * inv), fy = floor(py * inv), fz = floor(pz * inv);
        if (fabs(fx) < (double)kVoxOff && fabs(fy) < (double)kVoxOff && fabs(fz) < (double)kVoxOff) {
            const uint64_t key = ((uint64_t)((long long)fz + kVoxOff) << (2 * kVoxBits)) |
                                 ((uint64_t)((long long)fy + kVoxOff) << kVoxBits) | (uint64_t)((long long)fx + kVoxOff);
            int64_t lo = 0, hi = nu;   // first ukeys[lo] >= key
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (ukeys[mid] < key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            if (lo < nu && ukeys[lo] == key) {
                for (uint32_t j = ustart[lo]; j < ustart[lo + 1]; ++j) {
                    const uint32_t id = vals[j];
                    bool rm = true;
                    if (nrm) {
                        double n0 = nrm[3 * (size_t)id], n1 = nrm[3 * (size_t)id + 1], n2 = nrm[3 * (size_t)id + 2];
                        double u = n0 * n0;
                        double v = n1 * n1;
                        double z2 = u + v;
                        u = n2 * n2;
                        z2 = z2 + u;
                        if (z2 > 0.0) {
                            const double r = sqrt(z2);
                            n0 = n0 / r;
                            n1 = n1 / r;
                            n2 = n2 / r;
                        }
                        u = ux * n0;
                        v = uy * n1;
                        double d = u + v;
                        u = uz * n2;
                        d = d + u;
                        rm = fabs(d) > min_dot;
                    }
                    if (rm) mark[id] = 1u;   // idempotent: no atomic needed
                }
            }
        }
        distance += step;
    }
}
__global__ void k_carve_collect(const uint32_t* __restrict__ mark, const uint32_t* __restrict__ offs, int64_t m,
                                int32_t* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= m || !mark[i]) return;
    out[offs[i]] = (int32_t)i;
}
