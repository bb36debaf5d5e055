// kernels_normals.hpp -- surface normals / covariances by exact k-NN + PCA
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once

// =================================================================================================
// Next row: surface normals / covariances by exact k-NN + PCA on the voxel-bin table
//   libpointmatcher/pointmatcher/DataPointsFilters/SurfaceNormal.cpp:152-252 (self k-NN incl. the point itself,
//   mean, C = NN NN^T, eigenvector of the smallest eigenvalue, clamp to [-1,1]);
//   orientation towards the sensor: open3d_slam/src/CloudRegistration.cpp:37.
// 16 lanes per point.  Per radius level the group gathers every point of the bin box (within max_dist) into an
// LDS list, then extracts the k smallest (d2, original index) one by one; the k-th distance <= rho^2 proves the
// list held every closer point (same exactness argument as the 1-NN search).
// =================================================================================================
constexpr int kPcaGroup = 16;
// candidates per point and level held in LDS (8 B each): a template parameter of the kernel -- 128 (25 KB per workgroup,
// 6 workgroups per CU) is enough for small k, 256 (41 KB, 3 per CU) keeps the rescans rare for k up to 32
constexpr int kPcaMaxK = 32;

// Calls f(j, target point j, d2) on the lanes of one 16-lane group for every target point inside the bin box of
// level l around p that lies within max_dist.
template <class F>
__device__ __forceinline__ void pca_scan_box(const Grid& g, const float3 p, int l, int sub, int gbase, F&& f) {
    const float rb = g.rho_box[l];
    const int lox = (int)fminf(fmaxf(bin_coord_f(p.x - rb, g.ox, g.inv_c), 0.f), g.dimx - 1.f);
    const int loy = (int)fminf(fmaxf(bin_coord_f(p.y - rb, g.oy, g.inv_c), 0.f), g.dimy - 1.f);
    const int loz = (int)fminf(fmaxf(bin_coord_f(p.z - rb, g.oz, g.inv_c), 0.f), g.dimz - 1.f);
    const int hix = (int)fminf(fmaxf(bin_coord_f(p.x + rb, g.ox, g.inv_c), 0.f), g.dimx - 1.f);
    const int hiy = (int)fminf(fmaxf(bin_coord_f(p.y + rb, g.oy, g.inv_c), 0.f), g.dimy - 1.f);
    const int hiz = (int)fminf(fmaxf(bin_coord_f(p.z + rb, g.oz, g.inv_c), 0.f), g.dimz - 1.f);
    const int ny = hiy - loy + 1, nz = hiz - loz + 1;
    const int bx0 = lox >> kBrickLog2;
    const int nbx = (hix >> kBrickLog2) - bx0 + 1;
    const int nrow = nbx * ny;
    const int64_t total = (int64_t)nrow * nz;
    for (int64_t base = 0; base < total; base += kPcaGroup) {
        uint32_t s = 0, e = 0;
        const int64_t t = base + sub;
        if (t < total) {
            const int iz = (int)(t / nrow), rem = (int)(t - (int64_t)iz * nrow);
            const int iy = rem / nbx, ix = rem - iy * nbx;
            const int bx = bx0 + ix, cy = loy + iy, cz = loz + iz;
            const int bid = brick_lookup(g, bx, cy >> kBrickLog2, cz >> kBrickLog2);
            if (bid >= 0) {
                const int x0 = max(lox, bx << kBrickLog2) & (kBrickDim - 1);
                const int x1 = min(hix, (bx << kBrickLog2) + kBrickDim - 1) & (kBrickDim - 1);
                const uint32_t* cs = g.cell_start + (size_t)bid * kBrickCells +
                                     (((cz & (kBrickDim - 1)) << (2 * kBrickLog2)) |
                                      ((cy & (kBrickDim - 1)) << kBrickLog2));
                s = cs[x0];
                e = cs[x1 + 1];
            }
        }
        unsigned mask = (unsigned)((__ballot(e > s) >> gbase) & 0xffffull);
        while (mask) {
            const int it = __ffs((int)mask) - 1;
            mask &= mask - 1;
            const uint32_t si = (uint32_t)__shfl((int)s, gbase + it);
            const uint32_t ei = (uint32_t)__shfl((int)e, gbase + it);
            for (uint32_t j = si + (uint32_t)sub; j < ei; j += kPcaGroup) {
                const float4 tpt = g.pts[j];
                const float dx = p.x - tpt.x, dy = p.y - tpt.y, dz = p.z - tpt.z;
                float a = dx * dx;
                float b = dy * dy;
                float d2 = a + b;
                a = dz * dz;
                d2 = d2 + a;
                if (d2 <= g.max_d2) f(j, tpt, d2);
            }
        }
    }
}

// Per-point result of the neighbourhood kernel, consumed by k_pca_finish (one thread per point: the eigen-decomposition
// ran on ONE lane of a 16-lane group before -- 6 % lane utilisation for the most expensive arithmetic of the kernel).
struct PcaMoments {
    float mean[3];
    float C[6];      // xx xy xz yy yz zz of sum (q - mean)(q - mean)^T
    float m;         // neighbours found
    float p[3];      // the point itself
    float max_d2;    // largest squared distance of a neighbour from the mean (densities)
    uint32_t idx;    // original index
    uint32_t pad;
};
static_assert(sizeof(PcaMoments) == 64, "one 64-byte record per point");

template <int kPcaCap>
__global__ void __launch_bounds__(256)
k_knn_pca(Grid g, const float* __restrict__ raw_xyz, int64_t raw_stride, int64_t n, int k, int start_level,
          int32_t* __restrict__ ids_out, uint32_t* __restrict__ n_overflow, PcaMoments* __restrict__ mom) {
    constexpr int GP = 256 / kPcaGroup;   // points per workgroup
    __shared__ float l_d2[GP][kPcaCap];
    __shared__ uint32_t l_idx[GP][kPcaCap];
    __shared__ uint32_t l_cnt[GP];
    __shared__ uint32_t nb_idx[GP][kPcaMaxK];
    __shared__ float nb_xyz[GP][kPcaMaxK][3];
    const int grp = threadIdx.x / kPcaGroup, sub = threadIdx.x & (kPcaGroup - 1);
    const int gbase = (int)(threadIdx.x & 63) & ~(kPcaGroup - 1);
    const int64_t q = blockIdx.x * (int64_t)GP + grp;
    if (q >= n) return;   // whole groups leave together; nothing below synchronises across groups
    const float4 me = g.pts[q];
    const float3 p = make_float3(me.x, me.y, me.z);
    const uint32_t my_idx = __float_as_uint(me.w);
    int m = 0;
    bool overflow = false;
    for (int l = min(start_level, g.n_levels - 1); l < g.n_levels; ++l) {
        if (sub == 0) l_cnt[grp] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        pca_scan_box(g, p, l, sub, gbase, [&](uint32_t, const float4& tpt, float d2) {
            const uint32_t slot = atomicAdd(&l_cnt[grp], 1u);
            if (slot < (uint32_t)kPcaCap) {
                l_d2[grp][slot] = d2;
                l_idx[grp][slot] = __float_as_uint(tpt.w);
            }
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint32_t cnt = l_cnt[grp];
        const bool listed = cnt <= (uint32_t)kPcaCap;
        if (!listed) overflow = true;   // too many candidates for LDS: every extraction round rescans the box
        // extract the k smallest (d2, idx), ascending
        float last_d2 = -1.f;
        uint32_t last_idx = 0;
        m = 0;
        for (int r = 0; r < k; ++r) {
            float bd = INFINITY;
            uint32_t bi = 0xffffffffu;
            if (listed) {
                for (uint32_t t2 = sub; t2 < cnt; t2 += kPcaGroup) {
                    const float d = l_d2[grp][t2];
                    const uint32_t ix = l_idx[grp][t2];
                    const bool after = r == 0 || d > last_d2 || (d == last_d2 && ix > last_idx);
                    if (after && (d < bd || (d == bd && ix < bi))) {
                        bd = d;
                        bi = ix;
                    }
                }
            } else {
                pca_scan_box(g, p, l, sub, gbase, [&](uint32_t, const float4& tpt, float d) {
                    const uint32_t ix = __float_as_uint(tpt.w);
                    const bool after = r == 0 || d > last_d2 || (d == last_d2 && ix > last_idx);
                    if (after && (d < bd || (d == bd && ix < bi))) {
                        bd = d;
                        bi = ix;
                    }
                });
            }
#pragma unroll
            for (int x = 1; x < kPcaGroup; x <<= 1) {
                const float od = __shfl_xor(bd, x);
                const uint32_t oi = (uint32_t)__shfl_xor((int)bi, x);
                if (od < bd || (od == bd && oi < bi)) {
                    bd = od;
                    bi = oi;
                }
            }
            if (bi == 0xffffffffu) break;
            if (sub == 0) nb_idx[grp][r] = bi;
            last_d2 = bd;
            last_idx = bi;
            ++m;
        }
        const float r2 = g.rho[l] * g.rho[l];
        if ((m == k && last_d2 <= r2) || l == g.n_levels - 1) break;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // neighbour coordinates (input frame == table frame: the workspace table is not centred) and ids, in parallel
    for (int r = sub; r < k; r += kPcaGroup) {
        if (r < m) {
            const uint32_t ix = nb_idx[grp][r];
            const float* s = raw_xyz + (size_t)ix * raw_stride;
            nb_xyz[grp][r][0] = s[0];
            nb_xyz[grp][r][1] = s[1];
            nb_xyz[grp][r][2] = s[2];
            if (ids_out) ids_out[(size_t)my_idx * k + r] = (int32_t)ix;
        } else if (ids_out) {
            ids_out[(size_t)my_idx * k + r] = -1;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (sub != 0) return;
    if (overflow && n_overflow) atomicAdd(n_overflow, 1u);   // statistics only: the result is still exact
    // PCA: fp32 sequential sums in neighbour order (numeric contract), eigen-decomposition in fp64
    float mean[3] = {0.f, 0.f, 0.f};
    for (int r = 0; r < m; ++r) {
        mean[0] = mean[0] + nb_xyz[grp][r][0];
        mean[1] = mean[1] + nb_xyz[grp][r][1];
        mean[2] = mean[2] + nb_xyz[grp][r][2];
    }
    const float fm = (float)m;
    if (m > 0) {
        mean[0] = mean[0] / fm;
        mean[1] = mean[1] / fm;
        mean[2] = mean[2] / fm;
    }
    float C[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < m; ++r) {
        const float dx = nb_xyz[grp][r][0] - mean[0], dy = nb_xyz[grp][r][1] - mean[1], dz = nb_xyz[grp][r][2] - mean[2];
        float u;
        u = dx * dx; C[0] = C[0] + u;
        u = dx * dy; C[1] = C[1] + u;
        u = dx * dz; C[2] = C[2] + u;
        u = dy * dy; C[3] = C[3] + u;
        u = dy * dz; C[4] = C[4] + u;
        u = dz * dz; C[5] = C[5] + u;
    }
    float mx = 0.f;
    for (int r = 0; r < m; ++r) {
        const float dx = nb_xyz[grp][r][0] - mean[0], dy = nb_xyz[grp][r][1] - mean[1], dz = nb_xyz[grp][r][2] - mean[2];
        float u = dx * dx;
        float v2 = dy * dy;
        float s2 = u + v2;
        u = dz * dz;
        s2 = s2 + u;
        mx = fmaxf(mx, s2);
    }
    PcaMoments rec;
    rec.mean[0] = mean[0]; rec.mean[1] = mean[1]; rec.mean[2] = mean[2];
    for (int a = 0; a < 6; ++a) rec.C[a] = C[a];
    rec.m = fm;
    rec.p[0] = p.x; rec.p[1] = p.y; rec.p[2] = p.z;
    rec.max_d2 = mx;
    rec.idx = my_idx;
    rec.pad = 0;
    mom[q] = rec;
}

// One thread per point: eigen-decomposition of the scatter matrix and every output of the filter.
__global__ void __launch_bounds__(256)
k_pca_finish(const PcaMoments* __restrict__ mom, int64_t n, float vx, float vy, float vz, int has_vp, int regularise,
             float* __restrict__ normals, float* __restrict__ eigvals, float* __restrict__ covs,
             float* __restrict__ eigvecs, float* __restrict__ densities, float* __restrict__ mean_dists) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PcaMoments rec = mom[i];
    const float* mean = rec.mean;
    const float* C = rec.C;
    const int m = (int)rec.m;
    const float3 p = make_float3(rec.p[0], rec.p[1], rec.p[2]);
    const uint32_t my_idx = rec.idx;
    double M[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]}, V[9], lam[3];
    jacobi_eig_sym3(M, V, lam);
    int o0 = 0, o1 = 1, o2 = 2;
    if (lam[o1] < lam[o0]) { const int t = o0; o0 = o1; o1 = t; }
    if (lam[o2] < lam[o0]) { const int t = o0; o0 = o2; o2 = t; }
    if (lam[o2] < lam[o1]) { const int t = o1; o1 = o2; o2 = t; }
    const double lmax = fabs(lam[o2]);
    int rank = 0;
    for (int a = 0; a < 3; ++a)
        if (lmax > 0 && fabs(lam[a]) > lmax * 3.0 * 1.1920929e-07) ++rank;
    float nv[3] = {0.f, 0.f, 0.f};
    if (m >= 3 && rank + 1 >= 3) {
        const double v[3] = {V[0 * 3 + o0], V[1 * 3 + o0], V[2 * 3 + o0]};
        double sgn = 1.0;
        if (has_vp) {
            const double dot = v[0] * ((double)vx - (double)p.x) + v[1] * ((double)vy - (double)p.y) + v[2] * ((double)vz - (double)p.z);
            if (dot < 0) sgn = -1.0;
        } else {
            int big = 0;
            if (fabs(v[1]) > fabs(v[big])) big = 1;
            if (fabs(v[2]) > fabs(v[big])) big = 2;
            if (v[big] < 0) sgn = -1.0;
        }
        for (int a = 0; a < 3; ++a) {
            const float f = (float)(sgn * v[a]);
            nv[a] = f > 1.f ? 1.f : (f < -1.f ? -1.f : f);
        }
    }
    const size_t oi = (size_t)my_idx;
    normals[3 * oi + 0] = nv[0];
    normals[3 * oi + 1] = nv[1];
    normals[3 * oi + 2] = nv[2];
    const bool degenerate = !(m >= 3 && rank + 1 >= 3);
    if (eigvals) {
        eigvals[3 * oi + 0] = (float)lam[o0];
        eigvals[3 * oi + 1] = (float)lam[o1];
        eigvals[3 * oi + 2] = (float)lam[o2];
    }
    if (eigvecs) {   // "eigVectors": eigenvector k (ascending eigenvalue), components contiguous; zero when degenerate
        const int oo[3] = {o0, o1, o2};
        for (int kk = 0; kk < 3; ++kk)
            for (int a = 0; a < 3; ++a) eigvecs[9 * oi + 3 * kk + a] = degenerate ? 0.f : (float)V[a * 3 + oo[kk]];
    }
    if (densities) {   // utils.h:106-128: m / (4/3 pi r^3), r^2 = largest squared distance of a neighbour from the mean
        float dens = 0.f;
        if (!degenerate) {
            const float mx = rec.max_d2;
            const float tq = (float)(4. / 3.), pi = (float)3.14159265358979323846;
            const float c0 = tq * pi;
            const float r3 = mx * sqrtf(mx);
            const float volume = c0 * r3;
            dens = volume > 0.f ? (float)m / volume : 0.f;
        }
        densities[oi] = dens;
    }
    if (mean_dists) {   // SurfaceNormal.cpp:243-252
        float md = 18446744073709551615.0f;   // (float)std::numeric_limits<std::size_t>::max()
        if (!degenerate) {
            const float dx = p.x - mean[0], dy = p.y - mean[1], dz = p.z - mean[2];
            float u = dx * dx;
            float v2 = dy * dy;
            float s2 = u + v2;
            u = dz * dz;
            s2 = s2 + u;
            md = sqrtf(s2);
        }
        mean_dists[oi] = md;
    }
    if (covs) {
        double Cn[9];
        if (regularise) {
            const int oo[3] = {o0, o1, o2};
            const double w[3] = {1e-3, 1.0, 1.0};
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double t = 0;
                    for (int e2 = 0; e2 < 3; ++e2) t += w[e2] * V[a * 3 + oo[e2]] * V[b * 3 + oo[e2]];
                    Cn[3 * a + b] = t;
                }
        } else {
            const double inv = m > 0 ? 1.0 / (double)m : 0.0;
            const double Cd[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]};
            for (int a = 0; a < 9; ++a) Cn[a] = Cd[a] * inv;
        }
        covs[6 * oi + 0] = (float)Cn[0];
        covs[6 * oi + 1] = (float)Cn[1];
        covs[6 * oi + 2] = (float)Cn[2];
        covs[6 * oi + 3] = (float)Cn[4];
        covs[6 * oi + 4] = (float)Cn[5];
        covs[6 * oi + 5] = (float)Cn[8];
    }
}


// SurfaceNormalDataPointsFilter::smoothNormals (SurfaceNormal.cpp:259-283).  The reference smooths IN PLACE in index order:
// point i reads the already smoothed normals of its lower-indexed neighbours and the original ones of the others (itself
// included).  That recurrence is a DAG over the indices; it is evaluated here as a level-synchronous sweep: in pass p a point
// is computed when every lower-indexed neighbour finished in an EARLIER pass (level < p), with exactly the arithmetic of the
// sequential loop (fp32, neighbour order of the k-NN list, mean / float(n)).  Passes needed = longest dependency chain + 1
// (615 on the reference's 25 k-point scan in scan order, ~20 on unordered clouds).
__global__ void __launch_bounds__(256)
k_smooth_pass(const float* __restrict__ orig, float* cur, const int32_t* __restrict__ ids, int64_t n, int k, int pass,
              int* level, unsigned int* __restrict__ n_done) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (level[i] >= 0) return;
    const int32_t* row = ids + (size_t)i * k;
    for (int j = 0; j < k; ++j) {
        const int32_t r = row[j];
        if ((int64_t)r >= n) {   // an id beyond the cloud: reported (n_done[1]), never dereferenced
            n_done[1] = 1u;
            return;
        }
        if (r >= 0 && r < i) {
            const int lv = level[r];
            if (lv < 0 || lv >= pass) return;   // not finished before this pass started
        }
    }
    const float c0 = orig[3 * i], c1 = orig[3 * i + 1], c2 = orig[3 * i + 2];
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    int cnt = 0;
    for (int j = 0; j < k; ++j) {
        const int32_t r = row[j];
        if (r < 0) continue;
        const float* src = (r < i) ? cur : orig;
        const float a0 = src[3 * (size_t)r], a1 = src[3 * (size_t)r + 1], a2 = src[3 * (size_t)r + 2];
        float d = c0 * a0;
        float t = c1 * a1;
        d = d + t;
        t = c2 * a2;
        d = d + t;
        if (d > 0.f) {
            m0 = m0 + a0; m1 = m1 + a1; m2 = m2 + a2;
        } else {
            m0 = m0 - a0; m1 = m1 - a1; m2 = m2 - a2;
        }
        ++cnt;
    }
    const float fn = (float)cnt;
    cur[3 * i] = m0 / fn;
    cur[3 * i + 1] = m1 / fn;
    cur[3 * i + 2] = m2 / fn;
    level[i] = pass;
    atomicAdd(n_done, 1u);
}
