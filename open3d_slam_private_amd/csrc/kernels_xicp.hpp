// kernels_xicp.hpp -- R8x: first-iteration localizability analysis kernels
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once

// ---- R8x first-iteration analysis (ICP.cpp:2187-2444): matched pairs -> data frame, centre, alignment sums ----
// Vectors in fp32 with one rounding per operation (numeric contract), sums in fp64.
__device__ __forceinline__ float3 xicp_to_data_frame_point(const float* Trd, const float3 p) {
    const float q0 = p.x - Trd[3], q1 = p.y - Trd[7], q2 = p.z - Trd[11];
    float3 r;
    float a0, a1, a2, sacc;
    a0 = Trd[0] * q0; a1 = Trd[4] * q1; a2 = Trd[8] * q2; sacc = a0 + a1; r.x = sacc + a2;
    a0 = Trd[1] * q0; a1 = Trd[5] * q1; a2 = Trd[9] * q2; sacc = a0 + a1; r.y = sacc + a2;
    a0 = Trd[2] * q0; a1 = Trd[6] * q1; a2 = Trd[10] * q2; sacc = a0 + a1; r.z = sacc + a2;
    return r;
}
__device__ __forceinline__ float3 xicp_to_data_frame_vec(const float* Trd, const float x, const float y, const float z) {
    float3 r;
    float a0, a1, a2, sacc;
    a0 = Trd[0] * x; a1 = Trd[4] * y; a2 = Trd[8] * z; sacc = a0 + a1; r.x = sacc + a2;
    a0 = Trd[1] * x; a1 = Trd[5] * y; a2 = Trd[9] * z; sacc = a0 + a1; r.y = sacc + a2;
    a0 = Trd[2] * x; a1 = Trd[6] * y; a2 = Trd[10] * z; sacc = a0 + a1; r.z = sacc + a2;
    return r;
}

template <int NV>
__device__ __forceinline__ void xicp_block_add(double* v, double* dst) {
    __shared__ double red[4][NV];
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < NV) {
        const double t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (t != 0.0) unsafeAtomicAdd(&dst[threadIdx.x], t);
    }
}

__global__ void __launch_bounds__(256)
k_xicp_center(const float4* __restrict__ src, int64_t n, const IterState* __restrict__ it, const int* __restrict__ pos,
              const float* __restrict__ w, XicpState* __restrict__ xs) {
    if (it->done || it->xicp_stage != 2) return;
    const Xf T = load_xf(it);
    float Trd[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) Trd[k] = it->xicp_Trd[k];
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (pos[i] < 0 || w[i] == 0.f) continue;
        const float4 s = src[i];
        const float3 ps = xicp_to_data_frame_point(Trd, xf_point(T, s.x, s.y, s.z));
        v[0] += (double)ps.x;
        v[1] += (double)ps.y;
        v[2] += (double)ps.z;
        v[3] += 1.0;
    }
    xicp_block_add<4>(v, xs->center);
}

__global__ void __launch_bounds__(256)
k_xicp_detect(const float4* __restrict__ src, int64_t n, const IterState* __restrict__ it, const int* __restrict__ pos,
              const float* __restrict__ w, const float4* __restrict__ tgt_nrm, XicpState* __restrict__ xs) {
    if (it->done || it->xicp_stage != 2) return;
    const Xf T = load_xf(it);
    float Trd[12], vr[9], vt[9];
#pragma unroll
    for (int k = 0; k < 12; ++k) Trd[k] = it->xicp_Trd[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        vr[k] = xs->vr[k];
        vt[k] = xs->vt[k];
    }
    const double cnt = xs->center[3];
    float c[3] = {0.f, 0.f, 0.f};
    if (cnt > 0.0) {
        c[0] = (float)(xs->center[0] / cnt);
        c[1] = (float)(xs->center[1] / cnt);
        c[2] = (float)(xs->center[2] / cnt);
    }
    const float cos_min = it->xicp_cos_min, cos_strong = it->xicp_cos_strong;
    double v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = 0.0;   // comb[0..5], high[0..5]
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = pos[i];
        if (j < 0 || w[i] == 0.f) continue;
        const float4 s = src[i];
        float3 ps = xicp_to_data_frame_point(Trd, xf_point(T, s.x, s.y, s.z));
        ps.x = ps.x - c[0];
        ps.y = ps.y - c[1];
        ps.z = ps.z - c[2];
        const float4 nr = tgt_nrm[2 * (size_t)j + 1];   // {point, normal} pairs
        const float3 nn = xicp_to_data_frame_vec(Trd, nr.x, nr.y, nr.z);
        float cr[3];
        float u, q;
        u = ps.y * nn.z; q = ps.z * nn.y; cr[0] = u - q;
        u = ps.z * nn.x; q = ps.x * nn.z; cr[1] = u - q;
        u = ps.x * nn.y; q = ps.y * nn.x; cr[2] = u - q;
        float a = cr[0] * cr[0], b2 = cr[1] * cr[1];
        float s2 = a + b2;
        a = cr[2] * cr[2];
        s2 = s2 + a;
        const float nrm = sqrtf(s2);
        if (!(nrm < 1.0f)) {
            cr[0] = cr[0] / nrm;
            cr[1] = cr[1] / nrm;
            cr[2] = cr[2] / nrm;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a0 = cr[0] * vr[3 * k], a1 = cr[1] * vr[3 * k + 1], a2 = cr[2] * vr[3 * k + 2];
            float sacc = a0 + a1;
            const float ar = fabsf(sacc + a2);
            a0 = nn.x * vt[3 * k];
            a1 = nn.y * vt[3 * k + 1];
            a2 = nn.z * vt[3 * k + 2];
            sacc = a0 + a1;
            const float at = fabsf(sacc + a2);
            if (ar > cos_min) v[k] += (double)ar;
            if (ar > cos_strong) v[6 + k] += (double)ar;
            if (at > cos_min) v[3 + k] += (double)at;
            if (at > cos_strong) v[9 + k] += (double)at;
        }
    }
    xicp_block_add<12>(v, xs->comb);   // comb[6] and high[6] are contiguous
}
