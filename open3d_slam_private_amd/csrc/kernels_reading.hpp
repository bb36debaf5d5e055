// kernels_reading.hpp -- reading preparation (R2): T0, Morton keys, pre-transform
// Part of the single translation unit reg_core.hip (included there, in this order; not a standalone header).
#pragma once

// =================================================================================================
// kernels: reading preparation (R2)
// =================================================================================================

// Reading-preparation state computed on the device (no host round trip between the centroid reduction and the
// kernels that need it): centroid of the reading and T0 = T_refIn_refMean^-1 * T_init * T_readIn_readMean.
struct PrepState {
    float c_read[3];
    float pad;
    float T0[16];   // row-major
};
// Reading centroid (integer sums, NC1; or the given global centroid of a multi-GPU reading) and the pre-transform
// T0 = T_refIn_refMean^-1 * T_init * T_readIn_readMean (ICP.cpp:966-984).  Evaluated redundantly by every thread of
// k_prepare_source (uniform inputs, ~150 flops): a separate one-thread kernel cost a launch (2 us + a 4-6 us gap while
// the host enqueued the next one) in front of every registration.
struct PrepArgs {
    const unsigned long long* sums;   // 3 fixed-point sums (2^16 per unit)
    unsigned long long* sums_to_clear;   // the slot the NEXT registration accumulates into (or null)
    int64_t n;                        // points behind the sums (the whole reading)
    float3 c_ref;
    Xf4 T_init;
    int centre, use_override;
    float3 c_override;
    PrepState* out;
    PrepState* host_out;
};
__device__ __forceinline__ void prep_compute(const PrepArgs& a, float* c /*[3]*/, float* T0 /*[16]*/) {
    c[0] = c[1] = c[2] = 0.f;
    if (a.centre) {
        if (a.use_override) {
            c[0] = a.c_override.x; c[1] = a.c_override.y; c[2] = a.c_override.z;
        } else {
            for (int k = 0; k < 3; ++k) c[k] = (float)((double)(long long)a.sums[k] / (65536.0 * (double)a.n));
        }
    }
    float A[16], B[16], tmp[16];
    m4_identity(A);
    m4_identity(B);
    if (a.centre) {
        A[3] = -a.c_ref.x; A[7] = -a.c_ref.y; A[11] = -a.c_ref.z;   // T_refIn_refMean^-1
        B[3] = c[0]; B[7] = c[1]; B[11] = c[2];                     // T_readIn_readMean
        m4_mul(A, a.T_init.m, tmp);
        m4_mul(tmp, B, T0);
    } else {
        for (int i = 0; i < 16; ++i) T0[i] = a.T_init.m[i];
    }
}

// Morton key of the bin the (pre-transformed) reading point falls into: neighbouring lanes then search
// neighbouring bins (speed only -- results are reported in the caller's order).
__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// Morton key of a reading point in the reading's OWN frame (cells of edge `cell`, anchored at the first point, 10
// bits per axis): a rigid transform keeps neighbours together, so the order is computed once per reading
// (reg_set_source) and serves every initial guess.  Speed only -- results are reported in the caller's order.
__global__ void k_source_keys(const float* __restrict__ xyz, int64_t stride, int64_t n, float inv_cell,
                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = xyz + i * stride;
    const float ax = xyz[0], ay = xyz[1], az = xyz[2];
    const uint32_t bx = (uint32_t)fminf(fmaxf(floorf((p[0] - ax) * inv_cell) + 512.f, 0.f), 1023.f);
    const uint32_t by = (uint32_t)fminf(fmaxf(floorf((p[1] - ay) * inv_cell) + 512.f, 0.f), 1023.f);
    const uint32_t bz = (uint32_t)fminf(fmaxf(floorf((p[2] - az) * inv_cell) + 512.f, 0.f), 1023.f);
    keys[i] = spread10(bx) | (spread10(by) << 1) | (spread10(bz) << 2);
    vals[i] = (uint32_t)i;
}

// reading' = T0 * (p - c_read), normals' = R0 * n  (ICP.cpp:966-984); slot i holds input point perm[i]
// Also clears the per-registration scratch (level hints, trimmed-quantile histograms, accumulator replicas), so the
// registration needs no memset launches.
// Initial iteration state handed over as a kernel argument (dst == null: the caller copies the state itself).
struct StateInit {
    IterState init;
    IterState* dst;
};

__global__ void k_prepare_source(const float* __restrict__ xyz, int64_t stride, const float* __restrict__ nrm,
                                 int64_t nrm_stride, int64_t n, PrepArgs pa, int centre,
                                 const uint32_t* __restrict__ perm, float4* __restrict__ out_xyz,
                                 float4* __restrict__ out_nrm, uint8_t* __restrict__ hint, uint32_t* __restrict__ hist,
                                 double* __restrict__ acc, int n_acc, const StateInit si,
                                 float4* __restrict__ cache) {
    float c[3], T0f[16];
    prep_compute(pa, c, T0f);
    if (blockIdx.x == 0) {
        if (si.dst) {
            static_assert(sizeof(IterState) % 4 == 0, "word copy");
            const uint32_t* w = reinterpret_cast<const uint32_t*>(&si.init);
            for (int k = threadIdx.x; k < (int)(sizeof(IterState) / 4); k += blockDim.x)
                reinterpret_cast<uint32_t*>(si.dst)[k] = w[k];
        }
        for (int k = threadIdx.x; k < 3 * 2048; k += blockDim.x) hist[k] = 0u;
        for (int k = threadIdx.x; k < n_acc; k += blockDim.x) acc[k] = 0.0;
        if (threadIdx.x < 3 && pa.sums_to_clear) pa.sums_to_clear[threadIdx.x] = 0ull;
        if (threadIdx.x == 0) {
            for (int k = 0; k < 3; ++k) pa.out->c_read[k] = c[k];
            for (int i = 0; i < 16; ++i) pa.out->T0[i] = T0f[i];
            if (pa.host_out) {   // mapped pinned copy for the final composition on the host (a D2H memcpy costs ~50 us of host time)
                for (int k = 0; k < 3; ++k) pa.host_out->c_read[k] = c[k];
                for (int i = 0; i < 16; ++i) pa.host_out->T0[i] = T0f[i];
                __threadfence_system();
            }
        }
    }
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    hint[i] = 0;
    if (cache) cache[i] = make_float4(0.f, 0.f, 0.f, -1.f);   // no anchor yet: the first search of the registration sets it
    // R3 (TransformationsImpl.cpp:73-76): a pre-transform whose rotation block is off by |1 - det| > 1e-3 moves the POINTS
    // with the re-orthogonalised copy; the normals and the composed result keep the matrix as given
    float T0c[16];
    rigid_correct(T0f, T0c);
    Xf T0, T0p;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        T0.m[k] = T0f[k];
        T0p.m[k] = T0c[k];
    }
    const float cx = c[0], cy = c[1], cz = c[2];
    const int64_t src = perm ? (int64_t)perm[i] : i;
    const float* p = xyz + src * stride;
    float x = p[0], y = p[1], z = p[2];
    if (centre) {
        x = x - cx;
        y = y - cy;
        z = z - cz;
        const float3 q = xf_point(T0p, x, y, z);
        x = q.x; y = q.y; z = q.z;
    }
    out_xyz[i] = make_float4(x, y, z, 1.f);
    if (out_nrm) {
        const float* q = nrm + src * nrm_stride;
        float3 r = make_float3(q[0], q[1], q[2]);
        if (centre) r = xf_rot(T0, r.x, r.y, r.z);
        out_nrm[i] = make_float4(r.x, r.y, r.z, 0.f);
    }
}

__global__ void k_pack_cov(const float* __restrict__ cov, int64_t n, const uint32_t* __restrict__ perm,
                           float4* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = cov + (perm ? (int64_t)perm[i] : i) * 6;
    out[2 * i] = make_float4(q[0], q[1], q[2], q[3]);
    out[2 * i + 1] = make_float4(q[4], q[5], 0.f, 0.f);
}

