// host_math.hpp -- host-side (CPU) pieces of the registration path: 4x4 fp32 algebra, the
// 6x6 solve, x -> SE(3), the transformation checkers.  Product code (not the oracle).
//
// Reference behaviour restated (paths relative to the reference tree):
//   solve      libpointmatcher/pointmatcher/ErrorMinimizers/PointToPlane.cpp:112-265
//   x -> T     PointToPlane.cpp:327-381   (angle = atan(|w|), axis = w/|w|, t = x[3..5])
//   checkers   TransformationCheckersImpl.cpp:57-158
//   frames     ICP.cpp:883-890, 966-984, 1345
// All 4x4 matrices in this file are ROW-major float[16]; the C ABI converts from/to column-major.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define O3D_HD __host__ __device__
#else
#define O3D_HD
#endif

// NOTE: this header relies on being compiled with -ffp-contract=off (one rounding per fp32 operation).
namespace o3dreg {

O3D_HD inline void m4_identity(float* T) {
    memset(T, 0, 16 * sizeof(float));
    T[0] = T[5] = T[10] = T[15] = 1.f;
}

// C = A*B with one rounding per operation, k = 0..3 in order (numeric contract NC3).
O3D_HD inline void m4_mul(const float* A, const float* B, float* C) {
    float R[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = A[4 * i] * B[j];
            float t = A[4 * i + 1] * B[4 + j];
            s = s + t;
            t = A[4 * i + 2] * B[8 + j];
            s = s + t;
            t = A[4 * i + 3] * B[12 + j];
            s = s + t;
            R[4 * i + j] = s;
        }
    memcpy(C, R, sizeof(R));
}

O3D_HD inline void m4_transpose(const float* A, float* B) {
    float R[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) R[4 * i + j] = A[4 * j + i];
    memcpy(B, R, sizeof(R));
}

// R3: RigidTransformation::checkParameters / correctParameters (TransformationsImpl.cpp:105-166), row-major 4x4.
// |1 - det(R)| > 1e-3 in fp32 -> Tc = the re-orthogonalised copy the reference applies to the FEATURES (col1, col2
// normalised; newCol0 = col1 x col2; newCol1 = col2 x newCol0; newCol2 = col2; translation kept).  Descriptors keep
// the matrix as given (TransformationsImpl.cpp:83-101).  One rounding per operation; returns true when corrected.
O3D_HD inline bool rigid_correct(const float* T, float* Tc) {
    for (int i = 0; i < 16; ++i) Tc[i] = T[i];
    float m0 = T[5] * T[10], m1 = T[6] * T[9];
    const float c0 = m0 - m1;
    m0 = T[4] * T[10]; m1 = T[6] * T[8];
    const float c1 = m0 - m1;
    m0 = T[4] * T[9]; m1 = T[5] * T[8];
    const float c2 = m0 - m1;
    float a = T[0] * c0, b = T[1] * c1;
    float det = a - b;
    a = T[2] * c2;
    det = det + a;
    const float dev = 1.0f - det;
    if (!(fabsf(dev) > 0.001f)) return false;
    float n1[3] = {T[1], T[5], T[9]}, n2[3] = {T[2], T[6], T[10]}, n0[3], mv[3];
    for (int k = 0; k < 2; ++k) {   // Eigen 3.3 normalized(): z > 0 ? v / sqrt(z) : v
        float* v = k == 0 ? n1 : n2;
        float p = v[0] * v[0], q = v[1] * v[1];
        float z = p + q;
        p = v[2] * v[2];
        z = z + p;
        if (z > 0.f) {
            const float s = sqrtf(z);
            v[0] = v[0] / s; v[1] = v[1] / s; v[2] = v[2] / s;
        }
    }
    float u, v;
    u = n1[1] * n2[2]; v = n1[2] * n2[1]; n0[0] = u - v;
    u = n1[2] * n2[0]; v = n1[0] * n2[2]; n0[1] = u - v;
    u = n1[0] * n2[1]; v = n1[1] * n2[0]; n0[2] = u - v;
    u = n2[1] * n0[2]; v = n2[2] * n0[1]; mv[0] = u - v;
    u = n2[2] * n0[0]; v = n2[0] * n0[2]; mv[1] = u - v;
    u = n2[0] * n0[1]; v = n2[1] * n0[0]; mv[2] = u - v;
    for (int r = 0; r < 3; ++r) {
        Tc[4 * r + 0] = n0[r];
        Tc[4 * r + 1] = mv[r];
        Tc[4 * r + 2] = n2[r];
    }
    return true;
}

O3D_HD inline bool m4_is_finite(const float* T) {
    for (int i = 0; i < 16; ++i)
        if (!(T[i] - T[i] == 0.0f)) return false;  // NaN or +-inf
    return true;
}

// ---------------------------------------------------------------------------------------------
// 6x6 symmetric solve, fp64.  Full rank -> LDL^T with diagonal pivoting; rank deficient (pivot test
// with the fp32 threshold size*eps_f32 of fullPivHouseholderQr::isInvertible) -> minimum-norm
// solution through a Jacobi eigen-decomposition.  Returns the numerical rank.
// ---------------------------------------------------------------------------------------------
O3D_HD inline void jacobi_eig_sym(int n, double* A, double* V, double* lam) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) lam[i] = A[i * n + i];
}

// Fixed-size 3x3 variant of the cyclic Jacobi iteration above: constant indices keep A and V in registers on the
// device (the generic routine indexes at run time, which puts its arrays into scratch memory: 40 us per call on one
// lane), and the sweep loop stops as soon as the off-diagonal mass is below fp64 resolution relative to the diagonal
// (quadratic convergence: further sweeps would not change a bit of the result).
template <int P, int Q>
O3D_HD inline void jacobi3_rotate(double* A, double* V) {
    const double apq = A[3 * P + Q];
    if (fabs(apq) < 1e-300) return;
    const double theta = (A[3 * Q + Q] - A[3 * P + P]) / (2.0 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double akp = A[3 * k + P], akq = A[3 * k + Q];
        A[3 * k + P] = c * akp - s * akq;
        A[3 * k + Q] = s * akp + c * akq;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double apk = A[3 * P + k], aqk = A[3 * Q + k];
        A[3 * P + k] = c * apk - s * aqk;
        A[3 * Q + k] = s * apk + c * aqk;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double vkp = V[3 * k + P], vkq = V[3 * k + Q];
        V[3 * k + P] = c * vkp - s * vkq;
        V[3 * k + Q] = s * vkp + c * vkq;
    }
}
O3D_HD inline void jacobi_eig_sym3(double* A, double* V, double* lam) {
    V[0] = 1.0; V[1] = 0.0; V[2] = 0.0;
    V[3] = 0.0; V[4] = 1.0; V[5] = 0.0;
    V[6] = 0.0; V[7] = 0.0; V[8] = 1.0;
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        const double dg = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
        if (off < 1e-300 || off <= 1e-34 * dg) break;
        jacobi3_rotate<0, 1>(A, V);
        jacobi3_rotate<0, 2>(A, V);
        jacobi3_rotate<1, 2>(A, V);
    }
    lam[0] = A[0];
    lam[1] = A[4];
    lam[2] = A[8];
}

// rel_thr: eigenvalues <= rel_thr * max are treated as zero.
O3D_HD inline int solve_sym6(const double* H, const double* g, double* x, double rel_thr) {
    double M[36], V[36], lam[6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) M[6 * i + j] = 0.5 * (H[6 * i + j] + H[6 * j + i]);
    jacobi_eig_sym(6, M, V, lam);
    double lmax = 0;
    for (int k = 0; k < 6; ++k) lmax = fmax(lmax, fabs(lam[k]));
    int rank = 0;
    for (int i = 0; i < 6; ++i) x[i] = 0;
    for (int k = 0; k < 6; ++k) {
        if (!(fabs(lam[k]) > lmax * rel_thr)) continue;
        ++rank;
        double vb = 0;
        for (int i = 0; i < 6; ++i) vb += V[6 * i + k] * g[i];
        vb /= lam[k];
        for (int i = 0; i < 6; ++i) x[i] += V[6 * i + k] * vb;
    }
    return rank;
}

O3D_HD inline int solve6_p2pl(const float* A, const float* b, float* x) {
    double H[36], g[6], xd[6];
    for (int i = 0; i < 36; ++i) H[i] = A[i];
    for (int i = 0; i < 6; ++i) g[i] = b[i];
    const int rank = solve_sym6(H, g, xd, 6.0 * 1.1920929e-07);
    for (int i = 0; i < 6; ++i) x[i] = (float)xd[i];
    return rank;
}

// ---- R8x: X-ICP localizability (ICP.cpp:1580-1591, 2187-2444; PointToPlane.cpp:459-505) --------------------
// Eigenvectors of a symmetric 3x3 block in DESCENDING eigenvalue order (the order of JacobiSVD's U for a PSD
// matrix); V[3*r+k] = component r of eigenvector k.
O3D_HD inline void eig3_desc(const double* S, double* V) {
    double M[9], W[9], l[3];
    for (int i = 0; i < 9; ++i) M[i] = S[i];
    jacobi_eig_sym3(M, W, l);
    int o0 = 0, o1 = 1, o2 = 2;
    if (l[o1] > l[o0]) { const int t = o0; o0 = o1; o1 = t; }
    if (l[o2] > l[o0]) { const int t = o0; o0 = o2; o2 = t; }
    if (l[o2] > l[o1]) { const int t = o1; o1 = o2; o2 = t; }
    for (int r = 0; r < 3; ++r) {
        V[3 * r + 0] = W[3 * r + o0];
        V[3 * r + 1] = W[3 * r + o1];
        V[3 * r + 2] = W[3 * r + o2];
    }
}

// rotation (rows/cols 0-2) and translation (3-5) eigenvectors of the fp32 system matrix (row-major 6x6)
O3D_HD inline void xicp_eigvecs(const float* A, double* Vr, double* Vt) {
    double Sr[9], St[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Sr[3 * i + j] = 0.5 * ((double)A[6 * i + j] + (double)A[6 * j + i]);
            St[3 * i + j] = 0.5 * ((double)A[6 * (i + 3) + j + 3] + (double)A[6 * (j + 3) + i + 3]);
        }
    eig3_desc(Sr, Vr);
    eig3_desc(St, Vt);
}

// Equality-constrained solve: no update along the non-localizable eigen-directions (flags[k] == 0).  Null-space form
// of the reference's (6+c)x(6+c) KKT system: x = Z (Z^T A Z)^-1 Z^T b with Z = the localizable eigenvectors.
O3D_HD inline int solve6_xicp(const float* A, const float* b, const int* flags, float* x) {
    double Vr[9], Vt[9], Z[36];
    xicp_eigvecs(A, Vr, Vt);
    int m = 0;
    for (int k = 0; k < 3; ++k)
        if (flags[k]) {
            for (int r = 0; r < 6; ++r) Z[6 * r + m] = r < 3 ? Vr[3 * r + k] : 0.0;
            ++m;
        }
    for (int k = 0; k < 3; ++k)
        if (flags[3 + k]) {
            for (int r = 0; r < 6; ++r) Z[6 * r + m] = r >= 3 ? Vt[3 * (r - 3) + k] : 0.0;
            ++m;
        }
    for (int i = 0; i < 6; ++i) x[i] = 0.f;
    if (m == 0) return 0;
    double AZ[36], M[36], V[36], lam[6], g[6];
    for (int i = 0; i < 6; ++i)
        for (int c = 0; c < m; ++c) {
            double t = 0;
            for (int j = 0; j < 6; ++j) t += 0.5 * ((double)A[6 * i + j] + (double)A[6 * j + i]) * Z[6 * j + c];
            AZ[6 * i + c] = t;
        }
    for (int a = 0; a < m; ++a) {
        for (int c = 0; c < m; ++c) {
            double t = 0;
            for (int i = 0; i < 6; ++i) t += Z[6 * i + a] * AZ[6 * i + c];
            M[m * a + c] = t;
        }
        double t = 0;
        for (int i = 0; i < 6; ++i) t += Z[6 * i + a] * (double)b[i];
        g[a] = t;
    }
    for (int a = 0; a < m; ++a)
        for (int c = a + 1; c < m; ++c) {
            const double v = 0.5 * (M[m * a + c] + M[m * c + a]);
            M[m * a + c] = v;
            M[m * c + a] = v;
        }
    jacobi_eig_sym(m, M, V, lam);
    double lmax = 0;
    for (int k = 0; k < m; ++k) lmax = fmax(lmax, fabs(lam[k]));
    const double thr = lmax * (double)m * 1.1920929e-07;
    double y[6] = {0, 0, 0, 0, 0, 0};
    int rank = 0;
    for (int k = 0; k < m; ++k) {
        if (!(fabs(lam[k]) > thr)) continue;
        ++rank;
        double vb = 0;
        for (int a = 0; a < m; ++a) vb += V[m * a + k] * g[a];
        vb /= lam[k];
        for (int a = 0; a < m; ++a) y[a] += V[m * a + k] * vb;
    }
    for (int i = 0; i < 6; ++i) {
        double t = 0;
        for (int c = 0; c < m; ++c) t += Z[6 * i + c] * y[c];
        x[i] = (float)t;
    }
    return rank;
}

// x = [rx ry rz tx ty tz] -> row-major 4x4, fp32, one rounding per op (NC10).
O3D_HD inline void x_to_T(const float* x, float* T) {
    float a = x[0] * x[0], b = x[1] * x[1], c = x[2] * x[2];
    float s = a + b;
    s = s + c;
    const float nrm = sqrt((float)s);
    const float angle = atanf(nrm);
    float ax[3] = {x[0], x[1], x[2]};
    const float w = fmax(fabs(x[0]), fmax(fabs(x[1]), fabs(x[2])));
    const float y0 = x[0] / w, y1 = x[1] / w, y2 = x[2] / w;
    float z = y0 * y0, z1 = y1 * y1;
    z = z + z1;
    z1 = y2 * y2;
    z = z + z1;
    if (z > 0.f) {
        float d = sqrt((float)z);
        d = d * w;
        ax[0] = x[0] / d;
        ax[1] = x[1] / d;
        ax[2] = x[2] / d;
    }
    const float sn = sinf(angle), cs = cosf(angle);
    float sa0 = sn * ax[0], sa1 = sn * ax[1], sa2 = sn * ax[2];
    const float c1 = 1.0f - cs;
    float ca0 = c1 * ax[0], ca1 = c1 * ax[1], ca2 = c1 * ax[2];
    float R[9];
    float t;
    t = ca0 * ax[1];
    R[1] = t - sa2;
    R[3] = t + sa2;
    t = ca0 * ax[2];
    R[2] = t + sa1;
    R[6] = t - sa1;
    t = ca1 * ax[2];
    R[5] = t - sa0;
    R[7] = t + sa0;
    t = ca0 * ax[0];
    R[0] = t + cs;
    t = ca1 * ax[1];
    R[4] = t + cs;
    t = ca2 * ax[2];
    R[8] = t + cs;
    bool bad = false;
    for (int i = 0; i < 9; ++i) bad |= (R[i] != R[i]);
    for (int i = 3; i < 6; ++i) bad |= (x[i] != x[i]);
    m4_identity(T);
    if (!bad)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
    T[3] = x[3];
    T[7] = x[4];
    T[11] = x[5];
}

// SE(3) exponential (rotation first), fp64, row-major.
O3D_HD inline void se3_exp(const double* d, double* T) {
    const double w[3] = {d[0], d[1], d[2]}, v[3] = {d[3], d[4], d[5]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    double A, B, C;
    if (th < 1e-10) {
        A = 1.0 - th2 / 6.0;
        B = 0.5 - th2 / 24.0;
        C = 1.0 / 6.0 - th2 / 120.0;
    } else {
        A = sin(th) / th;
        B = (1 - cos(th)) / th2;
        C = (1 - A) / th2;
    }
    const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double K2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += K[3 * i + k] * K[3 * k + j];
            K2[3 * i + j] = s;
        }
    memset(T, 0, 16 * sizeof(double));
    T[15] = 1;
    for (int i = 0; i < 3; ++i) {
        double t = 0;
        for (int j = 0; j < 3; ++j) {
            T[4 * i + j] = (i == j ? 1.0 : 0.0) + A * K[3 * i + j] + B * K2[3 * i + j];
            t += ((i == j ? 1.0 : 0.0) + B * K[3 * i + j] + C * K2[3 * i + j]) * v[j];
        }
        T[4 * i + 3] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// Transformation checkers (TransformationCheckersImpl.cpp:57-158)
// ---------------------------------------------------------------------------------------------
O3D_HD inline void rot_to_quat(const float* T, float* q /* w x y z */) {
    const float m00 = T[0], m11 = T[5], m22 = T[10];
    const float tr = m00 + m11 + m22;
    if (tr > 0.f) {
        float t = sqrt(tr + 1.0f);
        q[0] = 0.5f * t;
        t = 0.5f / t;
        q[1] = (T[9] - T[6]) * t;
        q[2] = (T[2] - T[8]) * t;
        q[3] = (T[4] - T[1]) * t;
    } else {
        int i = 0;
        if (m11 > m00) i = 1;
        if (m22 > T[5 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        float t = sqrt(T[5 * i] - T[5 * j] - T[5 * k] + 1.0f);
        float v[3];
        v[i] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (T[4 * k + j] - T[4 * j + k]) * t;
        v[j] = (T[4 * j + i] + T[4 * i + j]) * t;
        v[k] = (T[4 * k + i] + T[4 * i + k]) * t;
        q[1] = v[0];
        q[2] = v[1];
        q[3] = v[2];
    }
}

O3D_HD inline float quat_angular_distance(const float* a, const float* b) {
    const float bw = b[0], bx = -b[1], by = -b[2], bz = -b[3];
    const float w = a[0] * bw - a[1] * bx - a[2] * by - a[3] * bz;
    const float x = a[0] * bx + a[1] * bw + a[2] * bz - a[3] * by;
    const float y = a[0] * by + a[2] * bw + a[3] * bx - a[1] * bz;
    const float z = a[0] * bz + a[3] * bw + a[1] * by - a[2] * bx;
    return 2.0f * atan2f(sqrt(x * x + y * y + z * z), fabs(w));
}

constexpr int kCheckerHist = 16;   // smooth_len is clamped to kCheckerHist - 1

// DifferentialTransformationChecker + CounterTransformationChecker state (fixed-size ring so the same
// code runs inside the device-side update kernel).
struct Checkers {
    int max_iter = 40;
    float min_diff_rot = 1e-3f, min_diff_trans = 1e-3f;
    int smooth_len = 3;
    float quats[kCheckerHist][4];
    float trans[kCheckerHist][3];
    int n_hist = 0;        // total poses pushed (ring holds the last kCheckerHist)
    int count = 0;
    bool converged = false, max_iter_reached = false;

    O3D_HD void init(const float* T) {
        n_hist = 0;
        count = 0;
        converged = max_iter_reached = false;
        if (smooth_len > kCheckerHist - 1) smooth_len = kCheckerHist - 1;
        push(T);
    }
    O3D_HD void push(const float* T) {
        const int slot = n_hist % kCheckerHist;
        rot_to_quat(T, quats[slot]);
        trans[slot][0] = T[3];
        trans[slot][1] = T[7];
        trans[slot][2] = T[11];
        ++n_hist;
    }
    // returns `iterate`
    O3D_HD bool check(const float* T) {
        bool iterate = true;
        push(T);
        if (smooth_len > 0 && n_hist > smooth_len) {
            float cr = 0.f, ct = 0.f;
            for (int i = n_hist - 1; i >= n_hist - smooth_len; --i) {
                const int a = i % kCheckerHist, b = (i - 1) % kCheckerHist;
                cr += fabsf(quat_angular_distance(quats[a], quats[b]));
                const float dx = trans[a][0] - trans[b][0], dy = trans[a][1] - trans[b][1],
                            dz = trans[a][2] - trans[b][2];
                ct += sqrtf(dx * dx + dy * dy + dz * dz);
            }
            cr /= (float)smooth_len;
            ct /= (float)smooth_len;
            if (cr < min_diff_rot && ct < min_diff_trans) {
                iterate = false;
                converged = true;
            }
        }
        ++count;
        if (count >= max_iter) {
            iterate = false;
            max_iter_reached = true;
        }
        return iterate;
    }
};

// LDL^T solve of a symmetric 6x6 system in fp64 (no pivoting).  Returns false when a pivot falls below
// rel_thr * (largest diagonal entry): the caller then uses the eigen-solve (minimum-norm) path.
O3D_HD inline bool solve_ldlt6(const double* H, const double* g, double* x, double rel_thr) {
    double L[36], d[6];
    double dmax = 0;
    for (int i = 0; i < 6; ++i) dmax = fmax(dmax, fabs(H[6 * i + i]));
    if (!(dmax > 0)) return false;
    for (int j = 0; j < 6; ++j) {
        double dj = H[6 * j + j];
        for (int k = 0; k < j; ++k) dj -= L[6 * j + k] * L[6 * j + k] * d[k];
        if (!(dj > rel_thr * dmax)) return false;
        d[j] = dj;
        L[6 * j + j] = 1.0;
        for (int i = j + 1; i < 6; ++i) {
            double v = 0.5 * (H[6 * i + j] + H[6 * j + i]);
            for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k] * d[k];
            L[6 * i + j] = v / dj;
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double v = g[i];
        for (int k = 0; k < i; ++k) v -= L[6 * i + k] * y[k];
        y[i] = v;
    }
    for (int i = 0; i < 6; ++i) y[i] /= d[i];
    for (int i = 5; i >= 0; --i) {
        double v = y[i];
        for (int k = i + 1; k < 6; ++k) v -= L[6 * k + i] * x[k];
        x[i] = v;
    }
    return true;
}

// Point-to-plane step (R8): A x = b.  Well-conditioned -> LDL^T; otherwise the eigen-solve with the fp32
// rank threshold (minimum-norm solution, PointToPlane.cpp:206-247).  Returns the numerical rank.
O3D_HD inline int solve6_p2pl_fast(const float* A, const float* b, float* x) {
    double H[36], g[6], xd[6];
    for (int i = 0; i < 36; ++i) H[i] = A[i];
    for (int i = 0; i < 6; ++i) g[i] = b[i];
    int rank = 6;
    if (!solve_ldlt6(H, g, xd, 1e-4)) rank = solve_sym6(H, g, xd, 6.0 * 1.1920929e-07);
    for (int i = 0; i < 6; ++i) x[i] = (float)xd[i];
    return rank;
}

}  // namespace o3dreg
