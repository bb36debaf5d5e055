// In-kernel clock check (MI355X_MICROARCH.md 'DVFS give-back' item 6): shader clock = d(s_memtime)/d(s_memrealtime)*100MHz
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_clock(unsigned long long* out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
__global__ void k_busy(float* o, int iters) {
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    if (x == 1.2345f) o[0] = x;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64); float* f; hipMalloc(&f, 64);
    unsigned long long h[3];
    auto probe = [&](const char* tag, int iters) {
        k_clock<<<256, 256>>>(d, iters); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("%s: iters=%d shader cycles=%llu realtime ticks=%llu -> clock %.0f MHz, duration %.1f us\n", tag, iters, h[0], h[1], 100.0 * h[0] / h[1], h[1] / 100.0);
    };
    probe("cold short", 2000);
    probe("cold short", 2000);
    probe("cold long", 200000);
    for (int r = 0; r < 5; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.5) { k_busy<<<1024, 256>>>(f, 20000); hipDeviceSynchronize(); }
        probe("after 0.5s busy", 2000);
    }
    // sparse short kernels with gaps (like an ICP iteration loop)
    for (int r = 0; r < 5; ++r) { for (int k = 0; k < 200; ++k) { k_busy<<<3000, 256>>>(f, 200); } hipDeviceSynchronize(); probe("after 200 short kernels", 2000); }
    return 0;
}
