// What a one-workgroup kernel costs before it does anything: back-to-back launches of empty kernels with the update
// kernel's shape (1024 threads, 87 KB of LDS) against leaner shapes.  hipcc -O3 --offload-arch=gfx950 -o tools_launch_probe ...
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS>
__global__ void __launch_bounds__(1024) k_empty(float* out) {
    __shared__ float s[LDS / 4];
    if (out == reinterpret_cast<float*>(1)) {   // never true: keeps the LDS allocation alive
        s[threadIdx.x] = 1.f;
        __syncthreads();
        out[0] = s[0];
    }
}
template <typename K>
static float run(K k, int threads, const char* name) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, (float*)nullptr);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, (float*)nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-34s %.2f us per launch (2000 back to back)\n", name, 1e3f * ms / 2000);
    return ms;
}
int main() {
    run(k_empty<87 * 1024>, 1024, "1024 threads, 87 KB LDS");
    run(k_empty<16 * 1024>, 1024, "1024 threads, 16 KB LDS");
    run(k_empty<16 * 1024>, 256, "256 threads, 16 KB LDS");
    run(k_empty<16 * 1024>, 64, "64 threads, 16 KB LDS");
    run(k_empty<87 * 1024>, 256, "256 threads, 87 KB LDS");
    return 0;
}
