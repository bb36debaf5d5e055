// Calibration of FETCH_SIZE / WRITE_SIZE on a known byte count (MI355X_MICROARCH.md, HBM section):
// (a) k_stream: 16 B/lane contiguous read of 1 GiB; (b) k_gather16: 16 B/lane reads of random 16-byte records from a
// 1 GiB table (one record per lane, like the candidate gathers of the registration kernels, footprint >> 256 MB L3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_stream(const float4* __restrict__ in, size_t n, float* out) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i].x;
    if (acc == 1.2345f) out[0] = acc;
}
__global__ void k_gather16(const float4* __restrict__ in, size_t n, size_t reads, float* out) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < reads; i += (size_t)gridDim.x * blockDim.x) {
        size_t h = i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        acc += in[h % n].x;
    }
    if (acc == 1.2345f) out[0] = acc;
}
int main() {
    const size_t n = (1ull << 30) / 16;
    float4* d; float* o;
    hipMalloc(&d, n * 16); hipMalloc(&o, 64); hipMemset(d, 0, n * 16);
    for (int r = 0; r < 3; ++r) k_stream<<<4096, 256>>>(d, n, o);
    for (int r = 0; r < 3; ++r) k_gather16<<<4096, 256>>>(d, n, n / 4, o);   // 256 MiB of 16-byte useful reads
    hipDeviceSynchronize();
    printf("stream bytes per launch = %zu ; gather useful bytes per launch = %zu (reads = %zu)\n", n * 16, (n / 4) * 16, n / 4);
    return 0;
}
