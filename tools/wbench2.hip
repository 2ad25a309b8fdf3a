// tools/wbench2.hip — why does a persistent store loop write slower than a flat fill?  Fill-size, workgroup-count and
// address-pattern matrix for pure float4 stores.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void fill_flat(float4 *p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void fill_stride(float4 *p, int64_t n) {  // persistent, grid-stride
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) p[k] = make_float4(1.f, 2.f, 3.f, (float)k);
}
__global__ void fill_block(float4 *p, int64_t n) {  // persistent, each WG owns one contiguous chunk
    const int64_t chunk = n / gridDim.x;
    float4 *q = p + (int64_t)blockIdx.x * chunk;
    for (int64_t k = threadIdx.x; k < chunk; k += blockDim.x) q[k] = make_float4(1.f, 2.f, 3.f, (float)k);
}
template <typename F> float best_ms(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}
int main() {
    const int64_t maxb = 8ll << 30;
    float4 *p; if (hipMalloc(&p, maxb) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int64_t gib : {1, 4, 8}) {
        const int64_t n = (gib << 30) / 16;
        float ms = best_ms([&] { fill_flat<<<(unsigned)(n / 256), 256>>>(p, n); });
        printf("flat    %lld GiB                 %7.1f GB/s\n", (long long)gib, n * 16 / ms / 1e6);
    }
    const int64_t n = (4ll << 30) / 16;
    for (int wgs : {1024, 2048, 4096, 8192, 16384, 65536}) {
        float ms = best_ms([&] { fill_stride<<<wgs, 256>>>(p, n); });
        printf("stride  4 GiB  %6d WGs x256   %7.1f GB/s\n", wgs, n * 16 / ms / 1e6);
        ms = best_ms([&] { fill_block<<<wgs, 256>>>(p, n); });
        printf("block   4 GiB  %6d WGs x256   %7.1f GB/s\n", wgs, n * 16 / ms / 1e6);
    }
    for (int wgs : {8192, 32768}) {
        float ms = best_ms([&] { fill_stride<<<wgs, 64>>>(p, n); });
        printf("stride  4 GiB  %6d WGs x64    %7.1f GB/s\n", wgs, n * 16 / ms / 1e6);
    }
    return 0;
}
