// tools/wbench4.hip — looking for the rule behind the placement modes: two equal-stride store streams (the rollout's pattern:
// single-wave workgroups, 2 envs per lane, XCD-contiguous tiles, K steps of N envs, 16 B per env each) carved out of ONE
// allocation at a relative byte offset D; time per step as a function of D.  A periodic structure in D would be the HBM
// bank/channel hash showing through (if the allocation is physically contiguous at that scale).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
constexpr int kWave = 64, E = 2, TILE = E * kWave;
__device__ __forceinline__ unsigned tile_of(unsigned bid, unsigned nt) {
    const unsigned x = bid % 8, idx = bid / 8, base = nt / 8, rem = nt % 8;
    return x * base + (x < rem ? x : rem) + idx;
}
__global__ void __launch_bounds__(kWave) two_streams(float4 *a, float4 *b, int64_t n, int K) {
    const int lane = threadIdx.x;
    const int64_t tile0 = (int64_t)tile_of(blockIdx.x, gridDim.x) * TILE;
    float x = (float)lane;
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int64_t e = (int64_t)k * n + tile0 + j * kWave + lane;
            x = x * 1.0001f + 0.5f;
            a[e] = make_float4(x, x + 1, x + 2, x + 3);
            b[e] = make_float4(x + 4, x + 5, x + 6, x + 7);
        }
}
int main() {
    const int64_t n = 1 << 20; const int K = 48;
    const size_t arr = (size_t)K * n * 16;            // 768 MiB per stream
    const size_t span = 2 * arr + (512ull << 20);
    char *p; if (hipMalloc((void **)&p, span) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t s, t; hipEventCreate(&s); hipEventCreate(&t);
    auto run = [&](size_t d) {
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(s);
            hipLaunchKernelGGL(two_streams, dim3((unsigned)(n / TILE)), dim3(kWave), 0, 0, (float4 *)p, (float4 *)(p + arr + d), n, K);
            hipEventRecord(t); hipEventSynchronize(t);
            float ms; hipEventElapsedTime(&ms, s, t);
            if (r && ms < best) best = ms;
        }
        return best * 1e3f / K;
    };
    run(0);
    printf("# D (KiB) : us per step (32 B/env-step), fine sweep 0..1 MiB step 4 KiB\n");
    for (size_t d = 0; d <= (1u << 20); d += 4096) printf("%zu:%.2f ", d >> 10, run(d));
    printf("\n# coarse sweep 0..256 MiB step 1 MiB\n");
    for (size_t d = 0; d <= (256ull << 20); d += (1u << 20)) printf("%zu:%.2f ", d >> 20, run(d));
    printf("\n# sub-4K sweep step 256 B\n");
    for (size_t d = 0; d <= 8192; d += 256) printf("%zu:%.2f ", d, run(d));
    printf("\n");
    return 0;
}
