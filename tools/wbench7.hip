// tools/wbench7.hip — does a box without a fast placement have one ANYWHERE in its 288 GB?  The pure-store replica of the CartPole
// trajectory launch (tools/wbench6.hip, pad = 0) on up to 26 tensor sets of 9.1 GB allocated one after the other and all HELD, so
// every set sits on different physical memory; each is timed twice.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/wbench7 tools/wbench7.hip && tools/_bin/wbench7
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(64, 4) stores(float4 *obs, double *rew, int64_t *act, uint8_t *term, uint8_t *trunc, int64_t row, int K) {
    const unsigned bid = blockIdx.x, ntiles = gridDim.x;
    const unsigned tile = (bid % 8) * (ntiles / 8) + bid / 8;
    const int lane = threadIdx.x;
    const int64_t e0 = (int64_t)tile * 128 + lane, e1 = e0 + 64;
    float x = (float)lane;
    for (int k = 0; k < K; ++k) {
        const int64_t so = (int64_t)k * row;
        x = x * 1.0001f + 0.5f;
        act[so + e0] = k & 1;
        act[so + e1] = (k >> 1) & 1;
        rew[so + e0] = 1.0;
        rew[so + e1] = 1.0;
        term[so + e0] = x > 1e30f;
        term[so + e1] = 0;
        trunc[so + e0] = 0;
        trunc[so + e1] = 0;
        obs[so + e0] = make_float4(x, x + 1, 0.f, 1.f);
        obs[so + e1] = make_float4(x + 2, x, 1.f, 0.f);
    }
}

struct Set { float4 *obs; double *rew; int64_t *act; uint8_t *term, *trunc; };

int main() {
    const int64_t n = 1 << 20;
    const int K = 256;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<Set> sets;
    for (int i = 0; i < 26; ++i) {
        size_t free_b = 0, total_b = 0;
        CK(hipMemGetInfo(&free_b, &total_b));
        if (free_b < (size_t)12 << 30) break;
        Set t;
        CK(hipMalloc(&t.obs, K * n * 16));
        CK(hipMalloc(&t.rew, K * n * 8));
        CK(hipMalloc(&t.act, K * n * 8));
        CK(hipMalloc(&t.term, K * n));
        CK(hipMalloc(&t.trunc, K * n));
        sets.push_back(t);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(stores, dim3(n / 128), dim3(64), 0, s, t.obs, t.rew, t.act, t.term, t.trunc, n, K);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("{\"set\": %d, \"held_GB\": %.1f, \"us_per_step\": %.3f}\n", i, (total_b - free_b) / 1e9, best * 1e3 / (4 * K));
        fflush(stdout);
    }
    for (auto &t : sets) { hipFree(t.obs); hipFree(t.rew); hipFree(t.act); hipFree(t.term); hipFree(t.trunc); }
    return 0;
}
