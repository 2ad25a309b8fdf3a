// tools/wbench3.hip — does the placement bimodality of the fused rollout's store pattern (profiles/r01h_placement_probe.txt)
// come from the NUMBER of concurrent output streams?  Pure-store replica of the pattern (single-wave workgroups, 2 envs per
// lane, XCD-contiguous tiles, K steps), repeated over fresh allocations:
//   A: five arrays  obs 16 B | reward 8 B | action 8 B | terminated 1 B | truncated 1 B          (the engine's layout)
//   B: three arrays obs 16 B | (reward, action) 16 B interleaved | (terminated, truncated) 2 B interleaved
//   C: two arrays   obs 16 B + (reward, action) 16 B interleaved per env = 32 B | flags 2 B
// Prints the per-allocation times: is there a layout whose speed does not depend on where the allocations land?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kWave = 64, E = 2, TILE = E * kWave;
__device__ int g_blk = 0;  // 0: XCD x owns the x-th contiguous eighth; B > 0: XCDs take turns in blocks of B tiles
__device__ __forceinline__ unsigned tile_of(unsigned bid, unsigned nt) {
    const unsigned x = bid % 8, idx = bid / 8, base = nt / 8, rem = nt % 8;
    if (g_blk > 0) {  // workgroup idx-th of XCD x -> block (idx / B) of that XCD, blocks dealt round-robin over the XCDs
        const unsigned B = (unsigned)g_blk;
        return ((idx / B) * 8 + x) * B + idx % B;
    }
    return x * base + (x < rem ? x : rem) + idx;
}
template <int LAYOUT>
__global__ void __launch_bounds__(kWave) pat(char *a0, char *a1, char *a2, char *a3, char *a4, int64_t n, int K) {
    const int lane = threadIdx.x;
    const int64_t tile0 = (int64_t)tile_of(blockIdx.x, gridDim.x) * TILE;
    float x = (float)lane;
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int64_t e = (int64_t)k * n + tile0 + j * kWave + lane;
            x = x * 1.0001f + 0.5f;
            const float4 o = make_float4(x, x + 1, x + 2, x + 3);
            if (LAYOUT == 0) {
                reinterpret_cast<float4 *>(a0)[e] = o;
                reinterpret_cast<double *>(a1)[e] = 1.0;
                reinterpret_cast<int64_t *>(a2)[e] = k & 1;
                reinterpret_cast<uint8_t *>(a3)[e] = 0;
                reinterpret_cast<uint8_t *>(a4)[e] = 0;
            } else if (LAYOUT == 1) {
                reinterpret_cast<float4 *>(a0)[e] = o;
                reinterpret_cast<double2 *>(a1)[e] = make_double2(1.0, (double)(k & 1));
                reinterpret_cast<uchar2 *>(a3)[e] = make_uchar2(0, 0);
            } else {
                reinterpret_cast<float4 *>(a0)[2 * e] = o;
                reinterpret_cast<double2 *>(a0)[2 * e + 1] = make_double2(1.0, (double)(k & 1));
                reinterpret_cast<uchar2 *>(a3)[e] = make_uchar2(0, 0);
            }
        }
    }
}
template <int LAYOUT>
float run_once(int64_t n, int K) {
    const size_t sz[5] = {(size_t)K * n * (LAYOUT == 2 ? 32 : 16), (size_t)K * n * (LAYOUT == 0 ? 8 : 16), (size_t)K * n * 8, (size_t)K * n * 2, (size_t)K * n};
    char *p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 5; ++i) {
        const bool used = LAYOUT == 0 || (LAYOUT == 1 && (i == 0 || i == 1 || i == 3)) || (LAYOUT == 2 && (i == 0 || i == 3));
        if (used && hipMalloc((void **)&p[i], sz[i]) != hipSuccess) return -1.f;
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned grid = (unsigned)(n / TILE);
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(pat<LAYOUT>, dim3(grid), dim3(kWave), 0, 0, p[0], p[1], p[2], p[3], p[4], n, K);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r > 0 && ms < best) best = ms;
    }
    for (int i = 0; i < 5; ++i) if (p[i]) hipFree(p[i]);
    return best * 1e3f / K;
}
int main(int argc, char **argv) {
    const int64_t n = 1 << 20; const int K = 128;
    std::vector<void *> junk;
    if (argc > 1) {  // tile-map granularity sweep on layout A
        for (int blk : {0, 4, 16, 64, 256}) {
            hipMemcpyToSymbol(HIP_SYMBOL(g_blk), &blk, sizeof blk);
            printf("layout A, XCD blocks of %3d tiles:", blk);
            for (int t = 0; t < 10; ++t) {
                void *j; hipMalloc(&j, (size_t)(37 + 29 * t) << 20); junk.push_back(j);
                printf(" %.2f", run_once<0>(n, K));
            }
            printf("\n");
        }
        return 0;
    }
    for (int l = 0; l < 3; ++l) {
        printf("layout %c:", "ABC"[l]);
        for (int t = 0; t < 14; ++t) {
            void *j; hipMalloc(&j, (size_t)(37 + 29 * t) << 20); junk.push_back(j);   // shift where the next allocations land
            float us = l == 0 ? run_once<0>(n, K) : (l == 1 ? run_once<1>(n, K) : run_once<2>(n, K));
            printf(" %.2f", us);
        }
        printf("  us/step (34 B/env-step)\n");
    }
    return 0;
}
