// tools/wbench6.hip — is the slow mode of the rollout's store pattern tied to its power-of-two ROW STRIDES?  The pure-store replica
// of the CartPole trajectory launch (one wave per workgroup, two envs per lane, XCD-contiguous tiles, K = 256 steps, obs float4 |
// reward f64 | action i64 | two flag bytes; every tensor its own hipMalloc, as torch allocates them) with the rows of every tensor
// padded by `pad` envs: [K][N + pad] instead of [K][N].  pad = 0 is the engine's layout (row strides 16 MiB / 8 MiB / 8 MiB / 1 MiB /
// 1 MiB at N = 2^20).  If a box that runs pad = 0 in the slow mode (> 6.3 us per step) runs a padded layout in the fast one
// (< 5.6), the mode is a persistent bank relation between equal-stride streams and padding is a layout RULE; if not, it is not.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/wbench6 tools/wbench6.hip && tools/_bin/wbench6
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MAP 0: XCD-contiguous tiles (the engine's map: XCD x owns the x-th contiguous eighth of every row); 1: natural order (consecutive
// tiles go to consecutive XCDs); 2: XCD x owns every eighth 16-tile group (32 KiB of obs)
template <int MAP>
__global__ void __launch_bounds__(64, 4) stores(float4 *obs, double *rew, int64_t *act, uint8_t *term, uint8_t *trunc, int64_t row, int K) {
    const unsigned bid = blockIdx.x, ntiles = gridDim.x;
    const unsigned xcd = bid % 8, j = bid / 8;
    const unsigned tile = MAP == 0 ? xcd * (ntiles / 8) + j : (MAP == 1 ? bid : ((j / 16) * 8 + xcd) * 16 + j % 16);
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

int main() {
    const int64_t n = 1 << 20;
    const int K = 256;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int64_t pads[] = {0, 128, 4096 + 128, 65536 + 4096 + 128, 3 * 128, 0, (1 << 18) + 128, 0};
    for (int round = 0; round < 2; ++round)
        for (int64_t pad : pads) {
            const int64_t row = n + pad;
            float4 *obs; double *rew; int64_t *act; uint8_t *term, *trunc;
            CK(hipMalloc(&obs, K * row * 16));
            CK(hipMalloc(&rew, K * row * 8));
            CK(hipMalloc(&act, K * row * 8));
            CK(hipMalloc(&term, K * row));
            CK(hipMalloc(&trunc, K * row));
            float best[3] = {1e9f, 1e9f, 1e9f};
            for (int rep = 0; rep < 3; ++rep)
                for (int map = 0; map < 3; ++map) {
                    CK(hipEventRecord(e0, s));
                    for (int i = 0; i < 4; ++i) {
                        if (map == 0) hipLaunchKernelGGL(stores<0>, dim3(n / 128), dim3(64), 0, s, obs, rew, act, term, trunc, row, K);
                        if (map == 1) hipLaunchKernelGGL(stores<1>, dim3(n / 128), dim3(64), 0, s, obs, rew, act, term, trunc, row, K);
                        if (map == 2) hipLaunchKernelGGL(stores<2>, dim3(n / 128), dim3(64), 0, s, obs, rew, act, term, trunc, row, K);
                    }
                    CK(hipEventRecord(e1, s));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best[map]) best[map] = ms;
                }
            printf("{\"round\": %d, \"pad_envs\": %lld, \"us_per_step\": %.3f, \"natural_map\": %.3f, \"grouped_map\": %.3f, \"obs\": \"%p\"}\n", round,
                   (long long)pad, best[0] * 1e3 / (4 * K), best[1] * 1e3 / (4 * K), best[2] * 1e3 / (4 * K), (void *)obs);
            fflush(stdout);
            CK(hipFree(obs)); CK(hipFree(rew)); CK(hipFree(act)); CK(hipFree(term)); CK(hipFree(trunc));
        }
    return 0;
}
