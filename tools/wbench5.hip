// tools/wbench5.hip — does the write path overlap with compute?  A synthetic twin of rollout_kernel_v3's loop structure: one wave64
// per workgroup, two envs per lane, XCD-contiguous tiles, K = 256 steps, the five output streams of the CartPole rollout
// (obs float4, reward f64, action i64, two flag bytes per env-step) — plus M dependent fp64 FMAs per env and step standing in for
// the physics.  M = 0 is the pure-store replica; M ~ 100 has the VALU work of the real kernel (210 VALU instructions per
// wave-step).  If time(M) = max(store time, compute time) the two overlap and a slower real kernel has something to fix; if
// time(M) grows like store + compute the platform does not overlap them.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/wbench5 tools/wbench5.hip && tools/_bin/wbench5
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int M, bool STORE>
__global__ void __launch_bounds__(64, 4) twin(float4 *obs, double *rew, int64_t *act, uint8_t *term, uint8_t *trunc, int64_t n, int K,
                                              double seed) {
    const unsigned bid = blockIdx.x, ntiles = gridDim.x;
    const unsigned tile = (bid % 8) * (ntiles / 8) + bid / 8;  // XCD x owns the x-th contiguous eighth
    const int lane = threadIdx.x;
    const int64_t e0 = (int64_t)tile * 128 + lane, e1 = e0 + 64;
    double x0 = seed + lane, x1 = seed - lane;
    for (int k = 0; k < K; ++k) {
        const int64_t so = (int64_t)k * n;
#pragma unroll
        for (int m = 0; m < M; ++m) {  // two independent dependent chains, like the two env chains of a lane
            x0 = __fma_rn(x0, 0.999999, 1e-9);
            x1 = __fma_rn(x1, 1.000001, -1e-9);
        }
        if (STORE) {
            act[so + e0] = k & 1;
            act[so + e1] = (k >> 1) & 1;
            rew[so + e0] = 1.0;
            rew[so + e1] = 1.0;
            term[so + e0] = x0 > 1e30;
            term[so + e1] = x1 > 1e30;
            trunc[so + e0] = 0;
            trunc[so + e1] = 0;
            obs[so + e0] = make_float4((float)x0, (float)x1, 0.f, 1.f);
            obs[so + e1] = make_float4((float)x1, (float)x0, 1.f, 0.f);
        }
    }
    if (x0 + x1 == 12345.678) obs[0] = make_float4((float)x0, 0, 0, 0);  // keep the chains alive when STORE is false
}

// the same with BLOCK lanes per workgroup (BLOCK / 64 waves, lane `tid` owns envs tile0 + j * BLOCK + tid): what round 1 measured with 256-lane workgroups
template <int M, int BLOCK>
__global__ void __launch_bounds__(BLOCK) twin_block(float4 *obs, double *rew, int64_t *act, uint8_t *term, uint8_t *trunc, int64_t n, int K,
                                                    double seed) {
    const unsigned bid = blockIdx.x, ntiles = gridDim.x;
    const unsigned tile = (bid % 8) * (ntiles / 8) + bid / 8;
    const int tid = threadIdx.x;
    const int64_t e0 = (int64_t)tile * (2 * BLOCK) + tid, e1 = e0 + BLOCK;
    double x0 = seed + tid, x1 = seed - tid;
    for (int k = 0; k < K; ++k) {
        const int64_t so = (int64_t)k * n;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            x0 = __fma_rn(x0, 0.999999, 1e-9);
            x1 = __fma_rn(x1, 1.000001, -1e-9);
        }
        act[so + e0] = k & 1;
        act[so + e1] = (k >> 1) & 1;
        rew[so + e0] = 1.0;
        rew[so + e1] = 1.0;
        term[so + e0] = x0 > 1e30;
        term[so + e1] = x1 > 1e30;
        trunc[so + e0] = 0;
        trunc[so + e1] = 0;
        obs[so + e0] = make_float4((float)x0, (float)x1, 0.f, 1.f);
        obs[so + e1] = make_float4((float)x1, (float)x0, 1.f, 0.f);
    }
}

template <int M, int BLOCK>
int run_block(float4 *obs, double *rew, int64_t *act, uint8_t *term, uint8_t *trunc, int64_t n, int K) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned grid = (unsigned)(n / (2 * BLOCK));
    for (int w = 0; w < 2; ++w) twin_block<M, BLOCK><<<grid, BLOCK>>>(obs, rew, act, term, trunc, n, K, 0.5);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 256 / K * 6;
    for (int r = 0; r < reps; ++r) twin_block<M, BLOCK><<<grid, BLOCK>>>(obs, rew, act, term, trunc, n, K, 0.5);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("M=%3d  %d lanes per workgroup, K=%3d      %7.3f us/step\n", M, BLOCK, K, ms / reps / K * 1e3);
    return 0;
}

template <int M, bool STORE>
int run(float4 *obs, double *rew, int64_t *act, uint8_t *term, uint8_t *trunc, int64_t n, int K) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned grid = (unsigned)(n / 128);
    for (int w = 0; w < 2; ++w) twin<M, STORE><<<grid, 64>>>(obs, rew, act, term, trunc, n, K, 0.5);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 6;
    for (int r = 0; r < reps; ++r) twin<M, STORE><<<grid, 64>>>(obs, rew, act, term, trunc, n, K, 0.5);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("M=%3d fma/env-step  %-14s %7.3f us/step\n", M, STORE ? "stores+compute" : "compute only", ms / reps / K * 1e3);
    return 0;
}

int main() {
    const int64_t n = 1 << 20;
    const int K = 256;
    float4 *obs; double *rew; int64_t *act; uint8_t *term, *trunc;
    CK(hipMalloc(&obs, n * K * 16)); CK(hipMalloc(&rew, n * K * 8)); CK(hipMalloc(&act, n * K * 8));
    CK(hipMalloc(&term, n * K)); CK(hipMalloc(&trunc, n * K));
    CK(hipMemset(obs, 0, n * K * 16)); CK(hipMemset(rew, 0, n * K * 8)); CK(hipMemset(act, 0, n * K * 8));
    CK(hipMemset(term, 0, n * K)); CK(hipMemset(trunc, 0, n * K));
    for (int warm = 0; warm < 40; ++warm) twin<0, true><<<(unsigned)(n / 128), 64>>>(obs, rew, act, term, trunc, n, K, 0.5);  // clock ramp
    CK(hipDeviceSynchronize());
#define BOTH(M) if (run<M, true>(obs, rew, act, term, trunc, n, K)) return 1; if (run<M, false>(obs, rew, act, term, trunc, n, K)) return 1;
    BOTH(0) BOTH(25) BOTH(50) BOTH(75) BOTH(100) BOTH(125) BOTH(150) BOTH(200)
    for (int K2 : {256, 64}) {
        if (run_block<0, 64>(obs, rew, act, term, trunc, n, K2)) return 1;
        if (run_block<0, 128>(obs, rew, act, term, trunc, n, K2)) return 1;
        if (run_block<0, 256>(obs, rew, act, term, trunc, n, K2)) return 1;
        if (run_block<0, 512>(obs, rew, act, term, trunc, n, K2)) return 1;
        if (run_block<100, 64>(obs, rew, act, term, trunc, n, K2)) return 1;
        if (run_block<100, 256>(obs, rew, act, term, trunc, n, K2)) return 1;
        if (run_block<100, 512>(obs, rew, act, term, trunc, n, K2)) return 1;
    }
    return 0;
}
