// tools/stepprobe.hip — what does the memory system sustain for the access pattern of ONE step(actions) launch (step_kernel<CartPole>,
// mxv_kernels.hip) with the physics removed?  Per env: read state 4 x f64 + elapsed i32 + action i64 (44 B), write state 4 x f64 + elapsed
// i32 + obs float4 + reward f64 + 2 flag bytes (62 B): 106 B, one launch per step, 2^20 envs.  Variants: block size, envs per lane, outputs
// into the same buffers every step (70 MB working set: inside the 256-MiB Infinity Cache) or rotating over [R][N] trajectory slices
// (the outputs then stream to HBM while the state stays resident), and the pieces alone (state round trip only / outputs only).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/stepprobe tools/stepprobe.hip && tools/_bin/stepprobe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Bufs {
    double *state;      // [4][n]
    int32_t *elapsed;   // [n]
    const int64_t *action;  // [n]
    float4 *obs;        // [R][n]
    double *rew;        // [R][n]
    uint8_t *term, *trunc;  // [R][n]
};

// WHAT: bit 0 = state round trip, bit 1 = action read + outputs
template <int BLOCK, int E, int WHAT>
__global__ void __launch_bounds__(BLOCK) step_like(Bufs b, int64_t n, int64_t out_off, int step) {
    const unsigned bid = blockIdx.x, ntiles = gridDim.x;
    const unsigned tile = (ntiles % 8 == 0) ? (bid % 8) * (ntiles / 8) + bid / 8 : bid;   // XCD-contiguous tiles, as the engine does
    const int64_t base = (int64_t)tile * (BLOCK * E) + threadIdx.x;
    double s[E][4];
    int32_t el[E];
    int64_t a[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t e = base + (int64_t)j * BLOCK;
        if (WHAT & 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s[j][k] = b.state[(int64_t)k * n + e];
            el[j] = b.elapsed[e];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) s[j][k] = (double)(e + k + step);
            el[j] = step;
        }
        a[j] = (WHAT & 2) ? b.action[e] : (int64_t)(e & 1);
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t e = base + (int64_t)j * BLOCK;
        const double f = a[j] ? 1e-9 : -1e-9;
#pragma unroll
        for (int k = 0; k < 4; ++k) s[j][k] = s[j][k] * 0.999999 + f;
        el[j] += 1;
        if (WHAT & 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) b.state[(int64_t)k * n + e] = s[j][k];
            b.elapsed[e] = el[j];
        }
        if (WHAT & 2) {
            b.obs[out_off + e] = make_float4((float)s[j][0], (float)s[j][1], (float)s[j][2], (float)s[j][3]);
            b.rew[out_off + e] = 1.0;
            b.term[out_off + e] = s[j][0] > 1e300;
            b.trunc[out_off + e] = el[j] >= 500;
        } else if (!(WHAT & 1)) {
            if (s[j][0] == 1.2345) b.rew[e] = s[j][1];   // keep the arithmetic alive
        }
    }
}

template <int BLOCK, int E, int WHAT>
static int run(const char *name, Bufs b, int64_t n, int R, hipStream_t st, int launches) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const dim3 grid((unsigned)(n / (BLOCK * E)));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < launches; ++i)
            hipLaunchKernelGGL((step_like<BLOCK, E, WHAT>), grid, dim3(BLOCK), 0, st, b, n, (int64_t)(i % R) * n, i);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double us = best * 1e3 / launches;
    const double bytes = ((WHAT & 1) ? 72.0 : 0.0) + ((WHAT & 2) ? 34.0 : 0.0);
    printf("{\"probe\": \"%s\", \"block\": %d, \"envs_per_lane\": %d, \"rotating_slices\": %d, \"us_per_launch\": %.3f, \"bytes_per_env\": %.0f, "
           "\"TBs\": %.3f}\n", name, BLOCK, E, R, us, bytes, bytes * n / us / 1e6);
    fflush(stdout);
    return 0;
}

int main() {
    const int64_t n = 1 << 20;
    const int RMAX = 64;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    Bufs b;
    int64_t *act;
    CK(hipMalloc(&b.state, 4 * n * 8));
    CK(hipMalloc(&b.elapsed, n * 4));
    CK(hipMalloc(&act, n * 8));
    CK(hipMalloc(&b.obs, RMAX * n * 16));
    CK(hipMalloc(&b.rew, RMAX * n * 8));
    CK(hipMalloc(&b.term, RMAX * n));
    CK(hipMalloc(&b.trunc, RMAX * n));
    CK(hipMemset(b.state, 0, 4 * n * 8));
    CK(hipMemset(b.elapsed, 0, n * 4));
    CK(hipMemset(act, 0, n * 8));
    b.action = act;
    const int L = 400;
    // clock ramp
    for (int i = 0; i < 3; ++i) run<256, 1, 3>("warm", b, n, 1, st, L);
    printf("--- full pattern (106 B per env), outputs into the same buffers every step\n");
    run<256, 1, 3>("full", b, n, 1, st, L);
    run<256, 2, 3>("full", b, n, 1, st, L);
    run<256, 4, 3>("full", b, n, 1, st, L);
    run<64, 1, 3>("full", b, n, 1, st, L);
    run<64, 2, 3>("full", b, n, 1, st, L);
    run<64, 4, 3>("full", b, n, 1, st, L);
    run<512, 1, 3>("full", b, n, 1, st, L);
    run<1024, 1, 3>("full", b, n, 1, st, L);
    printf("--- full pattern, outputs rotating over 64 trajectory slices (state resident, outputs stream to HBM)\n");
    run<256, 1, 3>("full", b, n, RMAX, st, L);
    run<256, 2, 3>("full", b, n, RMAX, st, L);
    run<64, 2, 3>("full", b, n, RMAX, st, L);
    printf("--- state round trip only (72 B per env)\n");
    run<256, 1, 1>("state", b, n, 1, st, L);
    run<256, 2, 1>("state", b, n, 1, st, L);
    run<64, 2, 1>("state", b, n, 1, st, L);
    printf("--- action read + outputs only (34 B per env), same buffers / rotating\n");
    run<256, 1, 2>("outputs", b, n, 1, st, L);
    run<256, 1, 2>("outputs", b, n, RMAX, st, L);
    printf("--- no memory traffic (launch + gap floor)\n");
    run<256, 1, 0>("empty", b, n, 1, st, L);
    return 0;
}
