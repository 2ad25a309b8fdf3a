// tools/vmm_probe7.hip — the pairwise conflict relation between 256-MiB chunks, and whether it predicts whole sets.
// (vmm_probe6.hip found that remapping a different chunk at an address that was mapped before is not reliable in this runtime — a value
// written through the previous mapping was read back — so here NOTHING is ever remapped: 96 chunks are mapped once, in creation order,
// into one reservation, and streams are placed by pointer.)
//   M[i][j] = time of a 16-step window with the observation stream (16 B/lane) in chunk i and the reward stream (8 B/lane) in chunk j;
//   then 14 placements of a whole trajectory set (obs = 16 consecutive chunks from a0, reward = 8 from b0, actions = 8 from c0, flags
//   from d0): whole-set time and the sixteen 16-step windows, to be compared with what M predicts (tools/vmm_predict.py).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/vmm_probe7 tools/vmm_probe7.hip && tools/_bin/vmm_probe7
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); fflush(stdout); exit(1); } } while (0)

__global__ void __launch_bounds__(64, 4) stores(float4 *obs, double *rew, int64_t *act, uint8_t *term, uint8_t *trunc, int64_t row, int K, int mask) {
    const unsigned bid = blockIdx.x, ntiles = gridDim.x;
    const unsigned tile = (bid % 8) * (ntiles / 8) + bid / 8;
    const int lane = threadIdx.x;
    const int64_t e0 = (int64_t)tile * 128 + lane, e1 = e0 + 64;
    float x = (float)lane;
    for (int k = 0; k < K; ++k) {
        const int64_t so = (int64_t)k * row;
        x = x * 1.0001f + 0.5f;
        if (mask & 4) { act[so + e0] = k & 1; act[so + e1] = (k >> 1) & 1; }
        if (mask & 2) { rew[so + e0] = 1.0; rew[so + e1] = 1.0; }
        if (mask & 8) { term[so + e0] = x > 1e30f; term[so + e1] = 0; trunc[so + e0] = 0; trunc[so + e1] = 0; }
        if (mask & 1) { obs[so + e0] = make_float4(x, x + 1, 0.f, 1.f); obs[so + e1] = make_float4(x + 2, x, 1.f, 0.f); }
    }
}

static const int64_t N = 1 << 20;
static const size_t MiB = 1 << 20, GiB = (size_t)1 << 30;
static hipStream_t s;
static hipEvent_t ev0, ev1;

static float tptr(char *obs, char *rew, char *act, char *term, char *trunc, int mask, int K, int launches = 6, int reps = 3) {
    float best = 1e30f;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipEventRecord(ev0, s));
        for (int j = 0; j < launches; ++j)
            hipLaunchKernelGGL(stores, dim3(N / 128), dim3(64), 0, s, (float4 *)obs, (double *)rew, (int64_t *)act, (uint8_t *)term, (uint8_t *)trunc, N, K, mask);
        CK(hipEventRecord(ev1, s));
        CK(hipEventSynchronize(ev1));
        float ms;
        CK(hipEventElapsedTime(&ms, ev0, ev1));
        best = std::min(best, ms * 1e3f / (launches * K));
    }
    return best;
}


int main(int argc, char **argv) {
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&s));
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    {   // six hipMalloc'ed sets held, touched, freed (as in the other probes)
        std::vector<void *> held;
        for (int i = 0; i < 6; ++i)
            for (size_t b : {(size_t)256 * N * 16, (size_t)256 * N * 8, (size_t)256 * N * 8, (size_t)256 * N, (size_t)256 * N}) { void *p; CK(hipMalloc(&p, b)); CK(hipMemset(p, 0, b)); held.push_back(p); }
        CK(hipDeviceSynchronize());
        for (void *p : held) CK(hipFree(p));
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t chunk = 256 * MiB;
    const int P = 96;
    char *b;
    CK(hipMemAddressReserve((void **)&b, P * chunk, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> pool(P);
    for (int j = 0; j < P; ++j) { CK(hipMemCreate(&pool[j], chunk, &prop, 0)); CK(hipMemMap(b + (size_t)j * chunk, chunk, 0, pool[j], 0)); }
    CK(hipMemSetAccess(b, P * chunk, &acc, 1));
    CK(hipMemset(b, 0, P * chunk));
    CK(hipDeviceSynchronize());
    for (int i = 0; i < 300; ++i) tptr(b, b + 20 * chunk, nullptr, nullptr, nullptr, 3, 16, 6, 1);
    printf("{\"exp\": \"base\", \"va\": \"0x%llx\", \"chunks\": %d}\n", (unsigned long long)(uintptr_t)b, P);
    printf("{\"exp\": \"pair_matrix\", \"what\": \"obs stream in chunk i (row), reward stream in chunk j (column), 16 steps, us per step\", \"us\": [");
    for (int i = 0; i < P; ++i) {
        printf("%s[", i ? ", " : "");
        for (int j = 0; j < P; ++j) printf("%s%.2f", j ? ", " : "", i == j ? 0.f : tptr(b + i * chunk, b + j * chunk, nullptr, nullptr, nullptr, 3, 16, 4, 2));
        printf("]");
    }
    printf("]}\n");
    fflush(stdout);
    uint64_t rs = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](int n) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (int)(rs % (uint64_t)n); };
    for (int c = 0; c < 14; ++c) {
        int a0, b0, c0, d0;
        for (;;) {
            a0 = rnd(P - 15); b0 = rnd(P - 7); c0 = rnd(P - 7); d0 = rnd(P - 1);
            if (c == 0) { a0 = 0; b0 = 16; c0 = 24; d0 = 32; }
            auto disjoint = [](int x, int nx, int y, int ny) { return x + nx <= y || y + ny <= x; };
            if (disjoint(a0, 16, b0, 8) && disjoint(a0, 16, c0, 8) && disjoint(b0, 8, c0, 8) && disjoint(d0, 2, a0, 16) && disjoint(d0, 2, b0, 8) && disjoint(d0, 2, c0, 8)) break;
        }
        char *obs = b + a0 * chunk, *rew = b + b0 * chunk, *act = b + c0 * chunk, *term = b + d0 * chunk, *trunc = term + chunk;
        tptr(obs, rew, act, term, trunc, 15, 256, 2, 1);
        printf("{\"exp\": \"placement\", \"obs_chunk\": %d, \"rew_chunk\": %d, \"act_chunk\": %d, \"flag_chunk\": %d, \"whole_us\": %.3f, \"windows16_us\": [", a0, b0, c0, d0,
               tptr(obs, rew, act, term, trunc, 15, 256, 3, 3));
        for (int w = 0; w < 16; ++w)
            printf("%s%.2f", w ? ", " : "", tptr(obs + w * 256 * MiB, rew + w * 128 * MiB, act + w * 128 * MiB, term + w * 16 * MiB, trunc + w * 16 * MiB, 15, 16, 6, 2));
        printf("]}\n");
        fflush(stdout);
    }
    printf("{\"exp\": \"done\"}\n");
    return 0;
}
