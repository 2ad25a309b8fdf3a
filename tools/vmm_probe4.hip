// tools/vmm_probe4.hip — where, in ADDRESS space, is the store pattern of the fused CartPole rollout slow?  vmm_probe.hip showed that the
// speed of a 16-step window does not depend on which physical chunks back it, that single streams are uniform, and that the combination
// obs + reward + actions carries the effect.  Here one 40-GiB range is reserved and fully mapped (160 chunks of 256 MiB, identity), and the
// streams' base pointers are simply moved around inside it — no remapping: scans of one stream's offset with the others fixed, pairs of
// streams, the launch length K.  JSON lines.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/vmm_probe4 tools/vmm_probe4.hip && tools/_bin/vmm_probe4
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
static char *base;

// us per vector step; offsets in bytes from base
static float tcfg(size_t o_obs, size_t o_rew, size_t o_act, size_t o_term, size_t o_trunc, int mask, int K, int launches = 6, int reps = 3) {
    float best = 1e30f;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipEventRecord(ev0, s));
        for (int j = 0; j < launches; ++j)
            hipLaunchKernelGGL(stores, dim3(N / 128), dim3(64), 0, s, (float4 *)(base + o_obs), (double *)(base + o_rew), (int64_t *)(base + o_act),
                               (uint8_t *)(base + o_term), (uint8_t *)(base + o_trunc), N, K, mask);
        CK(hipEventRecord(ev1, s));
        CK(hipEventSynchronize(ev1));
        float ms;
        CK(hipEventElapsedTime(&ms, ev0, ev1));
        best = std::min(best, ms * 1e3f / (launches * K));
    }
    return best;
}

int main() {
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&s));
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    {   // the same preamble as vmm_probe.hip (six hipMalloc'ed sets held, then freed): the runs with mixed windows all had it
        const int K = 256;
        std::vector<void *> held;
        for (int i = 0; i < 6; ++i)
            for (size_t b : {(size_t)K * N * 16, (size_t)K * N * 8, (size_t)K * N * 8, (size_t)K * N, (size_t)K * N}) { void *p; CK(hipMalloc(&p, b)); CK(hipMemset(p, 0, b)); held.push_back(p); }
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
    const size_t total = 40 * GiB, chunk = 256 * MiB;
    CK(hipMemAddressReserve((void **)&base, total, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> pool(total / chunk);
    for (size_t j = 0; j < pool.size(); ++j) { CK(hipMemCreate(&pool[j], chunk, &prop, 0)); CK(hipMemMap(base + j * chunk, chunk, 0, pool[j], 0)); }
    CK(hipMemSetAccess(base, total, &acc, 1));
    CK(hipMemset(base, 0, total));
    CK(hipDeviceSynchronize());
    printf("{\"exp\": \"base\", \"va\": \"0x%llx\"}\n", (unsigned long long)(uintptr_t)base);
    for (int i = 0; i < 30; ++i) tcfg(0, 4 * GiB, 6 * GiB, 8 * GiB, 8 * GiB + 256 * MiB, 15, 256, 2, 1);   // spin-up

    // packed set at offset 0: whole + 16-step windows (the reference picture)
    printf("{\"exp\": \"packed_whole\", \"us\": %.3f}\n", tcfg(0, 4 * GiB, 6 * GiB, 8 * GiB, 8 * GiB + 256 * MiB, 15, 256, 3, 3));
    printf("{\"exp\": \"packed_windows16\", \"us\": [");
    for (int w = 0; w < 16; ++w) printf("%s%.2f", w ? ", " : "", tcfg(w * 256 * MiB, 4 * GiB + w * 128 * MiB, 6 * GiB + w * 128 * MiB, 8 * GiB + w * 16 * MiB, 8 * GiB + 256 * MiB + w * 16 * MiB, 15, 16));
    printf("]}\n");
    fflush(stdout);

    // K dependence at window 0 and window 8 of the packed layout
    for (int w : {0, 4, 8, 12}) {
        printf("{\"exp\": \"K_dependence\", \"window\": %d, \"K\": [2, 4, 8, 16, 32], \"us\": [", w);
        for (int K : {2, 4, 8, 16, 32}) printf("%s%.2f", K > 2 ? ", " : "", tcfg(w * 256 * MiB, 4 * GiB + w * 128 * MiB, 6 * GiB + w * 128 * MiB, 8 * GiB + w * 16 * MiB, 8 * GiB + 256 * MiB + w * 16 * MiB, 15, K, 96 / K + 1));
        printf("]}\n");
    }
    fflush(stdout);

    // scans, K = 16: one stream moves in 64-MiB steps over [0, 16 GiB), the others fixed far away
    struct Scan { const char *name; int mask; int moving; size_t fixed[5]; };
    const Scan scans[] = {
        {"obs_alone", 1, 0, {0, 0, 0, 0, 0}},
        {"obs_moves__rew_at_20G", 3, 0, {0, 20 * GiB, 0, 0, 0}},
        {"obs_moves__rew_at_20.5G", 3, 0, {0, 20 * GiB + 512 * MiB, 0, 0, 0}},
        {"obs_moves__act_at_24G", 5, 0, {0, 0, 24 * GiB, 0, 0}},
        {"rew_moves__obs_at_20G", 3, 1, {20 * GiB, 0, 0, 0, 0}},
        {"obs_moves__rew_20G_act_24G", 7, 0, {0, 20 * GiB, 24 * GiB, 0, 0}},
        {"obs_moves__rew_20G_act_22G", 7, 0, {0, 20 * GiB, 22 * GiB, 0, 0}},
        {"rew_moves__obs_20G_act_24G", 7, 1, {20 * GiB, 0, 24 * GiB, 0, 0}},
        {"obs_moves__all_others_fixed", 15, 0, {0, 20 * GiB, 22 * GiB, 24 * GiB, 24 * GiB + 256 * MiB}},
        {"rew_act_alone_act_moves", 6, 2, {0, 20 * GiB, 0, 0, 0}},
    };
    for (const Scan &sc : scans) {
        printf("{\"exp\": \"scan\", \"name\": \"%s\", \"mask\": %d, \"step_MiB\": 64, \"us\": [", sc.name, sc.mask);
        for (size_t a = 0; a < 16 * GiB; a += 64 * MiB) {
            size_t o[5] = {sc.fixed[0], sc.fixed[1], sc.fixed[2], sc.fixed[3], sc.fixed[4]};
            o[sc.moving] = a;
            printf("%s%.2f", a ? ", " : "", tcfg(o[0], o[1], o[2], o[3], o[4], sc.mask, 16, 6, 2));
        }
        printf("]}\n");
        fflush(stdout);
    }
    // 2-D: obs offset x rew offset (256-MiB grid, 24 x 24), act fixed at 30 GiB, mask obs+rew+act
    printf("{\"exp\": \"grid_obs_x_rew\", \"step_MiB\": 256, \"obs_from_GiB\": 0, \"rew_from_GiB\": 12, \"act_GiB\": 30, \"us\": [");
    for (int i = 0; i < 24; ++i) {
        printf("%s[", i ? ", " : "");
        for (int j = 0; j < 24; ++j) printf("%s%.2f", j ? ", " : "", tcfg(i * 256 * MiB, 12 * GiB + j * 256 * MiB, 30 * GiB, 0, 0, 7, 16, 6, 2));
        printf("]");
    }
    printf("]}\n{\"exp\": \"done\"}\n");
    return 0;
}
