// tools/vmm_probe5.hip (pairwise 2-D grids; setup of vmm_probe4.hip) — where, in ADDRESS space, is the store pattern of the fused CartPole rollout slow?  vmm_probe.hip showed that the
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

    // packed set at offset 0: 16-step windows (the reference picture)
    printf("{\"exp\": \"packed_windows16\", \"us\": [");
    for (int w = 0; w < 16; ++w) printf("%s%.2f", w ? ", " : "", tcfg(w * 256 * MiB, 4 * GiB + w * 128 * MiB, 6 * GiB + w * 128 * MiB, 8 * GiB + w * 16 * MiB, 8 * GiB + 256 * MiB + w * 16 * MiB, 15, 16));
    printf("]}\n");
    fflush(stdout);
    // 2-D class structure, two streams at a time, 512-MiB grid over the whole 40 GiB
    struct Pair { const char *name; int mask; int s0, s1; };
    for (const Pair &pr : {Pair{"obs_x_rew", 3, 0, 1}, Pair{"obs_x_act", 5, 0, 2}, Pair{"rew_x_act", 6, 1, 2}}) {
        printf("{\"exp\": \"grid2\", \"name\": \"%s\", \"step_MiB\": 512, \"us\": [", pr.name);
        for (int i = 0; i < 79; ++i) {
            printf("%s[", i ? ", " : "");
            for (int j = 0; j < 79; ++j) {
                size_t o[5] = {0, 0, 0, 0, 0};
                o[pr.s0] = i * 512 * MiB;
                o[pr.s1] = j * 512 * MiB + (i == j ? 256 * MiB : 0);   // same cell: second half of it
                printf("%s%.2f", j ? ", " : "", tcfg(o[0], o[1], o[2], o[3], o[4], pr.mask, 16, 4, 2));
            }
            printf("]");
        }
        printf("]}\n");
        fflush(stdout);
    }
    printf("{\"exp\": \"done\"}\n");
    return 0;
}
