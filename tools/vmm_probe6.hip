// tools/vmm_probe6.hip — are the two "classes" of vmm_probe5.hip (a 16-B/lane stream and an 8-B/lane stream written concurrently are
// 12 % slower when both lie in the same class of address regions, GiB-scale runs of irregular length) a property of the PHYSICAL chunk?
// And can a trajectory set be made fast BY CONSTRUCTION: observations on chunks of one class, rewards + actions on chunks of the other?
//   1. 96 chunks of 256 MiB (hipMemCreate); every chunk is classified on its own, mapped at ONE scratch address (so the virtual
//      address is the same for all): obs-style stores into the chunk under test + reward-style stores into reference chunk 0.
//   2. sets of 34 chunks in fresh reservations (each chunk mapped once, never remapped):  "split" = obs on class-1 chunks, reward +
//      actions on class-0 chunks;  "same" = everything on one class;  "order" = chunks in creation order.  Whole-set time + windows.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/vmm_probe6 tools/vmm_probe6.hip && tools/_bin/vmm_probe6 [preamble=1]
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
    const bool preamble = argc < 2 || atoi(argv[1]) != 0;
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&s));
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    if (preamble) {   // six hipMalloc'ed sets held, touched, freed (as in the other probes)
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
    std::vector<hipMemGenericAllocationHandle_t> pool(P);
    for (auto &h : pool) CK(hipMemCreate(&h, chunk, &prop, 0));
    // scratch addresses: the chunk under test and the reference chunk, each a reservation of its own
    char *va_test, *va_ref;
    CK(hipMemAddressReserve((void **)&va_test, chunk, 0, nullptr, 0));
    CK(hipMemAddressReserve((void **)&va_ref, chunk, 0, nullptr, 0));
    CK(hipMemMap(va_ref, chunk, 0, pool[0], 0));
    CK(hipMemSetAccess(va_ref, chunk, &acc, 1));
    // spin-up
    CK(hipMemMap(va_test, chunk, 0, pool[1], 0));
    CK(hipMemSetAccess(va_test, chunk, &acc, 1));
    for (int i = 0; i < 400; ++i) tptr(va_test, va_ref, nullptr, nullptr, nullptr, 3, 16, 6, 1);
    CK(hipMemUnmap(va_test, chunk));
    printf("{\"exp\": \"scratch\", \"va_test\": \"0x%llx\", \"va_ref\": \"0x%llx\"}\n", (unsigned long long)(uintptr_t)va_test, (unsigned long long)(uintptr_t)va_ref);
    std::vector<float> us(P, 0.f);
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 1; i < P; ++i) {
            CK(hipMemMap(va_test, chunk, 0, pool[i], 0));
            CK(hipMemSetAccess(va_test, chunk, &acc, 1));
            if (pass == 0) {   // the remap really switches physical memory: a marker written through this mapping is read back later
                int marker = 1000 + i;
                CK(hipMemcpy(va_test + 4096, &marker, 4, hipMemcpyHostToDevice));
            } else {
                int got = 0;
                CK(hipMemcpy(&got, va_test + 4096, 4, hipMemcpyDeviceToHost));
                if (got != 1000 + i) printf("{\"error\": \"chunk %d: marker %d\"}\n", i, got);
            }
            if (pass == 1) {
                tptr(va_test, va_ref, nullptr, nullptr, nullptr, 3, 16, 2, 1);
                us[i] = tptr(va_test, va_ref, nullptr, nullptr, nullptr, 3, 16, 6, 3);   // obs stream -> chunk i, reward stream -> chunk 0 (first 128 MiB)
            }
            CK(hipMemUnmap(va_test, chunk));
        }
    }
    float lo = 1e30f, hi = 0.f;
    for (int i = 1; i < P; ++i) lo = std::min(lo, us[i]), hi = std::max(hi, us[i]);
    const float thr = 0.5f * (lo + hi);
    printf("{\"exp\": \"classify\", \"against\": \"chunk 0\", \"lo\": %.2f, \"hi\": %.2f, \"us\": [", lo, hi);
    for (int i = 1; i < P; ++i) printf("%s%.2f", i > 1 ? ", " : "", us[i]);
    printf("], \"same_class_as_chunk0\": \"");
    std::vector<int> same{0}, other;
    for (int i = 1; i < P; ++i) {
        const bool slow = us[i] > thr;
        const bool clear = us[i] < lo + 0.25f * (hi - lo) || us[i] > hi - 0.25f * (hi - lo);
        printf("%c", !clear ? '?' : slow ? '1' : '0');
        if (clear) (slow ? same : other).push_back(i);
    }
    printf("\"}\n");
    fflush(stdout);
    if (hi - lo < 0.25f) { printf("{\"exp\": \"note\", \"text\": \"no class spread among these chunks\"}\n"); }
    CK(hipMemUnmap(va_ref, chunk));

    // sets: obs 16 chunks, reward 8, actions 8, flags 1 + 1
    auto build_and_time = [&](const char *name, const std::vector<int> &obs_c, const std::vector<int> &ra_c, const std::vector<int> &flag_c) {
        if (obs_c.size() < 16 || ra_c.size() < 16 || flag_c.size() < 2) { printf("{\"exp\": \"set\", \"name\": \"%s\", \"skipped\": \"not enough chunks\"}\n", name); return; }
        char *b;
        const size_t total = 34 * chunk;
        CK(hipMemAddressReserve((void **)&b, total, 0, nullptr, 0));
        std::vector<int> order;
        for (int j = 0; j < 16; ++j) order.push_back(obs_c[j]);
        for (int j = 0; j < 16; ++j) order.push_back(ra_c[j]);
        order.push_back(flag_c[0]);
        order.push_back(flag_c[1]);
        for (int j = 0; j < 34; ++j) CK(hipMemMap(b + (size_t)j * chunk, chunk, 0, pool[order[j]], 0));
        CK(hipMemSetAccess(b, total, &acc, 1));
        char *obs = b, *rew = b + 4 * GiB, *act = b + 6 * GiB, *term = b + 8 * GiB, *trunc = term + 256 * MiB;
        tptr(obs, rew, act, term, trunc, 15, 256, 2, 1);
        printf("{\"exp\": \"set\", \"name\": \"%s\", \"whole_us\": %.3f, \"windows16_us\": [", name, tptr(obs, rew, act, term, trunc, 15, 256, 3, 3));
        for (int w = 0; w < 16; ++w)
            printf("%s%.2f", w ? ", " : "", tptr(obs + w * 256 * MiB, rew + w * 128 * MiB, act + w * 128 * MiB, term + w * 16 * MiB, trunc + w * 16 * MiB, 15, 16, 6, 2));
        printf("]}\n");
        fflush(stdout);
        for (int j = 0; j < 34; ++j) CK(hipMemUnmap(b + (size_t)j * chunk, chunk));   // exactly as mapped; the reservation itself is kept (frees corrupt the runtime's heap)
    };
    std::vector<int> seq;
    for (int i = 0; i < P; ++i) seq.push_back(i);
    auto tail = [](const std::vector<int> &v, size_t from) { return std::vector<int>(v.begin() + std::min(from, v.size()), v.end()); };
    build_and_time("creation_order", seq, tail(seq, 16), tail(seq, 32));
    build_and_time("split_obs_on_class1__rew_act_on_class0", same, other, tail(other, 16));
    build_and_time("split_obs_on_class0__rew_act_on_class1", other, same, tail(same, 16));
    build_and_time("same_everything_on_class1", same, tail(same, 16), tail(same, 32));
    build_and_time("same_everything_on_class0", other, tail(other, 16), tail(other, 32));
    build_and_time("split_again", same, other, tail(other, 16));
    printf("{\"exp\": \"done\"}\n");
    return 0;
}
