// tools/vmm_probe3.hip — (see below; shares the kernel and timing helpers of vmm_probe2.hip)
// tools/vmm_probe2.hip — follow-up of vmm_probe.hip.  That run showed: inside ONE reserved virtual range the speed of every 16-step
// window of the store pattern is the same whatever physical chunks back it (identity, rotated, random draws from a 2x pool) — the mode
// follows the VIRTUAL placement (or the page tables that serve it), not the data pages.  Questions here:
//   a  is the window profile reproducible across re-reservations (new page tables), and where does the range start?
//   b  does the alignment of the reservation change it?
//   c  does shifting the whole set inside a larger reservation change it?
//   d  one window's five pieces mapped at arbitrary offsets: which stream's position makes a window slow?
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/vmm_probe2 tools/vmm_probe2.hip && tools/_bin/vmm_probe2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); fflush(stdout); exit(1); } } while (0)

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

static const int64_t N = 1 << 20;
static const int K = 256;
static const size_t MiB = 1 << 20;
static hipStream_t s;
static hipEvent_t ev0, ev1;
struct Set { float4 *obs; double *rew; int64_t *act; uint8_t *term, *trunc; };

static float time_window(const Set &t, int t0, int k, int launches, int reps) {
    float best = 1e30f;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipEventRecord(ev0, s));
        for (int j = 0; j < launches; ++j)
            hipLaunchKernelGGL(stores, dim3(N / 128), dim3(64), 0, s, t.obs + (int64_t)t0 * N, t.rew + (int64_t)t0 * N, t.act + (int64_t)t0 * N,
                               t.term + (int64_t)t0 * N, t.trunc + (int64_t)t0 * N, N, k);
        CK(hipEventRecord(ev1, s));
        CK(hipEventSynchronize(ev1));
        float ms;
        CK(hipEventElapsedTime(&ms, ev0, ev1));
        best = std::min(best, ms * 1e3f / (launches * k));
    }
    return best;
}

static void windows(const char *tag, const Set &t, int W, uintptr_t base) {
    printf("{\"exp\": \"%s\", \"base\": \"0x%llx\", \"whole_us\": %.3f, \"windows_us\": [", tag, (unsigned long long)base, time_window(t, 0, K, 3, 2));
    for (int t0 = 0; t0 < K; t0 += W) printf("%s%.2f", t0 ? ", " : "", time_window(t, t0, W, 6, 2));
    printf("]}\n");
    fflush(stdout);
}


// tools/vmm_probe3.hip: ONE 16-step window of the store pattern (obs 256 MiB, reward 128, actions 128, two flag tensors 16 MiB each),
// its five pieces mapped at RANDOM offsets inside a 48-GiB reservation, timed; every configuration is timed with two different sets of
// physical pieces (A, B).  Output: one JSON line per configuration {offsets in MiB, us_A, us_B}.  Questions: is the time a function of the
// virtual offsets alone (us_A == us_B)?  which offset relations are slow?
int main(int argc, char **argv) {
    const int configs = argc > 1 ? atoi(argv[1]) : 300;
    const size_t gran = (size_t)(argc > 2 ? atoi(argv[2]) : 16) * MiB;   // offsets are multiples of this
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&s));
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t big = (size_t)48 << 30;
    char *b;
    CK(hipMemAddressReserve((void **)&b, big, 0, nullptr, 0));
    const size_t sz[5] = {256 * MiB, 128 * MiB, 128 * MiB, 16 * MiB, 16 * MiB};
    hipMemGenericAllocationHandle_t h[2][5];
    for (int p = 0; p < 2; ++p) {
        for (int i = 0; i < 5; ++i) CK(hipMemCreate(&h[p][i], sz[i], &prop, 0));
        // something between the two sets so that they do not sit next to each other physically
        void *spacer;
        CK(hipMalloc(&spacer, (size_t)3 << 30));
    }
    // hipMemSetAccess only accepts the START of a reservation (a range mapped further inside one is "invalid argument"), address hints of
    // hipMemAddressReserve are honoured, and reserve/free cycles corrupt the runtime's heap after a while (glibc aborts with "unsorted
    // double linked list corrupted").  So: the big range is probed for a free region and given back ONCE; then every stream gets a lane
    // of it, cut into slots of the stream's piece size, each slot a reservation of its own that is never freed.
    CK(hipMemAddressFree(b, big));
    const size_t lane_off[5] = {0, (size_t)10 << 30, (size_t)20 << 30, (size_t)30 << 30, (size_t)34 << 30};
    const size_t lane_len[5] = {(size_t)10 << 30, (size_t)10 << 30, (size_t)10 << 30, (size_t)4 << 30, (size_t)4 << 30};
    for (int i = 0; i < 5; ++i)
        for (size_t o = 0; o < lane_len[i]; o += sz[i]) {
            char *c = nullptr;
            CK(hipMemAddressReserve((void **)&c, sz[i], 0, b + lane_off[i] + o, 0));
            if (c != b + lane_off[i] + o) { printf("{\"error\": \"hint not honoured\"}\n"); exit(1); }
        }
    auto time_cfg = [&](const size_t o[5], int p, int reps) {
        for (int i = 0; i < 5; ++i) { CK(hipMemMap(b + o[i], sz[i], 0, h[p][i], 0)); CK(hipMemSetAccess(b + o[i], sz[i], &acc, 1)); }
        Set v{(float4 *)(b + o[0]), (double *)(b + o[1]), (int64_t *)(b + o[2]), (uint8_t *)(b + o[3]), (uint8_t *)(b + o[4])};
        time_window(v, 0, 16, 2, 1);
        const float us = time_window(v, 0, 16, 6, reps);
        for (int i = 0; i < 5; ++i) CK(hipMemUnmap(b + o[i], sz[i]));
        return us;
    };
    printf("{\"exp\": \"base\", \"va\": \"0x%llx\", \"granule_MiB\": %zu}\n", (unsigned long long)(uintptr_t)b, gran >> 20);
    // spin-up
    { const size_t o[5] = {lane_off[0], lane_off[1], lane_off[2], lane_off[3], lane_off[4]}; for (int i = 0; i < 40; ++i) time_cfg(o, 0, 1); }
    // the packed set's windows first (layout of vmm_probe.hip's C_identity)
    for (int w = 0; w < 16; ++w) {
        const size_t o[5] = {lane_off[0] + w * sz[0], lane_off[1] + w * sz[1], lane_off[2] + w * sz[2], lane_off[3] + w * sz[3], lane_off[4] + w * sz[4]};
        printf("{\"exp\": \"packed\", \"w\": %d, \"us_A\": %.2f, \"us_B\": %.2f}\n", w, time_cfg(o, 0, 3), time_cfg(o, 1, 3));
    }
    fflush(stdout);
    uint64_t rs = 88172645463325252ull;
    auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
    for (int c = 0; c < configs; ++c) {
        size_t o[5];
        for (int i = 0; i < 5; ++i) o[i] = lane_off[i] + (rnd() % (lane_len[i] / sz[i])) * sz[i];
        const float ua = time_cfg(o, 0, 3), ub = time_cfg(o, 1, 3), ua2 = time_cfg(o, 0, 2);
        printf("{\"exp\": \"random\", \"o_MiB\": [%zu, %zu, %zu, %zu, %zu], \"us_A\": %.2f, \"us_B\": %.2f, \"us_A_again\": %.2f}\n", o[0] >> 20, o[1] >> 20, o[2] >> 20,
               o[3] >> 20, o[4] >> 20, ua, ub, ua2);
        if (c % 32 == 0) fflush(stdout);
    }
    printf("{\"exp\": \"done\"}\n");
    return 0;
}
