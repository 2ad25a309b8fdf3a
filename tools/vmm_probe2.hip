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

int main() {
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&s));
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    const size_t b_obs = (size_t)K * N * 16, b_rew = (size_t)K * N * 8, b_act = b_rew, b_flag = (size_t)K * N;
    const size_t total = b_obs + b_rew + b_act + 2 * b_flag, chunk = 256 * MiB;
    const int per_set = (int)(total / chunk);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> pool(per_set);
    for (auto &h : pool) CK(hipMemCreate(&h, chunk, &prop, 0));
    auto set_at = [&](char *base) {
        Set v;
        v.obs = (float4 *)base;
        v.rew = (double *)(base + b_obs);
        v.act = (int64_t *)(base + b_obs + b_rew);
        v.term = (uint8_t *)(base + b_obs + b_rew + b_act);
        v.trunc = v.term + b_flag;
        return v;
    };
    auto map_all = [&](char *base) {
        for (int j = 0; j < per_set; ++j) CK(hipMemMap(base + (size_t)j * chunk, chunk, 0, pool[j], 0));
        CK(hipMemSetAccess(base, total, &acc, 1));
    };
    // spin-up on an ordinary mapping
    {
        char *b;
        CK(hipMemAddressReserve((void **)&b, total, 0, nullptr, 0));
        map_all(b);
        Set v = set_at(b);
        for (int i = 0; i < 8; ++i) time_window(v, 0, K, 4, 1);
        CK(hipMemUnmap(b, total));
        CK(hipMemAddressFree(b, total));
    }
    // a + b: re-reservations with different alignments
    // (the alignment argument of hipMemAddressReserve is ignored by this runtime: every reservation of this size came back at the same
    //  address; a 32-GiB reservation with a 32-GiB alignment crashed inside the runtime — so: align 0 only)
    {   // what hipMalloc gives on this box, for reference (six sets held, then freed — as vmm_probe.hip did before its VMM part)
        std::vector<Set> held;
        printf("{\"exp\": \"hipMalloc_sets_whole_us\", \"us\": [");
        for (int i = 0; i < 6; ++i) {
            Set t;
            CK(hipMalloc(&t.obs, b_obs)); CK(hipMalloc(&t.rew, b_rew)); CK(hipMalloc(&t.act, b_act)); CK(hipMalloc(&t.term, b_flag)); CK(hipMalloc(&t.trunc, b_flag));
            held.push_back(t);
            time_window(t, 0, K, 2, 1);
            printf("%s%.2f", i ? ", " : "", time_window(t, 0, K, 3, 2));
        }
        printf("]}\n");
        fflush(stdout);
        for (auto &t : held) { CK(hipFree(t.obs)); CK(hipFree(t.rew)); CK(hipFree(t.act)); CK(hipFree(t.term)); CK(hipFree(t.trunc)); }
    }
    for (size_t align : {(size_t)0, (size_t)0}) {
        char *b;
        CK(hipMemAddressReserve((void **)&b, total, align, nullptr, 0));
        map_all(b);
        Set v = set_at(b);
        time_window(v, 0, K, 2, 1);
        char tag[64];
        snprintf(tag, sizeof tag, "ab_align_%zuMiB", align >> 20);
        windows(tag, v, 16, (uintptr_t)b);
        CK(hipMemUnmap(b, total));
        CK(hipMemAddressFree(b, total));
    }
    // c: the set shifted inside a 32-GiB reservation
    {
        const size_t big = (size_t)32 << 30;
        char *b;
        CK(hipMemAddressReserve((void **)&b, big, 0, nullptr, 0));
        for (size_t shift : {(size_t)0, 256 * MiB, 512 * MiB, 1024 * MiB, 1536 * MiB, 2048 * MiB, 4096 * MiB, 8192 * MiB, 12288 * MiB, 16384 * MiB, 20480 * MiB}) {
            map_all(b + shift);
            Set v = set_at(b + shift);
            time_window(v, 0, K, 2, 1);
            char tag[64];
            snprintf(tag, sizeof tag, "c_shift_%zuMiB", shift >> 20);
            windows(tag, v, 16, (uintptr_t)(b + shift));
            CK(hipMemUnmap(b + shift, total));
        }
        // d: one window (16 steps) with its five pieces at arbitrary offsets inside the reservation
        hipMemGenericAllocationHandle_t h_obs, h_rew, h_act, h_term, h_trunc;
        const size_t w_obs = 256 * MiB, w_rew = 128 * MiB, w_flag = 16 * MiB;
        CK(hipMemCreate(&h_obs, w_obs, &prop, 0));
        CK(hipMemCreate(&h_rew, w_rew, &prop, 0));
        CK(hipMemCreate(&h_act, w_rew, &prop, 0));
        CK(hipMemCreate(&h_term, w_flag, &prop, 0));
        CK(hipMemCreate(&h_trunc, w_flag, &prop, 0));
        auto time_cfg = [&](size_t o_obs, size_t o_rew, size_t o_act, size_t o_term, size_t o_trunc) {
            CK(hipMemMap(b + o_obs, w_obs, 0, h_obs, 0));
            CK(hipMemMap(b + o_rew, w_rew, 0, h_rew, 0));
            CK(hipMemMap(b + o_act, w_rew, 0, h_act, 0));
            CK(hipMemMap(b + o_term, w_flag, 0, h_term, 0));
            CK(hipMemMap(b + o_trunc, w_flag, 0, h_trunc, 0));
            CK(hipMemSetAccess(b + o_obs, w_obs, &acc, 1));
            CK(hipMemSetAccess(b + o_rew, w_rew, &acc, 1));
            CK(hipMemSetAccess(b + o_act, w_rew, &acc, 1));
            CK(hipMemSetAccess(b + o_term, w_flag, &acc, 1));
            CK(hipMemSetAccess(b + o_trunc, w_flag, &acc, 1));
            Set v{(float4 *)(b + o_obs), (double *)(b + o_rew), (int64_t *)(b + o_act), (uint8_t *)(b + o_term), (uint8_t *)(b + o_trunc)};
            time_window(v, 0, 16, 2, 1);
            const float us = time_window(v, 0, 16, 6, 3);
            CK(hipMemUnmap(b + o_obs, w_obs));
            CK(hipMemUnmap(b + o_rew, w_rew));
            CK(hipMemUnmap(b + o_act, w_rew));
            CK(hipMemUnmap(b + o_term, w_flag));
            CK(hipMemUnmap(b + o_trunc, w_flag));
            return us;
        };
        // the layout of window w of the packed set
        auto packed = [&](int w, size_t o[5]) {
            o[0] = (size_t)w * 256 * MiB; o[1] = 4096 * MiB + (size_t)w * 128 * MiB; o[2] = 6144 * MiB + (size_t)w * 128 * MiB;
            o[3] = 8192 * MiB + (size_t)w * 16 * MiB; o[4] = 8448 * MiB + (size_t)w * 16 * MiB;
        };
        printf("{\"exp\": \"d_packed_windows\", \"us\": [");
        for (int w = 0; w < 16; ++w) { size_t o[5]; packed(w, o); printf("%s%.2f", w ? ", " : "", time_cfg(o[0], o[1], o[2], o[3], o[4])); }
        printf("]}\n");
        fflush(stdout);
        const char *names[5] = {"obs", "rew", "act", "term", "trunc"};
        for (int basew : {0, 8}) {
            for (int which = 0; which < 5; ++which) {
                printf("{\"exp\": \"d_move_one\", \"base_window\": %d, \"moved\": \"%s\", \"to_window_position\": [", basew, names[which]);
                for (int w = 0; w < 16; ++w) {
                    size_t o[5], m[5];
                    packed(basew, o);
                    packed(w, m);
                    o[which] = m[which];
                    printf("%s%.2f", w ? ", " : "", time_cfg(o[0], o[1], o[2], o[3], o[4]));
                }
                printf("]}\n");
                fflush(stdout);
            }
        }
        // everything far apart: pieces at 0, 9, 13, 17, 19 GiB
        printf("{\"exp\": \"d_far_apart\", \"us\": %.2f}\n", time_cfg(0, (size_t)9 << 30, (size_t)13 << 30, (size_t)17 << 30, (size_t)19 << 30));
        CK(hipMemAddressFree(b, big));
    }
    printf("{\"exp\": \"done\"}\n");
    return 0;
}
