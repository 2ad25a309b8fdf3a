// tools/vmm_probe.hip — can the speed mode of the fused CartPole rollout's store pattern be CHOSEN through HIP's virtual-memory API?
// (round-2 review, item 3).  DESIGN.md §6 established that the mode (5.4 vs 6.5 us per 2^20-env step for identical code and identical
// virtual addresses) belongs to the physical pages behind an allocation; hipMemCreate / hipMemAddressReserve / hipMemMap is the one
// software interface that decides which physical memory backs which virtual range.  Four experiments, JSON lines on stdout:
//   A  hipMalloc'ed sets (held): whole-set time, then the time of every 32-step window of the trajectory — is a slow set slow everywhere?
//   B  a pool of hipMemCreate'd chunks: flat-fill time of every chunk on its own — do single chunks separate into modes?
//   C  one reserved 8.5-GiB range backed by 34 chunks of the pool: whole-set time + windows; then the SAME 34 chunks in rotated
//      roles, then random draws of 34 from the whole pool — how much of the spread is the combination, how much the chunks?
//   D  greedy selection: per-window times attribute slowness to chunks; swap the worst for unused pool chunks; re-time.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/vmm_probe tools/vmm_probe.hip && tools/_bin/vmm_probe [chunk_MiB=256] [pool_factor_x10=20]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); fflush(stdout); exit(1); } } while (0)

// the store pattern of rollout_kernel_v3<CartPole, OUT=1> (= mxv_write_probe): one wave per workgroup, two envs per lane, XCD-contiguous tiles
template <int MASK>
__global__ void __launch_bounds__(64, 4) stores_m(float4 *obs, double *rew, int64_t *act, uint8_t *term, uint8_t *trunc, int64_t row, int K) {
    const unsigned bid = blockIdx.x, ntiles = gridDim.x;
    const unsigned tile = (bid % 8) * (ntiles / 8) + bid / 8;
    const int lane = threadIdx.x;
    const int64_t e0 = (int64_t)tile * 128 + lane, e1 = e0 + 64;
    float x = (float)lane;
    for (int k = 0; k < K; ++k) {
        const int64_t so = (int64_t)k * row;
        x = x * 1.0001f + 0.5f;
        if (MASK & 4) { act[so + e0] = k & 1; act[so + e1] = (k >> 1) & 1; }
        if (MASK & 2) { rew[so + e0] = 1.0; rew[so + e1] = 1.0; }
        if (MASK & 8) { term[so + e0] = x > 1e30f; term[so + e1] = 0; trunc[so + e0] = 0; trunc[so + e1] = 0; }
        if (MASK & 1) { obs[so + e0] = make_float4(x, x + 1, 0.f, 1.f); obs[so + e1] = make_float4(x + 2, x, 1.f, 0.f); }
    }
}

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

__global__ void __launch_bounds__(256) fill(float4 *p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

static const int64_t N = 1 << 20;
static const int K = 256;
static hipStream_t s;
static hipEvent_t ev0, ev1;

struct Set { float4 *obs; double *rew; int64_t *act; uint8_t *term, *trunc; };

// us per vector step of `launches` launches of steps [t0, t0 + k) of the set, best of `reps`
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

template <int MASK>
static float time_window_m(const Set &t, int t0, int k, int launches, int reps) {
    float best = 1e30f;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipEventRecord(ev0, s));
        for (int j = 0; j < launches; ++j)
            hipLaunchKernelGGL(stores_m<MASK>, dim3(N / 128), dim3(64), 0, s, t.obs + (int64_t)t0 * N, t.rew + (int64_t)t0 * N, t.act + (int64_t)t0 * N,
                               t.term + (int64_t)t0 * N, t.trunc + (int64_t)t0 * N, N, k);
        CK(hipEventRecord(ev1, s));
        CK(hipEventSynchronize(ev1));
        float ms;
        CK(hipEventElapsedTime(&ms, ev0, ev1));
        best = std::min(best, ms * 1e3f / (launches * k));
    }
    return best;
}

template <int MASK>
static void print_windows_m(const char *tag, const Set &t, int W) {
    printf("{\"exp\": \"%s\", \"mask\": %d, \"window\": %d, \"windows_us\": [", tag, MASK, W);
    for (int t0 = 0; t0 < K; t0 += W) printf("%s%.2f", t0 ? ", " : "", time_window_m<MASK>(t, t0, W, 8, 3));
    printf("]}\n");
    fflush(stdout);
}

static void print_windows_reversed(const char *tag, const Set &t, int W) {
    std::vector<float> r(K / W);
    for (int i = K / W - 1; i >= 0; --i) r[i] = time_window(t, i * W, W, 8, 3);
    printf("{\"exp\": \"%s\", \"order\": \"measured last window first\", \"window\": %d, \"windows_us\": [", tag, W);
    for (int i = 0; i < K / W; ++i) printf("%s%.2f", i ? ", " : "", r[i]);
    printf("]}\n");
    fflush(stdout);
}

static void print_windows(const char *tag, int id, const Set &t, int W) {
    printf("{\"exp\": \"%s\", \"id\": %d, \"whole_us\": %.3f, \"window\": %d, \"windows_us\": [", tag, id, time_window(t, 0, K, 3, 3), W);
    for (int t0 = 0; t0 < K; t0 += W) printf("%s%.3f", t0 ? ", " : "", time_window(t, t0, W, 8, 3));
    printf("]}\n");
    fflush(stdout);
}

int main(int argc, char **argv) {
    const size_t chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 256) << 20;
    const double pool_factor = (argc > 2 ? atoi(argv[2]) : 20) / 10.0;
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&s));
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    const size_t b_obs = (size_t)K * N * 16, b_rew = (size_t)K * N * 8, b_act = (size_t)K * N * 8, b_flag = (size_t)K * N;
    const size_t total = b_obs + b_rew + b_act + 2 * b_flag;

    // spin-up (clock ramp)
    {
        Set w;
        CK(hipMalloc(&w.obs, b_obs)); CK(hipMalloc(&w.rew, b_rew)); CK(hipMalloc(&w.act, b_act)); CK(hipMalloc(&w.term, b_flag)); CK(hipMalloc(&w.trunc, b_flag));
        for (int i = 0; i < 6; ++i) time_window(w, 0, K, 4, 1);
        // ---- A: hipMalloc'ed sets, held ----------------------------------------------------------------------------------------
        std::vector<Set> held{w};
        for (int i = 1; i < 6; ++i) {
            Set t;
            CK(hipMalloc(&t.obs, b_obs)); CK(hipMalloc(&t.rew, b_rew)); CK(hipMalloc(&t.act, b_act)); CK(hipMalloc(&t.term, b_flag)); CK(hipMalloc(&t.trunc, b_flag));
            held.push_back(t);
        }
        for (size_t i = 0; i < held.size(); ++i) { time_window(held[i], 0, K, 2, 1); print_windows("A_hipMalloc", (int)i, held[i], 32); }
        for (auto &t : held) { hipFree(t.obs); hipFree(t.rew); hipFree(t.act); hipFree(t.term); hipFree(t.trunc); }
    }

    // ---- VMM set-up ----------------------------------------------------------------------------------------------------------------
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    const int per_set = (int)(total / chunk);
    const int pool_n = (int)(per_set * pool_factor);
    printf("{\"exp\": \"vmm\", \"granularity_min\": %zu, \"granularity_recommended\": %zu, \"chunk_MiB\": %zu, \"chunks_per_set\": %d, \"pool\": %d}\n", gmin, grec,
           chunk >> 20, per_set, pool_n);
    if (total % chunk || b_flag % chunk) { printf("{\"error\": \"chunk must divide the flag tensors\"}\n"); return 1; }
    std::vector<hipMemGenericAllocationHandle_t> pool(pool_n);
    for (int i = 0; i < pool_n; ++i) CK(hipMemCreate(&pool[i], chunk, &prop, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    // ---- B: every chunk on its own: flat fill ---------------------------------------------------------------------------------------
    std::vector<float> chunk_us(pool_n);
    {
        void *va;
        CK(hipMemAddressReserve(&va, chunk, 0, nullptr, 0));
        for (int i = 0; i < pool_n; ++i) {
            CK(hipMemMap(va, chunk, 0, pool[i], 0));
            CK(hipMemSetAccess(va, chunk, &acc, 1));
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(ev0, s));
                for (int j = 0; j < 8; ++j) hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, s, (float4 *)va, chunk / 16);
                CK(hipEventRecord(ev1, s));
                CK(hipEventSynchronize(ev1));
                float ms;
                CK(hipEventElapsedTime(&ms, ev0, ev1));
                if (rep) best = std::min(best, ms * 1e3f / 8);
            }
            chunk_us[i] = best;
            CK(hipMemUnmap(va, chunk));
        }
        CK(hipMemAddressFree(va, chunk));
        printf("{\"exp\": \"B_chunk_fill\", \"GBs\": [");
        for (int i = 0; i < pool_n; ++i) printf("%s%.0f", i ? ", " : "", chunk / 1e3 / chunk_us[i]);
        printf("]}\n");
        fflush(stdout);
    }

    // ---- C: the set in one reserved range ---------------------------------------------------------------------------------------------
    char *base;
    CK(hipMemAddressReserve((void **)&base, total, 0, nullptr, 0));
    Set v;
    v.obs = (float4 *)base;
    v.rew = (double *)(base + b_obs);
    v.act = (int64_t *)(base + b_obs + b_rew);
    v.term = (uint8_t *)(base + b_obs + b_rew + b_act);
    v.trunc = v.term + b_flag;
    std::vector<int> mapped;
    auto map_set = [&](const std::vector<int> &assign) {
        if (!mapped.empty()) CK(hipMemUnmap(base, total));
        for (int j = 0; j < per_set; ++j) CK(hipMemMap(base + (size_t)j * chunk, chunk, 0, pool[assign[j]], 0));
        CK(hipMemSetAccess(base, total, &acc, 1));
        mapped = assign;
        time_window(v, 0, K, 2, 1);   // first touch
    };
    const int W = std::max(8, (int)(chunk / ((size_t)N * 16)));   // steps per obs chunk
    std::vector<int> ident(per_set);
    std::iota(ident.begin(), ident.end(), 0);
    map_set(ident);
    printf("{\"exp\": \"C_base\", \"va\": \"0x%llx\"}\n", (unsigned long long)(uintptr_t)base);
    print_windows("C_identity", 0, v, W);
    print_windows_reversed("C_identity_reversed", v, W);
    print_windows("C_identity_fine", 0, v, 4);
    print_windows_m<1>("C_only_obs", v, W);
    print_windows_m<2>("C_only_rew", v, W);
    print_windows_m<4>("C_only_act", v, W);
    print_windows_m<8>("C_only_flags", v, W);
    print_windows_m<6>("C_rew_act", v, W);
    print_windows_m<7>("C_obs_rew_act", v, W);
    print_windows_m<9>("C_obs_flags", v, W);
    print_windows_m<14>("C_rew_act_flags", v, W);
    print_windows_m<15>("C_all_masked_kernel", v, W);
    for (int rot : {1, per_set / 4, per_set / 2, per_set - 3}) {
        std::vector<int> a(per_set);
        for (int j = 0; j < per_set; ++j) a[j] = (j + rot) % per_set;
        map_set(a);
        printf("{\"exp\": \"C_rotated\", \"rot\": %d, \"whole_us\": %.3f}\n", rot, time_window(v, 0, K, 3, 3));
        fflush(stdout);
    }
    std::mt19937 rng(12345);
    std::vector<int> all(pool_n);
    std::iota(all.begin(), all.end(), 0);
    float best_us = 1e30f, worst_us = 0.f;
    std::vector<int> best_assign, worst_assign;
    for (int draw = 0; draw < 12; ++draw) {
        std::shuffle(all.begin(), all.end(), rng);
        std::vector<int> a(all.begin(), all.begin() + per_set);
        if (draw >= 8) a = std::vector<int>(mapped), std::shuffle(a.begin(), a.end(), rng);   // same chunks, shuffled roles
        map_set(a);
        const float us = time_window(v, 0, K, 3, 3);
        printf("{\"exp\": \"C_random\", \"draw\": %d, \"same_chunks_shuffled\": %s, \"whole_us\": %.3f}\n", draw, draw >= 8 ? "true" : "false", us);
        fflush(stdout);
        if (us < best_us) best_us = us, best_assign = a;
        if (us > worst_us) worst_us = us, worst_assign = a;
    }
    map_set(worst_assign);
    print_windows("C_worst_again", 0, v, W);
    map_set(best_assign);
    print_windows("C_best_again", 0, v, W);

    // ---- D: chunks sorted by their own fill speed: fastest 34 vs slowest 34 -------------------------------------------------------------
    {
        std::vector<int> order(pool_n);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return chunk_us[a] < chunk_us[b]; });
        std::vector<int> fast(order.begin(), order.begin() + per_set), slow(order.end() - per_set, order.end());
        map_set(fast);
        printf("{\"exp\": \"D_fastest_fill_chunks\", \"whole_us\": %.3f}\n", time_window(v, 0, K, 3, 3));
        map_set(slow);
        printf("{\"exp\": \"D_slowest_fill_chunks\", \"whole_us\": %.3f}\n", time_window(v, 0, K, 3, 3));
        fflush(stdout);
    }
    // ---- E: per-role hill climbing from the best draw: for each role j try 3 unused chunks, keep improvements (whole-set time) ---------------
    {
        map_set(best_assign);
        std::vector<int> cur = best_assign;
        float cur_us = time_window(v, 0, K, 3, 3);
        std::vector<char> used(pool_n, 0);
        for (int c : cur) used[c] = 1;
        int swaps = 0, tried = 0;
        for (int j = 0; j < per_set && tried < 40; j += std::max(1, per_set / 12)) {
            for (int c = 0; c < pool_n && tried < 40; ++c) {
                if (used[c]) continue;
                std::vector<int> a = cur;
                a[j] = c;
                map_set(a);
                const float us = time_window(v, 0, K, 3, 2);
                ++tried;
                if (us < 0.985f * cur_us) { used[cur[j]] = 0; used[c] = 1; cur = a; cur_us = us; ++swaps; }
                break;
            }
        }
        map_set(cur);
        printf("{\"exp\": \"E_hill_climb\", \"tried\": %d, \"swaps\": %d, \"whole_us\": %.3f}\n", tried, swaps, time_window(v, 0, K, 3, 3));
    }
    CK(hipMemUnmap(base, total));
    CK(hipMemAddressFree(base, total));
    for (auto h : pool) CK(hipMemRelease(h));
    printf("{\"exp\": \"done\"}\n");
    return 0;
}
