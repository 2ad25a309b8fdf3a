// server_probe.hip — feasibility probe of a RESIDENT step(actions) kernel.
//
// Question: can a persistent kernel that keeps the fp64 env state in LDS across vector steps and takes each step's actions
// from a device mailbox beat one step_kernel launch per step (18.7-19.0 us per 2^20-env CartPole step, 106 B of real traffic)?
// The resident form moves 34 B per env-step (actions in; obs, reward, flags out) but pays a hand-shake per step in each
// direction.  This probe runs the REAL CartPole step (Env<MXV_CARTPOLE>::step, TimeLimit, lazy-ordinal autoreset) under four
// hand-shakes and prints one JSON line per configuration:
//   free   : no hand-shake at all (the resident body's own time per step: the floor)
//   memops : hipStreamWriteValue64(mail) / hipStreamWaitValue64(done) on the policy's stream (command-processor packets)
//   kernels: a 1-lane post kernel / a 1-lane wait kernel on the policy's stream
//   +policy: either of the two with a bandwidth-shaped stand-in for the policy between wait and post (reads obs, writes actions)
// and, beside them, the same work as one ordinary kernel launch per step (fp64 state round trip) on the same buffers.
// Every spin is bounded by a wall-clock watchdog (the kernel parks itself: state back to HBM, exit), so a protocol bug costs
// 50 ms, never a hung box.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I gym_amd/csrc tools/server_probe.hip -o tools/_bin/server_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "mxv_device.hpp"

using namespace mxv;

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
            exit(2);                                                                               \
        }                                                                                          \
    } while (0)

typedef __attribute__((address_space(1))) uint64_t gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define RLX_SYS __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM

struct Ctl {                 // every polled word on its own 128-B line
    uint64_t mail;           // policy -> server: epoch e means "the actions of step e-1 are in memory"
    uint64_t pad0[15];
    uint64_t stop;           // host -> server: leave at the next poll
    uint64_t pad1[15];
    uint64_t top;            // arrivals of XCD leaders
    uint64_t pad2[15];
    uint64_t parked;         // workgroups that left (stop, step budget or watchdog)
    uint64_t gave_up;        // ... of them through the watchdog
    uint64_t pad3[14];
    uint64_t xcd[8][16];     // per-XCD arrival counters
};

struct ServerArgs {
    double *state;           // [4][N]
    int32_t *elapsed;        // [N]
    uint32_t *episodes;      // [N] reset ordinals (touched only by lanes that reset)
    const int64_t *actions;  // [N]
    float *obs;              // [N][4]
    double *reward;          // [N]
    uint8_t *terminated, *truncated;
    Ctl *ctl;
    uint64_t *done;          // signal word: = epoch once every workgroup has stored step epoch-1's outputs
    uint64_t *stamps;        // [max_steps][4] wall-clock stamps of workgroup 0 (mail seen, loads issued, stores drained, arrived)
    int64_t n;
    uint64_t base_seed;
    int32_t max_steps;       // TimeLimit
    int32_t step_budget;     // leave after this many steps
    int32_t handshake;       // 0 = free running
    int32_t write_through;   // 1 = sc1 stores, 0 = plain stores + release fence
    uint64_t watchdog_ticks; // wall_clock64 ticks (100 MHz) without mail before giving up
    EnvParams P;
};

#ifndef PROBE_SAFE
#define PROBE_SAFE false   // the autoreset invariant holds (|theta| <= pi/4 on entry), as in rollout_kernel_v3
#endif
#ifndef KTHREADS
#define KTHREADS 512
#endif
constexpr int kThreads = KTHREADS;
constexpr int kEnvsPerWg = 4096;
constexpr int kPerLane = kEnvsPerWg / kThreads;   // 8
constexpr int kLdsBytes = kEnvsPerWg * (4 * 8 + 2) + 16;   // + the go word (no static __shared__: it would shift the dynamic base off 16 B)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_obs_wt(float *p, float4 v, bool wt) {
    if (wt) {
        const u32x4 q = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(q) : "memory");
    } else {
        *reinterpret_cast<float4 *>(p) = v;
    }
}
__device__ __forceinline__ void store_u128_wt(void *p, uint4 v, bool wt) {
    if (wt) {
        const u32x4 q = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(q) : "memory");
    } else {
        *reinterpret_cast<uint4 *>(p) = v;
    }
}
__device__ __forceinline__ void store_f64_wt(double *p, double v, bool wt) {
    if (wt) {
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    } else {
        *p = v;
    }
}

// 16 flag bits -> 16 bytes of 0/1
__device__ __forceinline__ uint4 spread16(uint32_t bits) {
    auto four = [](uint32_t b) -> uint32_t { return ((b & 0xFu) * 0x00204081u) & 0x01010101u; };
    return make_uint4(four(bits), four(bits >> 4), four(bits >> 8), four(bits >> 12));
}

extern __shared__ __attribute__((aligned(16))) char smem[];

__global__ void __launch_bounds__(kThreads, kThreads / 256) server_kernel(const ServerArgs a) {
    using EV = Env<MXV_CARTPOLE>;
    double *ls = reinterpret_cast<double *>(smem);                                // [4][4096]
    uint16_t *lel = reinterpret_cast<uint16_t *>(smem + kEnvsPerWg * 32);         // [4096]
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t wg0 = (int64_t)blockIdx.x * kEnvsPerWg;
    const Par<PM_DEFAULT> P(a.P);
    const bool wt = a.write_through != 0;
    volatile int *s_go = reinterpret_cast<volatile int *>(smem + kEnvsPerWg * 34);
    // ---- park in: HBM -> LDS ----
#pragma unroll
    for (int j = 0; j < kPerLane; ++j) {
        const int el = j * kThreads + tid;
#pragma unroll
        for (int k = 0; k < 4; ++k) ls[k * kEnvsPerWg + el] = a.state[(int64_t)k * a.n + wg0 + el];
        lel[el] = (uint16_t)a.elapsed[wg0 + el];
    }
    __syncthreads();
    const unsigned xcd = blockIdx.x % 8u;
    const uint64_t per_xcd = gridDim.x / 8u + (xcd < gridDim.x % 8u ? 1u : 0u);
    int reason = 0;
    int step = 0;
    for (; step < a.step_budget; ++step) {
        const uint64_t epoch = (uint64_t)step + 1;
        if (a.handshake) {
            if (tid == 0) {
                const uint64_t t_in = wall_clock64();
                int go = 1;
                while (__hip_atomic_load((gu64 *)&a.ctl->mail, RLX_SYS) < epoch) {
                    __builtin_amdgcn_s_sleep(2);
                    if (__hip_atomic_load((gu64 *)&a.ctl->stop, RLX_AGENT) != 0) { go = 0; break; }
                    if (wall_clock64() - t_in > a.watchdog_ticks) { go = -1; break; }
                }
                if (go == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                *s_go = go;
            }
            __syncthreads();
            const int go = *s_go;
            if (go != 1) { reason = go == 0 ? 1 : 2; break; }
        }
        uint64_t st0 = 0;
        if (blockIdx.x == 0 && tid == 0) st0 = wall_clock64();
        int64_t act[kPerLane];
#pragma unroll
        for (int j = 0; j < kPerLane; ++j) act[j] = a.actions[wg0 + j * kThreads + tid];
        uint64_t st1 = 0;
        if (blockIdx.x == 0 && tid == 0) st1 = wall_clock64();
#pragma unroll(kThreads >= 1024 ? 1 : 2)
        for (int j = 0; j < kPerLane; ++j) {
            const int el = j * kThreads + tid;
            const int64_t e = wg0 + el;
            double s[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] = ls[k * kEnvsPerWg + el];
            int elapsed = lel[el];
            double rew;
            float o[4];
            bool term = EV::template step<PM_DEFAULT, PROBE_SAFE>(P, s, nullptr, elapsed == 0, (int)act[j], 0.0f, rew, o);
            elapsed += 1;
            const bool trunc = a.max_steps > 0 && elapsed >= a.max_steps;
            if (term || trunc) {   // sync_vector_env.py:152-156; the ordinal lives in HBM and only resetting lanes touch it
                const uint32_t k = landed(a.episodes[e]);
                a.episodes[e] = k + 1;
                const U4 w = episode_reset_words(a.base_seed + (uint64_t)e, k);
                EV::reset(w, -0.05, 0.05, s);
                EV::observe(s, o);
                elapsed = 0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) ls[k * kEnvsPerWg + el] = s[k];
            lel[el] = (uint16_t)elapsed;
            store_obs_wt(a.obs + e * 4, make_float4(o[0], o[1], o[2], o[3]), wt);
            store_f64_wt(a.reward + e, rew, wt);
            const uint64_t bt = __ballot(term), bu = __ballot(trunc);
            if (lane < 4) {   // 64 flag bytes of the wave as four 16-byte stores
                const int64_t e0 = e - lane + lane * 16;
                store_u128_wt(a.terminated + e0, spread16((uint32_t)(bt >> (16 * lane))), wt);
                store_u128_wt(a.truncated + e0, spread16((uint32_t)(bu >> (16 * lane))), wt);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains
        __syncthreads();
        uint64_t st2 = 0;
        if (blockIdx.x == 0 && tid == 0) st2 = wall_clock64();
        if (tid == 0) {
            if (!wt) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const uint64_t old = __hip_atomic_fetch_add((gu64 *)&a.ctl->xcd[xcd][0], 1ull, RLX_AGENT);
            if (old + 1 == per_xcd * epoch) {
                const uint64_t o2 = __hip_atomic_fetch_add((gu64 *)&a.ctl->top, 1ull, RLX_AGENT);
                const uint64_t groups = gridDim.x < 8u ? gridDim.x : 8u;
                if (o2 + 1 == groups * epoch) __hip_atomic_store((gu64 *)a.done, epoch, RLX_SYS);
            }
        }
        if (blockIdx.x == 0 && tid == 0 && a.stamps) {
            uint64_t *q = a.stamps + (int64_t)step * 4;
            q[0] = st0; q[1] = st1; q[2] = st2; q[3] = wall_clock64();
        }
    }
    // ---- park out: LDS -> HBM ----
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPerLane; ++j) {
        const int el = j * kThreads + tid;
#pragma unroll
        for (int k = 0; k < 4; ++k) a.state[(int64_t)k * a.n + wg0 + el] = ls[k * kEnvsPerWg + el];
        a.elapsed[wg0 + el] = lel[el];
    }
    if (tid == 0) {
        atomicAdd((unsigned long long *)&a.ctl->parked, 1ull);
        if (reason == 2) atomicAdd((unsigned long long *)&a.ctl->gave_up, 1ull);
    }
}

// The same work as ONE ordinary launch per step (state in HBM, fp64 round trip): what the resident kernel has to beat.
__global__ void __launch_bounds__(256) launch_per_step_kernel(const ServerArgs a) {
    using EV = Env<MXV_CARTPOLE>;
    const Par<PM_DEFAULT> P(a.P);
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n) return;
    double s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = a.state[(int64_t)k * a.n + e];
    int elapsed = a.elapsed[e];
    const int act = (int)a.actions[e];
    double rew;
    float o[4];
    bool term = EV::template step<PM_DEFAULT, PROBE_SAFE>(P, s, nullptr, elapsed == 0, act, 0.0f, rew, o);
    elapsed += 1;
    const bool trunc = a.max_steps > 0 && elapsed >= a.max_steps;
    if (term || trunc) {
        const uint32_t k = landed(a.episodes[e]);
        a.episodes[e] = k + 1;
        const U4 w = episode_reset_words(a.base_seed + (uint64_t)e, k);
        EV::reset(w, -0.05, 0.05, s);
        EV::observe(s, o);
        elapsed = 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) a.state[(int64_t)k * a.n + e] = s[k];
    a.elapsed[e] = elapsed;
    *reinterpret_cast<float4 *>(a.obs + e * 4) = make_float4(o[0], o[1], o[2], o[3]);
    a.reward[e] = rew;
    a.terminated[e] = term;
    a.truncated[e] = trunc;
}

__global__ void post_kernel(uint64_t *mail, uint64_t epoch) { __hip_atomic_store((gu64 *)mail, epoch, RLX_AGENT); }
__global__ void wait_kernel(uint64_t *done, uint64_t epoch, uint64_t ticks, uint64_t *timeouts) {
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load((gu64 *)done, RLX_SYS) < epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > ticks) { atomicAdd((unsigned long long *)timeouts, 1ull); break; }
    }
}
// post + wait in ONE kernel (one boundary per step on the policy's stream instead of two)
__global__ void postwait_kernel(uint64_t *mail, uint64_t *done, uint64_t epoch, uint64_t ticks, uint64_t *timeouts) {
    __hip_atomic_store((gu64 *)mail, epoch, RLX_AGENT);
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load((gu64 *)done, RLX_SYS) < epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > ticks) { atomicAdd((unsigned long long *)timeouts, 1ull); break; }
    }
}
// The best case for the mailbox: a policy that speaks the protocol itself.  Every workgroup waits for the server's outputs of the
// previous step at its top (relaxed poll + one agent acquire), stores its actions write-through, and the last workgroup to finish
// posts the mail — no extra kernel, no command-processor packet between policy and server.
__global__ void __launch_bounds__(256) policy_coop_kernel(const float *obs, int64_t *actions, int64_t n, uint64_t *mail, uint64_t *done,
                                                          uint64_t *ticket, uint64_t epoch, uint64_t ticks, uint64_t *timeouts) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int o = 1;
        if (epoch > 1) {
            const uint64_t t0 = wall_clock64();
            while (__hip_atomic_load((gu64 *)done, RLX_SYS) < epoch - 1) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > ticks) { o = 0; atomicAdd((unsigned long long *)timeouts, 1ull); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        ok = o;
    }
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n && ok) {
        const float4 o = reinterpret_cast<const float4 *>(obs)[e];
        const int64_t act = (o.z + o.w > 0.0f) ? 1 : 0;
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(actions + e), "v"(act) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t old = __hip_atomic_fetch_add((gu64 *)ticket, 1ull, RLX_AGENT);
        if (old + 1 == (uint64_t)gridDim.x * epoch) __hip_atomic_store((gu64 *)mail, epoch, RLX_AGENT);
    }
}
// stand-in for a policy: reads every observation row, writes every action (24 B per env, pure bandwidth)
__global__ void __launch_bounds__(256) policy_kernel(const float *obs, int64_t *actions, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float4 o = reinterpret_cast<const float4 *>(obs)[e];
    actions[e] = (o.z + o.w > 0.0f) ? 1 : 0;
}
__global__ void init_kernel(ServerArgs a) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n) return;
    const U4 w = episode_reset_words(a.base_seed + (uint64_t)e, 0);
    double s[4];
    Env<MXV_CARTPOLE>::reset(w, -0.05, 0.05, s);
    for (int k = 0; k < 4; ++k) a.state[(int64_t)k * a.n + e] = s[k];
    a.elapsed[e] = 0;
    a.episodes[e] = 1;
    const_cast<int64_t *>(a.actions)[e] = (w.x >> 7) & 1;
    reinterpret_cast<float4 *>(a.obs)[e] = make_float4((float)s[0], (float)s[1], (float)s[2], (float)s[3]);
}

struct Result {
    std::string name;
    double us_per_step;
    int steps;
    uint64_t parked, gave_up, wait_timeouts;
    double wg0_work_us, wg0_wait_us;
    uint64_t checksum;
};

static uint64_t checksum_state(const ServerArgs &a) {
    std::vector<double> h(4 * a.n);
    std::vector<uint32_t> ep(a.n);
    CK(hipMemcpy(h.data(), a.state, sizeof(double) * 4 * a.n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ep.data(), a.episodes, sizeof(uint32_t) * a.n, hipMemcpyDeviceToHost));
    uint64_t c = 1469598103934665603ull;
    for (double d : h) { uint64_t b; memcpy(&b, &d, 8); c = (c ^ b) * 1099511628211ull; }
    for (uint32_t v : ep) c = (c ^ v) * 1099511628211ull;
    return c;
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int steps = argc > 2 ? atoi(argv[2]) : 2000;
    const int wgs = (int)(n / kEnvsPerWg);
    if (n % kEnvsPerWg) { fprintf(stderr, "n must be a multiple of %d\n", kEnvsPerWg); return 2; }
    int can_wait = 0;
    CK(hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    fprintf(stderr, "device %s, CUs %d, canUseStreamWaitValue %d, wgs %d, lds %d B\n", prop.gcnArchName, prop.multiProcessorCount, can_wait, wgs,
            kLdsBytes);
    CK(hipFuncSetAttribute((const void *)server_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, server_kernel, kThreads, kLdsBytes));
    fprintf(stderr, "occupancy API: %d workgroup(s) per CU\n", occ);
    if (wgs > prop.multiProcessorCount * occ) { fprintf(stderr, "grid would not be resident\n"); return 2; }

    ServerArgs a{};
    a.n = n;
    a.base_seed = 1234;
    a.max_steps = 500;
    a.watchdog_ticks = 5000000;   // 50 ms at 100 MHz
    CK(hipMalloc(&a.state, sizeof(double) * 4 * n));
    CK(hipMalloc(&a.elapsed, 4 * n));
    CK(hipMalloc(&a.episodes, 4 * n));
    int64_t *actions;
    CK(hipMalloc(&actions, 8 * n));
    a.actions = actions;
    CK(hipMalloc(&a.obs, 16 * n));
    CK(hipMalloc(&a.reward, 8 * n));
    CK(hipMalloc(&a.terminated, n));
    CK(hipMalloc(&a.truncated, n));
    CK(hipMalloc(&a.ctl, sizeof(Ctl)));
    CK(hipMalloc(&a.stamps, sizeof(uint64_t) * 4 * (steps + 8)));
    uint64_t *sig = nullptr, *wait_timeouts;
    CK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
    CK(hipMalloc(&wait_timeouts, 8));
    a.done = sig;
    hipStream_t S, Pq;
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&Pq, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = (int)((n + 255) / 256);
    std::vector<Result> results;

    auto fresh = [&]() {
        hipLaunchKernelGGL(init_kernel, dim3(blocks), dim3(256), 0, Pq, a);
        CK(hipMemsetAsync(a.ctl, 0, sizeof(Ctl), Pq));
        CK(hipMemsetAsync(sig, 0, 8, Pq));
        CK(hipMemsetAsync(wait_timeouts, 0, 8, Pq));
        CK(hipStreamSynchronize(Pq));
    };
    auto finish = [&](Result &r) {
        Ctl h;
        uint64_t wt_ = 0;
        CK(hipMemcpyAsync(&h, a.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, Pq));
        CK(hipMemcpyAsync(&wt_, wait_timeouts, 8, hipMemcpyDeviceToHost, Pq));
        CK(hipStreamSynchronize(Pq));
        r.parked = h.parked; r.gave_up = h.gave_up; r.wait_timeouts = wt_;
        std::vector<uint64_t> st(4 * (size_t)r.steps);
        CK(hipMemcpy(st.data(), a.stamps, 8 * st.size(), hipMemcpyDeviceToHost));
        double work = 0, wait = 0;
        int cnt = 0;
        for (int i = r.steps / 2; i + 1 < r.steps; ++i) {   // second half: steady state
            work += (double)(st[4 * i + 3] - st[4 * i + 0]);
            wait += (double)(st[4 * (i + 1) + 0] - st[4 * i + 3]);
            ++cnt;
        }
        r.wg0_work_us = cnt ? work / cnt / 100.0 : 0;
        r.wg0_wait_us = cnt ? wait / cnt / 100.0 : 0;
        r.checksum = checksum_state(a);
        results.push_back(r);
    };

    // ---- (0) one ordinary launch per step ----
    for (int rep = 0; rep < 2; ++rep) {
        fresh();
        CK(hipEventRecord(e0, Pq));
        for (int t = 0; t < steps; ++t) hipLaunchKernelGGL(launch_per_step_kernel, dim3(blocks), dim3(256), 0, Pq, a);
        CK(hipEventRecord(e1, Pq));
        CK(hipStreamSynchronize(Pq));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 1) {
            Result r{"launch_per_step", ms * 1e3 / steps, steps, 0, 0, 0, 0, 0, checksum_state(a)};
            results.push_back(r);
        }
    }
    // ---- (0b) launch per step + policy stand-in ----
    {
        fresh();
        CK(hipEventRecord(e0, Pq));
        for (int t = 0; t < steps; ++t) {
            hipLaunchKernelGGL(launch_per_step_kernel, dim3(blocks), dim3(256), 0, Pq, a);
            hipLaunchKernelGGL(policy_kernel, dim3(blocks), dim3(256), 0, Pq, a.obs, actions, n);
        }
        CK(hipEventRecord(e1, Pq));
        CK(hipStreamSynchronize(Pq));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        Result r{"launch_per_step+policy", ms * 1e3 / steps, steps, 0, 0, 0, 0, 0, checksum_state(a)};
        results.push_back(r);
    }

    // ---- resident configurations ----
    struct Cfg { const char *name; int handshake; int wt; int mode; int policy; };   // mode 0 free, 1 memops, 2 kernels
    const Cfg cfgs[] = {
        {"resident_free_wt", 0, 1, 0, 0},     {"resident_free_plain", 0, 0, 0, 0},
        {"resident_kernels_wt", 1, 1, 2, 0},  {"resident_kernels_plain", 1, 0, 2, 0},
        {"resident_memops_wt", 1, 1, 1, 0},   {"resident_kernels_wt+policy", 1, 1, 2, 1},
        {"resident_memops_wt+policy", 1, 1, 1, 1},   {"resident_postwait_wt", 1, 1, 3, 0},
        {"resident_postwait_wt+policy", 1, 1, 3, 1}, {"resident_coop_policy_wt", 1, 1, 4, 1},
    };   // mode 3: post + wait in one kernel; mode 4: the policy kernel itself waits and posts
    for (const Cfg &c : cfgs) {
        if (c.mode == 1 && !can_wait) continue;
        fresh();
        a.handshake = c.handshake;
        a.write_through = c.wt;
        a.step_budget = steps;
        if (c.mode == 0) CK(hipEventRecord(e0, S));
        hipLaunchKernelGGL(server_kernel, dim3(wgs), dim3(kThreads), kLdsBytes, S, a);
        CK(hipGetLastError());
        if (c.mode != 0) CK(hipEventRecord(e0, Pq));
        if (c.mode != 0) {
            for (int t = 0; t < steps; ++t) {
                const uint64_t epoch = (uint64_t)t + 1;
                if (c.mode == 4) {
                    hipLaunchKernelGGL(policy_coop_kernel, dim3(blocks), dim3(256), 0, Pq, a.obs, actions, n, &a.ctl->mail, sig, &a.ctl->pad3[8],
                                       epoch, a.watchdog_ticks, wait_timeouts);
                    continue;
                }
                if (c.policy) hipLaunchKernelGGL(policy_kernel, dim3(blocks), dim3(256), 0, Pq, a.obs, actions, n);
                if (c.mode == 3) {
                    hipLaunchKernelGGL(postwait_kernel, dim3(1), dim3(1), 0, Pq, &a.ctl->mail, sig, epoch, a.watchdog_ticks, wait_timeouts);
                } else if (c.mode == 1) {
                    CK(hipStreamWriteValue64(Pq, &a.ctl->mail, epoch, 0));
                    CK(hipStreamWaitValue64(Pq, sig, epoch, hipStreamWaitValueGte, ~0ull));
                } else {
                    hipLaunchKernelGGL(post_kernel, dim3(1), dim3(1), 0, Pq, &a.ctl->mail, epoch);
                    hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(1), 0, Pq, sig, epoch, a.watchdog_ticks, wait_timeouts);
                }
            }
            CK(hipEventRecord(e1, Pq));
            CK(hipStreamSynchronize(Pq));
        } else {
            CK(hipEventRecord(e1, S));
        }
        CK(hipStreamSynchronize(S));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        Result r{c.name, ms * 1e3 / steps, steps, 0, 0, 0, 0, 0, 0};
        finish(r);
    }
    for (const Result &r : results)
        printf("{\"probe\": \"server\", \"config\": \"%s\", \"envs\": %lld, \"steps\": %d, \"us_per_step\": %.3f, \"wg0_work_us\": %.3f, "
               "\"wg0_wait_us\": %.3f, \"parked\": %llu, \"gave_up\": %llu, \"wait_timeouts\": %llu, \"state_checksum\": \"%016llx\"}\n",
               r.name.c_str(), (long long)n, r.steps, r.us_per_step, r.wg0_work_us, r.wg0_wait_us, (unsigned long long)r.parked,
               (unsigned long long)r.gave_up, (unsigned long long)r.wait_timeouts, (unsigned long long)r.checksum);
    return 0;
}
