// tools/wbench.hip — write-path microbenchmark: what the MI355X sustains for the fused rollout's store pattern
// with the physics removed.  Each lane "owns" E envs (wave-dense striding as in step_kernel) and, for K steps,
// writes obs (16 B), reward (RB B), action (AB B), terminated (1 B), truncated (1 B) of every env to [K][N]
// trajectory arrays.  Variants: plain stores, nontemporal stores, flags packed to one dword per 4 envs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/wbench tools/wbench.hip && tools/wbench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ void st(T *p, T v, bool nt) {
    if (nt) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ void st(float4 *p, float4 v, bool nt) {
    v4f w = {v.x, v.y, v.z, v.w};
    if (nt) __builtin_nontemporal_store(w, reinterpret_cast<v4f *>(p)); else *p = v;
}

template <int E, int BLOCK, typename RT, typename AT, bool NT, int FLAGMODE, int MAP = 0>
__global__ void __launch_bounds__(BLOCK) traj_write(float4 *obs, RT *rew, AT *act, uint8_t *term, uint8_t *trunc, int64_t n, int K, int k0 = 0, int k1 = -1) {
    const int tid = threadIdx.x;
    int64_t tile = blockIdx.x;
    if (MAP == 1) tile = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD x owns a contiguous eighth
    const int64_t tile0 = tile * (E * BLOCK);
    float x = (float)tid;
    if (k1 < 0) k1 = K;
    for (int kk = k0; kk < k1; ++kk) {
        const int k = MAP == 2 ? (int)((kk + blockIdx.x * 37) % K) : kk;
        const int64_t so = MAP == 3 ? (tile0 * K + (int64_t)k * (E * BLOCK) - tile0) : (int64_t)k * n;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int64_t e = so + tile0 + (int64_t)j * BLOCK + tid;
            x = x * 1.0001f + 0.5f;
            st(obs + e, make_float4(x, x + 1, x + 2, x + 3), NT);
            st(rew + e, (RT)1, NT);
            st(act + e, (AT)(k & 1), NT);
            if (FLAGMODE == 0) {
                st(term + e, (uint8_t)(x > 3.f), NT);
                st(trunc + e, (uint8_t)(x > 5.f), NT);
            } else if (FLAGMODE == 1) {  // 4 lanes' flag bytes gathered into one dword store by every 4th lane
                uint32_t a = (x > 3.f), b = (x > 5.f);
                a |= __shfl_down(a, 1) << 8;  a |= __shfl_down(a, 2) << 16;
                b |= __shfl_down(b, 1) << 8;  b |= __shfl_down(b, 2) << 16;
                if ((tid & 3) == 0) {
                    st(reinterpret_cast<uint32_t *>(term + e), a, NT);
                    st(reinterpret_cast<uint32_t *>(trunc + e), b, NT);
                }
            } else {  // one combined flag byte per env
                st(term + e, (uint8_t)((x > 3.f) | ((x > 5.f) << 1)), NT);
            }
        }
    }
}

__global__ void fill16(float4 *p, int64_t n, int reps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < reps; ++r)
        for (int64_t k = i; k < n; k += (int64_t)gridDim.x * blockDim.x) p[k] = make_float4(1.f, 2.f, 3.f, (float)r);
}
__global__ void fill16_flat(float4 *p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void copy16(const float4 *x, float4 *y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i];
}

template <typename F>
float time_ms(F f, int reps = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        f();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    const int64_t n = 1 << 20;
    const int K = 64;
    float4 *obs; double *rew; int64_t *act; uint8_t *term, *trunc;
    CK(hipMalloc(&obs, n * K * 16)); CK(hipMalloc(&rew, n * K * 8)); CK(hipMalloc(&act, n * K * 8));
    CK(hipMalloc(&term, n * K)); CK(hipMalloc(&trunc, n * K));
    {
        const int64_t m = n * K;  // 1 GiB of float4
        float ms = time_ms([&] { fill16_flat<<<(unsigned)((m + 255) / 256), 256>>>(obs, m); });
        printf("fill16_flat  %6.1f GB/s  (1 GiB, one float4 per lane, one launch)\n", m * 16 / ms / 1e6);
        ms = time_ms([&] { fill16<<<2048, 256>>>(obs, m, 1); });
        printf("fill16_grid  %6.1f GB/s  (1 GiB, grid-stride, 2048 WGs)\n", m * 16 / ms / 1e6);
        float4 *src = reinterpret_cast<float4 *>(rew);  // 512 MiB
        const int64_t c = n * K / 2;
        ms = time_ms([&] { copy16<<<(unsigned)((c + 255) / 256), 256>>>(src, obs, c); });
        printf("copy16       %6.1f GB/s  read+write (512 MiB each way)\n", 2.0 * c * 16 / ms / 1e6);
    }
#define RUN(E, BLOCK, RT, AT, NT, FM, label)                                                                         \
    {                                                                                                                 \
        const unsigned grid = (unsigned)(n / (E * BLOCK));                                                            \
        float ms = time_ms([&] {                                                                                      \
            traj_write<E, BLOCK, RT, AT, NT, FM><<<grid, BLOCK>>>(obs, (RT *)rew, (AT *)act, term, trunc, n, K);      \
        });                                                                                                           \
        const double bytes = (16.0 + sizeof(RT) + sizeof(AT) + (FM == 2 ? 1 : 2)) * n * K;                            \
        printf("%-44s %6.2f us/step  %6.1f GB/s (%.0f B/env-step)\n", label, ms * 1e3 / K, bytes / ms / 1e6,          \
               bytes / n / K);                                                                                        \
    }
    RUN(2, 256, double, int64_t, false, 0, "E2 B256 f64/i64 bytes");
#define RUNM(E, BLOCK, MAP, label)                                                                                    \
    {                                                                                                                 \
        const unsigned grid = (unsigned)(n / (E * BLOCK));                                                            \
        float ms = time_ms([&] {                                                                                      \
            traj_write<E, BLOCK, double, int64_t, false, 0, MAP><<<grid, BLOCK>>>(obs, rew, act, term, trunc, n, K);  \
        });                                                                                                           \
        printf("%-44s %6.2f us/step  %6.1f GB/s\n", label, ms * 1e3 / K, 34.0 * n * K / ms / 1e6);                    \
    }
    RUNM(2, 256, 1, "E2 B256 XCD-contiguous tiles");
    RUNM(2, 256, 2, "E2 B256 staggered steps (diagnostic)");
    RUNM(2, 256, 3, "E2 B256 tile-major layout [N/T][K][T]");
    RUNM(4, 256, 1, "E4 B256 XCD-contiguous tiles");
    RUNM(4, 256, 3, "E4 B256 tile-major layout");
    {
        const unsigned grid = (unsigned)(n / (2 * 256));
        float ms = time_ms([&] {
            for (int k = 0; k < K; ++k)
                traj_write<2, 256, double, int64_t, false, 0, 0><<<grid, 256>>>(obs, rew, act, term, trunc, n, K, k, k + 1);
        });
        printf("%-44s %6.2f us/step  %6.1f GB/s\n", "E2 B256 one launch per step (K launches)", ms * 1e3 / K, 34.0 * n * K / ms / 1e6);
    }
    {   // row pitch / base offset experiment (partition camping?): XCD-contiguous tiles, 1-wave workgroups like rollout_kernel
        const unsigned grid = (unsigned)(n / (2 * 64));
        for (int64_t padenv : {0ll, 64ll, 1024ll, 4096ll + 64, 65536ll + 1024}) {
            for (int stagger = 0; stagger < 2; ++stagger) {
                const int64_t pitch = n + padenv;
                // pitch-aware variant: reuse traj_write with n := pitch for the row stride (tiles still cover n envs)
                float4 *o = obs + (stagger ? 3 * 4096 / 16 : 0);
                double *r = rew + (stagger ? 5 * 4096 / 8 + 17 * 64 : 0);
                int64_t *a = act + (stagger ? 9 * 4096 / 8 + 33 * 64 : 0);
                uint8_t *te = term + (stagger ? 13 * 4096 + 49 * 512 : 0), *tr = trunc + (stagger ? 21 * 4096 + 77 * 512 : 0);
                const int Kp = 56;  // fewer rows so that padded rows still fit the allocations
                float ms = time_ms([&] { traj_write<2, 64, double, int64_t, false, 0, 1><<<grid, 64>>>(o, r, a, te, tr, pitch, Kp); });
                printf("E2 B64 XCD pitch=N+%-6lld stagger=%d          %6.2f us/step  %6.1f GB/s\n", (long long)padenv, stagger,
                       ms * 1e3 / Kp, 34.0 * n * Kp / ms / 1e6);
            }
        }
    }
    RUN(2, 256, double, int64_t, true, 0, "E2 B256 f64/i64 bytes nontemporal");
    RUN(2, 256, double, int64_t, false, 1, "E2 B256 f64/i64 packed-flag dwords");
    RUN(2, 256, double, int64_t, true, 1, "E2 B256 f64/i64 packed-flag dwords nt");
    RUN(2, 256, double, int64_t, false, 2, "E2 B256 f64/i64 one flag byte");
    RUN(2, 64, double, int64_t, false, 0, "E2 B64  f64/i64 bytes");
    RUN(2, 64, double, int64_t, true, 1, "E2 B64  f64/i64 packed nt");
    RUN(1, 256, double, int64_t, false, 0, "E1 B256 f64/i64 bytes");
    RUN(4, 256, double, int64_t, false, 0, "E4 B256 f64/i64 bytes");
    RUN(2, 256, float, int32_t, false, 0, "E2 B256 f32/i32 bytes");
    RUN(2, 256, float, int32_t, true, 1, "E2 B256 f32/i32 packed nt");
    RUN(2, 256, float, int32_t, false, 2, "E2 B256 f32/i32 one flag byte");
    return 0;
}
