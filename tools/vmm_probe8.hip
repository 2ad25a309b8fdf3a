// tools/vmm_probe8.hip — is the same-class interference between the observation stream and the reward stream (vmm_probe5/7) a matter of
// the STORE WIDTH mix (16-B/lane dwordx4 next to 8-B/lane dwordx2; two 8-B streams do not interfere)?  Same bytes, same addresses, other
// instructions: the observation rows written as float4 per lane (shipped), as two dense dwordx2 stores, as four dense dword stores; and
// the reward stream written as dwordx4 (two lanes' worth per lane, half the lanes).  Both streams inside ONE hipMalloc'ed GiB (one class).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/vmm_probe8 tools/vmm_probe8.hip && tools/_bin/vmm_probe8
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// VAR: 0 = obs float4 per lane; 1 = obs as 2 dense dwordx2 stores per 64-env block; 2 = obs as 4 dense dword stores; 3 = obs float4, reward as
// dwordx4 from half the lanes; mask bit 0 = obs, bit 1 = reward
template <int VAR>
__global__ void __launch_bounds__(64, 4) stores(char *obs, char *rew, int64_t row, int K, int mask) {
    const unsigned bid = blockIdx.x, ntiles = gridDim.x;
    const unsigned tile = (bid % 8) * (ntiles / 8) + bid / 8;
    const int lane = threadIdx.x;
    float x = (float)lane;
    for (int k = 0; k < K; ++k) {
        x = x * 1.0001f + 0.5f;
        char *o = obs + ((int64_t)k * row + (int64_t)tile * 128) * 16;   // this wave's 2 KiB of the observation row
        char *r = rew + ((int64_t)k * row + (int64_t)tile * 128) * 8;    // this wave's 1 KiB of the reward row
        if (mask & 2) {
            if (VAR == 3) {
                if (lane < 32) {
                    reinterpret_cast<double2 *>(r)[lane] = make_double2(1.0, 1.0);
                    reinterpret_cast<double2 *>(r)[32 + lane] = make_double2(1.0, 1.0);
                }
            } else {
                reinterpret_cast<double *>(r)[lane] = 1.0;
                reinterpret_cast<double *>(r)[64 + lane] = 1.0;
            }
        }
        if (mask & 1) {
            if (VAR == 0 || VAR == 3) {
                reinterpret_cast<float4 *>(o)[lane] = make_float4(x, x + 1, 0.f, 1.f);
                reinterpret_cast<float4 *>(o)[64 + lane] = make_float4(x + 2, x, 1.f, 0.f);
            } else if (VAR == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) reinterpret_cast<float2 *>(o)[q * 64 + lane] = make_float2(x + q, x);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) reinterpret_cast<float *>(o)[q * 64 + lane] = x + q;
            }
        }
    }
}

static hipStream_t s;
static hipEvent_t e0, e1;
template <int VAR>
static float t(char *obs, char *rew, int mask) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int j = 0; j < 6; ++j) hipLaunchKernelGGL(stores<VAR>, dim3(8192), dim3(64), 0, s, obs, rew, (int64_t)1 << 20, 16, mask);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = std::min(best, ms * 1e3f / (6 * 16));
    }
    return best;
}

int main() {
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&s));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    char *a;
    CK(hipMalloc(&a, (size_t)1 << 30));
    char *obs = a, *rew = a + ((size_t)512 << 20);
    for (int i = 0; i < 300; ++i) t<0>(obs, rew, 3);
    printf("{\"exp\": \"same_class_pair\", \"what\": \"us per 2^20-lane step, observation stream + reward stream inside one allocation\",\n");
    printf(" \"obs_float4__rew_dwordx2\": %.3f, \"obs_2x_dwordx2_dense__rew_dwordx2\": %.3f, \"obs_4x_dword_dense__rew_dwordx2\": %.3f, \"obs_float4__rew_dwordx4_half_lanes\": %.3f,\n",
           t<0>(obs, rew, 3), t<1>(obs, rew, 3), t<2>(obs, rew, 3), t<3>(obs, rew, 3));
    printf(" \"obs_alone_float4\": %.3f, \"obs_alone_2x_dwordx2\": %.3f, \"obs_alone_4x_dword\": %.3f, \"rew_alone_dwordx2\": %.3f, \"rew_alone_dwordx4\": %.3f}\n",
           t<0>(obs, rew, 1), t<1>(obs, rew, 1), t<2>(obs, rew, 1), t<0>(obs, rew, 2), t<3>(obs, rew, 2));
    return 0;
}
