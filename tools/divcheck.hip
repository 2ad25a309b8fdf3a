// tools/divcheck.hip — is the shared-reciprocal division of mxv_device.hpp (refined_rcp + div_with_rcp) bit-identical to
// the compiler's IEEE fp64 `/` on the MI355X?  Random dividends x and divisors d drawn over the ranges Acrobot's RK4 stage
// produces (and well beyond): |x| in [2^-60, 2^40], d in [2^-20, 2^20], both signs for x.  Prints mismatches.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/_bin/divcheck tools/divcheck.hip && tools/_bin/divcheck
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ double refined_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __fma_rn(-d, r, 1.0);
    r = __fma_rn(r, e, r);
    e = __fma_rn(-d, r, 1.0);
    return __fma_rn(r, e, r);
}
__device__ __forceinline__ double div_with_rcp(double x, double d, double r) {
    const double q0 = x * r;
    return __fma_rn(__fma_rn(-d, q0, x), r, q0);
}
__device__ __forceinline__ uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double rnd(uint64_t &s, int emin, int emax, bool sign) {
    const uint64_t w = splitmix(s);
    const uint64_t mant = w & 0xFFFFFFFFFFFFFull;
    const int e = emin + (int)((w >> 52) % (uint64_t)(emax - emin + 1));
    uint64_t bits = ((uint64_t)(e + 1023) << 52) | mant;
    if (sign && (w >> 63)) bits |= 1ull << 63;
    return __longlong_as_double((long long)bits);
}
__global__ void check(uint64_t seed, int iters, unsigned long long *bad, double *ex) {
    uint64_t s = seed + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x1234567ull;
    for (int i = 0; i < iters; ++i) {
        const double d = rnd(s, -20, 20, false);
        const double r = refined_rcp(d);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double x = rnd(s, -60, 40, true);
            const double a = x / d, b = div_with_rcp(x, d, r);
            if (__double_as_longlong(a) != __double_as_longlong(b)) {
                if (atomicAdd(bad, 1ull) == 0) { ex[0] = x; ex[1] = d; ex[2] = a; ex[3] = b; }
            }
        }
    }
}
int main() {
    unsigned long long *bad, hb = 0;
    double *ex, hex[4] = {0, 0, 0, 0};
    hipMalloc(&bad, 8); hipMalloc(&ex, 32);
    hipMemset(bad, 0, 8);
    const int blocks = 4096, threads = 256, iters = 1300;
    hipLaunchKernelGGL(check, dim3(blocks), dim3(threads), 0, 0, 0xC0FFEEull, iters, bad, ex);
    hipDeviceSynchronize();
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(hex, ex, 32, hipMemcpyDeviceToHost);
    printf("pairs checked: %.3e  mismatches vs `/`: %llu\n", 3.0 * blocks * threads * iters, hb);
    if (hb) printf("first: x=%a d=%a  x/d=%a  shared=%a\n", hex[0], hex[1], hex[2], hex[3]);
    return hb != 0;
}
