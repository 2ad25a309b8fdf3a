// tools/calib.hip — known-byte-count streaming kernels used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on
// gfx950 for THIS engine's access widths (MI355X_MICROARCH.md §HBM: FETCH_SIZE reads 1/2 of the bytes of a 16 B/lane
// stream; other widths and WRITE_SIZE are uncalibrated).  copy8: 8 B/lane loads+stores (the step kernel's fp64
// state pattern); copy16: 16 B/lane; copy4 / copy1: 4 B and 1 B per lane (elapsed[] and the flag arrays).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename T>
__global__ void copy_kernel(const T *__restrict__ x, T *__restrict__ y, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i];
}
__global__ void copy8(const double *x, double *y, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = x[i]; }
__global__ void copy16(const float4 *x, float4 *y, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = x[i]; }
__global__ void copy4(const float *x, float *y, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = x[i]; }
__global__ void copy1(const uint8_t *x, uint8_t *y, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = x[i]; }
int main() {
    const size_t bytes = 64ull << 20;  // 64 MiB read + 64 MiB written per launch, every kernel
    void *x, *y;
    hipMalloc(&x, bytes); hipMalloc(&y, bytes);
    hipMemset(x, 1, bytes); hipMemset(y, 0, bytes);
    for (int rep = 0; rep < 5; ++rep) {
        size_t n;
        n = bytes / 8;  copy8<<<(n + 255) / 256, 256>>>((const double *)x, (double *)y, n);
        n = bytes / 16; copy16<<<(n + 255) / 256, 256>>>((const float4 *)x, (float4 *)y, n);
        n = bytes / 4;  copy4<<<(n + 255) / 256, 256>>>((const float *)x, (float *)y, n);
        n = bytes;      copy1<<<(n + 255) / 256, 256>>>((const uint8_t *)x, (uint8_t *)y, n);
    }
    hipDeviceSynchronize();
    printf("calib done: 67108864 bytes read and written per launch\n");
    return 0;
}
