#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); return 1; } } while (0)
__global__ void wr(int *p, int v, size_t n) { size_t i = blockIdx.x * 256ull + threadIdx.x; if (i < n) p[i * 1024] = v; }   // one int per 4 KiB page
__global__ void rd(const int *p, int *out, size_t n) { size_t i = blockIdx.x * 256ull + threadIdx.x; if (i < n && p[i * 1024] != out[1]) atomicAdd(out, 1); }
int main() {
    const size_t MiB = 1 << 20, chunk = 256 * MiB, pages = chunk / 4096;
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    const int P = 12;
    std::vector<hipMemGenericAllocationHandle_t> h(P);
    for (auto &x : h) CK(hipMemCreate(&x, chunk, &prop, 0));
    int *out; CK(hipMalloc(&out, 8));
    auto check = [&](char *va, int expect) { int z[2] = {0, expect}; hipMemcpy(out, z, 8, hipMemcpyHostToDevice); hipLaunchKernelGGL(rd, dim3(pages / 256), dim3(256), 0, 0, (const int *)va, out, pages); hipMemcpy(z, out, 8, hipMemcpyDeviceToHost); return z[0]; };
    // T1: shared scratch VA, marker written by a kernel through it
    char *scratch; CK(hipMemAddressReserve((void **)&scratch, chunk, 0, nullptr, 0));
    for (int i = 0; i < P; ++i) {
        CK(hipMemMap(scratch, chunk, 0, h[i], 0)); CK(hipMemSetAccess(scratch, chunk, &acc, 1));
        hipLaunchKernelGGL(wr, dim3(pages / 256), dim3(256), 0, 0, (int *)scratch, 100 + i, pages);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(scratch, chunk));
    }
    // second mapping of every handle at a VA of its own
    std::vector<char *> va(P);
    for (int i = 0; i < P; ++i) { CK(hipMemAddressReserve((void **)&va[i], chunk, 0, nullptr, 0)); CK(hipMemMap(va[i], chunk, 0, h[i], 0)); CK(hipMemSetAccess(va[i], chunk, &acc, 1)); }
    printf("T1 shared scratch VA then own VA: pages NOT holding marker 100+i: ");
    for (int i = 0; i < P; ++i) printf("%d ", check(va[i], 100 + i));
    printf("\n");
    // T1b: shared scratch again, now READ through the scratch mapping what the own-VA mapping holds (write 200+i via own VA first)
    for (int i = 0; i < P; ++i) hipLaunchKernelGGL(wr, dim3(pages / 256), dim3(256), 0, 0, (int *)va[i], 200 + i, pages);
    CK(hipDeviceSynchronize());
    printf("T1b own VA written, read through remapped shared scratch: mismatching pages: ");
    for (int i = 0; i < P; ++i) {
        CK(hipMemMap(scratch, chunk, 0, h[i], 0)); CK(hipMemSetAccess(scratch, chunk, &acc, 1));
        printf("%d ", check(scratch, 200 + i));
        CK(hipMemUnmap(scratch, chunk));
    }
    printf("\n");
    // T2: unmap own VAs, map in a contiguous reservation (third mapping), verify
    for (int i = 0; i < P; ++i) CK(hipMemUnmap(va[i], chunk));
    char *big; CK(hipMemAddressReserve((void **)&big, P * chunk, 0, nullptr, 0));
    for (int i = 0; i < P; ++i) CK(hipMemMap(big + i * chunk, chunk, 0, h[P - 1 - i], 0));
    CK(hipMemSetAccess(big, P * chunk, &acc, 1));
    printf("T2 third mapping, reversed order in one reservation: mismatching pages: ");
    for (int i = 0; i < P; ++i) printf("%d ", check(big + i * chunk, 200 + (P - 1 - i)));
    printf("\n");
    // T3: unmap the whole reservation with ONE call (as vmm_probe.hip did), remap in forward order, verify
    CK(hipMemUnmap(big, P * chunk));
    for (int i = 0; i < P; ++i) CK(hipMemMap(big + i * chunk, chunk, 0, h[i], 0));
    CK(hipMemSetAccess(big, P * chunk, &acc, 1));
    printf("T3 one-call unmap of 12 mappings, remap forward: mismatching pages: ");
    for (int i = 0; i < P; ++i) printf("%d ", check(big + i * chunk, 200 + i));
    printf("\n");
    return 0;
}
