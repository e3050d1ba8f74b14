// cu_mask_probe.hip -- which CUs does a stream created with hipExtStreamCreateWithCUMask use?
// Build: hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.hip ; run: ./cu_mask_probe <bits set> [first bit]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <map>

__global__ void k_where(unsigned int* out, int spin)
{
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xffffu) | ((xcc & 0xfu) << 16);
}

int main(int argc, char** argv)
{
    int bits = argc > 1 ? atoi(argv[1]) : 8, first = argc > 2 ? atoi(argv[2]) : 0, stride = argc > 3 ? atoi(argv[3]) : 1;
    unsigned int mask[8] = {0};
    for (int i = 0; i < bits; ++i) { int b = first + i * stride; mask[b >> 5] |= 1u << (b & 31); }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    printf("create: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    const int n = 4096;
    unsigned int* d; hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k_where, dim3(n), dim3(256), 0, s, d, 2000);
    e = hipStreamSynchronize(s);
    printf("run: %s\n", hipGetErrorString(e));
    unsigned int* h = (unsigned int*)malloc(n * 4);
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    std::map<unsigned int, int> cus;
    for (int i = 0; i < n; ++i) { unsigned int v = h[i]; unsigned int key = ((v >> 16) << 8) | (((v >> 13) & 7) << 4) | ((v >> 8) & 15); cus[key]++; }
    printf("%d bits from %d stride %d -> %zu distinct CUs:", bits, first, stride, cus.size());
    for (auto& kv : cus) printf(" x%u.se%u.cu%u(%d)", kv.first >> 8, (kv.first >> 4) & 7, kv.first & 15, kv.second);
    printf("\n");
    return 0;
}
