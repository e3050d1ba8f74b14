// Probe: does hipStreamWaitValue64 on plain device memory order a consumer stream after a producer kernel's flag store,
// and what does it cost compared with hipEventRecord + hipStreamWaitEvent?  (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void producer(unsigned long long* flag, unsigned long long v, int* data, int spin)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)spin * 100) {}      // ~spin us at 100 MHz
    *data = (int)v;
    __threadfence_system();
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void consumer(const int* data, int* out, int i) { out[i] = *data; }
__global__ void filler(int* p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
int main()
{
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("CanUseStreamWaitValue = %d\n", can);
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned long long* flag; int *data, *out, *h_out, *fill;
    CK(hipMalloc(&flag, 8)); CK(hipMalloc(&data, 4)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&fill, 4));
    CK(hipMemset(flag, 0, 8)); CK(hipMemset(data, 0, 4)); CK(hipMemset(out, 0, 4096)); CK(hipMemset(fill, 0, 4));
    CK(hipHostMalloc(&h_out, 4096));
    const int N = 200;
    // ---- correctness: consumer i must see data == i + 1
    for (int i = 0; i < N; ++i) {
        hipError_t e = hipStreamWaitValue64(b, flag, (uint64_t)(i + 1), hipStreamWaitValueGte, 0xffffffffffffffffull);
        if (e != hipSuccess) { printf("hipStreamWaitValue64 -> %s\n", hipGetErrorString(e)); return 1; }
        hipLaunchKernelGGL(consumer, dim3(1), dim3(1), 0, b, data, out, i);
        hipLaunchKernelGGL(producer, dim3(1), dim3(1), 0, a, flag, (unsigned long long)(i + 1), data, 20);
    }
    CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
    CK(hipMemcpy(h_out, out, N * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < N; ++i) bad += h_out[i] < i + 1;
    printf("wait-value ordering violations: %d of %d\n", bad, N);
    // ---- cost on the producer stream: back-to-back tiny kernels with (a) nothing, (b) event record, between them
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 2000; ++i) {
            hipLaunchKernelGGL(filler, dim3(1), dim3(64), 0, a, fill);
            if (mode == 1) { CK(hipEventRecord(ev, a)); }
            if (mode == 2) { CK(hipStreamWriteValue64(a, flag, (uint64_t)(1000000 + i), 0)); }
        }
        CK(hipStreamSynchronize(a));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
        printf("mode %d (%s): %.2f us per kernel\n", mode, mode == 0 ? "kernels only" : mode == 1 ? "kernel + hipEventRecord" : "kernel + hipStreamWriteValue64", us);
    }
    return 0;
}
