// Can a kernel start while the previous kernel of the SAME stream still runs?  hipExtLaunchKernelGGL takes hipExtAnyOrderLaunch
// ("the launch may start out of order with respect to earlier launches of its stream": the AQL packet without its barrier bit).
// A spins for `hold` microseconds and stamps its end; B stamps its entry.  Forms: (1) B a plain launch behind A, (2) B with
// hipExtAnyOrderLaunch, (3) B on a second stream that waits for an event recorded in FRONT of A (what two streams would cost:
// the event wait in front of B), (4) three-kernel chain A0 -> A -> B(any order) where B must not pass A0: B is given its own
// flag word to wait for.  Prints B.entry - A.entry and B.entry - A.end in microseconds (100 MHz wall clock).
// Build: hipcc --offload-arch=gfx950 -O2 -o any_order_probe any_order_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned long long now() { return wall_clock64(); }

__global__ void k_hold(unsigned long long* st, int ticks)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long t0 = now();
        st[0] = t0;
        while (now() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
        st[1] = now();
    }
}
__global__ void k_mark(unsigned long long* st)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) st[2] = now();
    // a launch of the filter's size: 744 workgroups x 512 threads do nothing else
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    unsigned long long *d, h[3];
    CK(hipMalloc((void**)&d, 64));
    hipStream_t s, s2;
    CK(hipStreamCreate(&s));
    CK(hipStreamCreate(&s2));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int hold = 3000;                 // 30 us
    for (int form = 1; form <= 3; ++form) {
        double d_entry = 0, d_end = 0;
        const int reps = 20;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipMemsetAsync(d, 0, 64, s));
            CK(hipStreamSynchronize(s));
            if (form == 3) { CK(hipEventRecord(ev, s)); }
            hipLaunchKernelGGL(k_hold, dim3(1), dim3(64), 0, s, d, hold);
            if (form == 1) hipLaunchKernelGGL(k_mark, dim3(744), dim3(512), 0, s, d);
            else if (form == 2) hipExtLaunchKernelGGL(k_mark, dim3(744), dim3(512), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d);
            else { CK(hipStreamWaitEvent(s2, ev, 0)); hipLaunchKernelGGL(k_mark, dim3(744), dim3(512), 0, s2, d); }
            CK(hipStreamSynchronize(s));
            CK(hipStreamSynchronize(s2));
            CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
            if (r >= 3) { d_entry += ((double)h[2] - (double)h[0]) * 0.01; d_end += ((double)h[2] - (double)h[1]) * 0.01; }
        }
        printf("form %d (%s): B.entry - A.entry = %.2f us, B.entry - A.end = %.2f us\n", form,
               form == 1 ? "plain launch behind A" : form == 2 ? "hipExtAnyOrderLaunch, same stream" : "second stream behind an event in front of A",
               d_entry / reps, d_end / reps);
    }
    return 0;
}
