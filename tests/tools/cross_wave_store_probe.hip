// Does a wave see another wave's GLOBAL store without waiting for its acknowledgement, when the two are ordered by a workgroup barrier
// only?  (Same CU, same vector L1: the memory model asks the storing wave for s_waitcnt vmcnt(0) in front of the barrier -- ~500
// cycles a round; the question is what the hardware does when it is left out.)  Two waves of one workgroup, `iters` rounds:
//     wave 0: [reads line A_prev so that it sits in the vector L1]  store(A[i]) = tag    s_barrier    ...            s_barrier
//     wave 1:                                                                           s_barrier    load(A[i])     s_barrier
// A[i] walks a scratch of `span` bytes in strides that hit the same line again after a while (an L1-resident copy of an older value
// is what a stale read would return).  Forms: 16-bit and 32-bit stores (the open list's keys and payloads), plain and sc1 loads.
// Prints the mismatches.  Build: hipcc --offload-arch=gfx950 -O2 -o cross_wave_store_probe cross_wave_store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int FORM>
__global__ __launch_bounds__(128) void k_probe(unsigned int* buf, unsigned int words, int iters, unsigned int stride, unsigned long long* bad, int wait_ack)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int pos = 0;
    unsigned long long wrong = 0;
    for (int i = 1; i <= iters; ++i) {
        pos = (pos + stride) % words;
        const unsigned int idx = (pos + 17u * (unsigned)lane) % words;       // every lane its own word, several lines per instruction
        if (wave == 1) {
            // keep an older copy of the line in the vector L1: read it BEFORE the other wave stores
            unsigned int old;
            if (FORM & 1) asm volatile("global_load_ushort %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"((const unsigned short*)buf + 2 * idx) : "memory");
            else asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(buf + idx) : "memory");
            (void)old;
        }
        __builtin_amdgcn_s_barrier();                                         // (both have the old value's line, or not)
        if (wave == 0) {
            const unsigned int tag = ((unsigned)i * 2654435761u) ^ idx;
            if (FORM & 1) asm volatile("global_store_short %0, %1, off" :: "v"((unsigned short*)buf + 2 * idx), "v"(tag & 0xffffu) : "memory");
            else asm volatile("global_store_dword %0, %1, off" :: "v"(buf + idx), "v"(tag) : "memory");
            if (wait_ack) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                         // the store has been ISSUED (acknowledged only with wait_ack)
        if (wave == 1) {
            const unsigned int tag = ((unsigned)i * 2654435761u) ^ idx;
            unsigned int got;
            if (FORM & 1) {
                if (FORM & 2) asm volatile("global_load_ushort %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"((const unsigned short*)buf + 2 * idx) : "memory");
                else asm volatile("global_load_ushort %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"((const unsigned short*)buf + 2 * idx) : "memory");
                wrong += (got != (tag & 0xffffu)) ? 1 : 0;
            } else {
                if (FORM & 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(buf + idx) : "memory");
                else asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(buf + idx) : "memory");
                wrong += (got != tag) ? 1 : 0;
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    if (wave == 1) atomicAdd(bad, wrong);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const unsigned int spans[] = {4096, 1u << 16, 1u << 20, 1u << 24};        // bytes: inside one L1, beyond it, beyond the L2's hot set
    const unsigned int strides[] = {1, 7, 33, 4099};
    unsigned int* buf; unsigned long long* bad;
    CK(hipMalloc((void**)&buf, 1u << 24));
    CK(hipMalloc((void**)&bad, 8));
    const int iters = 200000;
    for (int form = 0; form < 4; ++form)
        for (int wait = 0; wait < 2; ++wait) {
            unsigned long long total = 0, trials = 0;
            for (unsigned int span : spans)
                for (unsigned int stride : strides) {
                    CK(hipMemset(buf, 0, 1u << 24));
                    CK(hipMemset(bad, 0, 8));
                    // 64 workgroups at once: other CUs' traffic beside the probe's own
                    if (form == 0) hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(128), 0, 0, buf, span / 4, iters, stride, bad, wait);
                    if (form == 1) hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(128), 0, 0, buf, span / 4, iters, stride, bad, wait);
                    if (form == 2) hipLaunchKernelGGL(k_probe<2>, dim3(1), dim3(128), 0, 0, buf, span / 4, iters, stride, bad, wait);
                    if (form == 3) hipLaunchKernelGGL(k_probe<3>, dim3(1), dim3(128), 0, 0, buf, span / 4, iters, stride, bad, wait);
                    CK(hipDeviceSynchronize());
                    unsigned long long h = 0;
                    CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
                    total += h; trials += (unsigned long long)iters * 64ull;
                }
            printf("%s store, %s load, %s: %llu mismatches in %llu lane-reads\n", (form & 1) ? "16-bit" : "32-bit", (form & 2) ? "sc1" : "plain",
                   wait ? "store acknowledged before the barrier (s_waitcnt vmcnt(0))" : "store only ISSUED before the barrier", total, trials);
        }
    return 0;
}
