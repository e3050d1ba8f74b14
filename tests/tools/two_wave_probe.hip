// two_wave_probe.hip -- what TWO wavefronts of one workgroup, alone on a CU, can do for a chain that is bound by instruction
// issue (k_astar2's loops: one instruction per ~4.2 cycles for a lone wave, tests/tools/lone_wave_probe.hip).  Do waves on
// different SIMDs issue side by side?  What does it cost them to meet (s_barrier) and to hand a value over through LDS?
//   hipcc --offload-arch=gfx950 -O2 -o two_wave_probe two_wave_probe.hip && ./two_wave_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define R4(X) X X X X
#define R16(X) R4(X) R4(X) R4(X) R4(X)
#define R64(X) R16(X) R16(X) R16(X) R16(X)
#define R256(X) R64(X) R64(X) R64(X) R64(X)

#define NOW(var) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")

__shared__ volatile unsigned int s_word[64];

// out[slot * 4 + wave] = cycles
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_probe(unsigned long long* out, int seed)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned v0 = seed + lane, v1 = 3, s0 = seed;
    unsigned long long t0, t1;
    if (threadIdx.x < 64) s_word[threadIdx.x] = 0;
    __syncthreads();
    // where the wave runs: HW_ID bits 4..5 = SIMD
    if (lane == 0) out[60 + wave] = __builtin_amdgcn_s_getreg(63492);
    // 0: 256 dependent VALU per wave, every wave at once (a lone wave: ~4.2 cycles each)
    __syncthreads();
    NOW(t0); asm volatile(R256("v_add_u32 %0, %0, %1\n") : "+v"(v0) : "v"(v1)); NOW(t1);
    if (lane == 0) out[0 * 4 + wave] = t1 - t0;
    // 1: 256 SALU per wave at once
    __syncthreads();
    NOW(t0); asm volatile(R256("s_add_u32 %0, %0, 1\n") : "+s"(s0) :: "scc"); NOW(t1);
    if (lane == 0) out[1 * 4 + wave] = t1 - t0;
    // 2: 64 barriers in a row (every wave arrives at once: the cost of the meeting itself)
    __syncthreads();
    NOW(t0); asm volatile(R64("s_barrier\n") ::: "memory"); NOW(t1);
    if (lane == 0) out[2 * 4 + wave] = t1 - t0;
    // 3: 16 x (wave 0: 16 VALU then barrier; the others: barrier at once) -- what the early wave waits, what the late one pays
    __syncthreads();
    NOW(t0);
    if (wave == 0) { asm volatile(R16(R16("v_add_u32 %0, %0, %1\n") "s_barrier\n") : "+v"(v0) : "v"(v1) : "memory"); }
    else { asm volatile(R16("s_barrier\n") ::: "memory"); }
    NOW(t1);
    if (lane == 0) out[3 * 4 + wave] = t1 - t0;
    // 4: 16 hand-overs through LDS behind a barrier: wave 0 writes, barrier, wave 1 reads and adds (the value is needed: a dependent read)
    __syncthreads();
    NOW(t0);
    {
        unsigned acc = 0;
        for (int i = 0; i < 16; ++i) {
            if (wave == 0) s_word[lane] = v0 + i;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (wave == 1) acc += s_word[lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        v0 += acc;
    }
    NOW(t1);
    if (lane == 0) out[4 * 4 + wave] = t1 - t0;
    // 5: 16 ping-pongs through an LDS flag, no barrier: wave 0 stores i, wave 1 polls for it and stores i to a second word, wave 0 polls
    __syncthreads();
    NOW(t0);
    if (WAVES >= 2 && wave < 2) {
        for (unsigned i = 1; i <= 16; ++i) {
            if (wave == 0) { if (lane == 0) s_word[0] = i; while (s_word[1] != i) { } }
            else { while (s_word[0] != i) { } if (lane == 0) s_word[1] = i; }
        }
    }
    NOW(t1);
    if (lane == 0) out[5 * 4 + wave] = t1 - t0;
    if (v0 + s0 == 0x12345678u) out[59] = 1;
}

int main()
{
    unsigned long long* d_out;
    CHECK(hipMalloc((void**)&d_out, 64 * 8));
    unsigned long long h[64];
    const char* names[] = {"256 dependent v_add_u32 in every wave at once", "256 s_add_u32 in every wave at once", "64 s_barrier in a row",
                           "16 x (wave 0: 16 v_add, then s_barrier; others: s_barrier)", "16 x (wave 0 writes LDS, s_barrier, wave 1 reads it)",
                           "16 ping-pongs through two LDS words (polling, no barrier)"};
    const double per[] = {256, 256, 64, 16, 16, 16};
    for (int waves = 1; waves <= 4; waves *= 2) {
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(d_out, 0, 64 * 8));
            if (waves == 1) hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(64), 0, 0, d_out, rep);
            else if (waves == 2) hipLaunchKernelGGL(k_probe<2>, dim3(1), dim3(128), 0, 0, d_out, rep);
            else hipLaunchKernelGGL(k_probe<4>, dim3(1), dim3(256), 0, 0, d_out, rep);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
        }
        printf("---- %d wave(s) in the workgroup; SIMD of each:", waves);
        for (int w = 0; w < waves; ++w) printf(" %llu", (h[60 + w] >> 4) & 3);
        printf("\n");
        for (int i = 0; i < 6; ++i) {
            if (waves == 1 && i >= 4) continue;
            printf("%-66s", names[i]);
            for (int w = 0; w < waves; ++w) printf("  wave %d: %5llu (%.1f each)", w, h[i * 4 + w], (double)h[i * 4 + w] / per[i]);
            printf("\n");
        }
    }
    return 0;
}
