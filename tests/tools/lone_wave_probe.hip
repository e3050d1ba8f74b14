// lone_wave_probe.hip -- what ONE wavefront alone on a CU pays per instruction of the kinds k_astar2's heap operations are made
// of (a dependent chain each, REP copies in straight-line code, timed by s_memtime).  The search is one such wave: its speed is
// the length of its dependent chain in these units, not a bandwidth.
//   hipcc --offload-arch=gfx950 -O2 -o lone_wave_probe lone_wave_probe.hip && ./lone_wave_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define R4(X) X X X X
#define R16(X) R4(X) R4(X) R4(X) R4(X)
#define R64(X) R16(X) R16(X) R16(X) R16(X)

#define TIC() unsigned long long t0_, t1_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_) :: "memory")
#define TOC(slot) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory"); if (threadIdx.x == 0) out[slot] = t1_ - t0_

extern __shared__ unsigned int lds[];

__global__ __launch_bounds__(64) void k_probe(unsigned long long* out, unsigned int* gbuf, int seed)
{
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = ((i * 37 + 11) & 4095) * 4;    // a permutation of byte offsets: pointer chase
    __syncthreads();
    unsigned v0 = seed + lane, v1 = 3, v2 = seed, v3 = 5, v4 = 7, v5 = 9, v6 = 1, v7 = 2;
    unsigned s0 = seed, s1 = 1;
    unsigned long long m = 0xFFull | (unsigned long long)seed;
    { TIC(); TOC(0); }                                                                            // clock overhead
    { TIC(); asm volatile(R64("v_add_u32 %0, %0, %1\n") : "+v"(v0) : "v"(v1)); TOC(1); }       // dependent VALU
    { TIC(); asm volatile(R16("v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4\n") : "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5) : "v"(v1)); TOC(2); }   // 4 independent chains
    { TIC(); asm volatile(R64("s_add_u32 %0, %0, %1\n") : "+s"(s0) : "s"(s1) : "scc"); TOC(3); }   // dependent SALU
    { TIC(); asm volatile(R16("v_cmp_lt_u32 vcc, %0, %1\ns_ff1_i32_b64 %2, vcc\nv_add_u32 %0, %2, %0\n") : "+v"(v0), "+v"(v6), "+s"(s0) :: "vcc"); TOC(4); }   // VALU -> SALU -> VALU
    { TIC(); asm volatile(R16("v_readlane_b32 %1, %0, 3\nv_add_u32 %0, %1, %0\n") : "+v"(v0), "+s"(s0)); TOC(5); }   // readlane -> VALU
    { unsigned a = (lane * 4) & 16383; TIC(); asm volatile(R16("ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\n") : "+v"(a)); TOC(6); v7 += a; }   // dependent LDS reads
    { unsigned a = (lane * 4) & 16383; TIC(); asm volatile(R16("ds_read_u16 %0, %0\ns_waitcnt lgkmcnt(0)\nv_and_b32 %0, 0x3ffc, %0\n") : "+v"(a)); TOC(7); v7 += a; }
    { unsigned a = 16384 + lane * 4; TIC(); asm volatile(R16("s_mov_b64 exec, %1\nds_write_b16 %0, %2\ns_mov_b64 exec, -1\n") :: "v"(a), "s"(m), "v"(v1) : "memory"); TOC(8); }   // masked stores, fire and forget
    { TIC(); asm volatile(R16("s_flbit_i32_b64 %1, %2\nv_readlane_b32 %1, %0, %1\ns_add_u32 %1, %1, 1\n") : "+v"(v0), "+s"(s0) : "s"(m) : "scc"); TOC(9); }   // flbit -> readlane(lane select) -> SALU
    { TIC(); asm volatile(R16("s_branch 1f\n1:\n")); TOC(10); }                                  // taken branches (to the next instruction)
    { TIC(); asm volatile(R64("s_nop 0\n")); TOC(11); }
    { unsigned a = 20000 + lane * 4, b = 0; TIC(); asm volatile(R16("ds_write_b32 %0, %1\nds_read_b32 %1, %0\ns_waitcnt lgkmcnt(0)\n") : "+v"(a), "+v"(b) :: "memory"); TOC(12); v7 += b; }   // store then load, same address
    { unsigned long long p = (unsigned long long)gbuf; unsigned off = (lane * 4) & 4095; TIC();
      asm volatile(R16("global_load_dword %0, %0, %1\ns_waitcnt vmcnt(0)\n") : "+v"(off) : "s"(p) : "memory"); TOC(13); v7 += off; }   // dependent global loads (L2 hits after the first touch)
    { unsigned long long p = (unsigned long long)gbuf; unsigned off = (lane * 4) & 4095; TIC();
      asm volatile(R16("global_load_dword %0, %0, %1\ns_waitcnt vmcnt(0)\n") : "+v"(off) : "s"(p) : "memory"); TOC(14); v7 += off; }   // again: warm
    { TIC(); asm volatile(R16("v_cmp_lt_u32 vcc, %0, %1\ns_not_b64 vcc, vcc\ns_ff1_i32_b64 %2, vcc\ns_bfm_b64 %3, %2, 0\ns_mov_b64 exec, %3\nds_write_b16 %4, %1\ns_mov_b64 exec, -1\nv_add_u32 %0, %2, %0\n")
                          : "+v"(v0), "+v"(v6), "+s"(s0), "+s"(m) : "v"(16384 + lane * 4) : "vcc", "scc", "memory"); TOC(15); }   // the tail of a push: compare .. masked store
    { TIC(); asm volatile(R16("v_lshl_add_u32 %0, %1, %2, %0\nv_and_b32 %0, 0xffff, %0\nv_min_u32 %0, %0, %3\n") : "+v"(v0) : "s"(s1), "v"(v1), "v"(v3)); TOC(16); }   // 3 dependent VALU with an SGPR operand
    { TIC(); asm volatile(R16("s_lshl_b32 %0, %0, 1\ns_and_b32 %0, %0, 0xffff\ns_or_b32 %0, %0, 1\ns_cmp_lt_u32 %0, 77\ns_cselect_b32 %0, %0, 5\n") : "+s"(s0) :: "scc"); TOC(17); }   // 5 dependent SALU incl. compare/select
    { TIC(); asm volatile(R16("s_branch 1f\n" R16("s_nop 0\n") "1:\n")); TOC(18); }                 // taken branches over 64 bytes
    { TIC(); asm volatile(R16("s_branch 1f\n" R64("s_nop 0\n") R64("s_nop 0\n") "1:\n")); TOC(19); }   // ... over 512 bytes
    { TIC(); asm volatile(R16("s_cmp_eq_u32 %0, 12345\ns_cbranch_scc1 1f\n" "s_nop 0\n1:\n") :: "s"(s1) : "scc"); TOC(20); }   // conditional branches not taken
    { TIC(); asm volatile(R16("s_cmp_lg_u32 %0, 12345\ns_cbranch_scc1 1f\n" R16("s_nop 0\n") "1:\n") :: "s"(s1) : "scc"); TOC(21); }   // conditional branches taken over 64 bytes
    { TIC(); asm volatile(R16("s_and_saveexec_b64 %0, vcc\ns_cbranch_execz 1f\ns_nop 0\n1:\ns_or_b64 exec, exec, %0\n") : "+s"(m) :: "memory"); TOC(22); }   // the compiler's divergent-if frame, body executed or skipped
    {   // does a vector-memory instruction issued with EXEC = 0 count in vmcnt?  A cold load, then a load under an empty mask, then
        // s_waitcnt vmcnt(1): if the masked one counts, the wait lasts as long as the cold load
        unsigned long long p = (unsigned long long)(gbuf + 1024 + 64 * seed); unsigned off = (lane * 4) & 255, r0 = 0, r1 = 0;
        TIC();
        asm volatile("global_load_dword %0, %2, %3 sc1 sc0\n\ts_mov_b64 exec, 0\n\tglobal_load_dword %1, %2, %3 offset:2048\n\ts_mov_b64 exec, -1\n\ts_waitcnt vmcnt(1)\n"
                     : "+v"(r0), "+v"(r1) : "v"(off), "s"(p) : "memory");
        TOC(23);
        v7 += r0 + r1;
    }
    {   unsigned long long p = (unsigned long long)(gbuf + 4096 + 64 * seed); unsigned off = (lane * 4) & 255, r0 = 0;
        TIC();
        asm volatile("global_load_dword %0, %1, %2 sc1 sc0\n\ts_waitcnt vmcnt(0)\n" : "+v"(r0) : "v"(off), "s"(p) : "memory");
        TOC(24);
        v7 += r0;
    }
    if (v0 + v2 + v3 + v4 + v5 + v6 + v7 + s0 + (unsigned)m == 0x12345678u) out[63] = 1;
}

int main()
{
    unsigned long long* d_out; unsigned int* d_g;
    CHECK(hipMalloc((void**)&d_out, 64 * 8)); CHECK(hipMalloc((void**)&d_g, 65536));
    std::vector<unsigned int> g(1024);
    for (int i = 0; i < 1024; ++i) g[i] = ((i * 37 + 11) & 1023) * 4;
    CHECK(hipMemcpy(d_g, g.data(), 4096, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_out, 0, 64 * 8));
    unsigned long long h[64];
    const char* names[] = {"clock overhead", "64 dependent v_add_u32", "64 v_add_u32 in 4 independent chains", "64 dependent s_add_u32",
                           "16 x (v_cmp -> s_ff1 -> v_add)", "16 x (v_readlane -> v_add)", "16 dependent ds_read_b32", "16 dependent ds_read_u16 + v_and",
                           "16 masked ds_write_b16 (exec set / restored)", "16 x (s_flbit -> v_readlane -> s_add)", "16 taken s_branch", "64 s_nop 0",
                           "16 x (ds_write, ds_read same address)", "16 dependent global loads (first touch)", "16 dependent global loads (warm)",
                           "16 x push tail (cmp, not, ff1, bfm, exec, ds_write, exec, v_add)", "16 x 3 dependent VALU", "16 x 5 dependent SALU",
                           "16 taken s_branch over 64 bytes", "16 taken s_branch over 512 bytes", "16 x (s_cmp, s_cbranch not taken, s_nop)", "16 x (s_cmp, s_cbranch taken over 64 bytes)", "16 x (saveexec, cbranch_execz, nop, or exec)",
                           "cold load + load under EXEC = 0 + vmcnt(1)", "cold load + vmcnt(0)"};
    const int per[] = {1, 64, 64, 64, 48, 32, 16, 32, 48, 48, 16, 64, 32, 16, 16, 128, 48, 80, 16, 16, 48, 32, 64, 1, 1};
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 65536, 0, d_out, d_g, rep);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < 25; ++i)
        printf("%-68s %6llu cycles  (%.1f per instruction after the clock's %llu)\n", names[i], h[i], i ? (double)((long long)h[i] - (long long)h[0]) / per[i] : 0.0, h[0]);
    return 0;
}
