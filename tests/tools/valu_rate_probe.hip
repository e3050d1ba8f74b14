// valu_rate_probe.hip -- issue cost of the vector instructions k_mcl_main's ray loop is made of, relative to v_fma_f32.
// Every kernel runs ITER x 8 independent copies of one instruction per lane, `wps` (argv[1], default 4) waves per SIMD on every CU; the figure
// printed is SIMD cycles per wave-instruction (time x clock x SIMDs / wave-instructions), 4.0 = full rate.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITER 2048
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define KERNEL32(NAME, ASM)                                                                      \
__global__ __launch_bounds__(256) void NAME(int* out, int seed)                                  \
{                                                                                                \
    int r0 = seed + threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    int a = seed | 0x3f800001, b = 0x3f800003;                                                   \
    for (int i = 0; i < ITER; ++i) {                                                             \
        asm volatile(ASM "\n" : "+v"(r0) : "v"(a), "v"(b)); asm volatile(ASM "\n" : "+v"(r1) : "v"(a), "v"(b));     \
        asm volatile(ASM "\n" : "+v"(r2) : "v"(a), "v"(b)); asm volatile(ASM "\n" : "+v"(r3) : "v"(a), "v"(b));     \
        asm volatile(ASM "\n" : "+v"(r4) : "v"(a), "v"(b)); asm volatile(ASM "\n" : "+v"(r5) : "v"(a), "v"(b));     \
        asm volatile(ASM "\n" : "+v"(r6) : "v"(a), "v"(b)); asm volatile(ASM "\n" : "+v"(r7) : "v"(a), "v"(b));     \
    }                                                                                            \
    if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345) out[0] = r0;                        \
}

// 64-bit destination / sources
#define KERNEL64(NAME, ASM)                                                                      \
__global__ __launch_bounds__(256) void NAME(int* out, int seed)                                  \
{                                                                                                \
    double r0 = seed + threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    double a = 1.0000001 + seed, b = 0.9999999;                                                  \
    for (int i = 0; i < ITER; ++i) {                                                             \
        asm volatile(ASM "\n" : "+v"(r0) : "v"(a), "v"(b)); asm volatile(ASM "\n" : "+v"(r1) : "v"(a), "v"(b));     \
        asm volatile(ASM "\n" : "+v"(r2) : "v"(a), "v"(b)); asm volatile(ASM "\n" : "+v"(r3) : "v"(a), "v"(b));     \
        asm volatile(ASM "\n" : "+v"(r4) : "v"(a), "v"(b)); asm volatile(ASM "\n" : "+v"(r5) : "v"(a), "v"(b));     \
        asm volatile(ASM "\n" : "+v"(r6) : "v"(a), "v"(b)); asm volatile(ASM "\n" : "+v"(r7) : "v"(a), "v"(b));     \
    }                                                                                            \
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 0.12345) out[0] = 1;                            \
}

// mixed widths: D = 32-bit dest from 64-bit source (or the reverse); the chain is broken on purpose (sources constant)
#define KERNEL_32_FROM_64(NAME, ASM)                                                             \
__global__ __launch_bounds__(256) void NAME(int* out, int seed)                                  \
{                                                                                                \
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;                           \
    double a = 1.0000001 + seed + threadIdx.x;                                                   \
    for (int i = 0; i < ITER; ++i) {                                                             \
        asm volatile(ASM "\n" : "=v"(r0) : "v"(a)); asm volatile(ASM "\n" : "=v"(r1) : "v"(a));  \
        asm volatile(ASM "\n" : "=v"(r2) : "v"(a)); asm volatile(ASM "\n" : "=v"(r3) : "v"(a));  \
        asm volatile(ASM "\n" : "=v"(r4) : "v"(a)); asm volatile(ASM "\n" : "=v"(r5) : "v"(a));  \
        asm volatile(ASM "\n" : "=v"(r6) : "v"(a)); asm volatile(ASM "\n" : "=v"(r7) : "v"(a));  \
    }                                                                                            \
    if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345) out[0] = r0;                        \
}
#define KERNEL_64_FROM_32(NAME, ASM)                                                             \
__global__ __launch_bounds__(256) void NAME(int* out, int seed)                                  \
{                                                                                                \
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;                        \
    int a = 0x3f800001 + seed + threadIdx.x;                                                     \
    for (int i = 0; i < ITER; ++i) {                                                             \
        asm volatile(ASM "\n" : "=v"(r0) : "v"(a)); asm volatile(ASM "\n" : "=v"(r1) : "v"(a));  \
        asm volatile(ASM "\n" : "=v"(r2) : "v"(a)); asm volatile(ASM "\n" : "=v"(r3) : "v"(a));  \
        asm volatile(ASM "\n" : "=v"(r4) : "v"(a)); asm volatile(ASM "\n" : "=v"(r5) : "v"(a));  \
        asm volatile(ASM "\n" : "=v"(r6) : "v"(a)); asm volatile(ASM "\n" : "=v"(r7) : "v"(a));  \
    }                                                                                            \
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 0.12345) out[0] = 1;                            \
}

KERNEL32(k_fma_f32, "v_fma_f32 %0, %1, %2, %0")
KERNEL32(k_add_u32, "v_add_u32 %0, %1, %0")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0")
KERNEL32(k_mul_i24, "v_mul_i32_i24 %0, %1, %0")
KERNEL32(k_med3_i32, "v_med3_i32 %0, %0, %1, %2")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp, "v_cmp_lt_i32 vcc, %1, %0")
KERNEL32(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %1")
KERNEL32(k_cvt_pk_i16_i32, "v_cvt_pk_i16_i32 %0, %1, %0")
KERNEL32(k_pk_sub_i16, "v_pk_sub_i16 %0, %1, %0")
KERNEL32(k_pk_max_i16, "v_pk_max_i16 %0, %1, %0")
KERNEL32(k_pk_ashr_i16, "v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]")
KERNEL32(k_dot2c_i16, "v_dot2c_i32_i16 %0, %1, %2")
KERNEL32(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8")
KERNEL32(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL32(k_cndmask_cmp, "v_cmp_lt_i32 vcc, %1, %0\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(k_cndmask_const, "v_cndmask_b32_e64 %0, -1, 1, vcc")
KERNEL32(k_bfi, "v_bfi_b32 %0, %1, %2, %0")
KERNEL32(k_ashr, "v_ashrrev_i32 %0, 31, %0")
KERNEL32(k_max_i32, "v_max_i32 %0, %1, %0")
KERNEL32(k_add3, "v_add3_u32 %0, %1, %2, %0")
KERNEL32(k_nop, "s_nop 0")
KERNEL32(k_sin_f32, "v_sin_f32 %0, %1")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %1, %2, %0")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %1, %0")
KERNEL64(k_add_f64, "v_add_f64 %0, %1, %0")
KERNEL64(k_trunc_f64, "v_trunc_f64 %0, %1")
KERNEL64(k_floor_f64, "v_floor_f64 %0, %1")
KERNEL64(k_pk_mul_f32, "v_pk_mul_f32 %0, %1, %0")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %1, %0")
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL_32_FROM_64(k_cvt_f32_f64, "v_cvt_f32_f64 %0, %1")
KERNEL_32_FROM_64(k_cvt_i32_f64, "v_cvt_i32_f64 %0, %1")
KERNEL_64_FROM_32(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1")
KERNEL_64_FROM_32(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %1")

struct entry { const char* name; void (*fn)(int*, int); };

int main(int argc, char** argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 4;    // waves per SIMD: workgroups of 4 waves, `wps` of them per CU
    int dev = 0; CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    int* out; CHECK(hipMalloc((void**)&out, 64));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = cus * wps;                    // wps workgroups x 4 waves per CU = wps waves per SIMD
    std::vector<entry> es = {
        {"v_fma_f32 (cold)", k_fma_f32}, {"v_fma_f32", k_fma_f32}, {"v_add_u32", k_add_u32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_i32_i24", k_mul_i24},
        {"v_med3_i32", k_med3_i32}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_i32", k_cmp}, {"v_cvt_i32_f32", k_cvt_i32_f32},
        {"v_cvt_pk_i16_i32", k_cvt_pk_i16_i32}, {"v_pk_sub_i16", k_pk_sub_i16}, {"v_pk_max_i16", k_pk_max_i16},
        {"v_pk_ashrrev_i16", k_pk_ashr_i16}, {"v_dot2c_i32_i16", k_dot2c_i16}, {"v_bitop3_b32", k_bitop3}, {"v_cndmask_e64 sgpr", k_cndmask_sgpr}, {"cmp+nop1+cndmask", k_cndmask_cmp}, {"v_cndmask -1,1", k_cndmask_const}, {"v_bfi_b32", k_bfi}, {"v_ashrrev_i32", k_ashr}, {"v_max_i32", k_max_i32}, {"v_add3_u32", k_add3}, {"s_nop 0", k_nop}, {"v_sin_f32", k_sin_f32},
        {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_trunc_f64", k_trunc_f64},
        {"v_floor_f64", k_floor_f64}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f64_i32", k_cvt_f64_i32}};
    printf("device: %s, %d CUs, clockRate %.2f GHz, %d waves per SIMD\n", prop.name, cus, ghz, wps);
    for (auto& e : es) {
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 0);          // warm
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, r);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_instr_per_simd = 5.0 * (double)wps * ITER * 8.0;      // launches x waves per SIMD x iterations x copies
        const double cyc = (ms * 1e-3) * ghz * 1e9 / wave_instr_per_simd;
        printf("%-18s %6.2f cycles per wave-instruction (%.3f ms)\n", e.name, cyc, ms / 5);
    }
    return 0;
}
