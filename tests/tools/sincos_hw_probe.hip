// sincos_hw_probe.hip -- how far is the hardware sine / cosine of an unwrapped ray angle from what the reference computes?
// Reference (sensor_model.cpp:34-37 through moving_laser_scan.cpp:33): theta' = wrap_to_pi(d), d = pose.theta - ray theta
// (a float), then sinf(theta'), cosf(theta').  Fast form: v_sin_f32 / v_cos_f32 of d * (1 / 2pi) (revolutions), no wrap.
// EXHAUSTIVE over every float d in [-3 pi - 0.01, pi + 0.01] (a wrapped pose angle less a scan angle in [0, 2 pi)): prints the
// largest |fast - reference| for sine and cosine, which is the trig part of the guard band in k_mcl_main's fast path.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -I botlab_amd/csrc -o sincos_hw_probe tests/tools/sincos_hw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "bl_math.h"

__device__ __forceinline__ float hw_sin_rev(float r) { float o; asm("v_sin_f32 %0, %1" : "=v"(o) : "v"(r)); return o; }
__device__ __forceinline__ float hw_cos_rev(float r) { float o; asm("v_cos_f32 %0, %1" : "=v"(o) : "v"(r)); return o; }

__global__ void k_probe(uint32_t lo_bits, uint32_t count, float* out_max, unsigned long long* out_arg)
{
    const float INV2PI = 0.15915494309189535f;
    float ms = 0.f, mc = 0.f;
    uint32_t as = 0, ac = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = lo_bits + (uint32_t)i;
        const float d = __uint_as_float(bits);
        float sn, cs;
        bl_sincosf_cells(bl_wrap_to_pi(d), &sn, &cs);
        const float r = d * INV2PI;
        const float es = fabsf(hw_sin_rev(r) - sn), ec = fabsf(hw_cos_rev(r) - cs);
        if (es > ms) { ms = es; as = bits; }
        if (ec > mc) { mc = ec; ac = bits; }
    }
    // block reduce through atomics on the bit patterns (non-negative floats order like their bits)
    atomicMax((unsigned int*)&out_max[0], __float_as_uint(ms));
    atomicMax((unsigned int*)&out_max[1], __float_as_uint(mc));
    if (ms > 0.f && __float_as_uint(ms) == atomicMax((unsigned int*)&out_max[2], __float_as_uint(ms))) out_arg[0] = as;
    (void)ac;
}

int main()
{
    float* d_max; unsigned long long* d_arg;
    hipMalloc((void**)&d_max, 16); hipMalloc((void**)&d_arg, 16);
    hipMemset(d_max, 0, 16); hipMemset(d_arg, 0, 16);
    const float hi_pos = 3.1515927f, hi_neg = 9.4347780f;        // pi + 0.01, 3 pi + 0.01
    uint32_t bp, bn;
    memcpy(&bp, &hi_pos, 4); memcpy(&bn, &hi_neg, 4);
    // positive floats 0 .. hi_pos: bit patterns 0 .. bp; negative floats -0 .. -hi_neg: 0x80000000 .. 0x80000000 + bn
    hipLaunchKernelGGL(k_probe, dim3(4096), dim3(256), 0, 0, 0u, bp + 1u, d_max, d_arg);
    hipLaunchKernelGGL(k_probe, dim3(4096), dim3(256), 0, 0, 0x80000000u, bn + 1u, d_max, d_arg);
    hipDeviceSynchronize();
    float h[4];
    hipMemcpy(h, d_max, 16, hipMemcpyDeviceToHost);
    printf("floats checked: %llu\n", (unsigned long long)bp + 1ull + (unsigned long long)bn + 1ull);
    printf("max |v_sin_f32(d/2pi) - sinf(wrap_to_pi(d))| = %.9g\n", h[0]);
    printf("max |v_cos_f32(d/2pi) - cosf(wrap_to_pi(d))| = %.9g\n", h[1]);
    return 0;
}
