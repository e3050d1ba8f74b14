// does "v_mov_b32_dpp wave_shl:1" give lane i the value of lane i + 1 across the whole wave (rows of 16 included) on gfx950?
// hipcc --offload-arch=gfx950 tests/tools/dpp_wave_shl_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o)
{
    int v = 100 + (int)threadIdx.x, r = -1;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(r) : "v"(v));
    o[threadIdx.x] = r;
}
int main()
{
    int* d; hipMalloc(&d, 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 63; ++i) bad += h[i] != 100 + i + 1;
    printf("wave_shl:1 : lane i <- lane i + 1 for i = 0..62: %s; lane 63 = %d (unwritten: -1)\n", bad ? "NO" : "yes", h[63]);
    for (int i = 0; i < 64; ++i) printf("%d ", h[i]);
    printf("\n");
    return bad != 0;
}
