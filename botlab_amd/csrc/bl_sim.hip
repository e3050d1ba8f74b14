// bl_sim.hip -- the simulator's lidar beam march (src/sim/lidar.py:106-138, src/sim/map.py:80-87) for many beams at once
// (SURVEY.md section 8 row f4): one thread per beam, exact double arithmetic, the reference's accumulation order.
// The trigonometry (numpy.cos / numpy.sin on a Python float in the reference) is done on the host with libm and the
// step vector handed to the kernel, so that everything the device computes is IEEE-exact and order-defined.
#include <math.h>
#include <string.h>
#include <vector>

#include "bl_internal.h"

struct sim_beam { double x, y, dx, dy; };

// Map.at_xy without bounds checks: index = row * width + col looked up among the occupied indices
__global__ __launch_bounds__(256) void k_sim_cast_beams(const int8_t* __restrict__ cells, int W, int H, double ox, double oy, double mpc,
                                                        const sim_beam* __restrict__ beams, int n, double max_distance,
                                                        double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sim_beam b = beams[i];
    const double step = mpc / 2;                                   // lidar.py:129
    const long long ncell = (long long)W * H;
    double dist = 0, result = max_distance;                        // lidar.py:124-126: no hit -> max distance
    double x = b.x, y = b.y;
    while (dist <= max_distance) {                                 // lidar.py:134-138
        const double row = floor((y - oy) / mpc);                  // map.py:81-82
        const double col = floor((x - ox) / mpc);
        // row * width + col in exact integer arithmetic; far outside the map the index is simply not in the set
        bool occ = false;
        if (fabs(row) < 1.0e9 && fabs(col) < 1.0e9) {
            const long long idx = (long long)row * W + (long long)col;
            if (idx >= 0 && idx < ncell) occ = cells[idx] > 0;
        }
        if (occ) { result = dist; break; }
        x += b.dx; y += b.dy; dist += step;
    }
    out[i] = result;
}

extern "C" int bl_sim_cast_beams(bl_ctx* ctx, const bl_grid* world, double origin_x, double origin_y, double meters_per_cell,
                                 const double* x, const double* y, const double* angle, int n, double max_distance, double* out_ranges)
{
    BL_CHECK_ARG(ctx != nullptr && world != nullptr && world->ctx == ctx && n >= 0 && meters_per_cell > 0);
    if (n == 0) return BL_OK;
    BL_CHECK_ARG(x != nullptr && y != nullptr && angle != nullptr && out_ranges != nullptr);
    BL_HIP(hipSetDevice(ctx->device));
    std::vector<sim_beam> h((size_t)n);
    const double step = meters_per_cell / 2;
    for (int i = 0; i < n; ++i) {                                   // lidar.py:129-131 (libm where the reference has numpy)
        h[i].x = x[i]; h[i].y = y[i];
        h[i].dx = cos(angle[i]) * step;
        h[i].dy = sin(angle[i]) * step;
    }
    sim_beam* d_beams = nullptr; double* d_out = nullptr;
    BL_HIP(hipMalloc((void**)&d_beams, (size_t)n * sizeof(sim_beam)));
    hipError_t e = hipMalloc((void**)&d_out, (size_t)n * sizeof(double));
    if (e != hipSuccess) { (void)hipFree(d_beams); bl_set_error("hipMalloc failed: %s", hipGetErrorString(e)); return BL_ERR_HIP; }
    int rc = BL_OK;
    do {
        if (hipMemcpyAsync(d_beams, h.data(), (size_t)n * sizeof(sim_beam), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = BL_ERR_HIP; break; }
        hipLaunchKernelGGL(k_sim_cast_beams, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, world->cells, world->frame.width,
                           world->frame.height, origin_x, origin_y, meters_per_cell, d_beams, n, max_distance, d_out);
        if (hipGetLastError() != hipSuccess) { rc = BL_ERR_HIP; break; }
        if (hipMemcpyAsync(out_ranges, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { rc = BL_ERR_HIP; break; }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = BL_ERR_HIP; break; }
    } while (0);
    if (rc) bl_set_error("bl_sim_cast_beams: HIP call failed: %s", hipGetErrorString(hipGetLastError()));
    (void)hipFree(d_beams); (void)hipFree(d_out);
    return rc;
}
