// bl_math.h -- scalar arithmetic shared by every kernel (and by the host code in this library), written so that the
// floating-point results are bit-identical to what the reference computes with glibc 2.35 on an FMA-capable x86-64.
// Compile with -ffp-contract=off: every rounding below is deliberate, fused operations are spelled __builtin_fma.
#ifndef BL_MATH_H
#define BL_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define BL_HD __host__ __device__ __forceinline__
#else
#define BL_HD inline
#endif

#define BL_PI 3.14159265358979323846   /* M_PI */

// ---------------------------------------------------------------- angle helpers
// wrap_to_pi (src/common/angle_functions.hpp:12-24): float angle compared against double M_PI, stepped by a double
// 2*M_PI and narrowed back to float on every iteration.
BL_HD float bl_wrap_to_pi(float angle)
{
    // (double)angle < -M_PI  <=>  angle <= -(float)M_PI, because (float)M_PI = 3.14159274... is the float just ABOVE
    // M_PI and the next float toward zero (3.14159250...) is below it; likewise (double)angle > M_PI <=> angle >= (float)M_PI.
    // The reference loops until the angle is in range and never returns for an angle so large that a float absorbs the
    // 2*pi step (|angle| >= 2^26) or for NaN-free but absurd inputs that need billions of steps.  On a GPU that is a hung
    // device, so the loop is cut after 2^16 steps (|angle| up to ~4e5 rad wraps exactly as in the reference; beyond that
    // the result is whatever the last step left -- inputs no lidar or pose produces).
    const float PI_F = 0x1.921fb6p+1f;
    int guard = 1 << 16;
    if (angle <= -PI_F) {
        do { angle = (float)((double)angle + 2.0 * BL_PI); } while (angle <= -PI_F && --guard > 0);
    } else if (angle >= PI_F) {
        do { angle = (float)((double)angle - 2.0 * BL_PI); } while (angle >= PI_F && --guard > 0);
    }
    return angle;
}
// angle_diff (angle_functions.hpp:78-87)
BL_HD double bl_angle_diff(double l, double r)
{
    double diff = l - r;
    if (__builtin_fabs(diff) > BL_PI) diff -= (diff > 0) ? BL_PI * 2 : BL_PI * -2;
    return diff;
}
// angle_sum (angle_functions.hpp:128-138)
BL_HD double bl_angle_sum(double a, double b)
{
    double sum = a + b;
    if (__builtin_fabs(sum) > BL_PI) sum -= (sum > 0) ? BL_PI * 2 : BL_PI * -2;
    return sum;
}

// ---------------------------------------------------------------- sinf / cosf
// The reference calls std::cos(float)/std::sin(float) (mapping.cpp:48-49, sensor_model.cpp:34-38,
// particle_filter.cpp:153-154), i.e. glibc's sinf/cosf.  glibc 2.35 implements them (sysdeps/ieee754/flt-32/s_sinf.c,
// s_cosf.c, sincosf.h -- the ARM optimized-routines algorithm) as a double-precision polynomial after a fast
// pi/2 reduction; the coefficient table and the placement of the fused multiply-adds below were read from the
// libm.so.6 the reference links against on this image (__sinf_fma/__cosf_fma).  Valid for |y| < 120 (the hot path
// only ever passes wrapped angles, |y| <= pi + ulp); larger arguments take the same reduction and are NOT bit-exact.
struct bl_sincos_tab { double c0, c1, c2, c3, c4, s1, s2, s3; };

// fma(a, k, c) with compile-time constants k and c.  On the device hipcc lowers __builtin_fma(a, k, c) to a v_mov_b64 of
// c followed by v_fmac_f64; the three-address v_fma_f64 (k from a scalar pair, c from a vector pair kept outside the loop)
// is the same IEEE operation in one instruction.
BL_HD double bl_fma_kc(double a, double k, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
    return r;
#else
    return __builtin_fma(a, k, c);
#endif
}

BL_HD double bl_sin_poly(double x, double x2, double s1c, double s2c, double s3c)
{
    double x3 = x * x2;
    double s1 = bl_fma_kc(x2, s3c, s2c);
    double x7 = x3 * x2;
    double s = __builtin_fma(x3, s1c, x);
    return __builtin_fma(s1, x7, s);
}
BL_HD double bl_cos_poly(double x2, double c0, double c1c, double c2c, double c3c, double c4c)
{
    double x4 = x2 * x2;
    double c1 = __builtin_fma(x2, c1c, c0);
    double c2 = bl_fma_kc(x2, c4c, c3c);
    double x6 = x4 * x2;
    double c = __builtin_fma(x4, c2c, c1);
    return __builtin_fma(c2, x6, c);
}

template <bool SIGNED_ZERO>
BL_HD void bl_sincosf_t(float y, float* sn, float* cs)
{
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16, S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7,
                 S3 = -0x1.994eb3774cf24p-13;
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    // One code path for every |y| < 120.  glibc branches three ways (|y| < 2^-12: returns y / 1.0f; |y| < pi/4: the
    // polynomials on x itself; else reduce_fast).  The reduction with n == 0 leaves x unchanged (fma(-0, hpi, x) == x),
    // so the middle case is the general case; for |y| < 2^-12 the polynomials round to y and 1.0f as well (the
    // exhaustive check in tests/tools/sincosf_exhaustive.cpp covers every float).  Signs: sign[n & 3] multiplies the
    // sine argument and table 1 negates the cosine coefficients; both polynomials are odd/linear in those signs under
    // round-to-nearest (fma(-a, b, -c) == -fma(a, b, c)), so the signs are applied to the float results instead.
    const double x = (double)y;
    const double r = x * HPI_INV;
    const int n = ((int32_t)r + 0x800000) >> 24;              // round(x * 2/pi)
    const double xr = __builtin_fma(-(double)n, HPI, x);      // x - n * pi/2, one fused step
    const double x2 = xr * xr;
    float ps = (float)bl_sin_poly(xr, x2, S1, S2, S3);
    float pc = (float)bl_cos_poly(x2, C0, C1, C2, C3, C4);
    union { float f; uint32_t u; } us, uc, uy;
    uy.f = y;
    if (SIGNED_ZERO) {
        const bool tiny = ((uy.u >> 20) & 0x7ff) < 0x398;     // |y| < 2^-12: glibc returns y and 1.0f (keeps sin(-0) = -0)
        us.f = tiny ? y : ps; uc.f = tiny ? 1.0f : pc;
    } else {
        us.f = ps; uc.f = pc;                                 // equal for every float but -0 (sin(-0) = +0 here)
    }
    us.u ^= ((uint32_t)(n + 1) & 2u) << 30;                   // sign[n & 3] = {1, -1, -1, 1}
    uc.u ^= ((uint32_t)n & 2u) << 30;                         // table 1 (n & 2): cosine coefficients negated
    if (n & 1) { *sn = uc.f; *cs = us.f; }
    else       { *sn = us.f; *cs = uc.f; }
}

BL_HD void bl_sincosf(float y, float* sn, float* cs) { bl_sincosf_t<true>(y, sn, cs); }

// For the ray scoring only: the one input on which it differs from bl_sincosf is y = -0 (sine +0 instead of -0; checked
// over every float by tests/tools/sincosf_exhaustive.cpp), and a zero of either sign times range * cellsPerMeter added to
// the start coordinate truncates to the same cell.  Saves the |y| < 2^-12 test and two selects per ray.
BL_HD void bl_sincosf_cells(float y, float* sn, float* cs) { bl_sincosf_t<false>(y, sn, cs); }

// ---------------------------------------------------------------- pose interpolation
struct bl_pose3 { float x, y, theta; };

// interpolate_pose_by_time (src/common/interpolation.hpp:23-50) for the case before.utime != after.utime;
// ratio = (double)(t - before.utime) / (double)(after.utime - before.utime) is formed by the caller.
BL_HD bl_pose3 bl_interpolate_pose(bl_pose3 before, bl_pose3 after, double ratio)
{
    double xStep = (double)(after.x - before.x) * ratio;          // float subtraction, double product
    double yStep = (double)(after.y - before.y) * ratio;
    double thetaStep = bl_angle_diff((double)after.theta, (double)before.theta) * ratio;
    bl_pose3 out;
    out.x = (float)((double)before.x + xStep);
    out.y = (float)((double)before.y + yStep);
    out.theta = (float)bl_angle_sum((double)before.theta, thetaStep);
    return out;
}

// ---------------------------------------------------------------- grid frame
struct bl_frame { int width, height; float mpc, cpm, ox, oy; };

// global_position_to_grid_position (src/common/grid_utils.hpp:49-55) narrowed to Point<float> by the caller
// (mapping.cpp:45, sensor_model.cpp:29): double arithmetic on float members.
BL_HD void bl_global_to_grid(float gx, float gy, const bl_frame& f, float* px, float* py)
{
    *px = (float)(((double)gx - (double)f.ox) * (double)f.cpm);
    *py = (float)(((double)gy - (double)f.oy) * (double)f.cpm);
}
// global_position_to_grid_cell (grid_utils.hpp:33-38): truncating cast of the double product
BL_HD void bl_global_to_cell(double gx, double gy, const bl_frame& f, int* cx, int* cy)
{
    *cx = (int)((gx - (double)f.ox) * (double)f.cpm);
    *cy = (int)((gy - (double)f.oy) * (double)f.cpm);
}

// One step of the reference's Bresenham variant from (x1,y1) toward (x2,y2) (sensor_model.cpp:61-86; the loop body
// of mapping.cpp:115-126 with err = dx - dy).  e2 = 2*err is exact in float/double for every reachable err.
BL_HD void bl_bresenham_first_step(int x1, int y1, int x2, int y2, int* ox, int* oy)
{
    int dx = x2 > x1 ? x2 - x1 : x1 - x2;
    int dy = y2 > y1 ? y2 - y1 : y1 - y2;
    int sx = x1 < x2 ? 1 : -1;
    int sy = y1 < y2 ? 1 : -1;
    int e2 = 2 * (dx - dy);
    int x = x1, y = y1;
    if (e2 >= -dy) x += sx;
    if (e2 <= dx) y += sy;
    *ox = x;
    *oy = y;
}

// ---------------------------------------------------------------- Philox4x32-10 (counter-based RNG for the
// action-model noise and the filter initialisation; the stream depends only on (seed, step, global particle index),
// never on the launch shape or the number of GPUs)
BL_HD void bl_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// a scan is "theta_simple" when every kept ray angle lies in [0, BL_THETA_SIMPLE_MAX]: below 2 pi, so that a wrapped pose angle less
// a ray angle needs at most ONE upward 2 pi step (the bound behind the fast ray loop: bl_mcl.hip, ray_cells_fast) -- one constant for
// the check (bl_ctx.hip), the loop and its probe
#define BL_THETA_SIMPLE_MAX 6.2831f

#endif  // BL_MATH_H
