// wrap_exhaustive.cpp -- bl_wrap_to_pi (float thresholds) against the reference formulation of wrap_to_pi
// (src/common/angle_functions.hpp:12-24: double comparisons against M_PI) for every float with |y| <= 100.
// Build+run: g++ -O2 -ffp-contract=off -fopenmp -I botlab_amd/csrc tests/tools/wrap_exhaustive.cpp -o /tmp/wrapt && /tmp/wrapt
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "bl_math.h"
static float ref_wrap(float angle)
{
    if (angle < -M_PI) { for (; angle < -M_PI; angle += 2.0 * M_PI); }
    else if (angle > M_PI) { for (; angle > M_PI; angle -= 2.0 * M_PI); }
    return angle;
}
int main()
{
    long long bad = 0, tot = 0;
#pragma omp parallel for reduction(+ : bad, tot)
    for (long long u = 0; u <= 0x42c80000LL; ++u) {
        for (int sg = 0; sg < 2; ++sg) {
            uint32_t b = (uint32_t)u | (sg ? 0x80000000u : 0);
            float y; memcpy(&y, &b, 4);
            float a = bl_wrap_to_pi(y), r = ref_wrap(y);
            if (memcmp(&a, &r, 4)) bad++;
            tot++;
        }
    }
    printf("wrap_to_pi: %lld floats with |y| <= 100, mismatches %lld\n", tot, bad);
    return bad != 0;
}
