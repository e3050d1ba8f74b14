// sincosf_exhaustive.cpp -- checks botlab_amd/csrc/bl_math.h's bl_sincosf against the host libm's sinf/cosf for EVERY
// float with |y| <= limit (default 4.0, which covers every wrapped angle the hot path can produce).
// Build+run:  g++ -O2 -ffp-contract=off -fopenmp -I botlab_amd/csrc tests/tools/sincosf_exhaustive.cpp -o /tmp/sce && /tmp/sce
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "bl_math.h"

int main(int argc, char** argv)
{
    float limit = argc > 1 ? (float)atof(argv[1]) : 4.0f;
    uint32_t top; memcpy(&top, &limit, 4);
    long long bad_s = 0, bad_c = 0, total = 0, diff_cells = 0;
#pragma omp parallel for reduction(+ : bad_s, bad_c, total, diff_cells) schedule(static)
    for (long long u = 0; u <= (long long)top; ++u) {
        for (int sgn = 0; sgn < 2; ++sgn) {
            uint32_t b = (uint32_t)u | (sgn ? 0x80000000u : 0u);
            float y; memcpy(&y, &b, 4);
            float s, c;
            bl_sincosf(y, &s, &c);
            float rs = sinf(y), rc = cosf(y);
            if (memcmp(&s, &rs, 4) != 0) bad_s++;
            if (memcmp(&c, &rc, 4) != 0) bad_c++;
            // the scoring variant may differ from bl_sincosf on -0 only (sine +0 there)
            float s2, c2;
            bl_sincosf_cells(y, &s2, &c2);
            if ((memcmp(&s2, &s, 4) != 0 || memcmp(&c2, &c, 4) != 0) && !(b == 0x80000000u && s2 == 0.0f && c2 == 1.0f)) diff_cells++;
            total++;
        }
    }
    printf("checked %lld floats with |y| <= %g: sinf mismatches %lld, cosf mismatches %lld, bl_sincosf_cells differences other "
           "than the sign of sin(-0): %lld\n", total, limit, bad_s, bad_c, diff_cells);
    return (bad_s || bad_c || diff_cells) ? 1 : 0;
}
