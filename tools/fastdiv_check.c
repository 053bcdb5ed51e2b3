/* Exhaustive check: for b = fl32(sqrt(W)) and y = fl32(1/b), is
 *     q0 = RN(f*y); rem = fma(-q0, b, f); q = fma(rem, y, q0)
 * equal to the IEEE quotient f/b for EVERY finite non-negative fp32 f?  (Markstein-style correction.)
 * Usage: fastdiv_check W [W...]   -> prints "W ok" or the first mismatch; exit code = number of failing W.
 * Build: gcc -O2 -fopenmp -ffp-contract=off -mfma tools/fastdiv_check.c -lm -o /tmp/fastdiv_check */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv)
{
    int bad_w = 0;
    for (int a = 1; a < argc; ++a) {
        int W = atoi(argv[a]);
        const float b = (float)sqrt((double)W);
        const float y = 1.0f / b;
        long long bad = 0;
        uint32_t first = 0, last = 0;
        const int64_t lo = getenv("FASTDIV_LO") ? strtoll(getenv("FASTDIV_LO"), 0, 0) : 0;
#pragma omp parallel for schedule(static) reduction(+ : bad) reduction(max : last)
        for (int64_t u = lo; u < 0x7f800000LL; ++u) {
            float f = u2f((uint32_t)u);
            float q0 = f * y;
            float rem = fmaf(-q0, b, f);
            float q = fmaf(rem, y, q0);
            float ref = f / b;
            if (q != ref) {
                ++bad;
                if ((uint32_t)u > last) last = (uint32_t)u;
            }
        }
        (void)first;
        if (bad) { printf("W=%d FAIL: %lld mismatches, largest failing f bits 0x%08x (%g)\n", W, bad, last, u2f(last)); ++bad_w; }
        else printf("W=%d ok\n", W);
        fflush(stdout);
    }
    return bad_w;
}
