/* verify_exact_math.c -- exhaustive / randomized CPU proof-by-enumeration that the cheap GPU sequences
 * used in optix-test-smallpt_amd/csrc/spt_device.h return the correctly rounded IEEE result:
 *   (1) sqrt fix-up:   s0 within +-1 ulp of sqrt(x)  ->  RN(sqrt(x))         (all 2^24 mantissas of [1,4))
 *   (2) reciprocal:    r0 within +-1 ulp of 1/y, two FMA Newton steps -> RN(1/y)  (all 2^23 mantissas; one exception,
 *                      the all-ones mantissa.  This is the hardware-independent MODEL; the kernels' rcp_exact uses one step
 *                      on gfx950's actual v_rcp_f32 and is enumerated on the device, tests/test_gpu_math.py)
 *   (3) double a/w via y=RN(1/w): q0=a*y; r=fma(-q0,w,a); q=fma(r,y,q0) == a/w   (w in [1,16384], random a)
 * Build: gcc -O2 -mfma -ffp-contract=off tools/verify_exact_math.c -lm -o /tmp/verify_exact_math
 * Exit code 0 = every case matched.  The list of reciprocal exceptions (if any) is printed. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef DMAX
#define DMAX 1   /* initial approximation within +-DMAX ulp of the correctly rounded value */
#endif

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static float sqrt_fix(float x, float s)
{
    float sd = u2f(f2u(s) - 1), su = u2f(f2u(s) + 1);
    float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
    s = rd <= 0.0f ? sd : s;
    s = ru > 0.0f ? su : s;
    return s;
}

static float rcp_fix(float y, float r)
{
    float e = fmaf(-y, r, 1.0f);
    r = fmaf(e, r, r);
    e = fmaf(-y, r, 1.0f);
    r = fmaf(e, r, r);
    return r;
}

int main(int argc, char** argv)
{
    long bad_sqrt = 0, bad_rcp = 0, bad_div = 0;
    int quick = argc > 1 && !strcmp(argv[1], "quick");
    int rcp_only = argc > 1 && !strcmp(argv[1], "rcp-only");   /* full reciprocal enumeration only */
    uint32_t step = quick ? 7 : 1;
    /* (1) sqrt: x in [1,4) covers both exponent parities */
    for (uint32_t u = f2u(1.0f); u < f2u(4.0f) && !rcp_only; u += step) {
        float x = u2f(u), ref = sqrtf(x);
        for (int d = -DMAX; d <= DMAX; ++d) {
            float s0 = u2f(f2u(ref) + d);
            if (sqrt_fix(x, s0) != ref) { if (bad_sqrt < 5) printf("sqrt mismatch x=%a s0=%a\n", x, s0); ++bad_sqrt; }
        }
    }
    /* zero and tiny-but-normal scale check */
    if (sqrt_fix(0.0f, 0.0f) != 0.0f) ++bad_sqrt;
    /* (2) reciprocal: y in [1,2) */
    long exc = 0;
    for (uint32_t u = f2u(1.0f); u < f2u(2.0f); u += step) {
        float y = u2f(u), ref = 1.0f / y;
        for (int d = -DMAX; d <= DMAX; ++d) {
            float r0 = u2f(f2u(ref) + d);
            if (rcp_fix(y, r0) != ref) { if (exc < 8) printf("rcp exception y=%a (mant 0x%06x) r0=%a got %a want %a\n", y, u & 0x7FFFFF, r0, rcp_fix(y, r0), ref); ++exc; }
        }
    }
    bad_rcp = exc;
    /* (3) double division by small integers */
    uint64_t st = 88172645463325252ull;
    long ndiv = rcp_only ? 0 : (quick ? 2000000 : 200000000);
    for (long i = 0; i < ndiv; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        int w = 1 + (int)((st >> 40) % 16384);
        double frac = (double)(st & 0xFFFFFF) * 0x1p-24;          /* tent offset granularity */
        float dx = (float)(frac * 2.0 - 1.0);
        int px = (int)((st >> 24) % (uint64_t)w);
        double a = (((double)(st & 1) + .5 + (double)dx) / 2.0 + (double)px);
        double y = 1.0 / (double)w;
        double q0 = a * y, r = fma(-q0, (double)w, a), q = fma(r, y, q0);
        if (q != a / (double)w) { if (bad_div < 5) printf("div mismatch a=%a w=%d\n", a, w); ++bad_div; }
    }
    printf("sqrt mismatches %ld, rcp exceptions %ld, div mismatches %ld\n", bad_sqrt, bad_rcp, bad_div);
    return (bad_sqrt || bad_rcp || bad_div) ? 1 : 0;
}
