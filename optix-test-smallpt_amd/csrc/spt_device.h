// spt_device.h -- device-side arithmetic of the MI355X path-tracing megakernel (gfx950 only).
//
// Arithmetic contract (DESIGN.md "Arithmetic spec"): every float operation below is a single IEEE
// binary32 operation; the translation unit is compiled with -ffp-contract=off so that hipcc never
// fuses a*b+c (the reference is host C++ on x86-64: separate mul/add), and with
// -fhip-fp32-correctly-rounded-divide-sqrt so '/' and sqrtf are the correctly rounded operations.
// Reference lines are cited as file:line relative to the reference tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spt {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 neg(f3 a) { return mk(-a.x, -a.y, -a.z); }
// optix dot(): x*x' + y*y' + z*z', left to right
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b)
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// ---- correctly rounded sqrtf without the compiler's generic wrapper -------------------------------------
// v_sqrt_f32 is within 1 ulp; the two FMA residuals pick between s-1ulp, s, s+1ulp exactly (the same
// fix-up LLVM emits for sqrtf, minus its unconditional 2^32 pre-scaling and class check).  Exhaustively
// checked on the CPU for every binary32 mantissa: tools/verify_exact_math.c.  x = 0, +inf, NaN and x < 0
// fall through with the IEEE result (0, inf, NaN, NaN).
__device__ __forceinline__ float sqrt_fix(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u);
    const float su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, x);
    const float ru = __builtin_fmaf(-su, s, x);
    s = rd <= 0.0f ? sd : s;
    s = ru > 0.0f ? su : s;
    return s;
}
// Correctly rounded sqrtf from v_rsq_f32 and one exact-residual correction: 1 quarter-rate + 5 full-rate instructions
// (8.3 issue slots) against 13 for v_sqrt_f32 + fix-up.  y ~ 1/sqrt(x); g = RN(x*y) ~ sqrt(x) to about
// 2 ulp; d = x - g*g (one FMA, so the residual carries no rounding of g*g); result = RN(g + d*(y/2)).  Whether that last
// rounding is the correct one for every input depends on the hardware's v_rsq_f32 table, so unlike the two forms above
// it cannot be enumerated on the CPU: tests/test_gpu_math.py compares it on the device with sqrt_fix for EVERY binary32
// input 2^-96 <= x < inf (1 879 048 192 values, 0 mismatches on gfx950; the uncorrected g is the test's negative control)
// and for x = 0, -0, x < 0 and NaN (0, -0, NaN, NaN as IEEE).  x = +inf gives NaN and 0 < x < 2^-96 is not correctly
// rounded: the API routes scenes that can produce either to the guarded build (sqrt_exact).
// The reciprocal square root is taken of x + 2^-125 (== x for x >= 2^-96 under round-to-nearest) so that x = 0 gives
// a finite y and g = 0 * y = 0 rather than 0 * inf.
// NONZERO: the caller guarantees x != 0 (x >= 2^-96, negative or NaN), which saves the addition.
template <bool CORRECT = true, bool NONZERO = false>
__device__ __forceinline__ float sqrt_rsq(float x)
{
    const float y = __builtin_amdgcn_rsqf(NONZERO ? x : x + 0x1p-125f);
    const float g = x * y;
    if (!CORRECT) return g;
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, 0.5f * y, g);
}
// General form: for 0 < x < 2^-96 the residual would underflow and x = +inf gives inf * 0, so those (practically never
// occurring) inputs take the compiler's scaled sequence.
__device__ __forceinline__ float sqrt_exact(float x)
{
    float s = sqrt_rsq(x);
    const uint32_t u = __float_as_uint(x);
    if (__builtin_expect((u - 1u) < (0x0F800000u - 1u) || u == 0x7F800000u, 0)) s = __builtin_sqrtf(x);
    return s;
}

// ---- correctly rounded 1.0f / y --------------------------------------------------------------------------
// v_rcp_f32 + ONE FMA Newton step.  With an arbitrary 1-ulp starting value a second step (Markstein) and a special case
// for the all-ones mantissa are needed (tools/verify_exact_math.c enumerates that model); gfx950's v_rcp_f32 is good enough
// that one step already gives RN(1/y) for every binary32 y in 2^-100 <= y < 2^100, all-ones mantissas included -- a
// property of the hardware table, established like sqrt_rsq by enumeration on the device: tests/test_gpu_math.py compares
// all 1 677 721 600 inputs with the compiler's IEEE division (0 mismatches; bare v_rcp_f32 is the negative control).
// Outside that range the residual could under/overflow: RANGE_CHECK takes the IEEE division there (rare branch);
// RANGE_CHECK = false is used where the caller guarantees the range (lengths of unit-scale vectors / sphere radii
// validated by the host).
template <bool RANGE_CHECK = true>
__device__ __forceinline__ float rcp_exact(float y)
{
    float r = __builtin_amdgcn_rcpf(y);
    const float e = __builtin_fmaf(-y, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    if (RANGE_CHECK && __builtin_expect((__float_as_uint(y) - 0x0D800000u) >= (0x71800000u - 0x0D800000u), 0)) r = 1.0f / y;
    return r;
}

// optix normalize(): v * (1.0f / sqrtf(dot(v, v)))
template <bool GUARD = true>
__device__ __forceinline__ f3 normalize(f3 v)
{
    const float q = dot(v, v);
    const float inv = GUARD ? rcp_exact<true>(sqrt_exact(q)) : rcp_exact<false>(sqrt_rsq<true, true>(q));
    return v * inv;
}

#include "spt_deal.h"   // task dealing: deal_task (shared with the CPU check of its bijection)

// ---- D7: counter-based RNG (replaces mt19937 + uniform_real_distribution, smallpt.cpp:157,319) ----
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
constexpr uint32_t kGolden = 0x9E3779B9u;
// ctr = [31:29] branch | [28] camera | [27:2] depth | [1:0] dimension.
// The kernel keeps rbase = k0 + ((branch << 29) | (depth << 2)) * kGolden per path (one add per bounce)
// and draws dimension j from rbase + j * kGolden: identical to k0 + ctr * kGolden mod 2^32.
__device__ __forceinline__ uint32_t rng_base(uint32_t k0, uint32_t branch, uint32_t depth)
{
    return k0 + ((branch << 29) | (depth << 2)) * kGolden;
}
// raw 32 random bits of a draw; the uniform is (bits >> 8) * 2^-24
__device__ __forceinline__ uint32_t rng_draw_bits(uint32_t x, uint32_t k1)
{
    x ^= x >> 16; x *= 0x21f0aaadu;
    x += k1;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
__device__ __forceinline__ float rng_draw(uint32_t x, uint32_t k1)
{
    x ^= x >> 16; x *= 0x21f0aaadu;
    x += k1;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return (float)(x >> 8) * 0x1p-24f;
}
__device__ __forceinline__ float rng_uniform(uint32_t k0, uint32_t k1, uint32_t ctr)
{
    uint32_t x = k0 + ctr * kGolden;
    x ^= x >> 16; x *= 0x21f0aaadu;
    x += k1;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return (float)(x >> 8) * 0x1p-24f;
}

// ---- D17: sin/cos of 2*pi*u, quadrant reduction + odd degree-9 polynomial, no FMA ----
__device__ __forceinline__ float sin_quarter(float z)
{
    const float c1 = 0x1.921fb4p+0f, c3 = -0x1.4abbb6p-1f, c5 = 0x1.46676ep-4f,
                c7 = -0x1.3232fap-8f, c9 = 0x1.3c4b2cp-13f;
    float z2 = z * z;
    float p = c9;
    p = p * z2 + c7;
    p = p * z2 + c5;
    p = p * z2 + c3;
    p = p * z2 + c1;
    return p * z;
}
// Same function evaluated from the draw's raw bits: u = k * 2^-24 with k = bits >> 8, so 4u = k * 2^-22,
// q = floor(4u) = bits >> 30 and f = 4u - q = (k mod 2^22) * 2^-22 -- bit-identical to the float route below
// (every step there is exact), with two conversions fewer; signs are applied by XOR-ing the sign bit.
__device__ __forceinline__ void sincos2pi_bits(uint32_t bits, float& s, float& c)
{
    const uint32_t q = bits >> 30;
    const float f = (float)((bits << 2) >> 10) * 0x1p-22f;
    const float S = sin_quarter(f);
    const float C = sin_quarter(1.0f - f);
    const bool odd = (q & 1u) != 0u;
    const float s0 = odd ? C : S;
    const float c0 = odd ? S : C;
    s = __uint_as_float(__float_as_uint(s0) ^ ((q & 2u) << 30));            // q: 0 S, 1 C, 2 -S, 3 -C
    c = __uint_as_float(__float_as_uint(c0) ^ (((q + 1u) & 2u) << 30));      // q: 0 C, 1 -S, 2 -C, 3 S
}
__device__ __forceinline__ void sincos2pi(float u, float& s, float& c)
{
    float t = 4.0f * u;
    int q = (int)t;
    float f = t - (float)q;
    float S = sin_quarter(f);
    float C = sin_quarter(1.0f - f);
    float s0 = (q & 1) ? C : S;
    float c0 = (q & 1) ? S : C;
    s = (q & 2) ? -s0 : s0;            // q: 0 S, 1 C, 2 -S, 3 -C
    c = ((q + 1) & 2) ? -c0 : c0;      // q: 0 C, 1 -S, 2 -C, 3 S
}

}  // namespace spt
