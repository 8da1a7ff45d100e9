/*
 * szg/fpmath.h — the GLSL built-in functions of the path, pinned to concrete fp32
 * algorithms.
 *
 * GLSL leaves exp / pow / sin / cos / asin / acos implementation-defined (a few ULP,
 * pow inherited from exp2(y*log2(x))), so "the reference's result" is a tolerance
 * class, not a bit pattern. The march of atmosphere/common.glinl:364-424 is
 * ill-conditioned near the ground (path segments of about one ULP of the planet
 * radius), which amplifies a 1-ULP difference in sin/cos/exp into percent-level
 * differences in the result. To make results reproducible bit for bit between the CPU
 * oracle and the GPU kernels, both evaluate the built-ins with the algorithms below,
 * which use only IEEE-754 binary32 +, -, *, /, sqrt, fma, rint and integer bit
 * manipulation — all correctly rounded and therefore identical on x86-64 and gfx950
 * (checked on hardware: DESIGN.md "numerics").
 *
 * Accuracy against float64 libm is measured in tests/test_fpmath.py:
 *   exp <= 1 ULP, log <= 3 ULP, sin/cos <= 1.6 ULP on |x| <= 10, asin/acos <= 2.5 ULP,
 *   pow = exp(y * log(x)) (error grows with |y log x| exactly as GLSL's definition).
 * Polynomial coefficients: exp/log after SLEEF 3 (Boost licence) single-precision
 * kernels; sin/cos/asin after Cephes single-precision kernels (S. Moshier).
 */
#ifndef SZG_FPMATH_H
#define SZG_FPMATH_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SZG_FP_FN __host__ __device__ __forceinline__
#else
#define SZG_FP_FN static inline
#endif

SZG_FP_FN float szg_bits_to_float(int i) { return __builtin_bit_cast(float, i); }
SZG_FP_FN int szg_float_to_bits(float f) { return __builtin_bit_cast(int, f); }
/* 2^k for k in [-126, 127] */
SZG_FP_FN float szg_pow2i(int k) { return szg_bits_to_float((k + 127) << 23); }

/* n / d for a denominator d in [1.75, 2.5] and |n| <= 0.5 (the only use: log's (m-1)/(m+1)).
 * Host: the IEEE operator. Device: the same correctly rounded quotient computed without hipcc's
 * generic denormal scaling and special-case fix-up (v_rcp_f32 seed, one Newton step, one exact fma
 * residual correction). Operands are normal and of moderate magnitude by construction, so the result is
 * bit-identical to `/` (tools/verify_div.hip, DESIGN.md "lean exact ops"). */
SZG_FP_FN float szg_div_moderate(float n, float d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    /* y = RN(1 / d) (v_rcp_f32 + one Newton step: exhaustively verified, tools/verify_div.hip), q0 = RN(n y), then ONE
     * correction with the exact residual (Markstein): the correctly rounded quotient for every operand pair of this domain
     * (the same sequence as szg_device.hpp divR0, whose verification covers it; a zero numerator gives +0 like `/`). */
    float y = __builtin_amdgcn_rcpf(d);
    y = __builtin_fmaf(__builtin_fmaf(-d, y, 1.0f), y, y);
    float const q0 = n * y;
    return __builtin_fmaf(__builtin_fmaf(-d, q0, n), y, q0);
#else
    return n / d;
#endif
}

/* All functions below are branch-free (selects only): they sit in the inner loops of
 * the GPU kernels. */

/* exp(x), natural. */
/* exp(x) for x that is not NaN (a NaN argument gives exp(-104)): szg_expf without its last select. */
SZG_FP_FN float szg_expf_notnan(float x)
{
    float const xc = __builtin_fminf(__builtin_fmaxf(x, -104.0f), 89.0f);
    float const q = __builtin_rintf(xc * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    float s = __builtin_fmaf(q, -0.693145751953125f, xc);
    s = __builtin_fmaf(q, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
    u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
    u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
    u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
    u = __builtin_fmaf(u, s, 0.5f);
    u = __builtin_fmaf(s * s, u, s) + 1.0f;
    int const qi = (int)q;
#if defined(__HIP_DEVICE_COMPILE__)
    /* The two-step scaling below in ONE instruction: the first product is exact (u in (0.5, 2), |q1| <= 76), so the value is
     * u * 2^q rounded once - which is what v_ldexp_f32 returns, denormal, zero and infinite results included. Checked on
     * gfx950 for every u in [0.5, 2) and every q in [-152, 130] (tools/verify_ldexp.hip, profiles/r02_verify_ldexp.txt). */
    return __builtin_ldexpf(u, qi);
#else
    int const q1 = qi >> 1;
    return (u * szg_pow2i(q1)) * szg_pow2i(qi - q1);
#endif
}
SZG_FP_FN float szg_expf(float x)
{
    float const r = szg_expf_notnan(x);
    return (x == x) ? r : x;
}

/* log(x), natural. x < 0 -> NaN, x == 0 -> -inf, +inf -> +inf, denormals handled. */
SZG_FP_FN float szg_logf(float x)
{
    int const tiny = x < 1.17549435e-38f;
    float const xs = tiny ? x * 16777216.0f : x; /* 2^24: exact */
    int const eadj = tiny ? -24 : 0;
    /* xs = m * 2^e with m in [0.75, 1.5) */
    int const bits = szg_float_to_bits(xs * 1.3333333333333333333333333333333333333f);
    int const e = ((bits >> 23) & 0xFF) - 127;
    float const m = szg_bits_to_float(szg_float_to_bits(xs) - (e << 23));
    float const t = szg_div_moderate(m - 1.0f, m + 1.0f);
    float const t2 = t * t;
    float p = 0.2392828464508056640625f;
    p = __builtin_fmaf(p, t2, 0.28518211841583251953125f);
    p = __builtin_fmaf(p, t2, 0.400005877017974853515625f);
    p = __builtin_fmaf(p, t2, 0.666666686534881591796875f);
    p = __builtin_fmaf(p, t2, 2.0f);
    float const fe = (float)(e + eadj);
    float r = __builtin_fmaf(t, p, 0.693147180559945286226764f * fe);
    r = (x == __builtin_inff()) ? x : r;
    r = (x == 0.0f) ? -__builtin_inff() : r;
    r = (!(x == x) || x < 0.0f) ? __builtin_nanf("") : r;
    return r;
}

/* pow(x, y) for x >= 0 as GLSL defines it: undefined for x < 0 (NaN here);
 * pow(0, y>0) = 0, pow(x, 0) = 1. */
SZG_FP_FN float szg_powf(float x, float y)
{
    float r = szg_expf(y * szg_logf(x));
    r = (x == 0.0f) ? (y > 0.0f ? 0.0f : __builtin_inff()) : r;
    r = (y == 0.0f) ? 1.0f : r;
    return r;
}

/* Argument reduction by pi/2 (three-part Cody-Waite, exact products for |x| < 2^13). */
SZG_FP_FN float szg_reduce_pio2(float x, int* quadrant)
{
    float const q = __builtin_rintf(x * 0.636619772367581343075535053490057448f);
    float r = __builtin_fmaf(q, -1.5703125f, x);
    r = __builtin_fmaf(q, -4.837512969970703125e-4f, r);
    r = __builtin_fmaf(q, -7.54978995489188216e-8f, r);
    *quadrant = (int)q;
    return r;
}
SZG_FP_FN float szg_sin_poly(float r)
{
    float const z = r * r;
    float p = -1.9515295891e-4f;
    p = __builtin_fmaf(p, z, 8.3321608736e-3f);
    p = __builtin_fmaf(p, z, -1.6666654611e-1f);
    return __builtin_fmaf(p * z, r, r);
}
SZG_FP_FN float szg_cos_poly(float r)
{
    float const z = r * r;
    float p = 2.443315711809948e-5f;
    p = __builtin_fmaf(p, z, -1.388731625493765e-3f);
    p = __builtin_fmaf(p, z, 4.166664568298827e-2f);
    return __builtin_fmaf(p * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
}
/* sin(x), cos(x); intended range |x| <= 8192 (the path uses |x| < 10), NaN outside. */
SZG_FP_FN float szg_sinf(float x)
{
    int const ok = __builtin_fabsf(x) <= 8192.0f;
    int n;
    float const r = szg_reduce_pio2(ok ? x : 0.0f, &n);
    float const v = (n & 1) ? szg_cos_poly(r) : szg_sin_poly(r);
    float const w = (n & 2) ? -v : v;
    return ok ? w : __builtin_nanf("");
}
SZG_FP_FN float szg_cosf(float x)
{
    int const ok = __builtin_fabsf(x) <= 8192.0f;
    int n;
    float const r = szg_reduce_pio2(ok ? x : 0.0f, &n);
    float const v = (n & 1) ? szg_sin_poly(r) : szg_cos_poly(r);
    float const w = ((n + 1) & 2) ? -v : v;
    return ok ? w : __builtin_nanf("");
}

/* asin on [0, 0.5]: x + x * z * P(z), z = x^2 */
SZG_FP_FN float szg_asin_core(float x, float z)
{
    float p = 4.2163199048e-2f;
    p = __builtin_fmaf(p, z, 2.4181311049e-2f);
    p = __builtin_fmaf(p, z, 4.5470025998e-2f);
    p = __builtin_fmaf(p, z, 7.4953002686e-2f);
    p = __builtin_fmaf(p, z, 1.6666752422e-1f);
    return __builtin_fmaf(p * z, x, x);
}
/* asin(x); |x| > 1 -> NaN */
SZG_FP_FN float szg_asinf(float x)
{
    float const a = __builtin_fabsf(x);
    int const small = a <= 0.5f;
    float const z = small ? a * a : 0.5f * (1.0f - a);
    float const s = small ? a : __builtin_sqrtf(z);
    float const t = szg_asin_core(s, z);
    float const r = small ? t : 1.5707963267948966192313216916397514f - (t + t);
    float const w = x < 0.0f ? -r : r;
    return (a <= 1.0f) ? w : __builtin_nanf("");
}
/* acos(x); |x| > 1 -> NaN */
SZG_FP_FN float szg_acosf(float x)
{
    float const a = __builtin_fabsf(x);
    int const small = a <= 0.5f;
    float const z = small ? a * a : 0.5f * (1.0f - a);
    float const s = small ? a : __builtin_sqrtf(z);
    float const t = szg_asin_core(s, z);
    float const hp = 1.5707963267948966192313216916397514f;
    float const rs = x < 0.0f ? (hp + t) : (hp - t);
    float const r2 = t + t;
    float const rl = x < 0.0f ? (3.14159265358979323846264338327950288f - r2) : r2;
    float const w = small ? rs : rl;
    return (a <= 1.0f) ? w : __builtin_nanf("");
}

#endif /* SZG_FPMATH_H */
