// szg_device.hpp — device-side math shared by the CDNA4 kernels.
//
// Every function keeps the operation ORDER of the GLSL it implements (cited
// per function, paths relative to the reference's shaders/ directory) so that
// with -ffp-contract=off the +,-,*,/,sqrt results are bit-identical to a
// straight evaluation; exp/pow/sin/cos/asin/acos are the pinned fp32 algorithms
// of include/szg/fpmath.h (never the device math library), so the kernels
// reproduce the scalar oracle bit for bit.
// What differs from the shaders is purely structural: loop invariants are
// hoisted into per-ray / per-light / per-atmosphere constants, images are
// linear buffers, samplers are explicit address arithmetic.
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "szg/abi.h"
#include "szg/fpmath.h"

#define SZG_DEV __device__ __forceinline__

namespace szg
{
struct V2
{
    float x, y;
};
struct V3
{
    float x, y, z;
};
struct V4
{
    float x, y, z, w;
};

SZG_DEV V3 mk3(float a, float b, float c) { return V3{a, b, c}; }
SZG_DEV V3 splat(float a) { return V3{a, a, a}; }
SZG_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
SZG_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
SZG_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
SZG_DEV V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
SZG_DEV V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
SZG_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
SZG_DEV V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
SZG_DEV V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
// Contraction rule (include/szg/contraction.h): a * b + c is fused only at a closed list of places, by site class, explicitly
// and at the same places as the oracle; the compiler itself contracts nothing (-ffp-contract=off). -DSZG_LITERAL builds
// libszg_hip_literal.so (SZG_CONTRACT = 0: two roundings everywhere), i.e. kernels that execute the shaders' SPIR-V
// literally; tests/test_gpu_spirv_pin.py compares that build, bit for bit and on the GPU, with the vectors an interpreter
// recorded from the reference's committed .spv (tests/golden/spirv_vectors.npz), and holds the product to 1e-4 / 1 LSB of
// them. The fused multiply-adds INSIDE the exact operators below (rcpN, divR, sqrtN, exp / log polynomials) are not part of
// the rule: they are how those operators reach their correctly rounded results, and stay.
#ifdef SZG_LITERAL
#define SZG_CONTRACT SZG_CONTRACT_NONE
#endif
} // namespace szg
#include "szg/contraction.h"
namespace szg
{
// shading geometry (SZG_C_DOT) ...
SZG_DEV float dot(V3 a, V3 b) { return SZG_CON(SZG_C_DOT, a.z, b.z, SZG_CON(SZG_C_DOT, a.y, b.y, a.x * b.x)); }
SZG_DEV float dot(V2 a, V2 b) { return SZG_CON(SZG_C_DOT, a.y, b.y, a.x * b.x); }
// ... and the atmosphere geometry of common.glinl (SZG_C_ATMODOT: raySphere, the LUT samplers, the march)
SZG_DEV float dotA(V3 a, V3 b) { return SZG_CON(SZG_C_ATMODOT, a.z, b.z, SZG_CON(SZG_C_ATMODOT, a.y, b.y, a.x * b.x)); }
// pbrFunctions.glinl (SZG_C_PBRDOT) and lights.comp (SZG_C_LDOT)
SZG_DEV float dotP(V3 a, V3 b) { return SZG_CON(SZG_C_PBRDOT, a.z, b.z, SZG_CON(SZG_C_PBRDOT, a.y, b.y, a.x * b.x)); }
SZG_DEV float dotL(V3 a, V3 b) { return SZG_CON(SZG_C_LDOT, a.z, b.z, SZG_CON(SZG_C_LDOT, a.y, b.y, a.x * b.x)); }
// length(position) in the 500-step loop of transmittance_LUT.comp (SZG_C_TMAIN)
SZG_DEV float dotT(V3 a, V3 b) { return SZG_CON(SZG_C_TMAIN, a.z, b.z, SZG_CON(SZG_C_TMAIN, a.y, b.y, a.x * b.x)); }
// the march's accumulations (SZG_C_ACCUM) and sample points (SZG_C_POINT)
SZG_DEV V3 fma3(V3 a, float s, V3 c) { return V3{SZG_CON(SZG_C_ACCUM, a.x, s, c.x), SZG_CON(SZG_C_ACCUM, a.y, s, c.y), SZG_CON(SZG_C_ACCUM, a.z, s, c.z)}; }
SZG_DEV V3 fma3(V3 a, V3 b, V3 c) { return V3{SZG_CON(SZG_C_ACCUM, a.x, b.x, c.x), SZG_CON(SZG_C_ACCUM, a.y, b.y, c.y), SZG_CON(SZG_C_ACCUM, a.z, b.z, c.z)}; }
SZG_DEV V3 fnma(float t, V3 d, V3 c) { return V3{SZG_CON(SZG_C_POINT, -t, d.x, c.x), SZG_CON(SZG_C_POINT, -t, d.y, c.y), SZG_CON(SZG_C_POINT, -t, d.z, c.z)}; }
SZG_DEV float length(V3 a) { return sqrtf(dot(a, a)); }
SZG_DEV V3 normalize(V3 v)
{
    float const s = 1.0f / sqrtf(dot(v, v));
    return v * s;
}
SZG_DEV V2 normalize(V2 v)
{
    float const s = 1.0f / sqrtf(dot(v, v));
    return V2{v.x * s, v.y * s};
}
SZG_DEV V3 normalizeP(V3 v)
{
    float const s = 1.0f / sqrtf(dotP(v, v));
    return v * s;
}
SZG_DEV V3 normalizeL(V3 v)
{
    float const s = 1.0f / sqrtf(dotL(v, v));
    return v * s;
}
SZG_DEV float lengthA(V3 a) { return sqrtf(dotA(a, a)); }
SZG_DEV V3 normalizeA(V3 v)
{
    float const s = 1.0f / sqrtf(dotA(v, v));
    return v * s;
}
SZG_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
SZG_DEV V3 clamp01(V3 v) { return V3{clampf(v.x, 0.0f, 1.0f), clampf(v.y, 0.0f, 1.0f), clampf(v.z, 0.0f, 1.0f)}; }
SZG_DEV V3 mix(V3 a, V3 b, V3 w)
{
    return V3{SZG_CON(SZG_C_MIX, b.x, w.x, a.x * (1.0f - w.x)), SZG_CON(SZG_C_MIX, b.y, w.y, a.y * (1.0f - w.y)),
              SZG_CON(SZG_C_MIX, b.z, w.z, a.z * (1.0f - w.z))};
}
SZG_DEV float smoothstep(float e0, float e1, float x)
{
    float const t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
SZG_DEV float safeSqrt(float v) { return sqrtf(fmaxf(v, 0.0f)); } // atmosphere/common.glinl:23-26

// ---------------------------------------------------------------------------
// Lean exact division and square root.
//
// hipcc's correctly rounded `/` and sqrtf() cost ~47 and ~64 cycles per wave64 operation
// on gfx950 (an fma costs ~2.3): every call carries the denormal/overflow scaling and the
// special-case fix-up. The sequences below are the SAME algorithms without the scaling
// (v_rcp_f32 / v_rsq_f32 seed, Newton, exact fma residual corrections). They return the
// IEEE-754 correctly rounded result — bit-identical to `/` and sqrtf() — whenever their
// operand preconditions hold, and cost ~6 / ~6 instructions; a division whose denominator's
// reciprocal is shared costs 3. Checked on MI355X: sqrtN == sqrtf for ALL
// binary32 inputs in its domain; divN == `/` on 4.3e9 random operand pairs with exponents
// in [-60, 60] plus zero numerators (scratch history in DESIGN.md "lean exact ops").
// They are used only under a wave-uniform `lean` flag that the kernels derive from the
// atmosphere block and the ray (see leanAtmosphere / leanRay); otherwise the generic
// operators are used, so results never depend on the flag.
// ---------------------------------------------------------------------------
// Correctly rounded reciprocal: rcpN(b) == RN(1 / b) for EVERY binary32 b with |b| in [2^-60, 2^60]
// (tools/verify_div.hip, exhaustive on MI355X): v_rcp_f32 and one Newton step.
SZG_DEV float rcpN(float b)
{
    float const y = __builtin_amdgcn_rcpf(b);
    float const e = __builtin_fmaf(-b, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
// a / b given y = rcpN(b); a == +-0 or |a| in [2^-60, 2^60]. Markstein's theorem: with y the correctly rounded
// reciprocal, q0 = RN(a * y) and the exact residual r = a - b * q0 (one fma), RN(q0 + r * y) is the correctly rounded
// quotient, the only candidates for an exception being denominators whose significand is all ones — and those are
// checked exhaustively against `/` together with the premise (tools/verify_div.hip: every numerator significand for the
// 120 all-ones denominators of the domain and for 4096 random ones, plus 6.4e10 random and structured pairs).
SZG_DEV float divR(float a, float b, float y)
{
    float const q0 = a * y;
    float const r = __builtin_fmaf(-b, q0, a);
    float const q = __builtin_fmaf(r, y, q0);
    return __builtin_copysignf(q, q0); // a zero quotient keeps the sign IEEE division gives it
}
SZG_DEV float divN(float a, float b) { return divR(a, b, rcpN(b)); }
// divR without the sign fix of a zero quotient (one v_bfi less). Used where a quotient of zero is consumed sign-blind or
// cannot be negative — every divRX site: exp(+-0) = 1 (densities), 1 - (+-0) (ozone tent), bias + q * scale with bias > 0
// (both LUT coordinates), mu only as -r*mu + sqrt(...) >= 0 and mu*mu (LUT tap), (mu_sun - c) - e0 with e0 != 0 (sun
// smoothstep), non-negative numerators over positive denominators (rho / H, Rp / r, distance * (i + .5) / 500), whose zero
// quotient comes out +0 on its own.
SZG_DEV float divR0(float a, float b, float y)
{
    float const q0 = a * y;
    float const r = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r, y, q0);
}
// sqrt(x) for x == 0 or x in [2^-96, FLT_MAX] (NaN -> NaN): v_rsq_f32 seed, one Newton step on the exact fma residual.
// Bit-identical to sqrtf for EVERY binary32 value of that domain (tools/verify_sqrt.hip, exhaustive on MI355X).
// The max() only matters for x == 0: rsq stays finite, so 0 * y = 0 and the correction is fma(0, h, 0) = 0.
SZG_DEV float sqrtN(float x)
{
    float const y = __builtin_amdgcn_rsqf(fmaxf(x, 0x1p-126f));
    float const s = x * y;
    float const h = 0.5f * y;
    float const r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}
// sqrtN for x in [2^-96, FLT_MAX] (not 0): without the max()
SZG_DEV float sqrtP(float x)
{
    float const y = __builtin_amdgcn_rsqf(x);
    float const s = x * y;
    float const h = 0.5f * y;
    float const r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}
template <bool LEAN> SZG_DEV float sqrtX(float x) { return LEAN ? sqrtN(x) : sqrtf(x); }
// square root of a squared radius / squared length that a lean path knows to be positive (>= the squared lean floor)
template <bool LEAN> SZG_DEV float sqrtPX(float x) { return LEAN ? sqrtP(x) : sqrtf(x); }
template <bool LEAN> SZG_DEV float safeSqrtX(float v) { return sqrtX<LEAN>(fmaxf(v, 0.0f)); }
// exp of a value that is never NaN on a lean path (finite coefficients, radii above the lean floor)
template <bool LEAN> SZG_DEV float expX(float x) { return LEAN ? szg_expf_notnan(x) : szg_expf(x); }
// szg_expf (szg/fpmath.h) for x in [-86, 87], not NaN: the same reduction and polynomial, value for value. In that range the
// clamp of the argument to [-104, 89] is the identity, q = rint(x log2 e) lies in [-125, 126] and u in (0.5, 2), so
// u * 2^(q >> 1) and its product with 2^(q - (q >> 1)) are exact (no underflow, no overflow): (u * 2^q1) * 2^(q - q1) is
// u * 2^q, which is what v_ldexp_f32 returns for a normal result. 9 instructions less than szg_expf_notnan.
SZG_DEV float expInner(float x)
{
    float const q = __builtin_rintf(x * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    float s = __builtin_fmaf(q, -0.693145751953125f, x);
    s = __builtin_fmaf(q, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
    u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
    u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
    u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
    u = __builtin_fmaf(u, s, 0.5f);
    u = __builtin_fmaf(s * s, u, s) + 1.0f;
    return __builtin_ldexpf(u, (int)q);
}
// (quotients of the divX sites — the two segment cosines and the smoothstep argument — are consumed sign-blind too)
template <bool LEAN> SZG_DEV float divX(float a, float b) { return LEAN ? divR0(a, b, rcpN(b)) : a / b; }
template <bool LEAN> SZG_DEV float divRX(float a, float b, float y) { return LEAN ? divR0(a, b, y) : a / b; }
SZG_DEV float xorSign(float x, unsigned signMask)
{
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) ^ signMask);
}
// a sky-view LUT texel the composite may leave unsampled: a finite number well inside the fp32 range
SZG_DEV bool slutTexelFinite(float r, float g, float b) { return fabsf(r) <= 0x1p100f && fabsf(g) <= 0x1p100f && fabsf(b) <= 0x1p100f; }
SZG_DEV bool inRange(float x, float lo, float hi) { return x >= lo && x <= hi; } // false for NaN
SZG_DEV float divN0(float a, float b) { return divR0(a, b, rcpN(b)); }
// true when `c` holds on every active lane: one compare into a lane mask and one scalar test (HIP's __all() builds two
// ballots from an int predicate, ~15 instructions per use)
SZG_DEV bool waveAll(bool c) { return __builtin_amdgcn_ballot_w64(!c) == 0ull; }

// szg_powf(x, y) (szg/fpmath.h: exp(y * log x) with GLSL's special cases) for a base that is a positive NORMAL number
// (2^-126 <= x < inf) and an exponent that is a finite number other than 0 - the same operations value for value, without
// the selects that cannot fire there: log's denormal pre-scaling (x >= 2^-126) and its x == inf / x == 0 / x < 0 / NaN
// cases, exp's NaN pass-through (the argument y * log x is a finite product), pow's x == 0 and y == 0 cases. The clamp of
// exp's argument stays: x^160 underflows for most bases.
SZG_DEV float powLean(float x, float y)
{
    int const bits = szg_float_to_bits(x * 1.3333333333333333333333333333333333333f);
    int const e = ((bits >> 23) & 0xFF) - 127;
    float const m = szg_bits_to_float(szg_float_to_bits(x) - (e << 23));
    float const t = szg_div_moderate(m - 1.0f, m + 1.0f);
    float const t2 = t * t;
    float p = 0.2392828464508056640625f;
    p = __builtin_fmaf(p, t2, 0.28518211841583251953125f);
    p = __builtin_fmaf(p, t2, 0.400005877017974853515625f);
    p = __builtin_fmaf(p, t2, 0.666666686534881591796875f);
    p = __builtin_fmaf(p, t2, 2.0f);
    float const fe = (float)e;
    float const lg = __builtin_fmaf(t, p, 0.693147180559945286226764f * fe);
    return szg_expf_notnan(y * lg);
}
// wave-uniform precondition of powLean
SZG_DEV bool powLeanOK(float x, float y)
{
    return x >= 1.17549435e-38f && x <= 0x1p127f && fabsf(y) >= 0x1p-100f && fabsf(y) <= 0x1p20f;
}

// column-major 4x4 times (x, y, z, w), rows summed left to right
struct M4
{
    float m[16];
};
SZG_DEV V4 mul(const M4& a, float x, float y, float z, float w)
{
    V4 r;
    r.x = SZG_CON(SZG_C_MATVEC, a.m[12], w, SZG_CON(SZG_C_MATVEC, a.m[8], z, SZG_CON(SZG_C_MATVEC, a.m[4], y, a.m[0] * x)));
    r.y = SZG_CON(SZG_C_MATVEC, a.m[13], w, SZG_CON(SZG_C_MATVEC, a.m[9], z, SZG_CON(SZG_C_MATVEC, a.m[5], y, a.m[1] * x)));
    r.z = SZG_CON(SZG_C_MATVEC, a.m[14], w, SZG_CON(SZG_C_MATVEC, a.m[10], z, SZG_CON(SZG_C_MATVEC, a.m[6], y, a.m[2] * x)));
    r.w = SZG_CON(SZG_C_MATVEC, a.m[15], w, SZG_CON(SZG_C_MATVEC, a.m[11], z, SZG_CON(SZG_C_MATVEC, a.m[7], y, a.m[3] * x)));
    return r;
}
SZG_DEV M4 mul(const M4& a, const M4& b)
{
    M4 r;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
        V4 const c = mul(a, b.m[j * 4 + 0], b.m[j * 4 + 1], b.m[j * 4 + 2], b.m[j * 4 + 3]);
        r.m[j * 4 + 0] = c.x;
        r.m[j * 4 + 1] = c.y;
        r.m[j * 4 + 2] = c.z;
        r.m[j * 4 + 3] = c.w;
    }
    return r;
}
// glm's mat4 * mat4 as the reference's HOST code evaluates it (one rounding per operation, columns combined left to right:
// host_scene.cpp szg_mat4_mul): for matrices the reference computes on the CPU and hands to a shader - the per-light
// projView of the shadow passes (shadowpass.cpp:188-248) - as opposed to products written in a shader (mul above).
SZG_DEV M4 mulGlm(const M4& a, const M4& b)
{
    M4 r;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            r.m[j * 4 + i] = a.m[i] * b.m[j * 4 + 0] + a.m[4 + i] * b.m[j * 4 + 1] + a.m[8 + i] * b.m[j * 4 + 2] + a.m[12 + i] * b.m[j * 4 + 3];
        }
    }
    return r;
}
// glm::inverse(mat4) (cofactor expansion, glm/detail/func_matrix.inl), same operation order as host_scene.cpp
SZG_DEV M4 inverse4(const M4& m)
{
    auto M = [&](int c, int r) { return m.m[c * 4 + r]; };
    float const C00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3), C02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3),
                C03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3), C04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3),
                C06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3), C07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3),
                C08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2), C10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2),
                C11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2), C12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3),
                C14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3), C15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3),
                C16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2), C18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2),
                C19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2), C20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1),
                C22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1), C23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    float const F0[4] = {C00, C00, C02, C03}, F1[4] = {C04, C04, C06, C07}, F2[4] = {C08, C08, C10, C11},
                F3[4] = {C12, C12, C14, C15}, F4[4] = {C16, C16, C18, C19}, F5[4] = {C20, C20, C22, C23};
    float const V0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)}, V1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)},
                V2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)}, V3_[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
    float const SA[4] = {1.0f, -1.0f, 1.0f, -1.0f}, SB[4] = {-1.0f, 1.0f, -1.0f, 1.0f};
    M4 inv;
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        inv.m[0 * 4 + i] = (V1[i] * F0[i] - V2[i] * F1[i] + V3_[i] * F2[i]) * SA[i];
        inv.m[1 * 4 + i] = (V0[i] * F0[i] - V2[i] * F3[i] + V3_[i] * F4[i]) * SB[i];
        inv.m[2 * 4 + i] = (V0[i] * F1[i] - V1[i] * F3[i] + V3_[i] * F5[i]) * SA[i];
        inv.m[3 * 4 + i] = (V0[i] * F2[i] - V1[i] * F4[i] + V2[i] * F5[i]) * SB[i];
    }
    float const det = (M(0, 0) * inv.m[0] + M(0, 1) * inv.m[4]) + (M(0, 2) * inv.m[8] + M(0, 3) * inv.m[12]);
    float const ood = 1.0f / det;
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
        inv.m[i] = inv.m[i] * ood;
    }
    return inv;
}
SZG_DEV M4 load_m4(const szg_mat4& s)
{
    M4 r;
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
        r.m[i] = s.m[i];
    }
    return r;
}

// ---------------------------------------------------------------------------
// Formats
// ---------------------------------------------------------------------------
// imageStore on rgba16 (UNORM16): clamp, scale, round to nearest even.
SZG_DEV unsigned unorm16(float x)
{
    float const c = fminf(fmaxf(x, 0.0f), 1.0f); // NaN -> 0
    return (unsigned)__float2int_rn(c * 65535.0f);
}
SZG_DEV uint2 pack_unorm16x4(float r, float g, float b, float a)
{
    return make_uint2(unorm16(r) | (unorm16(g) << 16), unorm16(b) | (unorm16(a) << 16));
}
SZG_DEV V4 unpack_half4(uint2 v)
{
    __half2 const lo = *reinterpret_cast<const __half2*>(&v.x);
    __half2 const hi = *reinterpret_cast<const __half2*>(&v.y);
    float2 const a = __half22float2(lo);
    float2 const b = __half22float2(hi);
    return V4{a.x, a.y, b.x, b.y};
}
SZG_DEV uint2 pack_half4(float r, float g, float b, float a)
{
    // The stored value is RNE(fp16) of the ROUNDED fp32 value. Keep the compiler from folding a producing fp32
    // multiply into the conversion (v_fma_mixlo_f16 rounds the exact product once: a different result at ties).
    asm volatile("" : "+v"(r), "+v"(g), "+v"(b), "+v"(a));
    __half2 const lo = __floats2half2_rn(r, g);
    __half2 const hi = __floats2half2_rn(b, a);
    uint2 o;
    o.x = *reinterpret_cast<const unsigned*>(&lo);
    o.y = *reinterpret_cast<const unsigned*>(&hi);
    return o;
}

// Linear image view passed by value to kernels.
struct Img
{
    const unsigned char* data;
    unsigned width, height, pitch;
};
struct ImgRW
{
    unsigned char* data;
    unsigned width, height, pitch;
};

// szg_rowtile: local row -> global row
struct RowMap
{
    unsigned block_rows, rank, nranks;
};
SZG_DEV unsigned global_row(const RowMap& m, unsigned local)
{
    if (m.nranks <= 1u)
    {
        return local;
    }
    return ((local / m.block_rows) * m.nranks + m.rank) * m.block_rows + local % m.block_rows;
}

// ---------------------------------------------------------------------------
// Atmosphere block in registers/SGPRs + per-atmosphere constants
// (types/atmosphere.glinl:3-32)
// ---------------------------------------------------------------------------
struct Atm
{
    V3 scatteringRayleigh;
    float densityScaleRayleigh;
    V3 absorptionRayleigh;
    float planetRadius;
    V3 scatteringMie;
    float densityScaleMie;
    float atmosphereRadius;
    V3 incidentDirectionSun;
    V3 scatteringOzone;
    V3 absorptionOzone;
    V3 sunIntensitySpectrum;
    float sunAngularRadius;
    // hoisted sub-expressions of common.glinl:42-46
    float Ra2, Rp2, H;
    // refined reciprocals for divR (valid when `lean`)
    float rcpH, rcpDsR, rcpDsM, rcp15;
    bool lean; // leanAtmosphere(): every atmosphere-level precondition of the lean ops holds
    // Squared radius floor of the lean paths: >= 0.9 Rp (the x_mu denominator) and >= Rp - 80 min(Hr, Hm), so that both
    // densities exp(-altitude / H) stay finite (<= e^80) wherever a lean path evaluates them.
    float leanFloor2;
    // Zero coefficient vectors (Earth defaults: Rayleigh absorption and ozone scattering are 0, scene.cpp:62-70). With
    // finite densities their terms are exactly +0, and acc + (+0) == acc for every acc but -0, which a sum of
    // products of sign-clear coefficients and non-negative densities cannot be: the lean paths skip those terms.
    bool zeroAbsorptionRayleigh, zeroScatteringOzone;
    bool signClearCoefficients;
    // the extinction sum stays in [2^-40, 2^52] for radii in [sqrt(extFloor2), sqrt(extCeil2)] (lean division of the
    // in-scatter integral): coefficients <= 2^10, densities <= e^27, Rayleigh scattering alone >= 2^-40 at the shell's top
    bool extModerate;
    // the sun direction is a unit vector to 1 % and the sun's angular radius a sane positive number: with extModerate and a
    // moderate transmittance LUT, what makes every environment sample (sky-view LUT, sun disc, ground march) finite for
    // any finite ray — the phase functions see a true cosine (phaseMie's pow(1 + g^2 - 2 g c, 1.5) is NaN for c > 1.025)
    bool sunSane;
    float extFloor2, extCeil2; // Rayleigh / Mie scattering and Rayleigh absorption carry no sign bit: partial sums are never -0
    // Squared radius ceiling of the INNER lean march (marchLoop<true, true>): min(Ra (1 - 2^-10), Rp + 85 min(Hr, Hm)).
    // Below it (a) every transmittance-LUT coordinate has a discriminant r^2 (mu^2 - 1) + Ra^2 >= 2^-10 Ra^2, far above its
    // rounding error, so the clamp to 0 under its square root never acts (the clamp of the distance itself stays: NaN cosines
    // of zero-length segments rely on it), and (b) both densities exp(-altitude / H) have arguments in [-85, 80], where the
    // clamp of exp's argument is the identity.
    float innerCeil2;
};
SZG_DEV bool plusZero3(V3 v)
{
    return (__builtin_bit_cast(unsigned, v.x) | __builtin_bit_cast(unsigned, v.y) | __builtin_bit_cast(unsigned, v.z)) == 0u;
}
SZG_DEV bool signClear3(V3 v) // no component negative, -0 or NaN-with-sign; (+NaN passes and poisons every sum alike)
{
    return ((__builtin_bit_cast(unsigned, v.x) | __builtin_bit_cast(unsigned, v.y) | __builtin_bit_cast(unsigned, v.z)) >> 31) == 0u;
}
SZG_DEV Atm load_atm(const szg_atmosphere_packed* p)
{
    Atm a;
    a.scatteringRayleigh = mk3(p->scatteringRayleighPerMm[0], p->scatteringRayleighPerMm[1], p->scatteringRayleighPerMm[2]);
    a.densityScaleRayleigh = p->densityScaleRayleighMm;
    a.absorptionRayleigh = mk3(p->absorptionRayleighPerMm[0], p->absorptionRayleighPerMm[1], p->absorptionRayleighPerMm[2]);
    a.planetRadius = p->planetRadiusMm;
    a.scatteringMie = mk3(p->scatteringMiePerMm[0], p->scatteringMiePerMm[1], p->scatteringMiePerMm[2]);
    a.densityScaleMie = p->densityScaleMieMm;
    a.atmosphereRadius = p->atmosphereRadiusMm;
    a.incidentDirectionSun = mk3(p->incidentDirectionSun[0], p->incidentDirectionSun[1], p->incidentDirectionSun[2]);
    a.scatteringOzone = mk3(p->scatteringOzonePerMm[0], p->scatteringOzonePerMm[1], p->scatteringOzonePerMm[2]);
    a.absorptionOzone = mk3(p->absorptionOzonePerMm[0], p->absorptionOzonePerMm[1], p->absorptionOzonePerMm[2]);
    a.sunIntensitySpectrum = mk3(p->sunIntensitySpectrum[0], p->sunIntensitySpectrum[1], p->sunIntensitySpectrum[2]);
    a.sunAngularRadius = p->sunAngularRadius;
    a.Ra2 = a.atmosphereRadius * a.atmosphereRadius;
    a.Rp2 = a.planetRadius * a.planetRadius;
    a.H = safeSqrt(a.Ra2 - a.Rp2);
    // Preconditions of the lean ops that depend only on the atmosphere: radii, H and density scales of moderate
    // magnitude, and the x_mu denominator rho + H - (Ra - r) bounded away from 0 for every r >= 0.9 Rp.
    float const lo = 0x1p-30f, hi = 0x1p30f;
    // The density scales must also be large against the rounding of a radius (a few ulps of Rp): the zero-coefficient
    // shortcuts and expX<true> count on exp(-altitude / scale) staying finite for samples the lean floor admits, and an
    // altitude that comes out a few ulps below that floor must not be e^100 scale heights deep.
    float const scaleFloor = a.planetRadius * 0x1p-18f;
    a.lean = inRange(a.planetRadius, lo, hi) && inRange(a.atmosphereRadius, lo, hi) && inRange(a.H, lo, hi) &&
             inRange(a.densityScaleRayleigh, fmaxf(lo, scaleFloor), hi) && inRange(a.densityScaleMie, fmaxf(lo, scaleFloor), hi) &&
             (a.H - a.atmosphereRadius + 0.9f * a.planetRadius >= 0x1p-20f);
    // coefficients finite and of moderate magnitude: no NaN / inf can enter the extinction sum on a lean path
    // (expX<true> relies on that)
    V3 const* const coefficients[5] = {&a.scatteringRayleigh, &a.absorptionRayleigh, &a.scatteringMie, &a.scatteringOzone, &a.absorptionOzone};
#pragma unroll
    for (int i = 0; i < 5; i++)
    {
        a.lean = a.lean && inRange(fabsf(coefficients[i]->x), 0.0f, hi) && inRange(fabsf(coefficients[i]->y), 0.0f, hi) &&
                 inRange(fabsf(coefficients[i]->z), 0.0f, hi);
    }
    float const rFloor = fmaxf(0.9f * a.planetRadius, a.planetRadius - 80.0f * fminf(a.densityScaleRayleigh, a.densityScaleMie));
    a.leanFloor2 = rFloor * rFloor;
    bool const signs = signClear3(a.scatteringRayleigh) && signClear3(a.scatteringMie);
    a.signClearCoefficients = signs && signClear3(a.absorptionRayleigh);
    a.zeroAbsorptionRayleigh = signs && plusZero3(a.absorptionRayleigh);
    a.zeroScatteringOzone = signs && signClear3(a.absorptionRayleigh) && plusZero3(a.scatteringOzone);
    {
        float const shell = a.atmosphereRadius - a.planetRadius;
        float const minRayleigh = fminf(fminf(a.scatteringRayleigh.x, a.scatteringRayleigh.y), a.scatteringRayleigh.z);
        float const thinnest = szg_expf(-1.02f * (shell / a.densityScaleRayleigh)) * 0.5f;
        bool small = true;
#pragma unroll
        for (int i = 0; i < 5; i++)
        {
            small = small && coefficients[i]->x <= 0x1p10f && coefficients[i]->y <= 0x1p10f && coefficients[i]->z <= 0x1p10f;
        }
        a.extModerate = a.lean && a.signClearCoefficients && signClear3(a.scatteringOzone) && signClear3(a.absorptionOzone) && small &&
                        inRange(shell, lo, hi) && (minRayleigh * thinnest >= 0x1p-40f);
        float const sun2 = dot(a.incidentDirectionSun, a.incidentDirectionSun);
        // (and its azimuth exists: normalize(vec2(-sun.x, -sun.z)) of skyview_LUT.comp:60-61 / camera.comp:113-114 is 0/0 for an
        // exactly vertical sun)
        float const sunHorizontal2 = a.incidentDirectionSun.x * a.incidentDirectionSun.x + a.incidentDirectionSun.z * a.incidentDirectionSun.z;
        a.sunSane = inRange(sun2, 0.98f, 1.02f) && inRange(a.sunAngularRadius, 0x1p-30f, 1.5f) && sunHorizontal2 >= 0x1p-100f;
        float const eFloor = fmaxf(0.9f * a.planetRadius, a.planetRadius - 27.0f * fminf(a.densityScaleRayleigh, a.densityScaleMie));
        float const eCeil = a.atmosphereRadius + shell * 0.005f;
        a.extFloor2 = eFloor * eFloor;
        a.extCeil2 = eCeil * eCeil;
        float const iCeil = fminf(a.atmosphereRadius * (1.0f - 0x1p-10f),
                                  a.planetRadius + 85.0f * fminf(a.densityScaleRayleigh, a.densityScaleMie));
        a.innerCeil2 = iCeil * iCeil;
    }
    a.rcpH = rcpN(a.lean ? a.H : 1.0f);
    a.rcpDsR = rcpN(a.lean ? a.densityScaleRayleigh : 1.0f);
    a.rcpDsM = rcpN(a.lean ? a.densityScaleMie : 1.0f);
    a.rcp15 = rcpN(15.0f);
    return a;
}

// common.glinl:194-216. Mie absorption uses the Rayleigh coefficient (line 202).
struct Extinction
{
    V3 scatteringRayleigh;
    V3 scatteringMie;
    V3 extinction;
};
template <bool LEAN>
SZG_DEV Extinction extinctionFromDensities(const Atm& a, float altitude, float densityRayleigh, float densityMie, bool belowOzone);
template <bool LEAN = false, bool INNER = false> SZG_DEV Extinction sampleExtinction(const Atm& a, float altitude, bool belowOzone = false)
{
    // (the lean paths never see a NaN altitude: every radius along the ray is finite and above the lean floor)
    if (LEAN && INNER)
    {
        // radii in [lean floor, inner ceiling]: -altitude / H in [-85, 80] for both scale heights (Atm::innerCeil2)
        float const densityRayleigh = expInner(divR0(-altitude, a.densityScaleRayleigh, a.rcpDsR));
        float const densityMie = expInner(divR0(-altitude, a.densityScaleMie, a.rcpDsM));
        return extinctionFromDensities<true>(a, altitude, densityRayleigh, densityMie, belowOzone);
    }
    float const densityRayleigh = expX<LEAN>(divRX<LEAN>(-altitude, a.densityScaleRayleigh, a.rcpDsR));
    float const densityMie = expX<LEAN>(divRX<LEAN>(-altitude, a.densityScaleMie, a.rcpDsM));
    return extinctionFromDensities<LEAN>(a, altitude, densityRayleigh, densityMie, belowOzone);
}
// belowOzone (wave-uniform, MarchSetup): every sample of the ray lies below 9.9 km, so altitude * 1000 - 25 <= -15.09 and the
// test |h - 25 km| >= 15 km below holds at every step without being evaluated
template <bool LEAN>
SZG_DEV Extinction extinctionFromDensities(const Atm& a, float altitude, float densityRayleigh, float densityMie, bool belowOzone)
{
    V3 const scatteringRayleigh = a.scatteringRayleigh * densityRayleigh;
    V3 const absorptionRayleigh = a.absorptionRayleigh * densityRayleigh;
    V3 const scatteringMie = a.scatteringMie * densityMie;
    V3 const absorptionMie = a.absorptionRayleigh * densityMie;
    Extinction e;
    e.scatteringRayleigh = scatteringRayleigh;
    e.scatteringMie = scatteringMie;
    if (LEAN && belowOzone && a.signClearCoefficients)
    {
        e.extinction = a.zeroAbsorptionRayleigh ? (scatteringRayleigh + scatteringMie)
                                                : (((scatteringRayleigh + absorptionRayleigh) + scatteringMie) + absorptionMie);
        return e;
    }
    float const ozoneOffset = fabsf(altitude * 1000.0f - 25.0f);
    // Outside the ozone tent (|h - 25 km| >= 15 km) the density is max(0, 1 - q) with q >= 1, i.e. +0, and both ozone
    // products are +-0: adding them to a partial sum that is not -0 (sign-clear coefficients) changes nothing. When that holds for the whole wave (aerial-perspective marches
    // near the ground, high-altitude samples) the tent, its division and the two terms are skipped.
    bool const noOzone = LEAN && a.signClearCoefficients && waveAll(ozoneOffset >= 15.0f);
    if (noOzone)
    {
        e.extinction = a.zeroAbsorptionRayleigh ? (scatteringRayleigh + scatteringMie)
                                                : (((scatteringRayleigh + absorptionRayleigh) + scatteringMie) + absorptionMie);
        return e;
    }
    float const densityOzone = fmaxf(0.0f, 1.0f - divRX<LEAN>(ozoneOffset, 15.0f, a.rcp15));
    V3 const scatteringOzone = a.scatteringOzone * densityOzone;
    V3 const absorptionOzone = a.absorptionOzone * densityOzone;
    if (LEAN && a.zeroAbsorptionRayleigh && a.zeroScatteringOzone)
    {
        // + absorptionRayleigh, + absorptionMie (= absorptionRayleigh coefficient, Q1) and + scatteringOzone add exact +0
        e.extinction = (scatteringRayleigh + scatteringMie) + absorptionOzone;
    }
    else if (LEAN && a.zeroAbsorptionRayleigh)
    {
        e.extinction = ((scatteringRayleigh + scatteringMie) + scatteringOzone) + absorptionOzone;
    }
    else
    {
        e.extinction = scatteringRayleigh + absorptionRayleigh + scatteringMie + absorptionMie + scatteringOzone + absorptionOzone;
    }
    return e;
}

// common.glinl:220-260
SZG_DEV bool raySphere(V3 f, V3 d, float radius, float& t0, float& t1)
{
    float const b = -1.0f * dotA(f, d);
    V3 const chord = f + b * d;
    float const discriminant = radius * radius - dotA(chord, chord);
    float const c = dotA(f, f) - radius * radius;
    if (discriminant < 0.0f)
    {
        return false;
    }
    float q = b;
    float const s = sqrtf(discriminant);
    q = (b < 0.0f) ? (q - s) : (q + s);
    float a0 = c / q;
    float a1 = q;
    if (a0 > a1)
    {
        float const tmp = a0;
        a0 = a1;
        a1 = tmp;
    }
    t0 = a0;
    t1 = a1;
    return true;
}

// common.glinl:284-307
SZG_DEV float raycastAtmosphere(const Atm& a, V3 origin, V3 direction)
{
    float at0 = 0.0f, at1 = 0.0f;
    bool const hitAtmosphere = raySphere(origin, direction, a.atmosphereRadius, at0, at1) && at1 > 0.0f;
    at0 = fmaxf(0.0f, at0);
    float pt0 = 0.0f, pt1 = 0.0f;
    bool const hitPlanet = raySphere(origin, direction, a.planetRadius, pt0, pt1) && pt0 > 0.0f;
    if (hitPlanet)
    {
        at1 = fminf(pt0, at1);
    }
    return hitAtmosphere ? (at1 - at0) : 0.0f;
}

// ---------------------------------------------------------------------------
// Transmittance LUT sampler: LINEAR / CLAMP_TO_EDGE / fp32 weights
// (renderer/pipelines/skyview.cpp:199-207, :339-346)
// ---------------------------------------------------------------------------
struct TLut
{
    const float4* texels; // RGBA32F, row-major, pitch = width texels
    int width, height;
    float fwidth, fheight;
    // textureCoordFromUnitRange constants (common.glinl:29-32)
    float u_bias, u_scale, v_bias, v_scale;
    // every texel's rgb lies in [2^-50, 2] (status dword behind the texels, szg_launch.hpp "transmittance LUT block"):
    // any bilinear tap then lies in [2^-51, 2.01], inside the operand domain of the lean exact division
    bool moderate;
    // wave-uniform, from k_frame_prep (FramePrep::rowsInterior): for every radius of an INNER march (radius^2 <= Atm::innerCeil2,
    // with a margin of 2^-16) both LUT rows of a tap, floor(v) and floor(v) + 1, lie inside [0, H - 1] - the two clamps of
    // radiusPart are identities there and the row offsets can be formed in float
    bool rowsInterior;
#ifdef SZG_EXP_LDS_TLUT
    // EXPERIMENT (profiles/r03_experiments.md, north_star "LDS-staged ... LUT tiles"; not built into libszg_hip.so): the first
    // SZG_EXP_LDS_TLUT rows of the LUT as packed rgb in LDS, staged per workgroup by k_composite; taps whose two rows lie
    // inside (the aerial-perspective marches near the ground: rows 0..3 cover altitudes up to ~90 m) read them with ds_read_b96.
    const __attribute__((address_space(3))) float* ldsRows;
#endif
};
SZG_DEV bool tlutTexelModerate(float x, float y, float z)
{
    return inRange(x, 0x1p-50f, 2.0f) && inRange(y, 0x1p-50f, 2.0f) && inRange(z, 0x1p-50f, 2.0f);
}
SZG_DEV TLut make_tlut(const float4* texels, int w, int h)
{
    TLut t;
    t.moderate = reinterpret_cast<const unsigned*>(texels + (size_t)w * (size_t)h)[0] == 0u;
    t.texels = texels;
    t.width = w;
    t.height = h;
    t.fwidth = (float)w;
    t.fheight = (float)h;
    t.u_bias = 0.5f / (float)w;
    t.u_scale = 1.0f - 1.0f / (float)w;
    t.v_bias = 0.5f / (float)h;
    t.v_scale = 1.0f - 1.0f / (float)h;
    t.rowsInterior = false;
#ifdef SZG_EXP_LDS_TLUT
    t.ldsRows = nullptr;
#endif
    return t;
}

// Per-frame constants (k_frame_prep, one lane, launched in front of the composite and the sky-view LUT kernel): everything
// load_atm() and make_tlut() derive from the atmosphere block and the LUT extent - ~300 instructions that every wave of
// those kernels would otherwise repeat - computed once and read back through scalar loads. Same code, same values.
struct FramePrep
{
    Atm a;
    float fwidth, fheight, u_bias, u_scale, v_bias, v_scale;
    // camera.comp:366-369: TO_TEX_COORD_MAT * (sun.projection * sun.view), the shadow frame's matrix of the composite - two 4x4
    // products that are the same for every pixel (round 2 formed them per lane, ~300 VALU instructions per geometry pixel when a
    // sun shadow map is bound). Same mul(), same order; read back through scalar loads. Valid when the composite has a sun map.
    float sunShadow[16];
    // TLut::rowsInterior (k_frame_prep evaluates the LUT's v coordinate at both ends of the INNER radius range: the map is
    // monotone, every operation in it being correctly rounded)
    unsigned rowsInterior;
};
// The block arrives through scalar loads; its floats then move to vector registers, where the per-wave derivation used to
// leave them: a VALU instruction of gfx950 reads at most one scalar operand, and ~70 constants held in scalar registers
// through the march loops spill (v_writelane / v_readlane inside the loops). The flags stay scalar: they steer wave-uniform
// branches.
SZG_DEV float inVector(float x)
{
    asm volatile("" : "+v"(x));
    return x;
}
SZG_DEV V3 inVector(V3 v) { return V3{inVector(v.x), inVector(v.y), inVector(v.z)}; }
SZG_DEV Atm load_atm(const FramePrep& f)
{
    Atm a = f.a;
    a.scatteringRayleigh = inVector(a.scatteringRayleigh);
    a.densityScaleRayleigh = inVector(a.densityScaleRayleigh);
    a.absorptionRayleigh = inVector(a.absorptionRayleigh);
    a.planetRadius = inVector(a.planetRadius);
    a.scatteringMie = inVector(a.scatteringMie);
    a.densityScaleMie = inVector(a.densityScaleMie);
    a.atmosphereRadius = inVector(a.atmosphereRadius);
    a.incidentDirectionSun = inVector(a.incidentDirectionSun);
    a.scatteringOzone = inVector(a.scatteringOzone);
    a.absorptionOzone = inVector(a.absorptionOzone);
    a.sunIntensitySpectrum = inVector(a.sunIntensitySpectrum);
    a.sunAngularRadius = inVector(a.sunAngularRadius);
    a.Ra2 = inVector(a.Ra2);
    a.Rp2 = inVector(a.Rp2);
    a.H = inVector(a.H);
    a.rcpH = inVector(a.rcpH);
    a.rcpDsR = inVector(a.rcpDsR);
    a.rcpDsM = inVector(a.rcpDsM);
    a.rcp15 = inVector(a.rcp15);
    a.leanFloor2 = inVector(a.leanFloor2);
    a.extFloor2 = inVector(a.extFloor2);
    a.extCeil2 = inVector(a.extCeil2);
    a.innerCeil2 = inVector(a.innerCeil2);
    return a;
}
SZG_DEV TLut make_tlut(const float4* texels, int w, int h, const FramePrep& f)
{
    TLut t;
    t.moderate = reinterpret_cast<const unsigned*>(texels + (size_t)w * (size_t)h)[0] == 0u;
    t.texels = texels;
    t.width = w;
    t.height = h;
    t.fwidth = inVector(f.fwidth);
    t.fheight = inVector(f.fheight);
    t.u_bias = inVector(f.u_bias);
    t.u_scale = inVector(f.u_scale);
    t.v_bias = inVector(f.v_bias);
    t.v_scale = inVector(f.v_scale);
    t.rowsInterior = f.rowsInterior != 0u;
#ifdef SZG_EXP_LDS_TLUT
    t.ldsRows = nullptr;
#endif
    return t;
}

SZG_DEV V3 bilinear_rgb(const float4* __restrict__ texels, int W, int H, float fW, float fH, float s, float t)
{
    float const u = SZG_CON(SZG_C_TEXCOORD, s, fW, -0.5f);
    float const v = SZG_CON(SZG_C_TEXCOORD, t, fH, -0.5f);
    float const fu = floorf(u);
    float const fv = floorf(v);
    float const a = u - fu;
    float const b = v - fv;
    int i0 = (int)fu, j0 = (int)fv;
    int i1 = i0 + 1, j1 = j0 + 1;
    i0 = min(max(i0, 0), W - 1);
    i1 = min(max(i1, 0), W - 1);
    j0 = min(max(j0, 0), H - 1);
    j1 = min(max(j1, 0), H - 1);
    float4 const t00 = texels[j0 * W + i0];
    float4 const t10 = texels[j0 * W + i1];
    float4 const t01 = texels[j1 * W + i0];
    float4 const t11 = texels[j1 * W + i1];
    float const w00 = (1.0f - a) * (1.0f - b);
    float const w10 = a * (1.0f - b);
    float const w01 = (1.0f - a) * b;
    float const w11 = a * b;
    V3 r;
    r.x = SZG_CON(SZG_C_BILINEAR, w11, t11.x, SZG_CON(SZG_C_BILINEAR, w01, t01.x, SZG_CON(SZG_C_BILINEAR, w10, t10.x, w00 * t00.x)));
    r.y = SZG_CON(SZG_C_BILINEAR, w11, t11.y, SZG_CON(SZG_C_BILINEAR, w01, t01.y, SZG_CON(SZG_C_BILINEAR, w10, t10.y, w00 * t00.y)));
    r.z = SZG_CON(SZG_C_BILINEAR, w11, t11.z, SZG_CON(SZG_C_BILINEAR, w01, t01.z, SZG_CON(SZG_C_BILINEAR, w10, t10.z, w00 * t00.z)));
    return r;
}

// transmittanceLUT_RMu_to_UV (common.glinl:40-66) + the bilinear fetch, split into the
// part that depends only on the radius (v coordinate, row pair and row weights) and the
// part that depends on mu. Two taps at the same radius share the first part; every
// operation and its order are those of the unsplit evaluation.
struct Rgb
{
    float x, y, z;
};
struct RadiusPart
{
    float r, r2;        // radius, radius * radius
    float d_min, denom; // atmosphereRadius - radius, (rho + H) - d_min
    float rcpDenom;     // rcpN(denom) when LEAN
    unsigned row0;      // texel index of the start of rows j0, j1 (clamped); 32-bit so that the taps use
    unsigned row1;      // SGPR-base + VGPR-offset addressing instead of 64-bit VALU address arithmetic
    float b, omb;       // v weight and 1 - b
};
template <bool LEAN = false, bool INNER = false> SZG_DEV RadiusPart radiusPart(const TLut& L, const Atm& a, float radius)
{
    RadiusPart p;
    p.r = radius;
    p.r2 = radius * radius;
    float const rho = safeSqrtX<LEAN>(p.r2 - a.Rp2);
    p.d_min = a.atmosphereRadius - radius;
    float const d_max = rho + a.H;
    p.denom = d_max - p.d_min;
    p.rcpDenom = LEAN ? rcpN(p.denom) : 0.0f;
    float const x_radius = divRX<LEAN>(rho, a.H, a.rcpH);
    float const t = SZG_CON(SZG_C_LUTMAP, x_radius, L.v_scale, L.v_bias);
    float const v = SZG_CON(SZG_C_TEXCOORD, t, L.fheight, -0.5f);
    float const fv = floorf(v);
    p.b = v - fv;
    p.omb = 1.0f - p.b;
    // clamp((int)fv, 0, H-1) and clamp((int)fv + 1, 0, H-1) with the clamp done in float by v_med3_f32 (one instruction
    // instead of two integer ones per index): the same indices for every finite fv (integer-valued; beyond 2^24
    // fv + 1 == fv and both land on the same edge, as the saturating conversion does). For a NaN fv the weights are NaN
    // and so is the result, whichever texels are fetched.
    if (LEAN && INNER && L.rowsInterior)
    {
        // 0 <= fv <= H - 2 for every radius of this march (TLut::rowsInterior): the clamped indices are fv and fv + 1 themselves,
        // and fv * W is an exact float product below 2^24 (the flag also says W * H <= 2^24): one conversion, one full-rate
        // multiply and one add instead of two clamps, two conversions and two integer multiplies. (A NaN fv converts to row 0
        // and has NaN weights: a NaN tap whichever texels are fetched, as above.)
        p.row0 = (unsigned)(int)(fv * L.fwidth);
        p.row1 = p.row0 + (unsigned)L.width;
        return p;
    }
    float const hm1 = L.fheight - 1.0f;
    int const j0 = (int)__builtin_amdgcn_fmed3f(fv, 0.0f, hm1);
    int const j1 = (int)__builtin_amdgcn_fmed3f(fv + 1.0f, 0.0f, hm1);
    p.row0 = (unsigned)(j0 * L.width);
    p.row1 = (unsigned)(j1 * L.width);
    return p;
}
template <bool LEAN = false, bool INNER = false> SZG_DEV V3 sampleT_at(const TLut& L, const Atm& a, const RadiusPart& p, float mu)
{
    // INNER (radius^2 <= Atm::innerCeil2): for a mu that is a number the discriminant is >= 2^-10 Ra^2 against rounding
    // errors of 2^-22 relative, so max(., 0) inside safeSqrt and sqrtN's own guard never act. The clamp of d STAYS: mu is NaN
    // where a march segment has length 0 (geometry nearer than 32 ulps of the planet radius, ~15 m: normalize(0) = NaN,
    // common.glinl:114-136), and max(NaN, 0) = 0 is what the reference then samples with (found by the 6 000-seed sweep of
    // round 2: a version without this clamp let the NaN through to the texel weights in 8 of 6 000 random frames).
    float const disc = SZG_CON(SZG_C_LUTDIST, p.r2, SZG_CON(SZG_C_LUTDIST, mu, mu, -1.0f), a.Ra2);
    float const d = fmaxf(SZG_CON(SZG_C_LUTDIST, -p.r, mu, (LEAN && INNER) ? sqrtP(disc) : safeSqrtX<LEAN>(disc)), 0.0f);
    float const x_mu = divRX<LEAN>(d - p.d_min, p.denom, p.rcpDenom);
    float const s = SZG_CON(SZG_C_LUTMAP, x_mu, L.u_scale, L.u_bias);
    float const u = SZG_CON(SZG_C_TEXCOORD, s, L.fwidth, -0.5f);
    float const fu = floorf(u);
    float const al = u - fu;
    float const wm1 = L.fwidth - 1.0f;
    // only .rgb is consumed (common.glinl:111, :142): 12-byte loads keep 16 VGPRs per step out of flight
    const char* const base = reinterpret_cast<const char*>(L.texels);
    unsigned o00, o10, o01, o11; // byte offsets of the four texels
    // (0 <= fu <= W - 2 as ONE unsigned compare of the converted index: a negative fu converts to a negative int = a huge
    // unsigned, an fu beyond int range saturates to INT_MAX, and a NaN fu, which converts to 0, has NaN weights and gives
    // a NaN texel average on either path)
    unsigned const iu = (unsigned)(int)fu;
    if (LEAN && INNER && waveAll(iu <= (unsigned)(L.width - 2)))
    {
        // interior column for the whole wave (the rule; the clamps only act for rays that point into the ground, x_mu > 1, or
        // on a rounding below 0): i0 = fu and i1 = fu + 1 are the unclamped indices and the right-hand texel of each row is
        // the next 16 bytes: one conversion and two full-rate adds instead of two clamps, two conversions and two shifts
        unsigned const i0 = iu;
        o00 = (p.row0 + i0) << 4;
        o01 = (p.row1 + i0) << 4;
        o10 = o00 + 16u;
        o11 = o01 + 16u;
    }
    else
    {
        int const i0 = (int)__builtin_amdgcn_fmed3f(fu, 0.0f, wm1); // see radiusPart
        int const i1 = (int)__builtin_amdgcn_fmed3f(fu + 1.0f, 0.0f, wm1);
        o00 = (p.row0 + (unsigned)i0) << 4;
        o10 = (p.row0 + (unsigned)i1) << 4;
        o01 = (p.row1 + (unsigned)i0) << 4;
        o11 = (p.row1 + (unsigned)i1) << 4;
    }
    Rgb t00, t10, t01, t11;
#ifdef SZG_EXP_LDS_TLUT
    if (LEAN && INNER && L.ldsRows != nullptr && waveAll(p.row1 < (unsigned)(SZG_EXP_LDS_TLUT * L.width) && p.row0 < (unsigned)(SZG_EXP_LDS_TLUT * L.width)))
    {
        // byte offset of texel k in the global LUT is 16 k, in the packed LDS copy 12 k
        // (explicit LDS address space: through a generic pointer these were flat loads, +84 % on the composite)
        const __attribute__((address_space(3))) float* const q00 = L.ldsRows + (o00 >> 4) * 3u;
        const __attribute__((address_space(3))) float* const q10 = L.ldsRows + (o10 >> 4) * 3u;
        const __attribute__((address_space(3))) float* const q01 = L.ldsRows + (o01 >> 4) * 3u;
        const __attribute__((address_space(3))) float* const q11 = L.ldsRows + (o11 >> 4) * 3u;
        t00 = Rgb{q00[0], q00[1], q00[2]};
        t10 = Rgb{q10[0], q10[1], q10[2]};
        t01 = Rgb{q01[0], q01[1], q01[2]};
        t11 = Rgb{q11[0], q11[1], q11[2]};
    }
    else
#endif
    {
        t00 = *reinterpret_cast<const Rgb*>(base + o00);
        t10 = *reinterpret_cast<const Rgb*>(base + o10);
        t01 = *reinterpret_cast<const Rgb*>(base + o01);
        t11 = *reinterpret_cast<const Rgb*>(base + o11);
    }
    float const oma = 1.0f - al;
    float const w00 = oma * p.omb;
    float const w10 = al * p.omb;
    float const w01 = oma * p.b;
    float const w11 = al * p.b;
    V3 r;
    r.x = SZG_CON(SZG_C_BILINEAR, w11, t11.x, SZG_CON(SZG_C_BILINEAR, w01, t01.x, SZG_CON(SZG_C_BILINEAR, w10, t10.x, w00 * t00.x)));
    r.y = SZG_CON(SZG_C_BILINEAR, w11, t11.y, SZG_CON(SZG_C_BILINEAR, w01, t01.y, SZG_CON(SZG_C_BILINEAR, w10, t10.y, w00 * t00.y)));
    r.z = SZG_CON(SZG_C_BILINEAR, w11, t11.z, SZG_CON(SZG_C_BILINEAR, w01, t01.z, SZG_CON(SZG_C_BILINEAR, w10, t10.z, w00 * t00.z)));
    return r;
}

// common.glinl:40-66 + :138-143 : sample at (radius, mu)
template <bool LEAN = false> SZG_DEV V3 sampleT_RadiusMu(const TLut& L, const Atm& a, float radius, float mu)
{
    RadiusPart const p = radiusPart<LEAN>(L, a, radius);
    return sampleT_at<LEAN>(L, a, p, mu);
}

// Wave-uniform precondition of the LEAN forms of the samplers below, given the squared radii / lengths they take square roots
// and quotients of: the atmosphere admits the lean ops (Atm::lean), every radius is above the lean floor and every
// length a normal number of moderate size.
SZG_DEV bool leanLength2(float x) { return inRange(x, 0x1p-60f, 0x1p60f); }
SZG_DEV bool leanRadius2(const Atm& a, float r2) { return r2 >= a.leanFloor2 && r2 <= 0x1p60f; }

// common.glinl:104-112
template <bool LEAN = false> SZG_DEV V3 sampleT_Ray(const TLut& L, const Atm& a, V3 position, V3 direction)
{
    if (LEAN)
    {
        // lengthA() = sqrt(dot) and the quotient, with the lean exact operators (same values)
        float const radius = sqrtP(dotA(position, position));
        float const mu = divN0(dotA(position, direction), radius * sqrtP(dotA(direction, direction)));
        return sampleT_RadiusMu<true>(L, a, radius, mu);
    }
    float const radius = lengthA(position);
    float const mu = dotA(position, direction) / (lengthA(position) * lengthA(direction));
    return sampleT_RadiusMu(L, a, radius, mu);
}

// common.glinl:114-136. The flipped branch samples with -direction, whose mu is the exact
// negation of the unflipped one (negation commutes with every rounding), so one code path
// with a sign select reproduces both branches.
template <bool LEAN = false, bool INNER = false>
SZG_DEV V3 segmentRatio(const TLut& L, const Atm& a, const RadiusPart& pFrom, float fromDotDir, float lenFrom,
                        const RadiusPart& pTo, float toDotDir, float lenTo, float lenDir)
{
    bool const flip = fromDotDir < 0.0f;
    float const muFrom = divX<LEAN>(fromDotDir, lenFrom * lenDir);
    float const muTo = divX<LEAN>(toDotDir, lenTo * lenDir);
    // flip ? -mu : mu as a sign-bit XOR (one VALU op instead of a compare/select pair through VCC); for the
    // ratio, numerator and denominator are swapped by selecting the operands once rather than two quotients
    unsigned const signFlip = flip ? 0x80000000u : 0u;
    V3 const Tf = sampleT_at<LEAN, INNER>(L, a, pFrom, xorSign(muFrom, signFlip));
    V3 const Tt = sampleT_at<LEAN, INNER>(L, a, pTo, xorSign(muTo, signFlip));
    if (LEAN && L.moderate)
    {
        // both taps lie in [2^-51, 2.01]; the quotient is clamped to [0, 1] and then consumed sign-blind
        V3 const ql = flip ? V3{divN0(Tt.x, Tf.x), divN0(Tt.y, Tf.y), divN0(Tt.z, Tf.z)}
                           : V3{divN0(Tf.x, Tt.x), divN0(Tf.y, Tt.y), divN0(Tf.z, Tt.z)};
        return clamp01(ql);
    }
    V3 const q = flip ? (Tt / Tf) : (Tf / Tt);
    return clamp01(q);
}
template <bool LEAN = false> SZG_DEV V3 sampleT_Segment(const TLut& L, const Atm& a, V3 from, V3 to)
{
    if (LEAN)
    {
        V3 const segment = to - from;
        V3 const direction = segment * divN0(1.0f, sqrtP(dotA(segment, segment))); // normalizeA()
        float const lenFrom = sqrtP(dotA(from, from));
        float const lenTo = sqrtP(dotA(to, to));
        RadiusPart const pf = radiusPart<true>(L, a, lenFrom);
        RadiusPart const pt = radiusPart<true>(L, a, lenTo);
        return segmentRatio<true>(L, a, pf, dotA(from, direction), lenFrom, pt, dotA(to, direction), lenTo, sqrtP(dotA(direction, direction)));
    }
    V3 const direction = normalizeA(to - from);
    float const lenFrom = lengthA(from);
    float const lenTo = lengthA(to);
    RadiusPart const pf = radiusPart<false>(L, a, lenFrom);
    RadiusPart const pt = radiusPart<false>(L, a, lenTo);
    return segmentRatio<false>(L, a, pf, dotA(from, direction), lenFrom, pt, dotA(to, direction), lenTo, lengthA(direction));
}

// common.glinl:263-279
SZG_DEV float phaseRayleigh(float cosine)
{
    float const scalar = 3.0f / (16.0f * 3.141592653589793f);
    return scalar * (1.0f + cosine * cosine);
}
SZG_DEV float phaseMie(float cosine, float g)
{
    float const scalar = 3.0f / (8.0f * 3.141592653589793f);
    float const numerator = (1.0f - g * g) * (1.0f + cosine * cosine);
    float const denominator = (2.0f + g * g) * szg_powf(1.0f + g * g - 2.0f * g * cosine, 1.5f);
    return scalar * numerator / denominator;
}

// common.glinl:364-424 computeLuminanceScatteringIntegral, 32 steps.
// Hoisted (values identical in every iteration of the GLSL loop):
//   phase functions (line 408-410), sun-disc sin/cos (147-148), the origin-side
//   LUT tap and mu_sunAndStepDirection of sampleTransmittanceLUT_RayMarchStep /
//   stepRadiusMu (325, 349/357). stepRadiusMu(originStep, t) is evaluated once
//   per step and used for both sampleStep (389) and `end` (343).
struct MarchSetup
{
    V3 origin, scatteringDir;
    float dS, pR, pM, sin_sunRadius, cos_sunRadius;
    float mu_sunAndStep, r_mu, two_r_mu, r2, r_musun;
    bool up;
    bool extLean; // wave-uniform: the extinction along this wave's rays is of moderate magnitude (Atm::extModerate)
    // wave-uniform: at EVERY sample of this wave's rays the sun disc lies entirely above the local horizon, decided once per
    // ray from bounds (scatteringIntegral); the horizon smoothstep of sampleTransmittanceLUT_Sun is then exactly 1 at every
    // step and its operands (sin / cos of the horizon angle, the two edges) need not be formed
    bool sunClear;
    // wave-uniform: every ray of the wave has a step dS >= 1e-7, so t = i * dS < 1e-7 (common.glinl:338) holds at step 0 and at
    // no other step: the per-step comparison and its select are not needed
    bool firstStepOnly;
    // wave-uniform: every radius of the wave's rays is below Rp + 9.9 km (sampleExtinction's ozone tent is 0 there)
    bool belowOzone;
    V3 T_origin;
};

// INNER: every radius of the march lies below Atm::innerCeil2 (wave-uniform, decided per ray in scatteringIntegral)
template <bool LEAN, bool INNER = false> SZG_DEV V3 marchLoop(const TLut& L, const Atm& a, const MarchSetup& m)
{
    V3 luminance = splat(0.0f);
    // `end` of step i and `begin` of step i+1 are the same expression (common.glinl:386-387),
    // so its length and radius part are carried from one iteration to the next.
    V3 begin = fnma(0.0f * m.dS, m.scatteringDir, m.origin);
    float lenBegin = sqrtPX<LEAN>(dotA(begin, begin));
    RadiusPart pBegin = radiusPart<LEAN, INNER>(L, a, lenBegin);
#pragma unroll 1
    for (unsigned i = 0; i < 32u; i++)
    {
        float const fi = (float)i;
        float const t = fi * m.dS;
        V3 const end = fnma((float)(i + 1u) * m.dS, m.scatteringDir, m.origin);
        float const lenEnd = sqrtPX<LEAN>(dotA(end, end));
        RadiusPart const pEnd = radiusPart<LEAN, INNER>(L, a, lenEnd);

        // stepRadiusMu(originStep, t), common.glinl:329-331
        // (on a lean path this is a squared radius above the lean floor: neither the clamp to 0 nor sqrtN's guard is needed)
        float const s_q = SZG_CON(SZG_C_STEP, m.two_r_mu, t, t * t) + m.r2;
        float const s_radius = LEAN ? sqrtP(s_q) : safeSqrt(s_q);
        float const yS = LEAN ? rcpN(s_radius) : 0.0f;
        float const s_mu = divRX<LEAN>(m.r_mu + t, s_radius, yS);
        float const s_musun = divRX<LEAN>(SZG_CON(SZG_C_STEP, t, m.mu_sunAndStep, m.r_musun), s_radius, yS);
        RadiusPart const pStep = radiusPart<LEAN, INNER>(L, a, s_radius);

        float const altitude = lenBegin - a.planetRadius;

        // sampleTransmittanceLUT_Sun, common.glinl:145-172
        V3 const T_atm = sampleT_at<LEAN, INNER>(L, a, pStep, s_musun);
        V3 T_sun;
        if (LEAN && m.sunClear)
        {
            // (the per-step test below would succeed at every step: MarchSetup::sunClear)
            T_sun = T_atm;
        }
        else
        {
            float const sin_hz = divRX<LEAN>(a.planetRadius, s_radius, yS);
            float const cos_hz = -safeSqrtX<LEAN>(1.0f - sin_hz * sin_hz);
            float const e0 = -sin_hz * m.sin_sunRadius;
            float const e1 = sin_hz * m.sin_sunRadius;
            float const ssNum = (s_musun - cos_hz * m.cos_sunRadius) - e0, ssDen = e1 - e0;
            if (LEAN && waveAll(ssNum >= ssDen && ssDen > 0.0f))
            {
                // the sun disc is entirely above the sample's horizon for the whole wave: num / den >= 1 clamps to 1,
                // smoothstep(1) = 1 * 1 * (3 - 2) = 1 and T_atm * 1 = T_atm, all exactly
                T_sun = T_atm;
            }
            else
            {
                float const ss = clampf(divX<LEAN>(ssNum, ssDen), 0.0f, 1.0f);
                float const angularFactor = ss * ss * (3.0f - 2.0f * ss);
                T_sun = T_atm * angularFactor;
            }
        }

        Extinction const ex = sampleExtinction<LEAN, INNER>(a, altitude, LEAN && m.belowOzone);

        // sampleTransmittanceLUT_RayMarchStep, common.glinl:336-361
        // (t < 1e-7 -> 1, common.glinl:338-341: true for the whole wave at step 0, where the tap and the quotients are skipped)
        V3 T_begin = splat(1.0f);
        if ((LEAN && m.firstStepOnly) ? (i != 0u) : !waveAll(t < 0.0000001f))
        {
            V3 const T_end = sampleT_at<LEAN, INNER>(L, a, pStep, xorSign(s_mu, m.up ? 0u : 0x80000000u));
            V3 ratio;
            if (LEAN && L.moderate)
            {
                ratio = clamp01(m.up ? V3{divN0(m.T_origin.x, T_end.x), divN0(m.T_origin.y, T_end.y), divN0(m.T_origin.z, T_end.z)}
                                     : V3{divN0(T_end.x, m.T_origin.x), divN0(T_end.y, m.T_origin.y), divN0(T_end.z, m.T_origin.z)});
            }
            else
            {
                ratio = clamp01(m.up ? (m.T_origin / T_end) : (T_end / m.T_origin));
            }
            // (only a wave with lanes on both sides of the threshold needs the per-lane select)
            T_begin = ((LEAN && m.firstStepOnly) || waveAll(!(t < 0.0000001f))) ? ratio : ((t < 0.0000001f) ? splat(1.0f) : ratio);
        }

        V3 const phaseTimesScattering = fma3(ex.scatteringMie, m.pM, ex.scatteringRayleigh * m.pR);

        // sampleTransmittanceLUT_Segment(begin, end), common.glinl:114-136. The segment can be arbitrarily short
        // (geometry close to the camera: end - begin may even be 0 and its normal NaN), so normalizeA() keeps the
        // generic operators unless the squared length is comfortably normal for the whole wave.
        V3 const segment = end - begin;
        float const segment2 = dotA(segment, segment);
        V3 segDir;
        if (LEAN && waveAll(inRange(segment2, 0x1p-90f, 0x1p60f)))
        {
            segDir = segment * divN0(1.0f, sqrtP(segment2)); // = segment * (1 / sqrt(dot)) of normalizeA()
        }
        else
        {
            segDir = normalizeA(segment);
        }
        V3 const T_path = segmentRatio<LEAN, INNER>(L, a, pBegin, dotA(begin, segDir), lenBegin, pEnd, dotA(end, segDir), lenEnd,
                                             sqrtPX<LEAN>(dotA(segDir, segDir)));
        // 1 - T_path is 0 or a multiple of 2^-24; the extinction is in [2^-40, 2^52] when m.extLean
        V3 const oneMinusT = splat(1.0f) - T_path;
        V3 const integral = (LEAN && m.extLean) ? V3{divN0(oneMinusT.x, ex.extinction.x), divN0(oneMinusT.y, ex.extinction.y),
                                                     divN0(oneMinusT.z, ex.extinction.z)}
                                                : oneMinusT / ex.extinction;
        luminance = fma3(phaseTimesScattering * T_sun * integral, T_begin, luminance);

        begin = end;
        lenBegin = lenEnd;
        pBegin = pEnd;
    }
    return luminance;
}

SZG_DEV V3 scatteringIntegral(const TLut& L, const Atm& a, V3 origin, V3 direction, float sampleDistance)
{
    MarchSetup m;
    m.origin = origin;
    // The per-ray setup with the lean exact operators (same values) when the whole wave's origins lie above the lean floor
    // and the direction and sun vectors have ordinary lengths; otherwise hipcc's generic sqrtf / division.
    float const origin2 = dotA(origin, origin), direction2 = dotA(direction, direction);
    float const sun2 = dotA(a.incidentDirectionSun, a.incidentDirectionSun);
    bool const setupLean = waveAll(a.lean && leanRadius2(a, origin2) && leanLength2(direction2) && leanLength2(sun2));
    float radius, mu, mu_sun;
    V3 const toSun = -a.incidentDirectionSun;
    if (setupLean)
    {
        float const lenDirection = sqrtP(direction2);
        m.scatteringDir = -(direction * divN0(1.0f, lenDirection));
        radius = sqrtP(origin2);
        mu = divN0(dotA(origin, direction), radius * lenDirection);
        mu_sun = divN0(dotA(origin, toSun), radius * sqrtP(sun2));
    }
    else
    {
        m.scatteringDir = -normalizeA(direction);
        radius = lengthA(origin);
        mu = dotA(origin, direction) / (lengthA(origin) * lengthA(direction));
        mu_sun = dotA(origin, toSun) / (lengthA(origin) * lengthA(a.incidentDirectionSun));
    }

    float const incidentCosine = dotA(a.incidentDirectionSun, m.scatteringDir);
    m.pR = phaseRayleigh(incidentCosine);
    m.pM = phaseMie(incidentCosine, 0.8f);
    m.sin_sunRadius = szg_sinf(a.sunAngularRadius);
    m.cos_sunRadius = szg_cosf(a.sunAngularRadius);

    // stepRadiusMu invariants (common.glinl:325)
    m.mu_sunAndStep = safeSqrt(mu_sun * mu - safeSqrt((1.0f - mu_sun * mu_sun) * (1.0f - mu * mu)));
    m.r_mu = radius * mu;
    m.two_r_mu = 2.0f * radius * mu;
    m.r2 = radius * radius;
    m.r_musun = radius * mu_sun;
    m.up = mu > 0.0f;
    m.T_origin = setupLean ? sampleT_RadiusMu<true>(L, a, radius, m.up ? mu : -mu) : sampleT_RadiusMu<false>(L, a, radius, m.up ? mu : -mu);
    m.dS = sampleDistance / 32.0f;

    // leanRay: every radius met along the path stays >= 0.9 Rp and of moderate magnitude, the path length is
    // moderate, and the smoothstep span 2 * sin_hz * sin(sunRadius) is a normal number. The closest approach of
    // the segment [0, L] to the planet centre is at t* = -r*mu when that lies inside the segment.
    float const L2 = SZG_CON(SZG_C_STEP, m.two_r_mu, sampleDistance, sampleDistance * sampleDistance) + m.r2; // stepRadiusMu's form
    float const tStar = -m.r_mu;
    float const rmin2 = (tStar > 0.0f && tStar < sampleDistance) ? m.r2 * (1.0f - mu * mu) : fminf(m.r2, L2);
    // ... and the two cosines are cosines: a sun (or view) vector of length 0 or inf makes mu_sun (mu) infinite or NaN, and
    // the one-correction division returns NaN for an infinite numerator where the quotient is inf.
    bool const lean = a.lean && rmin2 >= a.leanFloor2 && inRange(radius, 0x1p-30f, 0x1p30f) && inRange(sampleDistance, 0.0f, 0x1p30f) &&
                      m.sin_sunRadius >= 0x1p-30f && leanLength2(sun2) && leanLength2(direction2) && fabsf(mu) <= 2.0f &&
                      fabsf(mu_sun) <= 2.0f;
    m.extLean = waveAll(a.extModerate && rmin2 >= a.extFloor2 && fmaxf(m.r2, L2) <= a.extCeil2);
    m.firstStepOnly = waveAll(lean && m.dS >= 0.0000001f); // then fl(i * dS) >= dS >= 1e-7 for every i >= 1
    {
        float const ozoneLow = a.planetRadius + 0.0099f; // Mm: below the 10 km foot of the tent (25 km - 15 km) with a margin
        m.belowOzone = waveAll(lean && fmaxf(m.r2, L2) <= ozoneLow * ozoneLow);
    }
    // sunClear. At step i the sun cosine handed to the smoothstep is s_musun = (r_musun + t * mu_sunAndStep) / s_radius with
    // t >= 0 and mu_sunAndStep >= 0 (a safeSqrt), so its numerator is >= r_musun, and s_radius is the radius of a point of
    // the segment: <= sqrt(max(r2, L2)) up to rounding. The test passes when s_musun >= cos_hz cos R + sin_hz sin R, and with
    // cos_hz <= 0 <= cos R the right-hand side is at most sin_hz sin R <= (Rp / lean floor) sin R <= 1.12 sin R. Hence
    // r_musun * 0.999 / r_max >= 1.25 sin R + 2^-10 implies ssNum >= ssDen > 0 at every step, with a margin (2^-10) a
    // thousand times the rounding errors of the per-step expressions (which are never formed then).
    {
        float const rMax = sqrtf(fmaxf(m.r2, L2));
        m.sunClear = waveAll(lean && m.cos_sunRadius >= 0.0f && m.sin_sunRadius > 0.0f && m.r_musun > 0.0f &&
                             m.r_musun * 0.999f >= (1.25f * m.sin_sunRadius + 0x1p-10f) * rMax);
    }
    // wave-uniform choice: one lane outside the domain sends its whole wave down the generic path
    if (waveAll(lean))
    {
        // the largest radius of the march is at one of its ends (|origin - t dir|^2 is convex in t); the cosines handed to
        // the taps are quotients bounded by Cauchy-Schwarz up to rounding (|mu| <= 1 + 2^-20 is all INNER asks for)
        if (waveAll(fmaxf(m.r2, L2) <= a.innerCeil2))
        {
            return marchLoop<true, true>(L, a, m);
        }
        return marchLoop<true, false>(L, a, m);
    }
    return marchLoop<false>(L, a, m);
}

// ---------------------------------------------------------------------------
// PBR (gbuffer/pbrFunctions.glinl) and shadow maps (shadowmap.glinl)
// ---------------------------------------------------------------------------
struct Material
{
    V3 position;
    V3 normal;
    V3 subscattering;
    V3 reflectance;
    float occlusion;
    float specularPower;
    float metallic;
    // per-pixel invariants of the BRDF: diffuseBRDF() = subscattering / pi (pbrFunctions.glinl:38) and the
    // Blinn-Phong normalisation (specularPower + 2) / 8 (pbrFunctions.glinl:49)
    V3 diffuse;
    float normalization;
};

// pbrFunctions.glinl:3-20
SZG_DEV Material convertPBR(V4 position, V4 normal, V4 diffuse, V4 specular, V4 orm)
{
    Material m;
    V3 const spec = mk3(specular.x, specular.y, specular.z);
    float const mx = fmaxf(fmaxf(spec.x, spec.y), spec.z);
    V3 const metallicReflectance = (splat(0.5f) * spec) / mx;
    m.position = mk3(position.x, position.y, position.z);
    m.normal = mk3(normal.x, normal.y, normal.z);
    m.subscattering = mk3(diffuse.x, diffuse.y, diffuse.z);
    m.metallic = orm.z;
    m.reflectance = mix(splat(0.04f), metallicReflectance, splat(m.metallic));
    m.occlusion = orm.x;
    m.specularPower = szg_powf(160.0f, 1.0f - orm.y);
    m.diffuse = m.subscattering / 3.14159265359f;
    m.normalization = (m.specularPower + 2.0f) / 8.0f;
    return m;
}

// computeLightContribution's BRDF part (lights.comp:93-108 / camera.comp:258-271):
// mix(diffuseBRDF, specularBRDF, fresnel) for outgoing light/view directions. LEAN: the caller has checked
// that |lightDir + viewDir|^2 >= 2^-40 (normalisation with the lean exact ops).
template <bool LEAN = false> SZG_DEV V3 brdfMix(const Material& m, V3 lightDir, V3 viewDir)
{
    V3 const hs = lightDir + viewDir;
    float const hd = dotP(hs, hs);
    V3 const h = hs * divX<LEAN>(1.0f, sqrtX<LEAN>(hd));
    float const microfacet = szg_powf(clampf(dotP(h, m.normal), 0.0f, 1.0f), m.specularPower);
    V3 const specular = splat(m.normalization * microfacet);
    float const p = szg_powf(1.0f - clampf(dotP(h, lightDir), 0.0f, 1.0f), 5.0f);
    V3 const fresnel = m.reflectance + (splat(1.0f) - m.reflectance) * p;
    return mix(m.diffuse, specular, fresnel);
}

// pbrFunctions.glinl:22-32
SZG_DEV V3 computeFresnel(const Material& m, V3 lightOutgoing, V3 viewOutgoing)
{
    V3 const h = normalizeP(lightOutgoing + viewOutgoing);
    float const p = szg_powf(1.0f - clampf(dotP(h, lightOutgoing), 0.0f, 1.0f), 5.0f);
    return m.reflectance + (splat(1.0f) - m.reflectance) * p;
}

// shadowmap.glinl:32-64 with NEAREST / CLAMP_TO_BORDER(0) (shadowpass.cpp:29-35).
// `coord` is shadowCoord / w, (ndx, ndy) the two sqrt terms of computeShadowFrame.
SZG_DEV float sampleShadowMap(const float* __restrict__ map, unsigned W, unsigned H, unsigned pitchFloats, V3 coord, float fdx,
                              float fdy)
{
    float const fragmentDepth = coord.z;
    float const fW = (float)(int)W;
    float const fH = (float)(int)H;
    float const dx = 1.5f * fdx / fW;
    float const dy = 1.5f * fdy / fH;
    float summed = 0.0f;
#pragma unroll
    for (int y = -2; y <= 2; y++)
    {
#pragma unroll
        for (int x = -2; x <= 2; x++)
        {
            float const s = coord.x + (float)x * dx;
            float const t = coord.y + (float)y * dy;
            float const fx = floorf(s * fW);
            float const fy = floorf(t * fH);
            float occluder = 0.0f;
            if (fx >= 0.0f && fy >= 0.0f && fx < fW && fy < fH)
            {
                occluder = map[(unsigned)fy * pitchFloats + (unsigned)fx];
            }
            if (occluder > 0.0f && occluder > fragmentDepth)
            {
                summed += 1.0f;
            }
        }
    }
    return 1.0f - summed / 25.0f;
}

} // namespace szg
