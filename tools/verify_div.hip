// Checks szg_device.hpp's lean division (v_rcp + one Newton step = reciprocal; one residual correction = quotient)
// against hipcc's correctly rounded `/` on gfx950.
//   1. Premise of Markstein's theorem, exhaustively: rcpN(b) == RN(1 / b) for every b with |b| in [2^-60, 2^60].
//   2. The theorem's exception class, exhaustively: the 120 denominators 1.11..1 * 2^e of the domain against every numerator
//      significand (x 5 exponents x 2 signs).
//   3. 4096 random denominators against every numerator significand.
//   4. The operand space (2^64 pairs) cannot be enumerated: 4.3e9 random mantissa pairs per exponent window and family —
//      quotients next to 1 (a = b +- k ulp), denominators with an all-ones or all-zeros mantissa tail, numerators that are
//      exact multiples of the denominator, zero numerators.
// divN must match bit for bit; divN0 (no sign fix) must match except for the sign of a zero quotient.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Iinclude -Isyzygy_amd/csrc tools/verify_div.hip -o verify_div
#include <hip/hip_runtime.h>

#include <cstdio>
#include <utility>

#include "szg_device.hpp"

__device__ unsigned rng(unsigned& s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}
__device__ void compare(float a, float b, unsigned long long& bad, unsigned long long& bad0)
{
    float const ref = a / b;
    float const q = szg::divN(a, b);
    float const q0 = szg::divN0(a, b);
    if (__float_as_uint(q) != __float_as_uint(ref))
    {
        bad++;
    }
    bool const zeroBoth = (ref == 0.0f) && (q0 == 0.0f);
    if (__float_as_uint(q0) != __float_as_uint(ref) && !zeroBoth)
    {
        bad0++;
    }
}
__global__ void check(unsigned long long* out, int emin, int emax, int family)
{
    unsigned s = 0x9E3779B9u ^ ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + (unsigned)family * 977u);
    unsigned long long bad = 0, bad0 = 0;
    for (int it = 0; it < 4096; ++it)
    {
        unsigned const ma = rng(s), mb = rng(s);
        int const ea = emin + (int)(rng(s) % (unsigned)(emax - emin + 1)), eb = emin + (int)(rng(s) % (unsigned)(emax - emin + 1));
        float a = __uint_as_float((ma & 0x807FFFFFu) | ((unsigned)(ea + 127) << 23));
        float b = __uint_as_float((mb & 0x807FFFFFu) | ((unsigned)(eb + 127) << 23));
        if (family == 1) // quotient next to +-1: a = b +- k ulp
        {
            int const k = (int)(rng(s) % 65u) - 32;
            a = __uint_as_float((unsigned)((int)(__float_as_uint(b) & 0x7FFFFFFFu) + k)) * ((ma >> 31) ? -1.0f : 1.0f);
        }
        else if (family == 2) // mantissa tails of the denominator all ones / all zeros
        {
            unsigned const bits = (rng(s) % 23u) + 1u;
            unsigned const mask = (1u << bits) - 1u;
            unsigned bb = __float_as_uint(b);
            bb = (rng(s) & 1u) ? (bb | mask) : (bb & ~mask);
            b = __uint_as_float(bb);
        }
        else if (family == 3) // exact multiples: a = b * small integer (exact when it fits)
        {
            a = b * (float)((rng(s) % 4096u) + 1u);
        }
        else if (family == 4) // zero numerators of both signs
        {
            a = (ma & 1u) ? 0.0f : -0.0f;
        }
        compare(a, b, bad, bad0);
    }
    atomicAdd(out + 0, bad);
    atomicAdd(out + 1, bad0);
    atomicAdd(out + 2, 4096ull);
}

// 1. rcpN(b) == RN(1 / b)
__global__ void checkReciprocal(unsigned long long* out)
{
    unsigned long long cnt = 0, tot = 0;
    for (unsigned long long k = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; k < (1ull << 32);
         k += (unsigned long long)gridDim.x * blockDim.x)
    {
        float const b = __uint_as_float((unsigned)k), ab = fabsf(b);
        if (!(ab >= 0x1p-60f && ab <= 0x1p60f))
        {
            continue;
        }
        tot++;
        cnt += __float_as_uint(szg::rcpN(b)) != __float_as_uint(1.0f / b);
    }
    atomicAdd(out + 0, cnt);
    atomicAdd(out + 2, tot);
}
// 2. / 3. one denominator against every numerator significand
__global__ void checkDenominator(unsigned long long* out, unsigned denominatorBits)
{
    unsigned long long bad = 0, bad0 = 0, tot = 0;
    float const b = __uint_as_float(denominatorBits);
    int const eb = (int)((denominatorBits >> 23) & 0xFFu);
    for (unsigned long long k = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; k < (1ull << 24);
         k += (unsigned long long)gridDim.x * blockDim.x)
    {
        for (int de = -2; de <= 2; de++)
        {
            float a = __uint_as_float(((unsigned)(eb + de) << 23) | ((unsigned)k & 0x7FFFFFu));
            a = (k >> 23) ? -a : a;
            compare(a, b, bad, bad0);
            tot++;
        }
    }
    atomicAdd(out + 0, bad);
    atomicAdd(out + 1, bad0);
    atomicAdd(out + 2, tot);
}

int main()
{
    unsigned long long* d;
    (void)hipMalloc(&d, 24);
    {
        unsigned long long h[3];
        (void)hipMemset(d, 0, 24);
        checkReciprocal<<<4096, 256>>>(d);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        std::printf("rcpN(b) != RN(1/b): %llu of %llu values of b with |b| in [2^-60, 2^60] (exhaustive)\n", h[0], h[2]);
        int failedHere = h[0] != 0ull;
        (void)hipMemset(d, 0, 24);
        for (int e = -60; e <= 59; e++)
        {
            checkDenominator<<<512, 256>>>(d, ((unsigned)(e + 127) << 23) | 0x7FFFFFu);
        }
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        std::printf("all-ones denominators 1.11..1 * 2^e, e = -60..59, every numerator significand: %llu pairs, divN mismatches %llu, divN0 %llu\n",
                    h[2], h[0], h[1]);
        failedHere += (h[0] != 0ull) + (h[1] != 0ull);
        (void)hipMemset(d, 0, 24);
        unsigned s = 12345u;
        for (int i = 0; i < 4096; i++)
        {
            s = s * 1664525u + 1013904223u;
            unsigned const mant = (s >> 9) & 0x7FFFFFu;
            s = s * 1664525u + 1013904223u;
            unsigned const e = 127u - 60u + (s >> 8) % 120u;
            checkDenominator<<<512, 256>>>(d, (e << 23) | mant | ((s & 1u) << 31));
        }
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        std::printf("4096 random denominators, every numerator significand: %llu pairs, divN mismatches %llu, divN0 %llu\n", h[2], h[0], h[1]);
        failedHere += (h[0] != 0ull) + (h[1] != 0ull);
        if (failedHere != 0)
        {
            return 1;
        }
    }
    const char* names[5] = {"random mantissas", "quotients next to 1", "denominator tails 1..1 / 0..0", "exact multiples", "zero numerators"};
    int failed = 0;
    for (auto range : {std::pair<int, int>{-60, 60}, std::pair<int, int>{-30, 30}, std::pair<int, int>{-1, 1}})
    {
        for (int family = 0; family < 5; family++)
        {
            (void)hipMemset(d, 0, 24);
            check<<<4096, 256>>>(d, range.first, range.second, family);
            (void)hipDeviceSynchronize();
            unsigned long long h[3];
            (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            std::printf("exponents [%3d, %3d]  %-30s %llu pairs: divN mismatches %llu, divN0 mismatches beyond the sign of zero %llu\n",
                        range.first, range.second, names[family], h[2], h[0], h[1]);
            failed += (h[0] != 0ull) + (h[1] != 0ull);
        }
    }
    return failed == 0 ? 0 : 1;
}
