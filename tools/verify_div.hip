// Checks szg_device.hpp's lean division against hipcc's correctly rounded `/` on gfx950. The operand space (2^64 pairs)
// cannot be enumerated, so besides 4.3e9 uniformly random mantissa pairs per exponent window the check walks the families
// where Newton-Raphson division is known to be fragile: quotients next to 1 (a = b +- k ulp), denominators with an
// all-ones or all-zero mantissa tail, numerators that are exact multiples of the denominator, and zero numerators.
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

int main()
{
    unsigned long long* d;
    (void)hipMalloc(&d, 24);
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
