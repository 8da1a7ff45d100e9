// Exhaustive check on gfx950 that v_ldexp_f32 equals the two-step power-of-two scaling at the end of szg_expf
// (include/szg/fpmath.h): for every u in [0.5, 2) (all 2^24 bit patterns) and every q in [-152, 130],
//     (u * 2^(q >> 1)) * 2^(q - (q >> 1))  ==  ldexp(u, q)      bit for bit,
// including results that are denormal (one rounding in both forms: the first product is exact), zero and infinite.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Iinclude tools/verify_ldexp.hip -o verify_ldexp
#include <hip/hip_runtime.h>

#include <cstdio>

#include "szg/fpmath.h"

__global__ void k(unsigned long long* bad, int* firstQ, unsigned* firstU)
{
    int const q = (int)blockIdx.y - 152;
    unsigned long long cnt = 0;
    for (unsigned b = 0x3F000000u + blockIdx.x * blockDim.x + threadIdx.x; b < 0x40000000u; b += gridDim.x * blockDim.x)
    {
        float const u = __uint_as_float(b);
        int const q1 = q >> 1;
        float const two = (u * szg_pow2i(q1)) * szg_pow2i(q - q1);
        float const one = __builtin_ldexpf(u, q);
        if (__float_as_uint(two) != __float_as_uint(one))
        {
            if (cnt == 0)
            {
                atomicMin(firstQ, q);
                atomicMin(firstU, b);
            }
            cnt++;
        }
    }
    if (cnt)
    {
        atomicAdd(bad, cnt);
    }
}

int main()
{
    unsigned long long* bad;
    int* fq;
    unsigned* fu;
    (void)hipMalloc(&bad, 8);
    (void)hipMalloc(&fq, 4);
    (void)hipMalloc(&fu, 4);
    unsigned long long z = 0;
    int q = 1 << 30;
    unsigned u = 0xFFFFFFFFu;
    (void)hipMemcpy(bad, &z, 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(fq, &q, 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(fu, &u, 4, hipMemcpyHostToDevice);
    k<<<dim3(256, 283), 256>>>(bad, fq, fu); // q = -152 .. 130
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&z, bad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&q, fq, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&u, fu, 4, hipMemcpyDeviceToHost);
    std::printf("ldexp(u, q) vs (u * 2^(q>>1)) * 2^(q - (q>>1)), u in [0.5, 2) (2^24 values), q in [-152, 130]: %llu mismatches", z);
    if (z)
    {
        std::printf(" (smallest q %d, smallest u bits %08x)", q, u);
    }
    std::printf("\n");
    return z == 0 ? 0 : 1;
}
