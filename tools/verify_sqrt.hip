// Exhaustive check of szg_device.hpp sqrtN against hipcc's correctly rounded sqrtf on gfx950:
// every binary32 x in [2^-96, FLT_MAX], plus 0. Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
// -I../include -I../syzygy_amd/csrc tools/verify_sqrt.hip -o verify_sqrt
#include <hip/hip_runtime.h>

#include <cstdio>

#include "szg_device.hpp"

__global__ void k(unsigned long long* bad, unsigned* firstBad, unsigned lo, unsigned hi)
{
    unsigned long long cnt = 0;
    for (unsigned long long b = (unsigned long long)lo + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; b <= hi;
         b += (unsigned long long)gridDim.x * blockDim.x)
    {
        float const x = __uint_as_float((unsigned)b);
        if (__float_as_uint(sqrtf(x)) != __float_as_uint(szg::sqrtN(x)) || (x != 0.0f && __float_as_uint(sqrtf(x)) != __float_as_uint(szg::sqrtP(x))))
        {
            cnt++;
            atomicMin(firstBad, (unsigned)b);
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
    unsigned* fb;
    (void)hipMalloc(&bad, 8);
    (void)hipMalloc(&fb, 4);
    unsigned long long z = 0;
    unsigned f = 0xFFFFFFFFu;
    (void)hipMemcpy(bad, &z, 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(fb, &f, 4, hipMemcpyHostToDevice);
    k<<<4096, 256>>>(bad, fb, 0x0F800000u, 0x7F7FFFFFu); // [2^-96, FLT_MAX]
    k<<<1, 64>>>(bad, fb, 0u, 0u);                        // +0
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&z, bad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&f, fb, 4, hipMemcpyDeviceToHost);
    std::printf("sqrtN (and sqrtP, without 0) vs sqrtf over [2^-96, FLT_MAX] and 0: %llu mismatches (first %08x)\n", z, f);
    return z == 0 ? 0 : 1;
}
