// kernels_lut.hip — the two atmosphere LUT passes for gfx950.
//
//   k_transmittance : atmosphere/transmittance_LUT.comp:55-106
//   k_skyview       : atmosphere/skyview_LUT.comp:51-128
//
// Both are pure VALU/transcendental kernels (bytes written: 1 MiB and 32 MiB at
// the reference sizes); nothing here is HBM-bound. See DESIGN.md.

#include "szg_device.hpp"
#include "szg_launch.hpp"

namespace szg
{
// One texel per lane, 64-lane workgroups: 512x128 texels = 1024 workgroups, so
// the 500-step serial loop lands on every SIMD of the chip (1024 SIMDs).
__global__ __launch_bounds__(64) void k_transmittance(const szg_atmosphere_packed* __restrict__ atmospheres,
                                                      unsigned atmosphereIndex, float4* __restrict__ lut, int W, int H)
{
    int const id = (int)(blockIdx.x * 64u + threadIdx.x);
    if (id >= W * H)
    {
        return;
    }
    int const tx = id % W;
    int const ty = id / W;
    Atm const a = load_atm(atmospheres + atmosphereIndex);

    // transmittance_LUT.comp:66-67
    float const u = ((float)tx + 0.5f) / (float)W;
    float const v = ((float)ty + 0.5f) / (float)H;

    // transmittanceLUT_UV_to_RMu, common.glinl:69-102
    float const x_mu = (u - 0.5f / (float)W) / (1.0f - 1.0f / (float)W);
    float const x_radius = (v - 0.5f / (float)H) / (1.0f - 1.0f / (float)H);
    float const rho = a.H * x_radius;
    float const radius = sqrtf(rho * rho + a.Rp2);
    float const d_min = a.atmosphereRadius - radius;
    float const d_max = rho + a.H;
    float const d = (d_max - d_min) * x_mu + d_min;
    float mu = 1.0f;
    if (d != 0.0f)
    {
        mu = clampf((a.H * a.H - rho * rho - d * d) / (2.0f * radius * d), -1.0f, 1.0f);
    }

    V3 const origin = mk3(0.0f, radius, 0.0f);
    V3 const direction = mk3(sqrtf(1.0f - mu * mu), mu, 0.0f);

    float t0 = 0.0f, t1 = 0.0f;
    if (!raySphere(origin, direction, a.atmosphereRadius, t0, t1))
    {
        lut[id] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        return;
    }
    float const distance = t1;
    float const dt = distance / 500.0f;
    float const ndt = -fabsf(dt);
    V3 T = splat(1.0f);
#pragma unroll 2
    for (int i = 0; i < 500; i++)
    {
        float const t = distance * ((float)i + 0.5f) / 500.0f;
        V3 const position = origin + t * direction;
        float const altitude = length(position) - a.planetRadius;
        Extinction const e = sampleExtinction(a, altitude);
        T.x = T.x * szg_expf(ndt * e.extinction.x);
        T.y = T.y * szg_expf(ndt * e.extinction.y);
        T.z = T.z * szg_expf(ndt * e.extinction.z);
    }
    lut[id] = make_float4(T.x, T.y, T.z, 1.0f);
}

// 256-thread workgroups covering 32x8 texels (each wave an 8x8 patch, so the
// four transmittance-LUT taps of neighbouring lanes share cache lines).
__global__ __launch_bounds__(256) void k_skyview(const szg_atmosphere_packed* __restrict__ atmospheres, unsigned atmosphereIndex,
                                                 const szg_camera_packed* __restrict__ cameras, unsigned cameraIndex,
                                                 const float4* __restrict__ tlut, int tW, int tH, float4* __restrict__ lut,
                                                 int W, int H)
{
    unsigned const tid = threadIdx.x;
    unsigned const wave = tid >> 6, lane = tid & 63u;
    int const x = (int)(blockIdx.x * 32u + wave * 8u + (lane & 7u));
    int const y = (int)(blockIdx.y * 8u + (lane >> 3));
    if (x >= W || y >= H)
    {
        return;
    }
    Atm const a = load_atm(atmospheres + atmosphereIndex);
    TLut const L = make_tlut(tlut, tW, tH);
    const szg_camera_packed* cam = cameras + cameraIndex;
    float const PI = 3.141592653589793f;

    // skyview_LUT.comp:100-101
    float const u = ((float)x + 0.5f) / (float)W;
    float const v = ((float)y + 0.5f) / (float)H;

    // skyview_LUT.comp:110-112
    V3 origin = mk3(cam->position[0], cam->position[1], cam->position[2]) / 1000000.0f;
    origin.y *= -1.0f;
    origin.y += a.planetRadius;

    // uv_to_azimuthElevation, skyview_LUT.comp:51-89
    float const radius = length(origin);
    float const sinHorizonZenith = a.planetRadius / radius;
    float const horizonZenith = PI - szg_asinf(sinHorizonZenith);
    float const cosineViewLightProjected = (u - 0.5f) * 2.0f;
    V2 const lightDirectionProjected = normalize(V2{-a.incidentDirectionSun.x, -a.incidentDirectionSun.z});
    float azimuthSun = szg_asinf(lightDirectionProjected.x);
    if (lightDirectionProjected.y < 0.0f)
    {
        azimuthSun = PI - azimuthSun;
    }
    float const azimuth = szg_acosf(clampf(cosineViewLightProjected, -1.0f, 1.0f)) + azimuthSun;
    float viewZenith;
    float const unnormalized_v = 2.0f * v - 1.0f;
    if (v < 0.5f)
    {
        float const angleFraction = 1.0f - unnormalized_v * unnormalized_v;
        viewZenith = angleFraction * horizonZenith;
    }
    else
    {
        float const angleFraction = unnormalized_v * unnormalized_v;
        viewZenith = (PI - horizonZenith) * angleFraction + horizonZenith;
    }
    float const elevation = -(viewZenith - PI / 2.0f);

    // skyview_LUT.comp:118-119
    float const ce = szg_cosf(elevation);
    V3 const direction = normalize(mk3(szg_sinf(azimuth) * ce, szg_sinf(elevation), szg_cosf(azimuth) * ce));

    float const distance = raycastAtmosphere(a, origin, direction);
    V3 const luminance = scatteringIntegral(L, a, origin, direction, distance);
    lut[y * W + x] = make_float4(luminance.x, luminance.y, luminance.z, 1.0f);
}

hipError_t launch_transmittance(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, float* lut, unsigned W,
                                unsigned H)
{
    unsigned const n = W * H;
    hipLaunchKernelGGL(k_transmittance, dim3((n + 63u) / 64u), dim3(64), 0, s, d_atm, atmIndex, reinterpret_cast<float4*>(lut),
                       (int)W, (int)H);
    return hipGetLastError();
}

hipError_t launch_skyview(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, const szg_camera_packed* d_cam,
                          unsigned camIndex, const float* tlut, unsigned tW, unsigned tH, float* lut, unsigned W, unsigned H)
{
    dim3 const grid((W + 31u) / 32u, (H + 7u) / 8u);
    hipLaunchKernelGGL(k_skyview, grid, dim3(256), 0, s, d_atm, atmIndex, d_cam, camIndex, reinterpret_cast<const float4*>(tlut),
                       (int)tW, (int)tH, reinterpret_cast<float4*>(lut), (int)W, (int)H);
    return hipGetLastError();
}
} // namespace szg
