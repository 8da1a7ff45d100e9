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
// transmittance_LUT.comp:93-103. The 500 factors exp(-|dt| * extinction_i) are independent; only their
// product is sequential. 8 lanes share a texel: lane `sub` evaluates steps 8k + sub, then the running product is
// passed along the 8 lanes in ascending step order, i.e. (((1 * f0) * f1) * f2) ... exactly as the reference loop
// does; it ends up in the group's last lane. Steps >= 500
// contribute the factor 1.0, which is exact. This turns 1 wave per SIMD with a 500-deep serial chain into 8 waves
// per SIMD with a 63-deep one.
// v_mov_b32 with a DPP control (0x100 + n = row_shl:n, 0x110 + n = row_shr:n, rows of 16 lanes); the compiler
// folds it into the consuming v_mul_f32.
template <int CTRL> SZG_DEV float dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

constexpr int T_LANES = 8;
constexpr int T_STEPS = 500; // transmittance_LUT.comp:53

template <bool LEAN> SZG_DEV V3 transmittanceProduct(const Atm& a, V3 origin, V3 direction, float distance, float ndt, int sub)
{
    V3 T = splat(1.0f);
    float const rcp500 = rcpN(500.0f);
#pragma unroll 1
    for (int k = 0; k < (T_STEPS + T_LANES - 1) / T_LANES; k++)
    {
        int const i = k * T_LANES + sub;
        float const t = divRX<LEAN>(distance * ((float)i + 0.5f), 500.0f, rcp500);
        V3 const position = origin + t * direction;
        float const altitude = sqrtPX<LEAN>(dotT(position, position)) - a.planetRadius; // length(position), transmittance_LUT.comp:97
        Extinction const e = sampleExtinction<LEAN>(a, altitude);
        bool const valid = i < T_STEPS;
        float const fx = valid ? expX<LEAN>(ndt * e.extinction.x) : 1.0f;
        float const fy = valid ? expX<LEAN>(ndt * e.extinction.y) : 1.0f;
        float const fz = valid ? expX<LEAN>(ndt * e.extinction.z) : 1.0f;
        // Ordered product along the 8 lanes of the group with DPP (no LDS): position 0 takes the running
        // product from position 7 (row_shl:7), then position j takes position j-1's value (row_shr:1) and
        // multiplies its own factor. After 8 steps position 7 holds (((P * f0) * f1) ... * f7).
        T.x = dpp<0x107>(T.x) * fx;
        T.y = dpp<0x107>(T.y) * fy;
        T.z = dpp<0x107>(T.z) * fz;
#pragma unroll
        for (int j = 1; j < T_LANES; j++)
        {
            T.x = dpp<0x111>(T.x) * fx;
            T.y = dpp<0x111>(T.y) * fy;
            T.z = dpp<0x111>(T.z) * fz;
        }
    }
    return T;
}

// 256-thread workgroups = 32 texels x 8 lanes. 512x128 texels -> 2048 workgroups, 8 waves per SIMD.
__global__ __launch_bounds__(256) void k_transmittance(const szg_atmosphere_packed* __restrict__ atmospheres,
                                                      unsigned atmosphereIndex, float4* __restrict__ lut, int W, int H,
                                                      const unsigned* __restrict__ dirty)
{
    // LUT reuse (szg_launch.hpp "LUT reuse"): k_lut_key found the inputs byte-equal to those the texels were computed from
    if (dirty != nullptr && dirty[0] == 0u)
    {
        return;
    }
    int const sub = (int)(threadIdx.x & (unsigned)(T_LANES - 1));
    int const idRaw = (int)((blockIdx.x * 256u + threadIdx.x) / (unsigned)T_LANES);
    bool const inRangeTexel = idRaw < W * H;
    int const id = inRangeTexel ? idRaw : (W * H - 1); // out-of-range lanes shadow the last texel, never store
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
    bool const hit = raySphere(origin, direction, a.atmosphereRadius, t0, t1);
    // transmittance_LUT.comp:85-89: a miss stores 1; such a texel still walks the loop below with harmless
    // operands (the 8 lanes of a texel agree on `hit`, but shuffles need every lane of the wave in the loop)
    float const distance = hit ? t1 : 0.0f;
    float const dt = distance / 500.0f;
    float const ndt = -fabsf(dt);
    // lean exact ops (szg_device.hpp) when the atmosphere block and this ray are in their domain; the flag is
    // wave-uniform so that the loop body exists once per wave
    // smallest squared radius along the ray: at the closest approach when the ray points downwards, else at its origin
    float const rmin2 = mu < 0.0f ? radius * radius * (1.0f - mu * mu) : radius * radius;
    bool const lean = waveAll(a.lean && rmin2 >= a.leanFloor2 && inRange(radius, 0x1p-30f, 0x1p30f) && inRange(distance, 0.0f, 0x1p30f));
    V3 const T = lean ? transmittanceProduct<true>(a, origin, direction, distance, ndt, sub)
                      : transmittanceProduct<false>(a, origin, direction, distance, ndt, sub);
    if (inRangeTexel && sub == T_LANES - 1)
    {
        float4 const texel = hit ? make_float4(T.x, T.y, T.z, 1.0f) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        lut[id] = texel;
        if (!tlutTexelModerate(texel.x, texel.y, texel.z))
        {
            atomicOr(reinterpret_cast<unsigned*>(lut + (size_t)W * (size_t)H), 1u); // status dword (szg_launch.hpp)
        }
    }
}

// Status dword of a transmittance LUT whose texels were written by someone else (uploaded by the caller).
__global__ __launch_bounds__(256) void k_lut_range(float4* __restrict__ lut, unsigned n)
{
    unsigned const id = blockIdx.x * 256u + threadIdx.x;
    bool ok = true;
    if (id < n)
    {
        float4 const t = lut[id];
        ok = tlutTexelModerate(t.x, t.y, t.z);
    }
    if (!waveAll(ok) && (threadIdx.x & 63u) == 0u)
    {
        atomicOr(reinterpret_cast<unsigned*>(lut + n), 1u);
    }
}

// One lane: the per-frame constants of szg_device.hpp FramePrep.
__global__ __launch_bounds__(64) void k_frame_prep(const szg_atmosphere_packed* __restrict__ atmospheres, unsigned atmosphereIndex,
                                                   int tW, int tH, FramePrep* __restrict__ out,
                                                   const szg_directional_light_packed* __restrict__ sun)
{
    if (threadIdx.x != 0u)
    {
        return;
    }
    FramePrep f;
    f.a = load_atm(atmospheres + atmosphereIndex);
    f.fwidth = (float)tW;
    f.fheight = (float)tH;
    f.u_bias = 0.5f / (float)tW;
    f.u_scale = 1.0f - 1.0f / (float)tW;
    f.v_bias = 0.5f / (float)tH;
    f.v_scale = 1.0f - 1.0f / (float)tH;
    {
        // TLut::rowsInterior: radiusPart's v coordinate (same expressions, same contraction classes) at x_radius = 0 - every
        // radius at or below the planet's - and at the largest radius an INNER march can hand it, with a margin of 2^-16
        // for the few ulps by which a geometric length may exceed the analytic bound
        auto vOf = [&](float x_radius) {
            float const t = SZG_CON(SZG_C_LUTMAP, x_radius, f.v_scale, f.v_bias);
            return SZG_CON(SZG_C_TEXCOORD, t, f.fheight, -0.5f);
        };
        float const rMax = sqrtf(f.a.innerCeil2) * (1.0f + 0x1p-16f);
        float const rhoMax = safeSqrt(rMax * rMax - f.a.Rp2);
        float const vLo = vOf(0.0f), vHi = vOf(rhoMax / f.a.H);
        bool const ok = f.a.lean && vLo >= 0.0f && floorf(vHi) + 1.0f <= f.fheight - 1.0f && vHi == vHi && tW >= 2 && tH >= 2 &&
                        (float)tW * (float)tH <= 0x1p24f;
        f.rowsInterior = ok ? 1u : 0u;
    }
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
        f.sunShadow[k] = 0.0f;
    }
    if (sun != nullptr)
    {
        // shadowmap.glinl:2-7 TO_TEX_COORD_MAT (column-major literal); camera.comp:366-369
        M4 toTex;
        float const t[16] = {0.5f, 0.0f, 0.0f, 0.0f, 0.0f, 0.5f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.5f, 0.5f, 0.0f, 1.0f};
#pragma unroll
        for (int k = 0; k < 16; k++)
        {
            toTex.m[k] = t[k];
        }
        M4 const sm = mul(toTex, mul(load_m4(sun->projection), load_m4(sun->view)));
#pragma unroll
        for (int k = 0; k < 16; k++)
        {
            f.sunShadow[k] = sm.m[k];
        }
    }
    *out = f;
}

// 256-thread workgroups covering 32x8 texels (each wave an 8x8 patch, so the
// four transmittance-LUT taps of neighbouring lanes share cache lines).
__global__ __launch_bounds__(256, 4) void k_skyview(const szg_atmosphere_packed* __restrict__ atmospheres, unsigned atmosphereIndex,
                                                 const szg_camera_packed* __restrict__ cameras, unsigned cameraIndex,
                                                 const float4* __restrict__ tlut, int tW, int tH, float4* __restrict__ lut,
                                                 int W, int H, int rowBegin, int rowEnd, const unsigned* __restrict__ dirty,
                                                 const FramePrep* __restrict__ prep)
{
    if (dirty != nullptr && dirty[0] == 0u) // LUT reuse: inputs unchanged since these texels were computed
    {
        return;
    }
    unsigned const tid = threadIdx.x;
    unsigned const wave = tid >> 6, lane = tid & 63u;
    int const x = (int)(blockIdx.x * 32u + wave * 8u + (lane & 7u));
    int const y = rowBegin + (int)(blockIdx.y * 8u + (lane >> 3));
    if (x >= W || y >= rowEnd)
    {
        return;
    }
    Atm const a = load_atm(*prep); // (k_frame_prep: load_atm(atmospheres + atmosphereIndex), once per frame instead of once per wave)
    TLut const L = make_tlut(tlut, tW, tH, *prep);
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
    // status dword behind the texels (szg_launch.hpp "sky-view LUT block"): bit 0 = some texel is not a finite number of
    // moderate size. Ordinary LUTs never set it; the composite reads it before it skips a sample of this LUT unseen.
    if (!slutTexelFinite(luminance.x, luminance.y, luminance.z))
    {
        atomicOr(reinterpret_cast<unsigned*>(lut + (size_t)W * (size_t)H), 1u);
    }
}

// Status dword of a sky-view LUT whose texels were (partly) written by someone else: row slices gathered from other ranks,
// or an upload through szg_skyview_skyview_lut().
__global__ __launch_bounds__(256) void k_slut_check(float4* __restrict__ lut, unsigned n)
{
    unsigned const id = blockIdx.x * 256u + threadIdx.x;
    bool ok = true;
    if (id < n)
    {
        float4 const t = lut[id];
        ok = slutTexelFinite(t.x, t.y, t.z);
    }
    if (!waveAll(ok) && (threadIdx.x & 63u) == 0u)
    {
        atomicOr(reinterpret_cast<unsigned*>(lut + n), 1u);
    }
}

// The status dwords of the ranks' sky-view LUT slices, exchanged beside the slices (szg_skyview_allgather_lut_rows): each rank
// stages the status of ITS rows (or 1 = "not known to be moderate" when no slice launch of this pipeline produced it), the
// words are all-gathered, and the LUT's own status dword becomes their OR - instead of a 32 MiB re-scan of the gathered texels.
__global__ void k_slut_status_stage(unsigned* __restrict__ all, unsigned rank, const unsigned* __restrict__ status, unsigned known)
{
    if (threadIdx.x == 0u)
    {
        all[rank] = known != 0u ? status[0] : 1u;
    }
}
__global__ void k_slut_status_reduce(const unsigned* __restrict__ all, unsigned nranks, unsigned* __restrict__ status)
{
    if (threadIdx.x == 0u)
    {
        unsigned v = 0u;
        for (unsigned r = 0; r < nranks; r++)
        {
            v |= all[r];
        }
        status[0] = v;
    }
}

// Aerial-perspective froxel LUT (include/szg/abi.h "Aerial-perspective froxel LUT"): one froxel per lane, the
// exact reference math per froxel (32768 marches; cost ~ 1/64 of the sky-view LUT).
__global__ __launch_bounds__(256) void k_aerial_lut(const szg_atmosphere_packed* __restrict__ atmospheres, unsigned atmosphereIndex,
                                                    const szg_camera_packed* __restrict__ cameras, unsigned cameraIndex,
                                                    const float4* __restrict__ tlut, int tW, int tH, float4* __restrict__ luminance,
                                                    float4* __restrict__ transmittance, unsigned W, unsigned H, unsigned D,
                                                    float maxDistance)
{
    unsigned const id = blockIdx.x * 256u + threadIdx.x;
    if (id >= W * H * D)
    {
        return;
    }
    unsigned const i = id % W, j = (id / W) % H, k = id / (W * H);
    Atm const a = load_atm(atmospheres + atmosphereIndex);
    TLut const L = make_tlut(tlut, tW, tH);
    const szg_camera_packed* cam = cameras + cameraIndex;

    // camera.comp:320-328 with the froxel centre as the (continuous) pixel coordinate
    V3 position = mk3(cam->position[0], cam->position[1], cam->position[2]) / 1000000.0f;
    position.y *= -1.0f;
    position.y += a.planetRadius;
    float const clipx = (((float)i + 0.5f) / (float)W - 0.5f) * 2.0f;
    float const clipy = (((float)j + 0.5f) / (float)H - 0.5f) * 2.0f;
    V4 const dvs = mul(load_m4(cam->inverseProjection), clipx, clipy, 1.0f, 1.0f);
    V4 const rot = mul(load_m4(cam->rotation), dvs.x, dvs.y, dvs.z, dvs.w);
    V3 direction = normalize(mk3(rot.x, rot.y, rot.z));
    direction.y *= -1.0f;

    float const d = (((float)k + 0.5f) / (float)D) * maxDistance;
    V3 const lum = scatteringIntegral(L, a, position, direction, d);
    V3 const T = sampleT_Segment(L, a, position, position + d * direction);
    luminance[id] = make_float4(lum.x, lum.y, lum.z, 1.0f);
    transmittance[id] = make_float4(T.x, T.y, T.z, 1.0f);
}

hipError_t launch_aerial_lut(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, const szg_camera_packed* d_cam,
                             unsigned camIndex, const float* tlut, unsigned tW, unsigned tH, float* luminance, float* transmittance,
                             unsigned W, unsigned H, unsigned D, float maxDistance)
{
    unsigned const n = W * H * D;
    hipLaunchKernelGGL(k_aerial_lut, dim3((n + 255u) / 256u), dim3(256), 0, s, d_atm, atmIndex, d_cam, camIndex,
                       reinterpret_cast<const float4*>(tlut), (int)tW, (int)tH, reinterpret_cast<float4*>(luminance),
                       reinterpret_cast<float4*>(transmittance), W, H, D, maxDistance);
    return hipGetLastError();
}

// Multi-scattering LUT (include/szg/abi.h "Multi-scattering LUT"; extension, no reference counterpart).
// One wavefront per texel, one sphere direction per lane, butterfly reduction over the 64 lanes.
SZG_DEV float waveSum64(float v)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
        v = v + __shfl_xor(v, off); // every lane ends with the same tree sum ((l ^ off) partner, commutative add)
    }
    return v;
}

__global__ __launch_bounds__(64) void k_multiscatter(const szg_atmosphere_packed* __restrict__ atmospheres, unsigned atmosphereIndex,
                                                     const float4* __restrict__ tlut, int tW, int tH, float4* __restrict__ out,
                                                     unsigned dim)
{
    unsigned const tx = blockIdx.x, ty = blockIdx.y, lane = threadIdx.x;
    Atm const a = load_atm(atmospheres + atmosphereIndex);
    TLut const L = make_tlut(tlut, tW, tH);
    float const PI = 3.141592653589793f;

    float const cosSunZenith = (((float)tx + 0.5f) / (float)dim) * 2.0f - 1.0f;
    float const radius = a.planetRadius + (((float)ty + 0.5f) / (float)dim) * (a.atmosphereRadius - a.planetRadius);
    V3 const sunDir = mk3(safeSqrt(1.0f - cosSunZenith * cosSunZenith), cosSunZenith, 0.0f);
    V3 const pos = mk3(0.0f, radius, 0.0f);

    // 8 x 8 stratified sphere directions
    float const theta = 2.0f * PI * (((float)(lane & 7u) + 0.5f) / 8.0f);
    float const cosPhi = 1.0f - 2.0f * (((float)(lane >> 3) + 0.5f) / 8.0f);
    float const sinPhi = safeSqrt(1.0f - cosPhi * cosPhi);
    V3 const dir = mk3(szg_cosf(theta) * sinPhi, cosPhi, szg_sinf(theta) * sinPhi);

    float const tMax = raycastAtmosphere(a, pos, dir);
    float gt0 = 0.0f, gt1 = 0.0f;
    bool const hitGround = raySphere(pos, dir, a.planetRadius, gt0, gt1) && gt0 > 0.0f;

    float const sin_sunRadius = szg_sinf(a.sunAngularRadius);
    float const cos_sunRadius = szg_cosf(a.sunAngularRadius);
    float const isotropic = 1.0f / (4.0f * PI); // common.glinl:282
    V3 L2 = splat(0.0f), fms = splat(0.0f), Tacc = splat(1.0f);
    float const dt = tMax / 20.0f;
    for (int i = 0; i < 20; i++)
    {
        float const t = ((float)i + 0.5f) * dt;
        V3 const p = pos + t * dir;
        float const r = length(p);
        Extinction const ex = sampleExtinction(a, r - a.planetRadius);
        V3 const sigma_s = ex.scatteringRayleigh + ex.scatteringMie;
        float const mu_s = dot(p, sunDir) / r;
        // sampleTransmittanceLUT_Sun, common.glinl:145-172
        float const sin_hz = a.planetRadius / r;
        float const cos_hz = -safeSqrt(1.0f - sin_hz * sin_hz);
        V3 const T_sun = sampleT_RadiusMu(L, a, r, mu_s) * smoothstep(-sin_hz * sin_sunRadius, sin_hz * sin_sunRadius, mu_s - cos_hz * cos_sunRadius);
        V3 const T_step = mk3(szg_expf(-dt * ex.extinction.x), szg_expf(-dt * ex.extinction.y), szg_expf(-dt * ex.extinction.z));
        V3 const S = sigma_s * T_sun * isotropic;
        L2 = L2 + Tacc * ((S - S * T_step) / ex.extinction);
        fms = fms + Tacc * ((sigma_s - sigma_s * T_step) / ex.extinction);
        Tacc = Tacc * T_step;
    }
    if (hitGround)
    {
        V3 const pg = pos + tMax * dir;
        V3 const n = normalize(pg);
        float const NdotL = clampf(dot(n, sunDir), 0.0f, 1.0f);
        float const rg = length(pg);
        float const mu_g = dot(pg, sunDir) / rg;
        float const sin_hz = a.planetRadius / rg;
        float const cos_hz = -safeSqrt(1.0f - sin_hz * sin_hz);
        V3 const T_sun = sampleT_RadiusMu(L, a, rg, mu_g) * smoothstep(-sin_hz * sin_sunRadius, sin_hz * sin_sunRadius, mu_g - cos_hz * cos_sunRadius);
        L2 = L2 + Tacc * T_sun * (NdotL * (0.4f / PI)); // camera.comp:218 ground albedo
    }
    V3 const sumL = mk3(waveSum64(L2.x), waveSum64(L2.y), waveSum64(L2.z));
    V3 const sumF = mk3(waveSum64(fms.x), waveSum64(fms.y), waveSum64(fms.z));
    if (lane == 0u)
    {
        V3 const meanL = sumL / 64.0f;
        V3 const meanF = sumF / 64.0f;
        V3 const psi = meanL / (splat(1.0f) - meanF);
        out[ty * dim + tx] = make_float4(psi.x, psi.y, psi.z, 1.0f);
    }
}

hipError_t launch_multiscatter(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, const float* tlut, unsigned tW,
                               unsigned tH, float* out, unsigned dim)
{
    hipLaunchKernelGGL(k_multiscatter, dim3(dim, dim), dim3(64), 0, s, d_atm, atmIndex, reinterpret_cast<const float4*>(tlut), (int)tW,
                       (int)tH, reinterpret_cast<float4*>(out), dim);
    return hipGetLastError();
}

hipError_t launch_transmittance(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, float* lut, unsigned W,
                                unsigned H, const unsigned* d_dirty)
{
    unsigned const n = W * H * (unsigned)T_LANES;
    if (d_dirty == nullptr) // (with LUT reuse k_lut_key clears the dword, and only when the texels are recomputed)
    {
        hipError_t const e = hipMemsetAsync(lut + (size_t)W * H * 4u, 0, 4u, s); // status dword: set by texels out of range
        if (e != hipSuccess)
        {
            return e;
        }
    }
    hipLaunchKernelGGL(k_transmittance, dim3((n + 255u) / 256u), dim3(256), 0, s, d_atm, atmIndex, reinterpret_cast<float4*>(lut),
                       (int)W, (int)H, d_dirty);
    return hipGetLastError();
}

// LUT reuse across frames (szg_launch.hpp). One wave. `state` holds, as dwords: [0, 32) the atmosphere block the
// transmittance LUT was computed from, [32, 64) the one the sky-view LUT was computed from, [64, 67) its camera position,
// [67] the generation of the transmittance LUT, [68] the generation the sky-view LUT was computed from, [69] / [70] the
// dirty flags the two LUT kernels read. which == 0: transmittance; which == 1: sky-view. Comparisons are on the bit
// patterns: byte-equal inputs give byte-equal texels.
__global__ __launch_bounds__(64) void k_lut_key(const szg_atmosphere_packed* __restrict__ atmospheres, unsigned atmosphereIndex,
                                                const szg_camera_packed* __restrict__ cameras, unsigned cameraIndex,
                                                unsigned* __restrict__ state, unsigned which, unsigned force,
                                                unsigned* __restrict__ statusDword)
{
    unsigned const lane = threadIdx.x;
    const unsigned* const atm = reinterpret_cast<const unsigned*>(atmospheres + atmosphereIndex);
    unsigned* const key = state + (which == 0u ? 0u : 32u);
    bool differs = false;
    unsigned mine = 0u;
    if (lane < 32u)
    {
        mine = atm[lane];
        differs = mine != key[lane];
    }
    else if (which == 1u && lane < 35u)
    {
        mine = reinterpret_cast<const unsigned*>((cameras + cameraIndex)->position)[lane - 32u];
        differs = mine != state[64u + (lane - 32u)];
    }
    bool dirty = force != 0u || __builtin_amdgcn_ballot_w64(differs) != 0ull;
    if (which == 1u)
    {
        dirty = dirty || state[67] != state[68]; // the transmittance LUT has been recomputed since
    }
    if (dirty)
    {
        if (lane < 32u)
        {
            key[lane] = mine;
        }
        else if (which == 1u && lane < 35u)
        {
            state[64u + (lane - 32u)] = mine;
        }
    }
    if (lane == 0u)
    {
        if (dirty)
        {
            if (which == 0u)
            {
                state[67] = state[67] + 1u;
            }
            else
            {
                state[68] = state[67];
            }
            statusDword[0] = 0u; // the recomputing kernel sets it again where a texel is out of range
        }
        state[69u + which] = dirty ? 1u : 0u;
    }
}

hipError_t launch_lut_key(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, const szg_camera_packed* d_cam,
                          unsigned camIndex, unsigned* d_state, unsigned which, bool force, float* lutBlock, unsigned W, unsigned H)
{
    hipLaunchKernelGGL(k_lut_key, dim3(1), dim3(64), 0, s, d_atm, atmIndex, d_cam, camIndex, d_state, which, force ? 1u : 0u,
                       reinterpret_cast<unsigned*>(lutBlock + (size_t)W * H * 4u));
    return hipGetLastError();
}

size_t frame_prep_bytes() { return sizeof(FramePrep); }
hipError_t launch_frame_prep(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, unsigned tW, unsigned tH, void* d_prep,
                             const szg_directional_light_packed* d_sun)
{
    hipLaunchKernelGGL(k_frame_prep, dim3(1), dim3(64), 0, s, d_atm, atmIndex, (int)tW, (int)tH, static_cast<FramePrep*>(d_prep), d_sun);
    return hipGetLastError();
}

hipError_t launch_lut_range(hipStream_t s, float* lut, unsigned W, unsigned H)
{
    unsigned const n = W * H;
    hipError_t const e = hipMemsetAsync(lut + (size_t)n * 4u, 0, 4u, s);
    if (e != hipSuccess)
    {
        return e;
    }
    hipLaunchKernelGGL(k_lut_range, dim3((n + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<float4*>(lut), n);
    return hipGetLastError();
}

hipError_t launch_slut_check(hipStream_t s, float* lut, unsigned W, unsigned H)
{
    unsigned const n = W * H;
    hipError_t const e = hipMemsetAsync(lut + (size_t)n * 4u, 0, 4, s);
    if (e != hipSuccess)
    {
        return e;
    }
    hipLaunchKernelGGL(k_slut_check, dim3((n + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<float4*>(lut), n);
    return hipGetLastError();
}

hipError_t launch_slut_status_stage(hipStream_t s, unsigned* d_all, unsigned rank, const float* lut, unsigned W, unsigned H, bool known)
{
    hipLaunchKernelGGL(k_slut_status_stage, dim3(1), dim3(64), 0, s, d_all, rank,
                       reinterpret_cast<const unsigned*>(lut + (size_t)W * H * 4u), known ? 1u : 0u);
    return hipGetLastError();
}
hipError_t launch_slut_status_reduce(hipStream_t s, const unsigned* d_all, unsigned nranks, float* lut, unsigned W, unsigned H)
{
    hipLaunchKernelGGL(k_slut_status_reduce, dim3(1), dim3(64), 0, s, d_all, nranks, reinterpret_cast<unsigned*>(lut + (size_t)W * H * 4u));
    return hipGetLastError();
}

hipError_t launch_skyview(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, const szg_camera_packed* d_cam,
                          unsigned camIndex, const float* tlut, unsigned tW, unsigned tH, float* lut, unsigned W, unsigned H,
                          unsigned rowBegin, unsigned rowEnd, const unsigned* d_dirty, const void* d_prep)
{
    if (rowEnd > H)
    {
        rowEnd = H;
    }
    if (rowBegin >= rowEnd || W == 0u)
    {
        return hipSuccess;
    }
    if (d_dirty == nullptr)
    {
        // the status dword starts clear and the kernel sets it: for a whole LUT that is its status, for a row slice the status
        // of THAT slice, which szg_skyview_allgather_lut_rows exchanges with the other ranks' (any other consumer of a
        // partly written LUT re-scans it, launch_slut_check); with LUT reuse k_lut_key clears it
        hipError_t const e = hipMemsetAsync(lut + (size_t)W * H * 4u, 0, 4, s);
        if (e != hipSuccess)
        {
            return e;
        }
    }
    dim3 const grid((W + 31u) / 32u, (rowEnd - rowBegin + 7u) / 8u);
    hipLaunchKernelGGL(k_skyview, grid, dim3(256), 0, s, d_atm, atmIndex, d_cam, camIndex, reinterpret_cast<const float4*>(tlut),
                       (int)tW, (int)tH, reinterpret_cast<float4*>(lut), (int)W, (int)H, (int)rowBegin, (int)rowEnd, d_dirty,
                       static_cast<const FramePrep*>(d_prep));
    return hipGetLastError();
}
} // namespace szg
