// kernels_composite.hip — the full-screen sky / aerial-perspective composite
// ("perspective map") for gfx950: atmosphere/camera.comp:70-395.
//
// Structure (results unchanged, see DESIGN.md "composite"):
//   phase A  classify the pixel (sky / geometry), evaluate everything that is
//            not a ray march, and record up to two march requests
//              slot 0: primary ray  (geometry aerial perspective, or a sky ray
//                      that hits the ground: sampleGround)
//              slot 1: metal reflection ray that hits the ground
//   phase B  run the 32-step scattering integral for the active slots through
//            ONE copy of the march code
//   phase C  combine in the order camera.comp adds the terms, tonemap, store.

#include "szg_device.hpp"
#include "szg_launch.hpp"

namespace szg
{
namespace
{
template <typename T> SZG_DEV T* row_ptr(const szg_image& im, unsigned y)
{
    return reinterpret_cast<T*>(static_cast<unsigned char*>(im.data) + (size_t)y * im.pitch_bytes);
}

struct SkyLut
{
    const float4* texels;
    int width, height;
    bool finite; // status dword behind the texels (szg_launch.hpp "sky-view LUT block"): every texel a finite, moderate number
};

// sampleMap_Direction, camera.comp:70-121 (sky-view LUT: LINEAR, CLAMP_TO_EDGE,
// no half-texel squeeze: SURVEY Q14)
SZG_DEV V3 sampleMapDirection(const SkyLut& S, const Atm& a, V3 position, V3 direction)
{
    float const PI = 3.141592653589793f;
    V3 const normalized = normalize(direction);
    float const sinHorizonZenith = a.planetRadius / length(position);
    float const horizonZenith = PI - szg_asinf(sinHorizonZenith);
    float const cosViewZenith = normalized.y;
    float const cosHorizonZenith = -safeSqrt(1.0f - sinHorizonZenith * sinHorizonZenith);
    float const viewZenith = szg_acosf(normalized.y);
    float v;
    if (cosViewZenith > cosHorizonZenith)
    {
        float const angleFraction = viewZenith / horizonZenith;
        v = (1.0f - sqrtf(1.0f - angleFraction)) * 0.5f;
    }
    else
    {
        float const angleFraction = (viewZenith - horizonZenith) / (PI - horizonZenith);
        v = sqrtf(angleFraction) * 0.5f + 0.5f;
    }
    V2 const projectedLight = normalize(V2{-a.incidentDirectionSun.x, -a.incidentDirectionSun.z});
    V2 const projectedView = normalize(V2{direction.x, direction.z});
    float const u = clampf(dot(projectedLight, projectedView), -1.0f, 1.0f) * 0.5f + 0.5f;
    return bilinear_rgb(S.texels, S.width, S.height, (float)S.width, (float)S.height, u, v);
}

// sampleSunDisk, camera.comp:123-140
SZG_DEV V3 sampleSunDisk(const TLut& L, const Atm& a, V3 position, V3 direction)
{
    V3 const directionToSun = -a.incidentDirectionSun;
    float const cosDirectionSun = dot(direction, directionToSun) / (length(direction) * length(directionToSun));
    float const sinSunRadius = a.sunAngularRadius;
    float const sinDirectionSun = safeSqrt(1.0f - cosDirectionSun * cosDirectionSun);
    if (cosDirectionSun < 0.0f)
    {
        return splat(0.0f);
    }
    V3 const transmittanceToSun = sampleT_Ray(L, a, position, direction);
    return transmittanceToSun * (1.0f - smoothstep(0.2f * sinSunRadius, sinSunRadius, sinDirectionSun));
}

// raycastDistanceToGround, camera.comp:175-192
SZG_DEV bool raycastGround(const Atm& a, V3 origin, V3 direction, float& distance)
{
    float t0 = 0.0f, t1 = 0.0f;
    bool const hit = raySphere(origin, direction, a.planetRadius, t0, t1) && t0 > 0.0f;
    distance = t0;
    return hit;
}

// The part of sampleGround (camera.comp:203-235) that is not the march:
// surfaceLuminance * transmittanceToSurface.
SZG_DEV V3 groundSurfaceTerm(const TLut& L, const Atm& a, V3 origin, V3 direction, float distanceToGround)
{
    float const PI = 3.141592653589793f;
    V3 const surfacePosition = origin + distanceToGround * direction;
    V3 const surfaceNormal = normalize(surfacePosition);
    V3 const lightDirection = -a.incidentDirectionSun;
    V3 const viewDirection = -direction;
    V3 const h = normalize(lightDirection + viewDirection);
    float const specularPower = 160.0f;
    float const microfacet = szg_powf(clampf(dot(h, surfaceNormal), 0.0f, 1.0f), specularPower);
    float const normalization = (specularPower + 2.0f) / 8.0f;
    V3 const specular = splat(normalization * microfacet);
    V3 const diffuse = splat(0.4f) / PI;
    V3 const fresnel =
        splat(0.04f) + (splat(1.0f) - splat(0.04f)) * szg_powf(1.0f - clampf(dot(h, lightDirection), 0.0f, 1.0f), 5.0f);
    V3 const albedo = mix(diffuse, specular, fresnel);
    V3 const transmittanceToSun = sampleT_Ray(L, a, surfacePosition, lightDirection);
    V3 const surfaceLuminance = (transmittanceToSun * albedo) * clampf(dot(surfaceNormal, lightDirection), 0.0f, 1.0f);
    V3 const transmittanceToSurface = sampleT_Segment(L, a, origin, surfacePosition);
    return surfaceLuminance * transmittanceToSurface;
}
} // namespace

#ifdef SZG_TAIL_DIAG
// Diagnostic build only (tools/tail_histogram.py; never part of libszg_hip.so): every wave of k_composite stamps the constant-
// frequency clock (s_memrealtime, 100 MHz) when it starts and when it ends into a buffer the tool hands over, together
// with the hardware id of where it ran, so that the launch tail can be read off a histogram (profiles/r03_tail_*.txt).
__device__ unsigned long long* g_tailStamps = nullptr; // [workgroup * 4 + wave] * 3: start, end, HW_ID
extern "C" int szg_debug_tail_buffer(void* d_buffer)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tailStamps), &d_buffer, sizeof d_buffer);
}
struct TailStamp
{
    unsigned long long start;
    unsigned slot;
    SZG_DEV TailStamp(unsigned slot_) : start(wall_clock64()), slot(slot_) {}
    SZG_DEV ~TailStamp()
    {
        if (g_tailStamps != nullptr && (threadIdx.x & 63u) == 0u)
        {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            g_tailStamps[(size_t)slot * 3u + 0u] = start;
            g_tailStamps[(size_t)slot * 3u + 1u] = wall_clock64();
            g_tailStamps[(size_t)slot * 3u + 2u] = hw;
        }
    }
};
#endif

struct GBufferPtrsC
{
    szg_image diffuse, specular, normal, position, orm;
};

// Trilinear fetch of the aerial-perspective luminance volume at screen position (sx, sy) in [0, 1] and distance
// `dist` (Mm): clamp-to-edge in x, y; slices at d_k = (k + .5) / D * maxDistance, linear ramp to 0 below d_0.
SZG_DEV V3 sampleAerial(const AerialLut& A, float sx, float sy, float dist)
{
    const float4* vol = reinterpret_cast<const float4*>(A.luminance);
    float const fz = dist / A.maxDistance * (float)A.D - 0.5f;
    float const ramp = fz < 0.0f ? fmaxf(dist / (0.5f * A.maxDistance / (float)A.D), 0.0f) : 1.0f;
    float const z = fminf(fmaxf(fz, 0.0f), (float)A.D - 1.0f);
    int const k0 = (int)floorf(z);
    int const k1 = min(k0 + 1, (int)A.D - 1);
    float const wz = z - (float)k0;
    V3 const l0 = bilinear_rgb(vol + (size_t)k0 * A.W * A.H, (int)A.W, (int)A.H, (float)A.W, (float)A.H, sx, sy);
    V3 const l1 = bilinear_rgb(vol + (size_t)k1 * A.W * A.H, (int)A.W, (int)A.H, (float)A.W, (float)A.H, sx, sy);
    return (l0 * (1.0f - wz) + l1 * wz) * ramp;
}

template <bool FAST>
__global__ __launch_bounds__(256, 3) void k_composite(szg_image color, szg_image depth, szg_image debug, GBufferPtrsC g,
                                                   unsigned drawW, unsigned drawH, unsigned localRows, RowMap rm,
                                                   ShadowSlot sunSlot, const szg_atmosphere_packed* __restrict__ atmospheres,
                                                   unsigned atmosphereIndex, const szg_camera_packed* __restrict__ cameras,
                                                   unsigned cameraIndex,
                                                   const szg_directional_light_packed* __restrict__ dirLights,
                                                   unsigned sunLightIndex, const float4* __restrict__ tlut, int tW, int tH,
                                                   const float4* __restrict__ slut, int sW, int sH, AerialLut aerial,
                                                   const FramePrep* __restrict__ prep)
{
    unsigned const tid = threadIdx.x;
    unsigned const wave = tid >> 6, lane = tid & 63u;
#ifdef SZG_TAIL_DIAG
    TailStamp const stamp((blockIdx.y * gridDim.x + blockIdx.x) * 4u + wave);
#endif
#ifdef SZG_EXP_LDS_TLUT
    // EXPERIMENT: stage the first rows of the transmittance LUT (packed rgb) in LDS, once per workgroup
    __shared__ float s_tlutRows[SZG_EXP_LDS_TLUT * 512 * 3];
    if (tW == 512)
    {
        for (unsigned k = tid; k < (unsigned)(SZG_EXP_LDS_TLUT * 512); k += 256u)
        {
            float4 const t = tlut[k];
            s_tlutRows[k * 3u + 0u] = t.x;
            s_tlutRows[k * 3u + 1u] = t.y;
            s_tlutRows[k * 3u + 2u] = t.z;
        }
    }
    __syncthreads();
#endif
    unsigned const x = blockIdx.x * 32u + wave * 8u + (lane & 7u);
    // Workgroups are dispatched in blockIdx order; the rows are walked from the bottom of the image upwards so that the
    // cheap workgroups (sky: no march) tend to come last and fill the tail of the launch instead of its start.
    unsigned const y = (gridDim.y - 1u - blockIdx.y) * 8u + (lane >> 3);
    if (x >= drawW || y >= localRows)
    {
        return;
    }
    unsigned const gy = global_row(rm, y);

    Atm const a = load_atm(*prep); // (k_frame_prep: load_atm(atmospheres + atmosphereIndex), once per frame instead of once per wave)
#ifdef SZG_EXP_LDS_TLUT
    TLut L = make_tlut(tlut, tW, tH, *prep);
    L.ldsRows = tW == 512 ? (const __attribute__((address_space(3))) float*)s_tlutRows : nullptr;
#else
    TLut const L = make_tlut(tlut, tW, tH, *prep);
#endif
    SkyLut const S{slut, sW, sH, reinterpret_cast<const unsigned*>(slut + (size_t)sW * (size_t)sH)[0] == 0u};
    const szg_camera_packed* cam = cameras + cameraIndex;

    // camera.comp:315-316 (nearest at the pixel's own texel)
    float const sceneDepth = row_ptr<const float>(depth, y)[x];

    // camera.comp:320-322
    V3 position = mk3(cam->position[0], cam->position[1], cam->position[2]) / 1000000.0f;
    position.y *= -1.0f;
    position.y += a.planetRadius;

    // camera.comp:324-328 (no pixel-centre offset: SURVEY Q5; global row)
    float const clipx = ((float)x / (float)drawW - 0.5f) * 2.0f;
    float const clipy = ((float)gy / (float)drawH - 0.5f) * 2.0f;
    M4 const inverseProjection = load_m4(cam->inverseProjection);
    M4 const rotation = load_m4(cam->rotation);
    V4 const dvs = mul(inverseProjection, clipx, clipy, 1.0f, 1.0f);
    V4 const rot = mul(rotation, dvs.x, dvs.y, dvs.z, dvs.w);
    V3 direction = normalize(mk3(rot.x, rot.y, rot.z));
    direction.y *= -1.0f;

    // ---- phase A ---------------------------------------------------------
    bool isSky = (sceneDepth == 0.0f);
    Material m;
    if (!isSky)
    {
        // camera.comp:346-347; a pixel with depth 0 is sky whatever the G-buffer holds (:354)
        V4 const diffuse = unpack_half4(row_ptr<const uint2>(g.diffuse, y)[x]);
        V4 const specular = unpack_half4(row_ptr<const uint2>(g.specular, y)[x]);
        V4 const normal = unpack_half4(row_ptr<const uint2>(g.normal, y)[x]);
        V4 const orm = unpack_half4(row_ptr<const uint2>(g.orm, y)[x]);
        float4 const p4 = row_ptr<const float4>(g.position, y)[x];
        m = convertPBR(V4{p4.x, p4.y, p4.z, p4.w}, normal, diffuse, specular, orm);
        isSky = m.position.y > 0.0f; // underground, +y down (:354)
    }

    V3 surfaceLuminance = splat(0.0f);
    V3 base = splat(0.0f);   // non-march part of the primary term
    V3 coef = splat(0.0f);   // transmittanceToSurface * metallic * fresnel of the reflection
    V3 env2 = splat(0.0f);   // non-march part of the reflection's environment sample
    bool hasReflection = false;
    bool march0 = false, march1 = false;
    V3 o0 = position, d0 = direction, o1 = position, d1 = direction;
    float l0 = 0.0f, l1 = 0.0f;

    if (isSky)
    {
        // sampleEnvironmentLuminanceTransfer(position, direction, 1.0), camera.comp:286-301
        float dtg;
        if (raycastGround(a, position, direction, dtg))
        {
            base = groundSurfaceTerm(L, a, position, direction, dtg);
            march0 = true;
            l0 = dtg;
        }
        else
        {
            base = sampleMapDirection(S, a, position, direction) + sampleSunDisk(L, a, position, direction) * 1.0f;
        }
    }
    else
    {
        // camera.comp:364
        uint2 const prior = row_ptr<const uint2>(color, y)[x];
        surfaceLuminance = mk3((float)(prior.x & 0xFFFFu) / 65535.0f, (float)(prior.x >> 16) / 65535.0f,
                               (float)(prior.y & 0xFFFFu) / 65535.0f);

        // camera.comp:366-369: shadow frame from the engine-space position/normal
        float shadowFactor = 1.0f;
        if (sunSlot.map != nullptr)
        {
            // TO_TEX * (sun.projection * sun.view): once per frame by k_frame_prep (FramePrep::sunShadow), wave-uniform here
            M4 sm;
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
                sm.m[k] = prep->sunShadow[k];
            }
            V4 c = mul(sm, m.position.x, m.position.y, m.position.z, 1.0f);
            float const w = c.w;
            V3 const coord = mk3(c.x / w, c.y / w, c.z / w);
            V4 const pn = mul(sm, m.normal.x, m.normal.y, m.normal.z, 0.0f);
            float const fdx = sqrtf(1.0f - clampf(pn.x * pn.x, 0.0f, 1.0f));
            float const fdy = sqrtf(1.0f - clampf(pn.y * pn.y, 0.0f, 1.0f));
            shadowFactor = sampleShadowMap(sunSlot.map, sunSlot.width, sunSlot.height, sunSlot.pitchFloats, coord, fdx, fdy);
        }

        // camera.comp:371-374
        m.normal.y *= -1.0f;
        m.position.y *= -1.0f;
        m.position = m.position / 1000000.0f;
        m.position.y += a.planetRadius;

        // computeGeometryLuminanceTransfer, camera.comp:237-278
        V3 const surfacePosition = m.position;
        // the two transmittance samples of this branch with the lean exact operators when the whole wave allows it
        V3 const toSurface = surfacePosition - position;
        bool const samplesLean = waveAll(a.lean && leanRadius2(a, dot(position, position)) && leanRadius2(a, dot(surfacePosition, surfacePosition)) &&
                                         leanLength2(dot(toSurface, toSurface)) &&
                                         leanLength2(dot(a.incidentDirectionSun, a.incidentDirectionSun)));
        V3 const transmittanceToSurface = samplesLean ? sampleT_Segment<true>(L, a, position, surfacePosition)
                                                      : sampleT_Segment<false>(L, a, position, surfacePosition);
        // (normalize(), length() and the quotients below likewise: v * (1 / sqrt(dot)) and sqrt(dot) with the lean operators)
        V3 const minusSun = -a.incidentDirectionSun, minusDirection = -direction;
        float const direction2 = dot(minusDirection, minusDirection);
        bool const vectorsLean = samplesLean && waveAll(leanLength2(direction2));
        V3 const lightDirection = vectorsLean ? minusSun * divN0(1.0f, sqrtP(dot(minusSun, minusSun))) : normalize(minusSun);
        V3 const viewDirection = vectorsLean ? minusDirection * divN0(1.0f, sqrtP(direction2)) : normalize(minusDirection);
        float pt0 = 0.0f, pt1 = 0.0f;
        bool const shadowedByPlanet = raySphere(surfacePosition, lightDirection, a.planetRadius, pt0, pt1) && pt0 > 0.0f;
        V3 const halfSum = lightDirection + viewDirection;
        bool const brdfLean = vectorsLean && waveAll(inRange(dot(halfSum, halfSum), 0x1p-40f, 8.0f));
        V3 const brdf = brdfLean ? brdfMix<true>(m, lightDirection, viewDirection) : brdfMix<false>(m, lightDirection, viewDirection);
        V3 const transmittanceToSun = samplesLean ? sampleT_Ray<true>(L, a, surfacePosition, lightDirection)
                                                  : sampleT_Ray<false>(L, a, surfacePosition, lightDirection);
        float const fractionOfSunVisible = vectorsLean ? divN0(a.planetRadius, sqrtP(dot(m.position, m.position)))
                                                       : a.planetRadius / length(m.position); // camera.comp:144-147
        float const scalar = (shadowFactor * fractionOfSunVisible) * (shadowedByPlanet ? 0.0f : 1.0f);
        base = ((((scalar * transmittanceToSun) * transmittanceToSurface) * m.occlusion) * brdf) *
               clampf(dot(m.normal, lightDirection), 0.0f, 1.0f);
        l0 = vectorsLean ? sqrtP(dot(toSurface, toSurface)) : length(surfacePosition - position);
        if (FAST)
        {
            // APPROXIMATE extension (abi.h): aerial perspective from the froxel LUT instead of the inline march
            base = base + sampleAerial(aerial, (float)x / (float)drawW, (float)gy / (float)drawH, l0);
        }
        else
        {
            march0 = true;
        }

        // camera.comp:379-386: single-bounce reflection. With metallic == 0 the term is 0 * fresnel * environment: an
        // exact zero that may be skipped — unless the environment could be NaN or inf (0 * NaN = NaN poisons the pixel in
        // the reference). It cannot when the extinction is bounded away from 0 along every ray of the shell
        // (Atm::extModerate), the transmittance LUT is moderate, and both the camera (whose sky-view LUT is sampled)
        // and the surface (origin of the reflection ray) lie inside the shell, and the reflection direction itself is finite
        // (it is built from the G-buffer normal; found by tests/sweeps/random_sweep_mesh_frames.py: a NaN normal from a degenerate
        // triangle leaves the sun term at 0 through the clamps but makes the environment sample NaN); otherwise the term
        // is evaluated.
        float const cameraR2 = dot(position, position), surfaceR2 = dot(surfacePosition, surfacePosition);
        float const nBig = 0x1p60f;
        bool const normalFinite = fabsf(m.normal.x) <= nBig && fabsf(m.normal.y) <= nBig && fabsf(m.normal.z) <= nBig;
        // Both horizon angles asin(Rp / r) must exist: the camera's (every texel of the sky-view LUT depends on it) and the
        // surface's (where the LUT is sampled from); same expressions as skyview_LUT.comp:106-112 and camera.comp:75-77.
        float const cameraSinHorizon = vectorsLean ? divN0(a.planetRadius, sqrtP(cameraR2)) : a.planetRadius / length(position);
        bool const aboveGround = cameraSinHorizon <= 1.0f && fractionOfSunVisible <= 1.0f;
        bool const environmentFinite = a.extModerate && a.sunSane && L.moderate && S.finite && normalFinite && aboveGround &&
                                       inRange(cameraR2, a.extFloor2, a.extCeil2) && inRange(surfaceR2, a.extFloor2, a.extCeil2);
        if (m.metallic != 0.0f || !environmentFinite)
        {
            hasReflection = true;
            V3 const negDir = -direction;
            V3 const parallel = dot(m.normal, negDir) * m.normal;
            V3 const reflectionDirection = 2.0f * parallel - negDir;
            coef = (transmittanceToSurface * m.metallic) * computeFresnel(m, negDir, reflectionDirection);
            float dtg2;
            if (raycastGround(a, m.position, reflectionDirection, dtg2))
            {
                env2 = groundSurfaceTerm(L, a, m.position, reflectionDirection, dtg2);
                march1 = true;
                o1 = m.position;
                d1 = reflectionDirection;
                l1 = dtg2;
            }
            else
            {
                env2 = sampleMapDirection(S, a, m.position, reflectionDirection) +
                       sampleSunDisk(L, a, m.position, reflectionDirection) * shadowFactor;
            }
        }
    }

    // ---- phase B ---------------------------------------------------------
    // The march needs ~125 VGPRs by itself; what phase C wants back (12 floats) is parked in LDS meanwhile so
    // that the kernel fits 168 VGPRs = 3 waves per SIMD instead of 2 (LDS is otherwise unused here; [k][tid]
    // indexing is conflict-free).
    __shared__ float s_park[12][256];
    {
        float const park[12] = {base.x, base.y, base.z, coef.x, coef.y, coef.z, env2.x, env2.y, env2.z,
                                surfaceLuminance.x, surfaceLuminance.y, surfaceLuminance.z};
#pragma unroll
        for (int k = 0; k < 12; k++)
        {
            s_park[k][tid] = park[k];
        }
    }
    V3 ap0 = splat(0.0f), ap1 = splat(0.0f);
#pragma unroll 1
    for (int k = 0; k < 2; k++)
    {
        bool const active = (k == 0) ? march0 : march1;
        if (active)
        {
            V3 const o = (k == 0) ? o0 : o1;
            V3 const d = (k == 0) ? d0 : d1;
            float const l = (k == 0) ? l0 : l1;
            V3 const r = scatteringIntegral(L, a, o, d, l);
            if (k == 0)
            {
                ap0 = r;
            }
            else
            {
                ap1 = r;
            }
        }
    }

    base = mk3(s_park[0][tid], s_park[1][tid], s_park[2][tid]);
    coef = mk3(s_park[3][tid], s_park[4][tid], s_park[5][tid]);
    env2 = mk3(s_park[6][tid], s_park[7][tid], s_park[8][tid]);
    surfaceLuminance = mk3(s_park[9][tid], s_park[10][tid], s_park[11][tid]);
    // ---- phase C ---------------------------------------------------------
    // sky:      transfer = env(position, direction)
    // geometry: transfer = (surfaceTransfer + AP) + coef * env(reflection)
    V3 transfer = march0 ? (base + ap0) : base;
    if (hasReflection)
    {
        V3 const e2 = march1 ? (env2 + ap1) : env2;
        transfer = transfer + coef * e2;
    }
    V3 const luminance = transfer * a.sunIntensitySpectrum;
    V3 const pre = luminance * 10.0f + surfaceLuminance;
    V3 const out = mk3(szg_powf(pre.x, 1.2f), szg_powf(pre.y, 1.2f), szg_powf(pre.z, 1.2f));
    row_ptr<uint2>(color, y)[x] = pack_unorm16x4(out.x, out.y, out.z, 1.0f);
    if (debug.data != nullptr)
    {
        row_ptr<float4>(debug, y)[x] = make_float4(out.x, out.y, out.z, 1.0f);
    }
}

hipError_t launch_composite(hipStream_t s, const szg_scene_texture& scene, unsigned drawW, unsigned drawH, TileArgs tile,
                            const szg_gbuffer& g, ShadowSlot sunSlot, const szg_atmosphere_packed* d_atm, unsigned atmIndex, const szg_camera_packed* d_cam,
                            unsigned camIndex, const szg_directional_light_packed* d_dir, unsigned sunIndex, const float* tlut,
                            unsigned tW, unsigned tH, const float* slut, unsigned sW, unsigned sH, AerialLut aerial, const void* d_prep)
{
    unsigned const rows = tile.nranks <= 1u ? drawH : tile.local_rows;
    if (rows == 0u || drawW == 0u)
    {
        return hipSuccess;
    }
    dim3 const grid((drawW + 31u) / 32u, (rows + 7u) / 8u);
    RowMap const rm{tile.block_rows, tile.rank, tile.nranks};
    GBufferPtrsC const gp{g.diffuse, g.specular, g.normal, g.worldPosition, g.occlusionRoughnessMetallic};
    if (aerial.luminance != nullptr)
    {
        hipLaunchKernelGGL(k_composite<true>, grid, dim3(256), 0, s, scene.color, scene.depth, scene.debug_color, gp, drawW, drawH,
                           rows, rm, sunSlot, d_atm, atmIndex, d_cam, camIndex, d_dir, sunIndex,
                           reinterpret_cast<const float4*>(tlut), (int)tW, (int)tH, reinterpret_cast<const float4*>(slut),
                           (int)sW, (int)sH, aerial, static_cast<const FramePrep*>(d_prep));
    }
    else
    {
        hipLaunchKernelGGL(k_composite<false>, grid, dim3(256), 0, s, scene.color, scene.depth, scene.debug_color, gp, drawW, drawH,
                           rows, rm, sunSlot, d_atm, atmIndex, d_cam, camIndex, d_dir, sunIndex,
                           reinterpret_cast<const float4*>(tlut), (int)tW, (int)tH, reinterpret_cast<const float4*>(slut),
                           (int)sW, (int)sH, aerial, static_cast<const FramePrep*>(d_prep));
    }
    return hipGetLastError();
}
} // namespace szg
