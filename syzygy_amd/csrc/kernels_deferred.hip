// kernels_deferred.hip — G-buffer fill, light-list prep, deferred lights and the
// multi-GPU row-tile compose copy, for gfx950.
//
//   k_gbuffer_fill : output conventions of deferred/offscreen.frag:61-79 with the
//                    raster state of renderer/pipelines/deferred.cpp:342-392
//                    (analytic ray cast instead of the fixed-function rasteriser)
//   k_light_prep   : per-light invariants of deferred/lights.comp:141-161
//   k_lights       : deferred/lights.comp:110-164, fused with the clear of the
//                    scene colour (deferred.cpp:715-717)
//   k_compose      : HBM-bound row scatter after the RCCL gather

#include "szg_device.hpp"
#include "szg_launch.hpp"

namespace szg
{
namespace
{
// 256-thread workgroup = 32x8 pixels; each wave an 8x8 patch.
SZG_DEV void pixel_of_thread(unsigned& x, unsigned& y)
{
    unsigned const tid = threadIdx.x;
    unsigned const wave = tid >> 6, lane = tid & 63u;
    x = blockIdx.x * 32u + wave * 8u + (lane & 7u);
    y = blockIdx.y * 8u + (lane >> 3);
}
// The same tiling with the rows walked from the bottom of the image upwards (workgroups are dispatched in blockIdx order):
// the cheap workgroups - background texels, at the top of most frames - then come last and fill the tail of the launch.
SZG_DEV void pixel_of_thread_bottom_up(unsigned& x, unsigned& y)
{
    unsigned const tid = threadIdx.x;
    unsigned const wave = tid >> 6, lane = tid & 63u;
    x = blockIdx.x * 32u + wave * 8u + (lane & 7u);
    y = (gridDim.y - 1u - blockIdx.y) * 8u + (lane >> 3);
}

template <typename T> SZG_DEV T* row_ptr(const szg_image& im, unsigned y)
{
    return reinterpret_cast<T*>(static_cast<unsigned char*>(im.data) + (size_t)y * im.pitch_bytes);
}
} // namespace

struct GBufferPtrs
{
    szg_image diffuse, specular, normal, position, orm;
};

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gbuffer_fill(szg_image depth, GBufferPtrs g, unsigned drawW, unsigned drawH,
                                                      unsigned localRows, RowMap rm,
                                                      const szg_camera_packed* __restrict__ cameras, unsigned cameraIndex,
                                                      float ground_y, float ground_half_extent, float checker_cell,
                                                      float ground_roughness, const szg_fill_box* __restrict__ boxes,
                                                      unsigned boxCount)
{
    unsigned x, y;
    pixel_of_thread(x, y);
    if (x >= drawW || y >= localRows)
    {
        return;
    }
    unsigned const gy = global_row(rm, y);
    const szg_camera_packed* cam = cameras + cameraIndex;
    M4 const inverseProjection = load_m4(cam->inverseProjection);
    M4 const rotation = load_m4(cam->rotation);
    V3 const origin = mk3(cam->position[0], cam->position[1], cam->position[2]);

    float const ndcx = (((float)x + 0.5f) / (float)drawW - 0.5f) * 2.0f;
    float const ndcy = (((float)gy + 0.5f) / (float)drawH - 0.5f) * 2.0f;
    V4 const dv = mul(inverseProjection, ndcx, ndcy, 1.0f, 1.0f);
    V4 const dw = mul(rotation, dv.x, dv.y, dv.z, dv.w);
    V3 const dir = normalize(mk3(dw.x, dw.y, dw.z));

    float best_t = 3.0e38f;
    V3 best_n = splat(0.0f);
    float best_metallic = 0.0f, best_roughness = 0.0f;
    bool hit = false;

    if (dir.y != 0.0f)
    {
        float const t = (ground_y - origin.y) / dir.y;
        if (t > 0.0f)
        {
            V3 const p = origin + t * dir;
            if (fabsf(p.x) <= ground_half_extent && fabsf(p.z) <= ground_half_extent && t < best_t)
            {
                best_t = t;
                best_n = mk3(0.0f, -1.0f, 0.0f);
                best_metallic = 0.0f;
                best_roughness = ground_roughness;
                hit = true;
            }
        }
    }
    float const o[3] = {origin.x, origin.y, origin.z};
    float const d[3] = {dir.x, dir.y, dir.z};
    for (unsigned b = 0; b < boxCount; b++)
    {
        szg_fill_box const box = boxes[b];
        float tmin = -3.0e38f, tmax = 3.0e38f;
        int axis_min = 0;
        float sign_min = 0.0f;
        bool miss = false;
#pragma unroll
        for (int ax = 0; ax < 3; ax++)
        {
            float const lo = box.center[ax] - box.half_extent[ax];
            float const hi = box.center[ax] + box.half_extent[ax];
            if (d[ax] == 0.0f)
            {
                if (o[ax] < lo || o[ax] > hi)
                {
                    miss = true;
                }
                continue;
            }
            float t0 = (lo - o[ax]) / d[ax];
            float t1 = (hi - o[ax]) / d[ax];
            float s = -1.0f;
            if (t0 > t1)
            {
                float const tmp = t0;
                t0 = t1;
                t1 = tmp;
                s = 1.0f;
            }
            if (t0 > tmin)
            {
                tmin = t0;
                axis_min = ax;
                sign_min = s;
            }
            if (t1 < tmax)
            {
                tmax = t1;
            }
        }
        if (miss || tmin > tmax || tmin <= 0.0f)
        {
            continue;
        }
        if (tmin < best_t)
        {
            best_t = tmin;
            best_n = mk3(axis_min == 0 ? sign_min : 0.0f, axis_min == 1 ? sign_min : 0.0f, axis_min == 2 ? sign_min : 0.0f);
            best_metallic = box.metallic;
            best_roughness = box.roughness;
            hit = true;
        }
    }

    float depthOut = 0.0f;
    float4 pos4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint2 nrm = pack_half4(0.0f, 0.0f, 0.0f, 0.0f), dif = nrm, orm = nrm;
    if (hit)
    {
        V3 const p = origin + best_t * dir;
        M4 const projection = load_m4(cam->projection);
        M4 const view = load_m4(cam->view);
        V4 const pv = mul(view, p.x, p.y, p.z, 1.0f);
        V4 const clip = mul(projection, pv.x, pv.y, pv.z, pv.w);
        float const dz = clip.z / clip.w;
        if (dz > 0.0f && dz <= 1.0f)
        {
            depthOut = dz;
            float const cx = floorf(p.x / checker_cell), cy = floorf(p.y / checker_cell), cz = floorf(p.z / checker_cell);
            float const s = cx + cy + cz;
            bool const light = (s - 2.0f * floorf(s * 0.5f)) == 0.0f;
            float const grey = light ? (200.0f / 255.0f) : (100.0f / 255.0f);
            pos4 = make_float4(p.x, p.y, p.z, 1.0f);
            nrm = pack_half4(best_n.x, best_n.y, best_n.z, 0.0f);
            dif = pack_half4(grey, grey, grey, 1.0f);
            orm = pack_half4(1.0f, best_roughness, best_metallic, 1.0f);
        }
    }
    row_ptr<uint2>(g.diffuse, y)[x] = dif;
    row_ptr<uint2>(g.specular, y)[x] = dif;
    row_ptr<uint2>(g.normal, y)[x] = nrm;
    row_ptr<uint2>(g.orm, y)[x] = orm;
    row_ptr<float4>(g.position, y)[x] = pos4;
    row_ptr<float>(depth, y)[x] = depthOut;
}

// ---------------------------------------------------------------------------
// One thread per light. Slot numbering follows lights.comp:138-161: the shadow
// map index starts at directionalLightSkipCount and runs over the directional
// lights [skip, count) then the spot lights.
__global__ void k_light_prep(const szg_directional_light_packed* __restrict__ dirs, unsigned dirCount, unsigned dirSkip,
                             const szg_spot_light_packed* __restrict__ spots, unsigned spotCount,
                             const ShadowSlot* __restrict__ slots, unsigned slotCount, LightRec* __restrict__ out)
{
    unsigned const i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned const nDir = dirCount > dirSkip ? dirCount - dirSkip : 0u;
    if (i >= nDir + spotCount)
    {
        return;
    }
    // shadowmap.glinl:2-7
    M4 toTex;
    {
        float const t[16] = {0.5f, 0.0f, 0.0f, 0.0f, 0.0f, 0.5f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.5f, 0.5f, 0.0f, 1.0f};
#pragma unroll
        for (int k = 0; k < 16; k++)
        {
            toTex.m[k] = t[k];
        }
    }
    LightRec r;
    M4 projection, view;
    V3 forward, color;
    float strength;
    if (i < nDir)
    {
        const szg_directional_light_packed* l = dirs + dirSkip + i;
        projection = load_m4(l->projection);
        view = load_m4(l->view);
        forward = mk3(l->forward[0], l->forward[1], l->forward[2]);
        color = mk3(l->color[0], l->color[1], l->color[2]);
        strength = l->strength;
        r.isSpot = 0u;
        r.falloffFactor = 0.0f;
        r.falloffDistance = 1.0f;
        r.position[0] = r.position[1] = r.position[2] = 0.0f;
    }
    else
    {
        const szg_spot_light_packed* l = spots + (i - nDir);
        projection = load_m4(l->projection);
        view = load_m4(l->view);
        forward = mk3(l->forward[0], l->forward[1], l->forward[2]);
        color = mk3(l->color[0], l->color[1], l->color[2]);
        strength = l->strength;
        r.isSpot = 1u;
        r.falloffFactor = l->falloffFactor;
        r.falloffDistance = l->falloffDistance;
        r.position[0] = l->position[0];
        r.position[1] = l->position[1];
        r.position[2] = l->position[2];
    }
    // computeShadowFrame(light.projection * light.view, ...): TO_TEX * (projection * view)
    M4 const shadowMatrix = mul(toTex, mul(projection, view));
#pragma unroll
    for (int row = 0; row < 4; row++)
    {
#pragma unroll
        for (int col = 0; col < 4; col++)
        {
            r.shadowRows[row * 4 + col] = shadowMatrix.m[col * 4 + row];
        }
    }
    V3 const dirUnit = normalizeL(-forward);
    r.dir[0] = dirUnit.x;
    r.dir[1] = dirUnit.y;
    r.dir[2] = dirUnit.z;
    V3 const cs = color * strength;
    r.colorStrength[0] = cs.x;
    r.colorStrength[1] = cs.y;
    r.colorStrength[2] = cs.z;
    unsigned const slot = dirSkip + i;
    if (slot < slotCount && slots[slot].map != nullptr)
    {
        r.map = slots[slot].map;
        r.mapWidth = slots[slot].width;
        r.mapHeight = slots[slot].height;
        r.mapPitchFloats = slots[slot].pitchFloats;
    }
    else
    {
        r.map = nullptr;
        r.mapWidth = r.mapHeight = r.mapPitchFloats = 0u;
    }
    {
        float const lo = 0x1p-30f, hi = 0x1p30f;
        // colour * strength is the numerator of lean divisions (an infinite one would come out NaN instead of inf) and a
        // factor of the culled term: a spot light's term for a pixel outside its cone is (colour * strength / falloff) * 0 *
        // brdf, an exact zero only if every factor is finite. The light's own factors are checked here, the pixel's in k_lights.
        auto numerator = [&](float a) { return a == 0.0f || (fabsf(a) >= lo && fabsf(a) <= hi); };
        // rows of the shadow matrix: with |position| <= 2^30 (checked per pixel) entries up to 2^28 keep every projected
        // coordinate — the numerators of the lean divisions by w — below 2^60
        bool rowsModerate = true;
        for (int k = 0; k < 16; k++)
        {
            rowsModerate = rowsModerate && fabsf(r.shadowRows[k]) <= 0x1p28f;
        }
        bool const finite = numerator(r.colorStrength[0]) && numerator(r.colorStrength[1]) && numerator(r.colorStrength[2]) &&
                            fabsf(r.position[0]) <= hi && fabsf(r.position[1]) <= hi && fabsf(r.position[2]) <= hi &&
                            fabsf(r.dir[0]) <= 2.0f && fabsf(r.dir[1]) <= 2.0f && fabsf(r.dir[2]) <= 2.0f;
        r.leanOK = (r.isSpot != 0u && finite && rowsModerate && inRange(r.falloffDistance, lo, hi) && inRange(r.falloffFactor, lo, hi)) ? 1u : 0u;
    }
    r.rcpFalloffDistance = r.leanOK != 0u ? rcpN(r.falloffDistance) : 0.0f;
    r.falloffBound = r.falloffFactor * r.rcpFalloffDistance * r.rcpFalloffDistance;
    // The falloff of a lean pair is falloffBound * d^2 up to rounding, with d^2 in [2^-30, 2^30] (tested per pair): it is the
    // denominator of colour * strength / falloff, so it has to stay inside the domain of the exact reciprocal, [2^-60, 2^60]
    // (round 3: factor and distance were only bounded one by one before, which admitted falloffs up to 2^120 and quotients in
    // the denormal range, where the one-correction division is not the IEEE quotient).
    if (r.leanOK != 0u && !inRange(r.falloffBound, 0x1p-30f, 0x1p30f))
    {
        r.leanOK = 0u;
        r.rcpFalloffDistance = 0.0f;
        r.falloffBound = 0.0f;
    }
    if (r.leanOK != 0u)
    {
        bool tight = fabsf(r.position[0]) <= 0x1p12f && fabsf(r.position[1]) <= 0x1p12f && fabsf(r.position[2]) <= 0x1p12f &&
                     inRange(r.falloffBound, 0x1p-20f, 0x1p20f);
        for (int k = 0; k < 16; k++)
        {
            tight = tight && fabsf(r.shadowRows[k]) <= 0x1p14f;
        }
        r.leanOK |= tight ? 2u : 0u;
    }
    out[i] = r;
}

// ---------------------------------------------------------------------------
// deferred/lights.comp:110-164.
//
// Light records are wave-uniform, so they are read with scalar loads into SGPRs (VALU operands for free,
// no VGPR or LDS cost); the next light's cull rows are prefetched while the current light is evaluated.
// Per light a lane first runs a division-free conservative test ("surely outside the cone"), then the exact
// evaluation (the reference's arithmetic, including its exact zero for pixels outside the cone), in
// ascending light order, so the per-pixel sum adds the same non-zero terms in the same order as the reference
// loop. Measured on the C3 scene: 29 % of pixel-light pairs are lit (~19 lights per pixel), so the pass is bound
// by the exact BRDF evaluation rather than by culling; an LDS-staged variant (records copied to LDS once per
// workgroup, broadcast reads) was measured 20 % slower than scalar loads and dropped.
//
// A spot light contributes exactly 0 when distance(st, 0.5) / 0.5 >= 1 (lights.comp:84-88). |s - 0.5| > 0.505 on
// either axis implies that with a 1 % margin, far above any rounding of the exact evaluation, so rejecting there
// never changes a result — unless the other axis is NaN. cw == 0 (st = inf/NaN) is not rejected.

// Runtime (wave-uniform) selection between the lean exact ops of szg_device.hpp and the generic operators;
// both give the IEEE correctly rounded result, the lean ones only inside their operand ranges.
SZG_DEV float divU(bool lean, float a, float b, float y) { return lean ? divR(a, b, y) : a / b; }
// numerator of a lean division: 0, or not tiny (the residual of the one-correction division underflows below ~2^-100; its
// upper bound comes from the light's rows, checked once in k_light_prep)
SZG_DEV bool leanNumerator(float a) { return a == 0.0f || fabsf(a) >= 0x1p-60f; }
SZG_DEV float sqrtU(bool lean, float x) { return lean ? sqrtN(x) : sqrtf(x); }

// One light's term of the sum (lights.comp:141-161), exact.
// `clip`: shadowMatrix * vec4(position, 1), rows x, y, w summed left to right (projectRows below: the cone test and the
// exact evaluation share one evaluation of the three rows; R[k] * 1.0f of the shader's product is R[k] itself)
// `backCull` (per lane): the pixel's own factors are of ordinary magnitude (k_lights, pixelModerate), so that a term whose last
// factor clamp(N.L, 0, 1) is 0 is an exact zero and its BRDF (half of the cost of a lit pair) need not be evaluated.
//
// OPTIMISTIC (k_lights' first pass, only for lights whose record is "tight" and waves whose pixels all are, see there): the lean
// exact operators are used without testing their operand ranges pair by pair. Upper bounds hold by construction; the LOWER
// bounds (|w| and d^2 >= 2^-30, |half vector|^2 >= 2^-40, projected numerators >= 2^-60 in magnitude - a numerator of exactly
// 0 counts as out of range here, which only costs time) are folded into `track`, a running minimum per lane, scaled to the
// common threshold 2^-30: three v_min3 and three multiplies instead of fourteen compares and the mask arithmetic around
// them. Every operand is a finite number on this path (no NaN can hide from the minimum). k_lights checks `track` once after
// the loop and repeats the wave's loop with the tested form if it ever fell below the threshold.
template <bool OPTIMISTIC>
SZG_DEV V3 lightContribution(const LightRec& L, const Material& m, bool positionModerate, V3 viewDirection, bool cullable, V3 clip,
                             bool backCull, float d2, float& track)
{
    const float* R = L.shadowRows;
    float const cx = clip.x, cy = clip.y, cw = clip.z;
    bool const isSpot = L.isSpot != 0u;
    V3 const lightDir = mk3(L.dir[0], L.dir[1], L.dir[2]);
    // (d2 = dot(toLight, toLight) of distance(), lights.comp:80, formed by the caller in front of the cone test)
    // The surface faces away from the light: clamp(dot(N, L), 0, 1) = 0 (also for a NaN, fmax(NaN, 0) = 0) is the last factor
    // of ((occlusion * brdf) * spectral) * clamp(N.L) (lights.comp:106-107), so the term is +-0 - and sum + (+-0) == sum, the sum
    // being never -0 - provided the other factors are finite numbers whose product does not overflow:
    //   |occlusion * brdf| < 2^36 for a pixel of ordinary magnitude (k_lights, pixelModerate; clamp() turns a NaN half
    //   vector into 0, so brdf is a number whatever the directions are);
    //   spectral = ((colour * strength / falloff) * edge) * shadow with colour * strength <= 2^30 (k_light_prep, leanOK), edge and
    //   shadow in [0, 1] whatever the projected coordinates are, and falloff = factor * (dist / falloffDistance)^2 >= 2^-30:
    //   tested as falloffBound * d^2 >= 2^-28, the same quantity up to rounding, with a factor of 4 to spare (an infinite
    //   falloff gives spectral = 0).
    // Then nothing of this light needs evaluating for the pixel: a quarter of the lit pixel-light pairs of the bench scenes.
    float const ndl = dotL(m.normal, lightDir);
    if (backCull && cullable && isSpot && (OPTIMISTIC || L.leanOK != 0u) && !(ndl > 0.0f) && L.falloffBound * d2 >= 0x1p-28f)
    {
        return splat(0.0f);
    }
    V3 const hs = lightDir + viewDirection;
    float const hd = dotP(hs, hs);
    // Operand ranges of the lean ops used below: w of the projected position, squared distance to the light
    // (=> lightFalloff = factor * (dist / falloffDistance)^2 with the per-light constants checked by k_light_prep),
    // the half-vector length, and the position magnitude. One lane outside sends the wave down the generic path.
    float const lo = 0x1p-30f, hi = 0x1p30f;
    // (numerators too: the one-correction division returns NaN for an infinite numerator where the quotient is inf — a
    // projection with entries near FLT_MAX — and is not verified for denormal ones)
    float const cz = (L.map != nullptr)
                         ? SZG_CON(SZG_C_MATVEC, R[10], m.position.z, SZG_CON(SZG_C_MATVEC, R[9], m.position.y, R[8] * m.position.x)) + R[11]
                         : 0.0f;
    bool lean = true;
    if (OPTIMISTIC)
    {
        track = fminf(track, fminf(fabsf(cw), d2));
        track = fminf(track, fminf(hd * 0x1p10f, fabsf(cx) * 0x1p30f));
        track = fminf(track, fabsf(cy) * 0x1p30f);
        if (L.map != nullptr)
        {
            track = fminf(track, fabsf(cz) * 0x1p30f);
        }
    }
    else
    {
        lean = waveAll(L.leanOK != 0u && positionModerate && inRange(fabsf(cw), lo, hi) && inRange(d2, lo, hi) &&
                       inRange(hd, 0x1p-40f, 8.0f) && leanNumerator(cx) && leanNumerator(cy) && leanNumerator(cz));
    }

    float const ycw = lean ? rcpN(cw) : 0.0f;
    // (OPTIMISTIC: the quotients below are never zero - numerators >= 2^-60 over denominators < 2^28, positive lengths - so the
    // sign fix of a zero quotient, divR's last instruction, has nothing to do: divR0; and the square roots of d^2 and |h|^2 are
    // of numbers >= 2^-40: sqrtP)
    float const sx = OPTIMISTIC ? divR0(cx, cw, ycw) : divU(lean, cx, cw, ycw);
    float const sy = OPTIMISTIC ? divR0(cy, cw, ycw) : divU(lean, cy, cw, ycw);
    float edgeSoftening = 1.0f;
    float lightFalloff = 1.0f;
    if (isSpot)
    {
        // lights.comp:80-85. distanceUV = clamp(sqrt(q) / 0.5, 0, 1) and edgeSoftening = 1 - distanceUV^2 is
        // exactly 0 iff sqrt(q) * 2 >= 1 iff q >= 0.25 (sqrt is monotonic, sqrt(0.25) = 0.5 exactly), so the
        // cull needs no square root. x / 0.5 == x * 2 exactly.
        float const ddx = sx - 0.5f, ddy = sy - 0.5f;
        float const q = SZG_CON(SZG_C_LDOT, ddy, ddy, ddx * ddx); // dot(d, d) of distance()
        // (an exact zero only while colour * strength / falloff is finite: see falloffAwayFromZero in lightLoop)
        if (q >= 0.25f && cullable && L.falloffBound * d2 >= 0x1p-28f)
        {
            return splat(0.0f);
        }
        float const distanceUV = clampf(sqrtU(lean, q) * 2.0f, 0.0f, 1.0f);
        edgeSoftening = 1.0f - distanceUV * distanceUV;
        float const dist = OPTIMISTIC ? sqrtP(d2) : sqrtU(lean, d2);
        float const nd = OPTIMISTIC ? divR0(dist, L.falloffDistance, L.rcpFalloffDistance)
                                    : (lean ? divR(dist, L.falloffDistance, L.rcpFalloffDistance) : dist / L.falloffDistance);
        lightFalloff = L.falloffFactor * nd * nd;
    }
    float shadow = 1.0f;
    if (L.map != nullptr)
    {
        float const sz = OPTIMISTIC ? divR0(cz, cw, ycw) : divU(lean, cz, cw, ycw);
        // projectedNormal = shadowMatrix * vec4(normal, 0)
        float const nx = SZG_CON(SZG_C_MATVEC, R[3], 0.0f, SZG_CON(SZG_C_MATVEC, R[2], m.normal.z, SZG_CON(SZG_C_MATVEC, R[1], m.normal.y, R[0] * m.normal.x)));
        float const ny = SZG_CON(SZG_C_MATVEC, R[7], 0.0f, SZG_CON(SZG_C_MATVEC, R[6], m.normal.z, SZG_CON(SZG_C_MATVEC, R[5], m.normal.y, R[4] * m.normal.x)));
        float const fdx = sqrtf(1.0f - clampf(nx * nx, 0.0f, 1.0f));
        float const fdy = sqrtf(1.0f - clampf(ny * ny, 0.0f, 1.0f));
        shadow = sampleShadowMap(L.map, L.mapWidth, L.mapHeight, L.mapPitchFloats, mk3(sx, sy, sz), fdx, fdy);
    }
    V3 const cs = mk3(L.colorStrength[0], L.colorStrength[1], L.colorStrength[2]);
    // lights.comp:68 / :87-88
    V3 spectral;
    if (isSpot)
    {
        float const yf = lean ? rcpN(lightFalloff) : 0.0f;
        V3 const perFalloff = mk3(divU(lean, cs.x, lightFalloff, yf), divU(lean, cs.y, lightFalloff, yf), divU(lean, cs.z, lightFalloff, yf));
        spectral = (perFalloff * edgeSoftening) * shadow;
    }
    else
    {
        spectral = cs * shadow;
    }
    // computeLightContribution, lights.comp:93-108 (brdfMix of szg_device.hpp with the half vector shared above)
    float const hinv = OPTIMISTIC ? divN0(1.0f, sqrtP(hd)) : (lean ? divN(1.0f, sqrtN(hd)) : 1.0f / sqrtf(hd));
    V3 const h = hs * hinv;
    // both pow() without their special-case selects when no lane of the wave has a zero / denormal / non-finite base or a
    // zero exponent (szg_device.hpp powLean: the same values)
    float const baseSpecular = clampf(dotP(h, m.normal), 0.0f, 1.0f);
    float const baseFresnel = 1.0f - clampf(dotP(h, lightDir), 0.0f, 1.0f);
    // (OPTIMISTIC: both bases lie in [0, 1] and must be >= 2^-126 for powLean: tracked like the other lower bounds, x * 2^96
    // against 2^-30; the exponent's range is part of the wave's precondition in k_lights)
    bool powsLean = true;
    if (OPTIMISTIC)
    {
        track = fminf(track, fminf(baseSpecular * 0x1p96f, baseFresnel * 0x1p96f));
    }
    else
    {
        powsLean = waveAll(powLeanOK(baseSpecular, m.specularPower) && powLeanOK(baseFresnel, 5.0f));
    }
    float const microfacet = powsLean ? powLean(baseSpecular, m.specularPower) : szg_powf(baseSpecular, m.specularPower);
    V3 const specular = splat(m.normalization * microfacet);
    float const p = powsLean ? powLean(baseFresnel, 5.0f) : szg_powf(baseFresnel, 5.0f);
    V3 const fresnel = m.reflectance + (splat(1.0f) - m.reflectance) * p;
    V3 const brdf = mix(m.diffuse, specular, fresnel);
    // lights.comp:106-107
    return ((m.occlusion * brdf) * spectral) * clampf(ndl, 0.0f, 1.0f);
}

// The three rows of the shadow matrix the cone test needs (x, y, w) + the spot flag: what is prefetched.
struct LightCull
{
    float rx[4], ry[4], rw[4];
    float position[3], falloffBound;
    unsigned isSpot;
};
SZG_DEV LightCull loadCull(const LightRec* __restrict__ L)
{
    LightCull c;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        c.rx[k] = L->shadowRows[k];
        c.ry[k] = L->shadowRows[4 + k];
        c.rw[k] = L->shadowRows[12 + k];
    }
    c.position[0] = L->position[0];
    c.position[1] = L->position[1];
    c.position[2] = L->position[2];
    c.falloffBound = L->falloffBound;
    c.isSpot = L->isSpot;
    return c;
}
// rows x, y and w of shadowMatrix * vec4(p, 1), each summed left to right as the shader does
SZG_DEV V3 projectRows(const LightCull& c, V3 p)
{
    // (M * vec4(p, 1)).row as the oracle's operator* evaluates it: an fma chain over x, y, z, then + m3 * 1
    float const cx = SZG_CON(SZG_C_MATVEC, c.rx[2], p.z, SZG_CON(SZG_C_MATVEC, c.rx[1], p.y, c.rx[0] * p.x)) + c.rx[3];
    float const cy = SZG_CON(SZG_C_MATVEC, c.ry[2], p.z, SZG_CON(SZG_C_MATVEC, c.ry[1], p.y, c.ry[0] * p.x)) + c.ry[3];
    float const cw = SZG_CON(SZG_C_MATVEC, c.rw[2], p.z, SZG_CON(SZG_C_MATVEC, c.rw[1], p.y, c.rw[0] * p.x)) + c.rw[3];
    return V3{cx, cy, cw};
}
SZG_DEV bool surelyOutsideCone(V3 clip)
{
    float const cx = clip.x, cy = clip.y, cw = clip.z;
    float const acw = fabsf(cw);
    float const ax = fabsf(cx - 0.5f * cw);
    float const ay = fabsf(cy - 0.5f * cw);
    // (one axis outside implies q = dx^2 + dy^2 >= 0.25 only if the other axis is a number. It is: the test is only consulted
    // for a `cullable` pair - every position of the wave finite and <= 2^30, every row of the light <= 2^28 - so the three
    // projected coordinates are finite numbers below 2^60 and no inf - inf can occur.)
    return acw > 0.0f && (ax > 0.505f * acw || ay > 0.505f * acw);
}

// The loop over the lights for one pixel (lights.comp:141-161): terms added in ascending light order.
template <bool OPTIMISTIC>
SZG_DEV V3 lightLoop(const LightRec* lights, unsigned lightCount, const Material& m, bool positionModerate, V3 viewDirection,
                     bool waveFinite, bool pixelModerate, float& track)
{
    V3 sum = splat(0.0f);
#pragma unroll 1
    for (unsigned i = 0; i < lightCount; i++)
    {
        const LightRec* __restrict__ L = lights + i;
        LightCull const cur = loadCull(L); // (prefetching light i+1's rows here measured 20 % slower)
        unsigned const flags = L->leanOK;
        bool const cullable = waveFinite && (flags & 1u) != 0u;
        V3 const clip = projectRows(cur, m.position);
        // A pixel outside the cone contributes (colour * strength / falloff) * 0 * ...: an exact zero ONLY while the quotient
        // is finite. The falloff factor * (distance / falloffDistance)^2 is 0 AT the light's position and underflows next to
        // it, and inf * 0 = NaN poisons the pixel in the reference (found in round 3 by a test that puts G-buffer positions
        // on the lights: rounds 1 and 2 returned 0 there). So the squared distance is formed in front of the cone test, and
        // the two cone culls - like the back-face cull - require falloffBound * d^2 >= 2^-28 (falloff >= 2^-30 with a factor of
        // 4 for the roundings): closer pixels are evaluated in full.
        V3 const toLight = mk3(cur.position[0], cur.position[1], cur.position[2]) - m.position;
        float const d2 = dotL(toLight, toLight);
        bool const falloffAwayFromZero = cur.falloffBound * d2 >= 0x1p-28f;
        if (cur.isSpot != 0u && cullable && falloffAwayFromZero && surelyOutsideCone(clip))
        {
            continue;
        }
        if (OPTIMISTIC && (flags & 2u) != 0u)
        {
            sum = sum + lightContribution<true>(*L, m, positionModerate, viewDirection, cullable, clip, pixelModerate, d2, track);
        }
        else
        {
            sum = sum + lightContribution<false>(*L, m, positionModerate, viewDirection, cullable, clip, pixelModerate, d2, track);
        }
    }
    return sum;
}

__global__ __launch_bounds__(256, 6) void k_lights(szg_image color, szg_image debug, GBufferPtrs g, unsigned drawW,
                                                unsigned localRows, const szg_camera_packed* __restrict__ cameras,
                                                unsigned cameraIndex, const LightRec* __restrict__ lights, unsigned lightCount)
{
#ifdef SZG_EXP_LDS_LIGHTS
    // EXPERIMENT (profiles/r03_experiments.md, north_star "LDS-staged light lists"; not built into libszg_hip.so): the light
    // records copied to LDS once per workgroup and read from there (broadcast ds_read into VGPRs) instead of scalar loads
    // into SGPRs. Up to 264 lights (37 KiB: the 256 spots + the moon of C5).
    __shared__ LightRec s_lights[264];
    {
        unsigned const n = min(lightCount, 264u) * (unsigned)(sizeof(LightRec) / 4u);
        const unsigned* src = reinterpret_cast<const unsigned*>(lights);
        unsigned* dst = reinterpret_cast<unsigned*>(s_lights);
        for (unsigned k = threadIdx.x; k < n; k += 256u)
        {
            dst[k] = src[k];
        }
        __syncthreads();
    }
#define SZG_LIGHTS_PTR s_lights /* (the experiment is run with <= 264 lights) */
#else
#define SZG_LIGHTS_PTR lights
#endif
    unsigned x, y;
    pixel_of_thread_bottom_up(x, y);
    if (x >= drawW || y >= localRows)
    {
        return;
    }
    V4 const diffuse = unpack_half4(row_ptr<const uint2>(g.diffuse, y)[x]);
    V3 sum = splat(0.0f);
    // lights.comp:126-129: background texels keep the clear colour (0,0,0,1)
    if (!(diffuse.w < 1.0f))
    {
        V4 const specular = unpack_half4(row_ptr<const uint2>(g.specular, y)[x]);
        V4 const normal = unpack_half4(row_ptr<const uint2>(g.normal, y)[x]);
        V4 const orm = unpack_half4(row_ptr<const uint2>(g.orm, y)[x]);
        float4 const p4 = row_ptr<const float4>(g.position, y)[x];
        Material const m = convertPBR(V4{p4.x, p4.y, p4.z, p4.w}, normal, diffuse, specular, orm);
        const szg_camera_packed* cam = cameras + cameraIndex;
        V3 const viewDirection = normalizeL(mk3(cam->position[0], cam->position[1], cam->position[2]) - m.position);

        float const hi = 0x1p30f;
        bool const positionModerate = fabsf(m.position.x) <= hi && fabsf(m.position.y) <= hi && fabsf(m.position.z) <= hi;
        // every per-pixel factor of a light's term is finite (see k_light_prep): only then is 0 * term an exact zero and a
        // pixel outside a spot light's cone may skip that light (a NaN anywhere poisons the sum in the reference)
        float const big = 0x1p120f;
        bool const pixelFinite = positionModerate && fabsf(viewDirection.x) <= 2.0f && fabsf(viewDirection.y) <= 2.0f &&
                                 fabsf(viewDirection.z) <= 2.0f && fabsf(m.normal.x) <= big && fabsf(m.normal.y) <= big &&
                                 fabsf(m.normal.z) <= big && fabsf(m.diffuse.x) <= big && fabsf(m.diffuse.y) <= big &&
                                 fabsf(m.diffuse.z) <= big && fabsf(m.reflectance.x) <= big && fabsf(m.reflectance.y) <= big &&
                                 fabsf(m.reflectance.z) <= big && fabsf(m.occlusion) <= big && fabsf(m.normalization) <= big &&
                                 fabsf(m.specularPower) <= big;
        // ... and of ordinary magnitude (every real material: roughness in [0, 1] gives a specular power <= 160): then
        // |occlusion * mix(diffuse, specular, fresnel)| < 2^36, see lightContribution's back-face test
        bool const pixelModerate = pixelFinite && fabsf(m.diffuse.x) <= 0x1p16f && fabsf(m.diffuse.y) <= 0x1p16f &&
                                   fabsf(m.diffuse.z) <= 0x1p16f && fabsf(m.reflectance.x) <= 4.0f && fabsf(m.reflectance.y) <= 4.0f &&
                                   fabsf(m.reflectance.z) <= 4.0f && fabsf(m.occlusion) <= 0x1p16f && fabsf(m.normalization) <= 0x1p8f &&
                                   fabsf(m.normal.x) <= 0x1p16f && fabsf(m.normal.y) <= 0x1p16f && fabsf(m.normal.z) <= 0x1p16f;
        // Light records are wave-uniform: they are fetched with scalar loads and live in SGPRs. Culling is decided per
        // wave: when some pixel of the wave has a non-finite factor the whole wave evaluates every light (no term of its
        // sum may be dropped); the flag of the light itself is wave-uniform anyway.
        bool const waveFinite = waveAll(pixelFinite);
        // The optimistic pass (lightContribution<true>): every pixel of the wave finite and within 4096 units of the origin.
        bool const tightPixel = pixelFinite && fabsf(m.position.x) <= 0x1p12f && fabsf(m.position.y) <= 0x1p12f && fabsf(m.position.z) <= 0x1p12f &&
                                inRange(fabsf(m.specularPower), 0x1p-100f, 0x1p20f); // (the exponent half of powLeanOK)
        bool done = false;
        if (waveAll(tightPixel))
        {
            float track = 0x1p100f;
            sum = lightLoop<true>(SZG_LIGHTS_PTR, lightCount, m, positionModerate, viewDirection, waveFinite, pixelModerate, track);
            done = waveAll(track >= 0x1p-30f);
        }
        if (!done)
        {
            float unused = 0.0f;
            sum = lightLoop<false>(SZG_LIGHTS_PTR, lightCount, m, positionModerate, viewDirection, waveFinite, pixelModerate, unused);
        }
    }
    row_ptr<uint2>(color, y)[x] = pack_unorm16x4(sum.x, sum.y, sum.z, 1.0f);
    if (debug.data != nullptr)
    {
        row_ptr<float4>(debug, y)[x] = make_float4(sum.x, sum.y, sum.z, 1.0f);
    }
}

// ---------------------------------------------------------------------------
// Row scatter after the gather: rank r's packed tile holds its rows in local
// order; local row l of rank r is global row ((l / B) * N + r) * B + l % B.
// 16 B per lane, fully coalesced both sides.
__global__ __launch_bounds__(256) void k_compose(const uint4* __restrict__ gathered, size_t tileStrideVec, unsigned nranks,
                                                 unsigned blockRows, unsigned char* __restrict__ dst, unsigned dstPitch,
                                                 unsigned rowVecs, unsigned height)
{
    unsigned const gy = blockIdx.y;
    unsigned const blk = gy / blockRows;
    unsigned const rank = blk % nranks;
    unsigned const local = (blk / nranks) * blockRows + gy % blockRows;
    const uint4* src = gathered + (size_t)rank * tileStrideVec + (size_t)local * rowVecs;
    uint4* out = reinterpret_cast<uint4*>(dst + (size_t)gy * dstPitch);
    for (unsigned v = blockIdx.x * 256u + threadIdx.x; v < rowVecs; v += gridDim.x * 256u)
    {
        out[v] = src[v];
    }
    (void)height;
}

// ---------------------------------------------------------------------------
// Shadow-map generation for the analytic scene (abi.h szg_deferred_record_shadow_maps).
__global__ void k_shadow_prep(const szg_directional_light_packed* __restrict__ dirs, unsigned dirCount,
                              const szg_spot_light_packed* __restrict__ spots, unsigned spotCount,
                              const ShadowSlot* __restrict__ owned, unsigned slotCount, ShadowGen* __restrict__ gen)
{
    unsigned const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slotCount)
    {
        return;
    }
    ShadowGen g;
    g.map = nullptr;
    g.dim = g.pitchFloats = g.pad = 0u;
    M4 pv;
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
        pv.m[k] = 0.0f;
    }
    if (i < dirCount + spotCount && owned[i].map != nullptr)
    {
        // light.projection * light.view (shadowpass.cpp:207-215)
        // (a host-side product in the reference: glm semantics, not the shaders' contraction rule)
        pv = (i < dirCount) ? mulGlm(load_m4(dirs[i].projection), load_m4(dirs[i].view))
                            : mulGlm(load_m4(spots[i - dirCount].projection), load_m4(spots[i - dirCount].view));
        g.map = const_cast<float*>(owned[i].map);
        g.dim = owned[i].width;
        g.pitchFloats = owned[i].pitchFloats;
    }
    M4 const inv = inverse4(pv);
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
        g.projView[k] = pv.m[k];
        g.invProjView[k] = inv.m[k];
    }
    gen[i] = g;
}

__global__ __launch_bounds__(256) void k_shadow_fill(const ShadowGen* __restrict__ gen, const szg_fill_box* __restrict__ boxes,
                                                     unsigned boxCount)
{
    const ShadowGen* G = gen + blockIdx.z;
    if (G->map == nullptr)
    {
        return;
    }
    unsigned x, y;
    pixel_of_thread(x, y);
    unsigned const dim = G->dim;
    if (x >= dim || y >= dim)
    {
        return;
    }
    M4 pv, inv;
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
        pv.m[k] = G->projView[k];
        inv.m[k] = G->invProjView[k];
    }
    // texel centre -> NDC (inverse of TO_TEX_COORD_MAT, shadowmap.glinl:2-7): s = 0.5 ndc + 0.5
    float const ndcx = (((float)x + 0.5f) / (float)dim) * 2.0f - 1.0f;
    float const ndcy = (((float)y + 0.5f) / (float)dim) * 2.0f - 1.0f;
    V4 const h1 = mul(inv, ndcx, ndcy, 1.0f, 1.0f); // reverse-Z: depth 1 = near plane
    V4 const h2 = mul(inv, ndcx, ndcy, 0.5f, 1.0f);
    V3 const p1 = mk3(h1.x / h1.w, h1.y / h1.w, h1.z / h1.w);
    V3 const p2 = mk3(h2.x / h2.w, h2.y / h2.w, h2.z / h2.w);
    V3 const dir = normalize(p2 - p1);
    float const o[3] = {p1.x, p1.y, p1.z};
    float const d[3] = {dir.x, dir.y, dir.z};
    float best = 0.0f; // cleared depth = far
    for (unsigned b = 0; b < boxCount; b++)
    {
        szg_fill_box const box = boxes[b];
        float tmin = -3.0e38f, tmax = 3.0e38f;
        bool miss = false;
#pragma unroll
        for (int ax = 0; ax < 3; ax++)
        {
            float const lo = box.center[ax] - box.half_extent[ax];
            float const hi = box.center[ax] + box.half_extent[ax];
            if (d[ax] == 0.0f)
            {
                if (o[ax] < lo || o[ax] > hi)
                {
                    miss = true;
                }
                continue;
            }
            float t0 = (lo - o[ax]) / d[ax];
            float t1 = (hi - o[ax]) / d[ax];
            if (t0 > t1)
            {
                float const tmp = t0;
                t0 = t1;
                t1 = tmp;
            }
            tmin = fmaxf(tmin, t0);
            tmax = fminf(tmax, t1);
        }
        if (miss || tmin > tmax || tmax <= 0.0f)
        {
            continue;
        }
        V3 const pe = p1 + tmax * dir; // back face: what front-face culling leaves (pipelines.cpp:656-659)
        V4 const clip = mul(pv, pe.x, pe.y, pe.z, 1.0f);
        float const depth = clip.z / clip.w;
        if (depth > 0.0f && depth <= 1.0f && depth >= best) // GREATER_OR_EQUAL (pipelines.cpp:661)
        {
            best = depth;
        }
    }
    G->map[(size_t)y * G->pitchFloats + x] = best;
}

// ---------------------------------------------------------------------------
// transfer/oetf_srgb.comp:9-19, transfer/oetf_pure_gamma.comp:9 — in place on RGBA16 UNORM. Two pixels (16 B)
// per lane per access, fully coalesced; 16 B/px of traffic.
SZG_DEV float oetf(float linear, unsigned function)
{
    if (function == SZG_OETF_SRGB)
    {
        float const lower = 12.92f * linear;
        float const higher = szg_powf(linear, (float)(1.0 / 2.4)) * 1.055f - 0.055f;
        return (linear <= 0.0031308f) ? lower : higher; // mix(higher, lower, cutoff)
    }
    // (float)(1.0 / 2.2), not 1.0f / 2.2f: the constant of the reference's SPIR-V, folded in double (oracle_oetf)
    return szg_powf(linear, (float)(1.0 / 2.2));
}
SZG_DEV unsigned oetf_pair(unsigned packed, unsigned function, bool hiIsAlpha)
{
    float const a = (float)(packed & 0xFFFFu) / 65535.0f;
    float const b = (float)(packed >> 16) / 65535.0f;
    unsigned const lo = unorm16(oetf(a, function));
    unsigned const hi = hiIsAlpha ? (packed >> 16) : unorm16(oetf(b, function));
    return lo | (hi << 16);
}
// The OETF is a pure function of a 16-bit channel value: 65 536 possible inputs. k_oetf_table evaluates the
// expression above once per input; k_oetf then maps every channel through the 128 KiB table (L2-resident gathers),
// which leaves the pass with ~20 instructions per pixel against 16 B of traffic: HBM-bound, and bit-identical to
// evaluating the three pow() per pixel.
__global__ __launch_bounds__(256) void k_oetf_table(unsigned short* __restrict__ table, unsigned function)
{
    unsigned const i = blockIdx.x * 256u + threadIdx.x; // 256 x 256 threads
    table[i] = (unsigned short)unorm16(oetf((float)i / 65535.0f, function));
}
SZG_DEV unsigned oetf_pair_lut(unsigned packed, const unsigned short* __restrict__ table, bool hiIsAlpha)
{
    unsigned const lo = table[packed & 0xFFFFu];
    unsigned const hi = hiIsAlpha ? (packed >> 16) : (unsigned)table[packed >> 16];
    return lo | (hi << 16);
}
__global__ __launch_bounds__(256) void k_oetf(unsigned char* __restrict__ data, unsigned pitch, unsigned width, unsigned height,
                                             const unsigned short* __restrict__ table)
{
    unsigned const pairs = (width + 1u) / 2u; // uint4 = 2 pixels
    for (unsigned y = blockIdx.y; y < height; y += gridDim.y)
    {
        uint4* row = reinterpret_cast<uint4*>(data + (size_t)y * pitch);
        for (unsigned v = blockIdx.x * 256u + threadIdx.x; v < pairs; v += gridDim.x * 256u)
        {
            if (2u * v + 1u < width)
            {
                uint4 t = row[v];
                t.x = oetf_pair_lut(t.x, table, false);
                t.y = oetf_pair_lut(t.y, table, true);
                t.z = oetf_pair_lut(t.z, table, false);
                t.w = oetf_pair_lut(t.w, table, true);
                row[v] = t;
            }
            else
            {
                uint2* px = reinterpret_cast<uint2*>(row) + 2u * v; // odd width: last single pixel
                uint2 t = *px;
                t.x = oetf_pair_lut(t.x, table, false);
                t.y = oetf_pair_lut(t.y, table, true);
                *px = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------
static GBufferPtrs gptrs(const szg_gbuffer& g)
{
    return GBufferPtrs{g.diffuse, g.specular, g.normal, g.worldPosition, g.occlusionRoughnessMetallic};
}
static unsigned local_rows_of(const TileArgs& t, unsigned drawH) { return t.nranks <= 1u ? drawH : t.local_rows; }

hipError_t launch_gbuffer_fill(hipStream_t s, const szg_scene_texture& scene, unsigned drawW, unsigned drawH, TileArgs tile,
                               const szg_gbuffer& g, const szg_camera_packed* d_cam, unsigned camIndex, float ground_y,
                               float ground_half_extent, float checker_cell, float ground_roughness,
                               const szg_fill_box* d_boxes, unsigned boxCount)
{
    unsigned const rows = local_rows_of(tile, drawH);
    if (rows == 0u || drawW == 0u)
    {
        return hipSuccess;
    }
    dim3 const grid((drawW + 31u) / 32u, (rows + 7u) / 8u);
    RowMap const rm{tile.block_rows, tile.rank, tile.nranks};
    hipLaunchKernelGGL(k_gbuffer_fill, grid, dim3(256), 0, s, scene.depth, gptrs(g), drawW, drawH, rows, rm, d_cam, camIndex,
                       ground_y, ground_half_extent, checker_cell, ground_roughness, d_boxes, boxCount);
    return hipGetLastError();
}

hipError_t launch_light_prep(hipStream_t s, const szg_directional_light_packed* d_dir, unsigned dirCount, unsigned dirSkip,
                             const szg_spot_light_packed* d_spot, unsigned spotCount, const ShadowSlot* d_slots,
                             unsigned slotCount, LightRec* d_out)
{
    unsigned const nDir = dirCount > dirSkip ? dirCount - dirSkip : 0u;
    unsigned const n = nDir + spotCount;
    if (n == 0u)
    {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_light_prep, dim3((n + 63u) / 64u), dim3(64), 0, s, d_dir, dirCount, dirSkip, d_spot, spotCount, d_slots,
                       slotCount, d_out);
    return hipGetLastError();
}

hipError_t launch_lights(hipStream_t s, const szg_scene_texture& scene, unsigned drawW, unsigned drawH, TileArgs tile,
                         const szg_gbuffer& g, const szg_camera_packed* d_cam, unsigned camIndex, const LightRec* d_lights,
                         unsigned lightCount)
{
    unsigned const rows = local_rows_of(tile, drawH);
    if (rows == 0u || drawW == 0u)
    {
        return hipSuccess;
    }
    dim3 const grid((drawW + 31u) / 32u, (rows + 7u) / 8u);
    hipLaunchKernelGGL(k_lights, grid, dim3(256), 0, s, scene.color, scene.debug_color, gptrs(g), drawW, rows, d_cam, camIndex,
                       d_lights, lightCount);
    return hipGetLastError();
}

hipError_t launch_shadow_maps(hipStream_t s, const szg_directional_light_packed* d_dir, unsigned dirCount,
                              const szg_spot_light_packed* d_spot, unsigned spotCount, const ShadowSlot* d_ownedSlots,
                              unsigned slotCount, ShadowGen* d_gen, const szg_fill_box* d_boxes, unsigned boxCount, unsigned maxDim)
{
    if (slotCount == 0u || maxDim == 0u)
    {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_shadow_prep, dim3((slotCount + 63u) / 64u), dim3(64), 0, s, d_dir, dirCount, d_spot, spotCount, d_ownedSlots,
                       slotCount, d_gen);
    hipLaunchKernelGGL(k_shadow_fill, dim3((maxDim + 31u) / 32u, (maxDim + 7u) / 8u, slotCount), dim3(256), 0, s, d_gen, d_boxes,
                       boxCount);
    return hipGetLastError();
}

hipError_t launch_shadow_prep(hipStream_t s, const szg_directional_light_packed* d_dir, unsigned dirCount,
                              const szg_spot_light_packed* d_spot, unsigned spotCount, const ShadowSlot* d_ownedSlots,
                              unsigned slotCount, ShadowGen* d_gen)
{
    if (slotCount == 0u)
    {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_shadow_prep, dim3((slotCount + 63u) / 64u), dim3(64), 0, s, d_dir, dirCount, d_spot, spotCount, d_ownedSlots,
                       slotCount, d_gen);
    return hipGetLastError();
}

hipError_t launch_oetf_table(hipStream_t s, unsigned short* table, unsigned function)
{
    hipLaunchKernelGGL(k_oetf_table, dim3(256), dim3(256), 0, s, table, function);
    return hipGetLastError();
}

hipError_t launch_oetf(hipStream_t s, const szg_image& image, unsigned width, unsigned height, const unsigned short* table)
{
    if (width == 0u || height == 0u)
    {
        return hipSuccess;
    }
    unsigned const pairs = (width + 1u) / 2u;
    unsigned gx = (pairs + 255u) / 256u;
    gx = gx > 8u ? 8u : gx;
    unsigned const gy = height > 65535u ? 65535u : height;
    hipLaunchKernelGGL(k_oetf, dim3(gx, gy), dim3(256), 0, s, static_cast<unsigned char*>(image.data), image.pitch_bytes, width, height,
                       table);
    return hipGetLastError();
}

hipError_t launch_compose_rowtiles(hipStream_t s, const void* gathered, size_t tileStrideBytes, unsigned nranks,
                                   unsigned blockRows, const szg_image& dst, unsigned width, unsigned height)
{
    if (width == 0u || height == 0u)
    {
        return hipSuccess;
    }
    unsigned const rowBytes = width * 8u; // RGBA16_UNORM
    unsigned const rowVecs = rowBytes / 16u;
    unsigned const gx = (rowVecs + 255u) / 256u;
    hipLaunchKernelGGL(k_compose, dim3(gx > 8u ? 8u : gx, height), dim3(256), 0, s, static_cast<const uint4*>(gathered),
                       tileStrideBytes / 16u, nranks, blockRows, static_cast<unsigned char*>(dst.data), dst.pitch_bytes, rowVecs,
                       height);
    return hipGetLastError();
}
} // namespace szg
