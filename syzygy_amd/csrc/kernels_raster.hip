// kernels_raster.hip — compute rasteriser for gfx950: the G-buffer pass and the shadow depth passes of
// DeferredShadingPipeline::recordDrawCommands (include/szg/raster.h states the rules and cites the reference).
//
//   k_raster_setup   one lane per primitive in submission order: vertex stage for its three vertices, homogeneous
//                    edge functions, facing / trivial rejection, screen bounding box, and a sort key (box size class,
//                    Morton code of the box centre)
//   (rocPRIM radix sort of the keys when there are more than a few thousand primitives: kernels_raster_sort.hip)
//   k_raster_chunks / k_raster_superchunks   box hierarchy over the (sorted) order: 64 primitives per chunk,
//                    64 chunks per super-chunk, union boxes by wave reductions
//   k_raster_tile    256 threads = 32x8 pixels, each wave an 8x8 patch that walks the hierarchy: super-chunk box vs
//                    patch (scalar), 64 chunk boxes in parallel (one per lane, ballot), 64 primitive boxes in parallel,
//                    then the surviving primitives one by one with wave-uniform coefficients (scalar loads) — depth
//                    test in registers, no atomics; the winner is shaded (offscreen.frag) and all five planes + depth
//                    are written once: 52 B/px, coalesced
//   k_shadow_tile    the same walk, depth only (front faces culled, GREATER_OR_EQUAL, depth bias)

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

SZG_DEV V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// submission-order primitive -> (draw, instance, triangle): draws are sorted by firstPrim
SZG_DEV unsigned findDraw(const RasterDraw* __restrict__ draws, unsigned drawCount, unsigned prim)
{
    unsigned lo = 0u, hi = drawCount;
    while (hi - lo > 1u)
    {
        unsigned const mid = (lo + hi) >> 1;
        if (draws[mid].firstPrim <= prim)
        {
            lo = mid;
        }
        else
        {
            hi = mid;
        }
    }
    return lo;
}

struct VertexOut
{
    V4 clip;
    V3 world, normal;
    V2 uv;
};

// offscreen.vert:41-56 (shadow == false) / depthpass.vert:30-38 (shadow == true)
template <bool SHADOW>
SZG_DEV VertexOut vertexStage(const RasterDraw& d, unsigned instance, unsigned index, const M4& projView)
{
    VertexOut o;
    szg_vertex_packed const v = d.vertices[index];
    M4 const model = load_m4(d.models[instance]);
    if (SHADOW)
    {
        M4 const pvm = mul(projView, model);
        o.clip = mul(pvm, v.position[0], v.position[1], v.position[2], 1.0f);
        o.world = splat(0.0f);
        o.normal = splat(0.0f);
        o.uv = V2{0.0f, 0.0f};
    }
    else
    {
        V4 const position = mul(model, v.position[0], v.position[1], v.position[2], 1.0f);
        o.world = mk3(position.x, position.y, position.z);
        o.clip = mul(projView, position.x, position.y, position.z, position.w);
        M4 const mit = load_m4(d.mits[instance]);
        V4 const n = mul(mit, v.normal[0], v.normal[1], v.normal[2], 0.0f);
        o.normal = normalize(mk3(n.x, n.y, n.z));
        o.uv = V2{v.uv_x, v.uv_y};
    }
    return o;
}

SZG_DEV unsigned packBox(int lo, int hi) { return (unsigned)lo | ((unsigned)hi << 16); }
SZG_DEV unsigned spread14(unsigned v) // 14 bits -> every other bit of 28
{
    v &= 0x3FFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
SZG_DEV unsigned morton14(unsigned x, unsigned y) { return spread14(x) | (spread14(y) << 1); }
unsigned const EMPTY_BOX = 65535u | (0u << 16); // min 65535 > max 0: overlaps no patch

SZG_DEV int waveMin(int v)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
        v = min(v, __shfl_xor(v, off));
    }
    return v;
}
SZG_DEV int waveMax(int v)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
        v = max(v, __shfl_xor(v, off));
    }
    return v;
}
} // namespace

// ---------------------------------------------------------------------------
// One wave per chunk of 64 primitives. `target` selects the projection: the camera (G-buffer) or one shadow slot.
template <bool SHADOW>
__global__ __launch_bounds__(64) void k_raster_setup(const RasterDraw* __restrict__ draws, unsigned drawCount, unsigned primCount,
                                                     const szg_camera_packed* __restrict__ cameras, unsigned cameraIndex,
                                                     const ShadowGen* __restrict__ gen, unsigned W, unsigned H,
                                                     PrimRec* __restrict__ prims, uint2* __restrict__ boxes,
                                                     unsigned* __restrict__ keys, unsigned* __restrict__ order)
{
    unsigned const p = blockIdx.x * 64u + threadIdx.x;
    int minX = 65535, maxX = 0, minY = 65535, maxY = 0; // empty
    if (p < primCount)
    {
        M4 projView;
        if (SHADOW)
        {
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
                projView.m[k] = gen->projView[k];
            }
        }
        else
        {
            // offscreen.vert:51: `camera.projection * camera.view * position` multiplies the matrices first
            projView = mul(load_m4(cameras[cameraIndex].projection), load_m4(cameras[cameraIndex].view));
        }
        unsigned const di = findDraw(draws, drawCount, p);
        RasterDraw const d = draws[di];
        unsigned const local = p - d.firstPrim;
        unsigned const instance = local / d.triCount;
        unsigned const tri = local - instance * d.triCount;
        unsigned const i0 = d.indices[d.firstIndex + tri * 3u], i1 = d.indices[d.firstIndex + tri * 3u + 1u],
                       i2 = d.indices[d.firstIndex + tri * 3u + 2u];
        PrimRec r;
        r.draw = di;
        r.instance = instance;
        r.tri = tri;
        r.pad[0] = r.pad[1] = 0u;
        bool valid = i0 < d.vertexCount && i1 < d.vertexCount && i2 < d.vertexCount;
        if (valid)
        {
            V4 clip[3];
            clip[0] = vertexStage<SHADOW>(d, instance, i0, projView).clip;
            clip[1] = vertexStage<SHADOW>(d, instance, i1, projView).clip;
            clip[2] = vertexStage<SHADOW>(d, instance, i2, projView).clip;
            // raster.h "coverage"
            float const halfW = (float)W * 0.5f, halfH = (float)H * 0.5f;
            float hx[3], hy[3], hw[3];
            bool allBehind = true, xl = true, xr = true, yt = true, yb = true, zn = true, zf = true, allFront = true;
#pragma unroll
            for (int i = 0; i < 3; i++)
            {
                V4 const c = clip[i];
                valid = valid && (c.x == c.x) && (c.y == c.y) && (c.z == c.z) && (c.w == c.w);
                allBehind = allBehind && (c.w <= 0.0f);
                allFront = allFront && (c.w > 0.0f);
                xl = xl && (c.x < -c.w);
                xr = xr && (c.x > c.w);
                yt = yt && (c.y < -c.w);
                yb = yb && (c.y > c.w);
                zf = zf && (c.z < 0.0f);
                zn = zn && (c.z > c.w);
                hx[i] = (c.x + c.w) * halfW;
                hy[i] = (c.y + c.w) * halfH;
                hw[i] = c.w;
                r.z[i] = c.z;
                r.w[i] = c.w;
            }
            valid = valid && !(allBehind || xl || xr || yt || yb || zn || zf);
#pragma unroll
            for (int i = 0; i < 3; i++)
            {
                int const j = (i + 1) % 3, k = (i + 2) % 3;
                r.a[i] = hy[j] * hw[k] - hy[k] * hw[j];
                r.b[i] = hx[k] * hw[j] - hx[j] * hw[k];
                r.c[i] = hx[j] * hy[k] - hx[k] * hy[j];
            }
            float const det = (hx[0] * r.a[0] + hy[0] * r.b[0]) + hw[0] * r.c[0];
            valid = valid && (det > 0.0f || det < 0.0f);
            bool const front = det > 0.0f;
            valid = valid && (SHADOW ? !front : front); // shadow pass culls FRONT faces, G-buffer pass BACK faces
            if (!front)
            {
#pragma unroll
                for (int i = 0; i < 3; i++)
                {
                    r.a[i] = -r.a[i];
                    r.b[i] = -r.b[i];
                    r.c[i] = -r.c[i];
                }
            }
            if (valid)
            {
                // Conservative pixel box when every vertex is in front of the eye, else the viewport. Coverage is DEFINED by
                // the fp32 edge functions (raster.h), and |error of e_i| <= noise_i = 2^-22 (A px + B py + C) with A, B, C the sums
                // of the magnitudes of the products a_i, b_i, c_i are made of (two roundings each, three more in the
                // evaluation). Every pixel the fp32 test can accept therefore lies in {e_i > -noise_i for all i}: the
                // triangle T' bounded by the three edge lines pushed out by their noise. The box is the bounding box of T'
                // (+- 1 px for its own rounding): the projected triangle itself for ordinary primitives, a few pixels more
                // for distant tiny ones, and — because nearly parallel lines meet far away — the whole viewport where the
                // edge functions are rounding noise (a camera a million units away) or the triangle projects to a line (a
                // zero column in the projection): regions the oracle, which tests every pixel, shades too.
                float fx0 = 0.0f, fx1 = (float)W, fy0 = 0.0f, fy1 = (float)H;
                bool boxed = false; // the box below is the bounding box of T' (not the viewport fallback)
                if (allFront)
                {
                    float pushed[3];
#pragma unroll
                    for (int i = 0; i < 3; i++)
                    {
                        int const j = (i + 1) % 3, k = (i + 2) % 3;
                        float const A = fabsf(hy[j] * hw[k]) + fabsf(hy[k] * hw[j]);
                        float const B = fabsf(hx[k] * hw[j]) + fabsf(hx[j] * hw[k]);
                        float const Cc = fabsf(hx[j] * hy[k]) + fabsf(hx[k] * hy[j]);
                        pushed[i] = r.c[i] + 0x1p-22f * ((A * ((float)W + 1.0f) + B * ((float)H + 1.0f)) + Cc);
                    }
                    float x[3], y[3];
                    bool defined = true;
#pragma unroll
                    for (int v = 0; v < 3; v++)
                    {
                        // vertex v of T': where the pushed lines of the other two edges meet
                        int const i = (v + 1) % 3, j = (v + 2) % 3;
                        float const p0 = r.a[i] * r.b[j], p1 = r.a[j] * r.b[i];
                        float const D = p0 - p1;
                        x[v] = (r.b[i] * pushed[j] - r.b[j] * pushed[i]) / D;
                        y[v] = (r.a[j] * pushed[i] - r.a[i] * pushed[j]) / D;
                        // The three half-planes bound a triangle only if their normals turn the same way round (for the exact
                        // coefficients D = det * w_v > 0); with coefficients that are themselves rounding noise they may
                        // enclose an unbounded wedge instead, which no three corner points contain. NaN fails the test too.
                        defined = defined && (D > 0x1p-21f * (fabsf(p0) + fabsf(p1))) && (x[v] == x[v]) && (y[v] == y[v]);
                    }
                    if (defined)
                    {
                        boxed = true;
                        fx0 = fmaxf(fminf(fminf(x[0], x[1]), x[2]) - 1.0f, 0.0f);
                        fx1 = fminf(fmaxf(fmaxf(x[0], x[1]), x[2]) + 1.0f, (float)W);
                        fy0 = fmaxf(fminf(fminf(y[0], y[1]), y[2]) - 1.0f, 0.0f);
                        fy1 = fminf(fmaxf(fmaxf(y[0], y[1]), y[2]) + 1.0f, (float)H);
                    }
                }
                minX = (int)fx0;
                minY = (int)fy0;
                maxX = min((int)fx1, (int)W - 1);
                maxY = min((int)fy1, (int)H - 1);
                if (boxed && (fx0 > fx1 || fy0 > fy1))
                {
                    // the bounding box of T' is finite (its corners are numbers: `defined`) and misses the viewport: a
                    // primitive wholly off-screen that the per-plane rejection did not catch. No pixel can pass the
                    // edge tests, so it is culled like an invalid one instead of being handed to every pixel patch.
                    minX = 65535;
                    maxX = 0;
                    minY = 65535;
                    maxY = 0;
                }
                else if (!(fx0 <= fx1) || !(fy0 <= fy1)) // NaN-safe: keep the viewport
                {
                    minX = 0;
                    minY = 0;
                    maxX = (int)W - 1;
                    maxY = (int)H - 1;
                }
            }
        }
        if (!valid)
        {
            minX = 65535;
            maxX = 0;
            minY = 65535;
            maxY = 0;
        }
        prims[p] = r;
        boxes[p] = make_uint2(packBox(minX, maxX), packBox(minY, maxY));
        // sort key: large boxes first (they would bloat every chunk they land in), then Morton order of the centre
        unsigned key = 0xFFFFFFFFu; // culled primitives last
        if (minX <= maxX)
        {
            unsigned const extent = (unsigned)max(maxX - minX, maxY - minY) + 1u;
            unsigned const sizeClass = 15u - min(15u, 31u - (unsigned)__builtin_clz(extent)); // 0 = largest
            key = (sizeClass << 28) | morton14((unsigned)(minX + maxX) >> 2, (unsigned)(minY + maxY) >> 2);
        }
        keys[p] = key;
        order[p] = p;
    }
}

// Box hierarchy over `order` (submission order, or the sorted one): one wave per chunk of 64 primitives.
__global__ __launch_bounds__(64) void k_raster_chunks(const unsigned* __restrict__ order, const uint2* __restrict__ boxes,
                                                      unsigned primCount, uint2* __restrict__ orderedBoxes,
                                                      uint2* __restrict__ chunkBoxes)
{
    unsigned const pos = blockIdx.x * 64u + threadIdx.x;
    uint2 box = make_uint2(EMPTY_BOX, EMPTY_BOX);
    if (pos < primCount)
    {
        box = boxes[order[pos]];
        orderedBoxes[pos] = box;
    }
    int const x0 = waveMin((int)(box.x & 0xFFFFu)), x1 = waveMax((int)(box.x >> 16));
    int const y0 = waveMin((int)(box.y & 0xFFFFu)), y1 = waveMax((int)(box.y >> 16));
    if (threadIdx.x == 0u)
    {
        chunkBoxes[blockIdx.x] = (x0 <= x1 && y0 <= y1) ? make_uint2(packBox(x0, x1), packBox(y0, y1)) : make_uint2(EMPTY_BOX, EMPTY_BOX);
    }
}
// one wave per super-chunk of 64 chunks
__global__ __launch_bounds__(64) void k_raster_superchunks(const uint2* __restrict__ chunkBoxes, unsigned chunkCount,
                                                           uint2* __restrict__ superBoxes)
{
    unsigned const c = blockIdx.x * 64u + threadIdx.x;
    uint2 const box = c < chunkCount ? chunkBoxes[c] : make_uint2(EMPTY_BOX, EMPTY_BOX);
    int const x0 = waveMin((int)(box.x & 0xFFFFu)), x1 = waveMax((int)(box.x >> 16));
    int const y0 = waveMin((int)(box.y & 0xFFFFu)), y1 = waveMax((int)(box.y >> 16));
    if (threadIdx.x == 0u)
    {
        superBoxes[blockIdx.x] = (x0 <= x1 && y0 <= y1) ? make_uint2(packBox(x0, x1), packBox(y0, y1)) : make_uint2(EMPTY_BOX, EMPTY_BOX);
    }
}

namespace
{
SZG_DEV bool boxOverlaps(uint2 box, int x0, int x1, int y0, int y1)
{
    int const bx0 = (int)(box.x & 0xFFFFu), bx1 = (int)(box.x >> 16);
    int const by0 = (int)(box.y & 0xFFFFu), by1 = (int)(box.y >> 16);
    return bx0 <= x1 && bx1 >= x0 && by0 <= y1 && by1 >= y0;
}

// wave-uniform primitive record (scalar loads), per-lane pixel centre
SZG_DEV void edgeFunctions(const PrimRec* __restrict__ t, float px, float py, float e[3])
{
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        e[i] = (t->a[i] * px + t->b[i] * py) + t->c[i];
    }
}
SZG_DEV bool coversPixel(const PrimRec* __restrict__ t, const float e[3])
{
    bool in = true;
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        // top-left rule for a centre exactly on the edge (raster.h)
        in = in && (e[i] > 0.0f || (e[i] == 0.0f && (t->a[i] > 0.0f || (t->a[i] == 0.0f && t->b[i] > 0.0f))));
    }
    return in;
}
SZG_DEV bool fragmentDepth(const PrimRec* __restrict__ t, const float e[3], float& depth)
{
    float const zc = (e[0] * t->z[0] + e[1] * t->z[1]) + e[2] * t->z[2];
    float const wc = (e[0] * t->w[0] + e[1] * t->w[1]) + e[2] * t->w[2];
    bool const ok = zc >= 0.0f && zc <= wc && wc > 0.0f;
    depth = zc / wc;
    return ok;
}

// The box hierarchy a tile kernel walks (all in the order of `order`).
struct Hierarchy
{
    const uint2* superBoxes;
    const uint2* chunkBoxes;
    const uint2* orderedBoxes;
    const unsigned* order;
    unsigned primCount;
};
// Walk for one 8x8 patch; `visit(primIndex)` is called with a wave-uniform submission-order index.
template <typename F> SZG_DEV void walkPrimitives(const Hierarchy& h, int x0, int x1, int y0, int y1, F&& visit)
{
    unsigned const lane = threadIdx.x & 63u;
    unsigned const chunks = (h.primCount + 63u) / 64u;
    unsigned const supers = (chunks + 63u) / 64u;
    for (unsigned sc = 0; sc < supers; sc++)
    {
        if (!boxOverlaps(h.superBoxes[sc], x0, x1, y0, y1)) // uniform address: scalar load
        {
            continue;
        }
        unsigned const c = sc * 64u + lane;
        unsigned long long chunkMask = __ballot(c < chunks && boxOverlaps(h.chunkBoxes[c], x0, x1, y0, y1));
        while (chunkMask != 0ull)
        {
            unsigned const chunk = sc * 64u + (unsigned)__builtin_ctzll(chunkMask);
            chunkMask &= chunkMask - 1ull;
            unsigned const pos = chunk * 64u + lane;
            unsigned long long mask = __ballot(pos < h.primCount && boxOverlaps(h.orderedBoxes[pos], x0, x1, y0, y1));
            while (mask != 0ull)
            {
                unsigned const bit = (unsigned)__builtin_ctzll(mask);
                mask &= mask - 1ull;
                visit(h.order[chunk * 64u + bit]);
            }
        }
    }
}

struct Varyings
{
    V3 world, normal;
    V2 uv;
};
SZG_DEV float lerp3(const float l[3], float a0, float a1, float a2) { return (l[0] * a0 + l[1] * a1) + l[2] * a2; }
SZG_DEV Varyings interpolate(const PrimRec& t, const VertexOut v[3], float px, float py)
{
    float e[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        e[i] = (t.a[i] * px + t.b[i] * py) + t.c[i];
    }
    float const S = (e[0] + e[1]) + e[2];
    float const l[3] = {e[0] / S, e[1] / S, e[2] / S};
    Varyings o;
    o.world = mk3(lerp3(l, v[0].world.x, v[1].world.x, v[2].world.x), lerp3(l, v[0].world.y, v[1].world.y, v[2].world.y),
                  lerp3(l, v[0].world.z, v[1].world.z, v[2].world.z));
    o.normal = mk3(lerp3(l, v[0].normal.x, v[1].normal.x, v[2].normal.x), lerp3(l, v[0].normal.y, v[1].normal.y, v[2].normal.y),
                   lerp3(l, v[0].normal.z, v[1].normal.z, v[2].normal.z));
    o.uv = V2{lerp3(l, v[0].uv.x, v[1].uv.x, v[2].uv.x), lerp3(l, v[0].uv.y, v[1].uv.y, v[2].uv.y)};
    return o;
}

// raster.h "textures": RGBA8, LINEAR, REPEAT, one level
SZG_DEV float decode8(unsigned b, bool srgb)
{
    float const c = (float)b / 255.0f;
    if (!srgb)
    {
        return c;
    }
    return c <= 0.04045f ? c / 12.92f : szg_powf((c + 0.055f) / 1.055f, 2.4f);
}
SZG_DEV int wrapIndex(float f, int n)
{
    float const fn = (float)n;
    float const m = f - fn * floorf(f / fn);
    int i = (int)m;
    if (i >= n || i < 0)
    {
        i = 0;
    }
    return i;
}
// `unormTable[b]` = decode8(b, false), `srgbTable[b]` = decode8(b, true): the 36 texel decodes of a pixel are LDS
// look-ups of values each computed once per workgroup by the same expression.
SZG_DEV V3 sampleTexture(const szg_texture& tex, V2 st, const float* unormTable, const float* srgbTable)
{
    if (tex.data == nullptr || tex.width == 0u || tex.height == 0u)
    {
        return splat(0.0f);
    }
    int const W = (int)tex.width, H = (int)tex.height;
    float const u = st.x * (float)W - 0.5f;
    float const v = st.y * (float)H - 0.5f;
    float const fu = floorf(u), fv = floorf(v);
    float const a = u - fu, b = v - fv;
    int const i0 = wrapIndex(fu, W), j0 = wrapIndex(fv, H);
    int const i1 = (i0 + 1 == W) ? 0 : i0 + 1, j1 = (j0 + 1 == H) ? 0 : j0 + 1;
    const unsigned char* base = static_cast<const unsigned char*>(tex.data);
    unsigned const t00 = *reinterpret_cast<const unsigned*>(base + (size_t)j0 * tex.pitch_bytes + (size_t)i0 * 4u);
    unsigned const t10 = *reinterpret_cast<const unsigned*>(base + (size_t)j0 * tex.pitch_bytes + (size_t)i1 * 4u);
    unsigned const t01 = *reinterpret_cast<const unsigned*>(base + (size_t)j1 * tex.pitch_bytes + (size_t)i0 * 4u);
    unsigned const t11 = *reinterpret_cast<const unsigned*>(base + (size_t)j1 * tex.pitch_bytes + (size_t)i1 * 4u);
    const float* const table = tex.srgb != 0u ? srgbTable : unormTable;
    float const w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float r[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
    {
        unsigned const sh = (unsigned)ch * 8u;
        r[ch] = w00 * table[(t00 >> sh) & 0xFFu] + w10 * table[(t10 >> sh) & 0xFFu] + w01 * table[(t01 >> sh) & 0xFFu] +
                w11 * table[(t11 >> sh) & 0xFFu];
    }
    return mk3(r[0], r[1], r[2]);
}

// deferred/offscreen.frag:25-59
SZG_DEV V3 perturbNormal(const szg_texture& normalMap, V3 N, V3 dPosDx, V3 dPosDy, V2 dUvDx, V2 dUvDy, V2 texcoord,
                         const float* unormTable, const float* srgbTable)
{
    V3 map = sampleTexture(normalMap, texcoord, unormTable, srgbTable);
    float const k = 128.0f / 127.0f;
    map = mk3(map.x * 255.0f / 127.0f - k, map.y * 255.0f / 127.0f - k, map.z * 255.0f / 127.0f - k); // :47
    map.y = -map.y;                                                                                    // :50
    V3 const dp1 = -dPosDx; // cotangentFrame(N, -V, uv) with V = inWorldPosition (:54, :65)
    V3 const dp2 = -dPosDy;
    V3 const dp2perp = cross3(dp2, N);
    V3 const dp1perp = cross3(N, dp1);
    V3 const T = dp2perp * dUvDx.x + dp1perp * dUvDy.x;
    V3 const B = dp2perp * dUvDx.y + dp1perp * dUvDy.y;
    float const invmax = 1.0f / sqrtf(fmaxf(dot(T, T), dot(B, B)));
    V3 const c0 = T * invmax, c1 = B * invmax;
    V3 const v = (c0 * map.x + c1 * map.y) + N * map.z;
    return normalize(v);
}
} // namespace

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_raster_tile(szg_image depth, szg_image gDiffuse, szg_image gSpecular, szg_image gNormal,
                                                     szg_image gPosition, szg_image gOrm, unsigned drawW, unsigned drawH,
                                                     unsigned localRows, RowMap rm, const RasterDraw* __restrict__ draws,
                                                     const PrimRec* __restrict__ prims, Hierarchy hier,
                                                     const szg_camera_packed* __restrict__ cameras, unsigned cameraIndex)
{
    unsigned const tid = threadIdx.x;
    // RGBA8 decode tables (one entry per thread, both by the expression of decode8)
    __shared__ float s_unorm[256];
    __shared__ float s_srgb[256];
    s_unorm[tid] = decode8(tid, false);
    s_srgb[tid] = decode8(tid, true);
    __syncthreads();
    unsigned const wave = tid >> 6, lane = tid & 63u;
    unsigned const patchX = blockIdx.x * 32u + wave * 8u;
    unsigned const patchY = blockIdx.y * 8u;
    if (patchX >= drawW || patchY >= localRows)
    {
        return; // whole wave
    }
    unsigned const x = patchX + (lane & 7u);
    unsigned const y = patchY + (lane >> 3);
    bool const inFrame = x < drawW && y < localRows;
    unsigned const gy = global_row(rm, inFrame ? y : patchY);
    // patch rectangle in global pixels (the row map is monotonic)
    unsigned const lastRow = min(patchY + 7u, localRows - 1u);
    int const x0 = (int)patchX, x1 = (int)min(patchX + 7u, drawW - 1u);
    int const y0 = (int)global_row(rm, patchY), y1 = (int)global_row(rm, lastRow);

    float const px = (float)x + 0.5f, py = (float)gy + 0.5f;
    float best = 0.0f; // cleared depth; compare GREATER (deferred.cpp:383-386)
    unsigned winner = 0xFFFFFFFFu;
    walkPrimitives(hier, x0, x1, y0, y1, [&](unsigned p) {
        const PrimRec* t = prims + p;
        float e[3];
        edgeFunctions(t, px, py, e);
        float d;
        bool const ok = fragmentDepth(t, e, d);
        // GREATER in submission order: the walk order is arbitrary, so an equal depth goes to the earlier primitive — among
        // fragments; a depth equal to the clear value (0, also as an underflow of a huge primitive) is not GREATER than it
        if (coversPixel(t, e) && ok && (d > best || (d == best && winner != 0xFFFFFFFFu && p < winner)))
        {
            best = d;
            winner = p;
        }
    });
    if (!inFrame)
    {
        return;
    }

    float4 pos4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint2 nrm = pack_half4(0.0f, 0.0f, 0.0f, 0.0f), dif = nrm, orm = nrm;
    if (winner != 0xFFFFFFFFu)
    {
        PrimRec const t = prims[winner];
        RasterDraw const d = draws[t.draw];
        M4 const projView = mul(load_m4(cameras[cameraIndex].projection), load_m4(cameras[cameraIndex].view));
        unsigned const base = d.firstIndex + t.tri * 3u;
        VertexOut v[3];
        v[0] = vertexStage<false>(d, t.instance, d.indices[base], projView);
        v[1] = vertexStage<false>(d, t.instance, d.indices[base + 1u], projView);
        v[2] = vertexStage<false>(d, t.instance, d.indices[base + 2u], projView);
        Varyings const in = interpolate(t, v, px, py);
        // fine derivatives over the 2x2 quad (raster.h "derivatives"): right minus left, bottom minus top. One of the
        // two pixels of each difference is this pixel itself, so only the other one is interpolated.
        bool const isLeft = (x & 1u) == 0u, isTop = (gy & 1u) == 0u;
        Varyings const ox = interpolate(t, v, isLeft ? px + 1.0f : px - 1.0f, py);
        Varyings const oy = interpolate(t, v, px, isTop ? py + 1.0f : py - 1.0f);
        Varyings const xl = isLeft ? in : ox, xr = isLeft ? ox : in;
        Varyings const yt = isTop ? in : oy, yb = isTop ? oy : in;
        V2 const dUvDx{xr.uv.x - xl.uv.x, xr.uv.y - xl.uv.y}, dUvDy{yb.uv.x - yt.uv.x, yb.uv.y - yt.uv.y};
        V3 const N = perturbNormal(d.tex[1], in.normal, xr.world - xl.world, yb.world - yt.world, dUvDx, dUvDy, in.uv, s_unorm, s_srgb);
        V3 const color = sampleTexture(d.tex[0], in.uv, s_unorm, s_srgb);
        V3 const o = sampleTexture(d.tex[2], in.uv, s_unorm, s_srgb);
        dif = pack_half4(color.x, color.y, color.z, 1.0f);  // offscreen.frag:72, :75
        nrm = pack_half4(N.x, N.y, N.z, 0.0f);               // :68
        orm = pack_half4(o.x, o.y, o.z, 1.0f);               // :79
        pos4 = make_float4(in.world.x, in.world.y, in.world.z, 1.0f); // :63
    }
    row_ptr<uint2>(gDiffuse, y)[x] = dif;
    row_ptr<uint2>(gSpecular, y)[x] = dif;
    row_ptr<uint2>(gNormal, y)[x] = nrm;
    row_ptr<uint2>(gOrm, y)[x] = orm;
    row_ptr<float4>(gPosition, y)[x] = pos4;
    row_ptr<float>(depth, y)[x] = best;
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_shadow_tile(const ShadowGen* __restrict__ gen, const PrimRec* __restrict__ prims,
                                                     Hierarchy hier, float biasConstant, float biasSlope)
{
    if (gen->map == nullptr)
    {
        return;
    }
    unsigned const dim = gen->dim;
    unsigned const tid = threadIdx.x;
    unsigned const wave = tid >> 6, lane = tid & 63u;
    unsigned const patchX = blockIdx.x * 32u + wave * 8u;
    unsigned const patchY = blockIdx.y * 8u;
    if (patchX >= dim || patchY >= dim)
    {
        return;
    }
    unsigned const x = patchX + (lane & 7u);
    unsigned const y = patchY + (lane >> 3);
    int const x0 = (int)patchX, x1 = (int)min(patchX + 7u, dim - 1u);
    int const y0 = (int)patchY, y1 = (int)min(patchY + 7u, dim - 1u);
    float const px = (float)x + 0.5f, py = (float)y + 0.5f;
    bool const biased = biasConstant != 0.0f || biasSlope != 0.0f;
    float best = 0.0f; // cleared depth; compare GREATER_OR_EQUAL (pipelines.cpp:663)
    walkPrimitives(hier, x0, x1, y0, y1, [&](unsigned p) {
        const PrimRec* t = prims + p;
        float e[3];
        edgeFunctions(t, px, py, e);
        float d;
        bool const ok = fragmentDepth(t, e, d) && coversPixel(t, e);
        if (biased)
        {
            // Vulkan depth bias o = m * slope + r * constant (raster.h / oracle_shadow_raster)
            float ex[3], ey[3];
            edgeFunctions(t, px + 1.0f, py, ex);
            edgeFunctions(t, px, py + 1.0f, ey);
            float const zx = ((ex[0] * t->z[0] + ex[1] * t->z[1]) + ex[2] * t->z[2]) / ((ex[0] * t->w[0] + ex[1] * t->w[1]) + ex[2] * t->w[2]);
            float const zy = ((ey[0] * t->z[0] + ey[1] * t->z[1]) + ey[2] * t->z[2]) / ((ey[0] * t->w[0] + ey[1] * t->w[1]) + ey[2] * t->w[2]);
            float const m = fmaxf(fabsf(zx - d), fabsf(zy - d));
            int exponent = 0;
            (void)frexpf(d, &exponent);
            float const r = ldexpf(1.0f, (exponent - 1) - 23);
            float const o = m * biasSlope + r * biasConstant;
            d = fminf(fmaxf(d + o, 0.0f), 1.0f);
        }
        if (ok && d >= best)
        {
            best = d;
        }
    });
    if (x < dim && y < dim)
    {
        gen->map[(size_t)y * gen->pitchFloats + x] = best;
    }
}

// ---------------------------------------------------------------------------
namespace
{
Hierarchy hierarchyOf(const RasterBuffers& b, unsigned primCount)
{
    return Hierarchy{b.superBoxes, b.chunkBoxes, b.orderedBoxes, b.order, primCount};
}
} // namespace

// Setup + (sort) + box hierarchy for `primCount` primitives; afterwards b.order is the walk order.
hipError_t launch_raster_setup(hipStream_t s, bool shadow, const RasterDraw* d_draws, unsigned drawCount, unsigned primCount,
                               const szg_camera_packed* d_cam, unsigned camIndex, const ShadowGen* d_gen, unsigned W, unsigned H,
                               RasterBuffers& b)
{
    if (primCount == 0u)
    {
        return hipSuccess;
    }
    unsigned const chunks = (primCount + 63u) / 64u;
    b.order = b.valsA;
    if (shadow)
    {
        hipLaunchKernelGGL(k_raster_setup<true>, dim3(chunks), dim3(64), 0, s, d_draws, drawCount, primCount, d_cam, camIndex, d_gen, W, H,
                           b.prims, b.boxes, b.keysA, b.valsA);
    }
    else
    {
        hipLaunchKernelGGL(k_raster_setup<false>, dim3(chunks), dim3(64), 0, s, d_draws, drawCount, primCount, d_cam, camIndex, d_gen, W, H,
                           b.prims, b.boxes, b.keysA, b.valsA);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        return e;
    }
    // Few primitives: every patch can afford to look at all chunk boxes, keep the submission order.
    if (primCount > RASTER_SORT_THRESHOLD)
    {
        e = raster_sort_pairs(s, b.sortTemp, b.sortTempBytes, b.keysA, b.keysB, b.valsA, b.valsB, primCount);
        if (e != hipSuccess)
        {
            return e;
        }
        b.order = b.valsB;
    }
    hipLaunchKernelGGL(k_raster_chunks, dim3(chunks), dim3(64), 0, s, b.order, b.boxes, primCount, b.orderedBoxes, b.chunkBoxes);
    hipLaunchKernelGGL(k_raster_superchunks, dim3((chunks + 63u) / 64u), dim3(64), 0, s, b.chunkBoxes, chunks, b.superBoxes);
    return hipGetLastError();
}

hipError_t launch_raster_tile(hipStream_t s, const szg_scene_texture& scene, unsigned drawW, unsigned drawH, TileArgs tile,
                              const szg_gbuffer& g, const RasterDraw* d_draws, const RasterBuffers& b, unsigned primCount,
                              const szg_camera_packed* d_cam, unsigned camIndex)
{
    unsigned const rows = tile.nranks <= 1u ? drawH : tile.local_rows;
    if (rows == 0u || drawW == 0u)
    {
        return hipSuccess;
    }
    dim3 const grid((drawW + 31u) / 32u, (rows + 7u) / 8u);
    RowMap const rm{tile.block_rows, tile.rank, tile.nranks};
    hipLaunchKernelGGL(k_raster_tile, grid, dim3(256), 0, s, scene.depth, g.diffuse, g.specular, g.normal, g.worldPosition,
                       g.occlusionRoughnessMetallic, drawW, drawH, rows, rm, d_draws, b.prims, hierarchyOf(b, primCount), d_cam, camIndex);
    return hipGetLastError();
}

hipError_t launch_shadow_tile(hipStream_t s, const ShadowGen* d_gen, unsigned dim, const RasterBuffers& b, unsigned primCount,
                              float biasConstant, float biasSlope)
{
    if (dim == 0u)
    {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_shadow_tile, dim3((dim + 31u) / 32u, (dim + 7u) / 8u), dim3(256), 0, s, d_gen, b.prims, hierarchyOf(b, primCount),
                       biasConstant, biasSlope);
    return hipGetLastError();
}
} // namespace szg
