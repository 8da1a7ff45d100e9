// szg_oracle_raster.cpp — CPU oracle of the real-mesh G-buffer and shadow raster passes.
//
// TEST INFRASTRUCTURE ONLY (same rules as szg_oracle.cpp): imported by tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg, never by the product path.
//
// PARITY UNPINNED: the reference rasterises with the Vulkan fixed-function pipeline
// (renderer/pipelines/deferred.cpp:493-713, renderer/pipelines.cpp:593-806), whose results are
// implementation-defined at the bit level, and holds no fixtures for it. This file restates the raster RULES of
// include/szg/raster.h (Vulkan's, made exact) as a scalar pixel-major loop, plus a literal restatement of the two
// shaders (deferred/offscreen.vert:41-56, deferred/offscreen.frag:25-79, offscreenpass/depthpass.vert:30-38).
// tests/test_raster.py checks it against facts that do not come from this code: watertightness, the analytic
// ray-cast fill of the same boxes, closed-form depths.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "szg/abi.h"
#include "szg/fpmath.h"
#include "szg/raster.h"

// The contraction rule as ONE switch: -DSZG_ORACLE_LITERAL evaluates every a * b + c of szg_oracle.cpp's contraction rule with two roundings,
// i.e. executes the shaders' SPIR-V literally (libszg_oracle_literal.so; tests/test_spirv_pin.py compares that build, bit for
// bit, with an interpreter run over the reference's committed .spv). The default build fuses.
#ifdef SZG_ORACLE_LITERAL
#define SZG_CONTRACT SZG_CONTRACT_NONE
#endif
#include "szg/contraction.h"

extern "C" uint16_t oracle_float_to_half(float f);

namespace
{
#ifdef SZG_ORACLE_LIBM
#define GL_POW(x, y) powf((x), (y))
#else
#define GL_POW(x, y) szg_powf((x), (y))
#endif

struct vec2
{
    float x, y;
};
struct vec3
{
    float x, y, z;
};
struct vec4
{
    float x, y, z, w;
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline float dot(vec3 a, vec3 b) { return SZG_CON(SZG_C_DOT, a.z, b.z, SZG_CON(SZG_C_DOT, a.y, b.y, a.x * b.x)); } // OpDot (szg_oracle.cpp header)
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float inversesqrt(float x) { return 1.0f / sqrtf(x); }
inline vec3 normalize(vec3 v) { return v * inversesqrt(dot(v, v)); }

struct mat4
{
    float m[16]; // column-major
};
inline mat4 load(const szg_mat4& s)
{
    mat4 r;
    std::memcpy(r.m, s.m, sizeof r.m);
    return r;
}
inline vec4 operator*(const mat4& a, vec4 v)
{
    vec4 r;
    r.x = SZG_CON(SZG_C_MATVEC, a.m[12], v.w, SZG_CON(SZG_C_MATVEC, a.m[8], v.z, SZG_CON(SZG_C_MATVEC, a.m[4], v.y, a.m[0] * v.x)));
    r.y = SZG_CON(SZG_C_MATVEC, a.m[13], v.w, SZG_CON(SZG_C_MATVEC, a.m[9], v.z, SZG_CON(SZG_C_MATVEC, a.m[5], v.y, a.m[1] * v.x)));
    r.z = SZG_CON(SZG_C_MATVEC, a.m[14], v.w, SZG_CON(SZG_C_MATVEC, a.m[10], v.z, SZG_CON(SZG_C_MATVEC, a.m[6], v.y, a.m[2] * v.x)));
    r.w = SZG_CON(SZG_C_MATVEC, a.m[15], v.w, SZG_CON(SZG_C_MATVEC, a.m[11], v.z, SZG_CON(SZG_C_MATVEC, a.m[7], v.y, a.m[3] * v.x)));
    return r;
}
inline mat4 operator*(const mat4& a, const mat4& b)
{
    mat4 r;
    for (int j = 0; j < 4; j++)
    {
        vec4 const c = a * vec4{b.m[j * 4 + 0], b.m[j * 4 + 1], b.m[j * 4 + 2], b.m[j * 4 + 3]};
        r.m[j * 4 + 0] = c.x;
        r.m[j * 4 + 1] = c.y;
        r.m[j * 4 + 2] = c.z;
        r.m[j * 4 + 3] = c.w;
    }
    return r;
}

// One assembled primitive: raster.h "coverage" / "depth". Coefficients carry the facing sign, so the interior
// is where all three edge functions are positive.
struct Primitive
{
    float a[3], b[3], c[3];
    float z[3], w[3];
    // vertex-stage outputs of offscreen.vert for the fragment stage
    vec3 world[3];
    vec3 normal[3];
    vec2 uv[3];
    const szg_material* material;
};

// raster.h "coverage": returns false when the primitive produces no fragments at all.
bool assemble(const vec4 clip[3], float W, float H, bool cullFront, Primitive& out)
{
    float const halfW = W * 0.5f, halfH = H * 0.5f;
    float hx[3], hy[3], hw[3];
    bool allBehind = true, xl = true, xr = true, yt = true, yb = true, zn = true, zf = true;
    for (int i = 0; i < 3; i++)
    {
        vec4 const p = clip[i];
        if (!(p.x == p.x) || !(p.y == p.y) || !(p.z == p.z) || !(p.w == p.w))
        {
            return false;
        }
        allBehind = allBehind && (p.w <= 0.0f);
        xl = xl && (p.x < -p.w);
        xr = xr && (p.x > p.w);
        yt = yt && (p.y < -p.w);
        yb = yb && (p.y > p.w);
        zf = zf && (p.z < 0.0f);
        zn = zn && (p.z > p.w);
        hx[i] = (p.x + p.w) * halfW;
        hy[i] = (p.y + p.w) * halfH;
        hw[i] = p.w;
        out.z[i] = p.z;
        out.w[i] = p.w;
    }
    if (allBehind || xl || xr || yt || yb || zn || zf)
    {
        return false;
    }
    for (int i = 0; i < 3; i++)
    {
        int const j = (i + 1) % 3, k = (i + 2) % 3;
        out.a[i] = hy[j] * hw[k] - hy[k] * hw[j];
        out.b[i] = hx[k] * hw[j] - hx[j] * hw[k];
        out.c[i] = hx[j] * hy[k] - hx[k] * hy[j];
    }
    float const det = (hx[0] * out.a[0] + hy[0] * out.b[0]) + hw[0] * out.c[0];
    if (!(det > 0.0f) && !(det < 0.0f))
    {
        return false; // degenerate or NaN
    }
    bool const front = det > 0.0f; // clockwise in framebuffer space (deferred.cpp:380)
    if (cullFront ? front : !front)
    {
        return false;
    }
    if (!front)
    {
        for (int i = 0; i < 3; i++)
        {
            out.a[i] = -out.a[i];
            out.b[i] = -out.b[i];
            out.c[i] = -out.c[i];
        }
    }
    return true;
}

inline void edges(const Primitive& t, float px, float py, float e[3])
{
    for (int i = 0; i < 3; i++)
    {
        e[i] = (t.a[i] * px + t.b[i] * py) + t.c[i];
    }
}
inline bool covers(const Primitive& t, const float e[3])
{
    for (int i = 0; i < 3; i++)
    {
        bool const in = e[i] > 0.0f || (e[i] == 0.0f && (t.a[i] > 0.0f || (t.a[i] == 0.0f && t.b[i] > 0.0f)));
        if (!in)
        {
            return false;
        }
    }
    return true;
}
// raster.h "depth"; false outside the depth clip volume
inline bool fragmentDepth(const Primitive& t, const float e[3], float& depth)
{
    float const zc = (e[0] * t.z[0] + e[1] * t.z[1]) + e[2] * t.z[2];
    float const wc = (e[0] * t.w[0] + e[1] * t.w[1]) + e[2] * t.w[2];
    if (!(zc >= 0.0f && zc <= wc && wc > 0.0f))
    {
        return false;
    }
    depth = zc / wc;
    return true;
}

struct Varyings
{
    vec3 world, normal;
    vec2 uv;
};
inline float lerp3(const float l[3], float a0, float a1, float a2) { return (l[0] * a0 + l[1] * a1) + l[2] * a2; }
inline Varyings interpolate(const Primitive& t, float px, float py)
{
    float e[3];
    edges(t, px, py, e);
    float const S = (e[0] + e[1]) + e[2];
    float const l[3] = {e[0] / S, e[1] / S, e[2] / S};
    Varyings v;
    v.world = {lerp3(l, t.world[0].x, t.world[1].x, t.world[2].x), lerp3(l, t.world[0].y, t.world[1].y, t.world[2].y),
               lerp3(l, t.world[0].z, t.world[1].z, t.world[2].z)};
    v.normal = {lerp3(l, t.normal[0].x, t.normal[1].x, t.normal[2].x), lerp3(l, t.normal[0].y, t.normal[1].y, t.normal[2].y),
                lerp3(l, t.normal[0].z, t.normal[1].z, t.normal[2].z)};
    v.uv = {lerp3(l, t.uv[0].x, t.uv[1].x, t.uv[2].x), lerp3(l, t.uv[0].y, t.uv[1].y, t.uv[2].y)};
    return v;
}

// raster.h "textures": RGBA8, LINEAR, REPEAT, one level
inline float decode8(uint8_t b, bool srgb)
{
    float const c = (float)b / 255.0f;
    if (!srgb)
    {
        return c;
    }
    return c <= 0.04045f ? c / 12.92f : GL_POW((c + 0.055f) / 1.055f, 2.4f);
}
inline int wrapIndex(float f, int n)
{
    float const fn = (float)n;
    float const m = f - fn * floorf(f / fn); // in [0, n] for finite f
    int i = (int)m;
    if (i >= n || i < 0)
    {
        i = 0;
    }
    return i;
}
vec3 sampleTexture(const szg_texture& tex, vec2 st)
{
    if (tex.data == nullptr || tex.width == 0u || tex.height == 0u)
    {
        return {0.0f, 0.0f, 0.0f};
    }
    int const W = (int)tex.width, H = (int)tex.height;
    float const u = st.x * (float)W - 0.5f;
    float const v = st.y * (float)H - 0.5f;
    float const fu = floorf(u), fv = floorf(v);
    float const a = u - fu, b = v - fv;
    int const i0 = wrapIndex(fu, W), j0 = wrapIndex(fv, H);
    int const i1 = (i0 + 1 == W) ? 0 : i0 + 1, j1 = (j0 + 1 == H) ? 0 : j0 + 1;
    auto texel = [&](int i, int j, int ch) {
        const uint8_t* p = static_cast<const uint8_t*>(tex.data) + (size_t)j * tex.pitch_bytes + (size_t)i * 4u;
        return decode8(p[ch], tex.srgb != 0u);
    };
    float const w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float r[3];
    for (int ch = 0; ch < 3; ch++)
    {
        r[ch] = w00 * texel(i0, j0, ch) + w10 * texel(i1, j0, ch) + w01 * texel(i0, j1, ch) + w11 * texel(i1, j1, ch);
    }
    return {r[0], r[1], r[2]};
}

// deferred/offscreen.frag:25-59
vec3 perturbNormal(const szg_material& mat, vec3 N, vec3 dPosDx, vec3 dPosDy, vec2 dUvDx, vec2 dUvDy, vec2 texcoord)
{
    vec3 map = sampleTexture(mat.normal, texcoord);
    float const k = 128.0f / 127.0f;
    map = {map.x * 255.0f / 127.0f - k, map.y * 255.0f / 127.0f - k, map.z * 255.0f / 127.0f - k}; // :47
    map.y = -map.y;                                                                                  // :50
    // cotangentFrame(N, -V, texcoord) with V = inWorldPosition (:54, :65): p = -worldPosition
    vec3 const dp1 = -dPosDx;
    vec3 const dp2 = -dPosDy;
    vec3 const dp2perp = cross(dp2, N);
    vec3 const dp1perp = cross(N, dp1);
    vec3 const T = dp2perp * dUvDx.x + dp1perp * dUvDy.x;
    vec3 const B = dp2perp * dUvDx.y + dp1perp * dUvDy.y;
    float const invmax = inversesqrt(fmaxf(dot(T, T), dot(B, B)));
    vec3 const c0 = T * invmax, c1 = B * invmax;
    vec3 const v = (c0 * map.x + c1 * map.y) + N * map.z; // mat3 * vec3
    return normalize(v);
}

inline uint32_t global_row(const szg_rowtile* tile, uint32_t local)
{
    if (tile == nullptr || tile->nranks <= 1)
    {
        return local;
    }
    return ((local / tile->block_rows) * tile->nranks + tile->rank) * tile->block_rows + local % tile->block_rows;
}
inline uint32_t local_rows(const szg_rowtile* tile, uint32_t height)
{
    if (tile == nullptr || tile->nranks <= 1)
    {
        return height;
    }
    return tile->local_rows;
}

void parallel_rows(uint32_t rows, int threads, const std::function<void(uint32_t, uint32_t)>& fn)
{
    if (threads <= 1 || rows < 2)
    {
        fn(0, rows);
        return;
    }
    uint32_t const n = std::min<uint32_t>((uint32_t)threads, rows);
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < n; t++)
    {
        pool.emplace_back([=, &fn]() {
            for (uint32_t r0 = t * 4u; r0 < rows; r0 += n * 4u)
            {
                fn(r0, std::min(rows, r0 + 4u));
            }
        });
    }
    for (auto& th : pool)
    {
        th.join();
    }
}

// Primitive assembly in submission order (raster.h "order"). `shadow`: depthpass.vert instead of offscreen.vert.
std::vector<Primitive> assembleScene(const szg_mesh_instanced* meshes, uint32_t meshCount, const mat4& projection, const mat4& view,
                                     bool shadow, float W, float H)
{
    std::vector<Primitive> prims;
    mat4 const projView = projection * view; // offscreen.vert:51 `P * V * p`; for the shadow pass `projection` IS projView
    for (uint32_t mi = 0; mi < meshCount; mi++)
    {
        szg_mesh_instanced const& mesh = meshes[mi];
        if (mesh.render == 0u || (shadow && mesh.casts_shadow == 0u) || mesh.d_vertices == nullptr || mesh.d_indices == nullptr ||
            mesh.d_models == nullptr || (!shadow && mesh.d_model_inverse_transposes == nullptr))
        {
            continue;
        }
        for (uint32_t si = 0; si < mesh.surface_count; si++)
        {
            szg_surface const& surf = mesh.surfaces[si];
            uint32_t const first = surf.first_index;
            uint32_t const count = first >= mesh.index_count ? 0u : std::min(surf.index_count, mesh.index_count - first);
            for (uint32_t inst = 0; inst < mesh.instance_count; inst++)
            {
                mat4 const model = load(mesh.d_models[inst]);
                for (uint32_t t = 0; t + 3u <= count; t += 3u)
                {
                    uint32_t const idx[3] = {mesh.d_indices[first + t], mesh.d_indices[first + t + 1u], mesh.d_indices[first + t + 2u]};
                    if (idx[0] >= mesh.vertex_count || idx[1] >= mesh.vertex_count || idx[2] >= mesh.vertex_count)
                    {
                        continue;
                    }
                    Primitive prim{};
                    vec4 clip[3];
                    for (int v = 0; v < 3; v++)
                    {
                        szg_vertex_packed const& vert = mesh.d_vertices[idx[v]];
                        vec4 const local{vert.position[0], vert.position[1], vert.position[2], 1.0f};
                        if (shadow)
                        {
                            clip[v] = (projection * model) * local; // depthpass.vert:37
                        }
                        else
                        {
                            vec4 const position = model * local; // offscreen.vert:46
                            prim.world[v] = {position.x, position.y, position.z};
                            clip[v] = projView * position;
                            mat4 const mit = load(mesh.d_model_inverse_transposes[inst]);
                            vec4 const n = mit * vec4{vert.normal[0], vert.normal[1], vert.normal[2], 0.0f};
                            prim.normal[v] = normalize(vec3{n.x, n.y, n.z}); // :53
                            prim.uv[v] = {vert.uv_x, vert.uv_y};             // :55
                        }
                    }
                    prim.material = &surf.material;
                    if (assemble(clip, W, H, /*cullFront=*/shadow, prim))
                    {
                        prims.push_back(prim);
                    }
                }
            }
        }
    }
    (void)view;
    return prims;
}

inline uint8_t* rowp(const szg_image& im, uint32_t y) { return static_cast<uint8_t*>(im.data) + (size_t)y * im.pitch_bytes; }
inline void storeHalf4(const szg_image& im, uint32_t x, uint32_t y, float r, float g, float b, float a)
{
    uint16_t* p = reinterpret_cast<uint16_t*>(rowp(im, y)) + (size_t)x * 4u;
    p[0] = oracle_float_to_half(r);
    p[1] = oracle_float_to_half(g);
    p[2] = oracle_float_to_half(b);
    p[3] = oracle_float_to_half(a);
}
} // namespace

extern "C" {

// The G-buffer pass (deferred.cpp:493-713) into host images laid out like the device ones.
void oracle_gbuffer_raster(const szg_scene_texture* scene, szg_rect drawRect, const szg_rowtile* tile, const szg_gbuffer* gbuffer,
                           const szg_camera_packed* cameras, uint32_t cameraIndex, const szg_mesh_instanced* meshes,
                           uint32_t meshCount, int threads)
{
    szg_camera_packed const& camera = cameras[cameraIndex];
    float const W = (float)drawRect.width, H = (float)drawRect.height;
    std::vector<Primitive> const prims =
        assembleScene(meshes, meshCount, load(camera.projection), load(camera.view), false, W, H);
    uint32_t const rows = local_rows(tile, drawRect.height);

    parallel_rows(rows, threads, [&](uint32_t r0, uint32_t r1) {
        for (uint32_t y = r0; y < r1; y++)
        {
            uint32_t const gy = global_row(tile, y);
            for (uint32_t x = 0; x < drawRect.width; x++)
            {
                float const px = (float)x + 0.5f, py = (float)gy + 0.5f;
                float best = 0.0f; // clear value; GREATER
                const Primitive* winner = nullptr;
                for (Primitive const& t : prims)
                {
                    float e[3];
                    edges(t, px, py, e);
                    float depth;
                    if (covers(t, e) && fragmentDepth(t, e, depth) && depth > best)
                    {
                        best = depth;
                        winner = &t;
                    }
                }
                {
                    float* pos = reinterpret_cast<float*>(rowp(gbuffer->worldPosition, y)) + (size_t)x * 4u;
                    if (winner == nullptr)
                    {
                        storeHalf4(gbuffer->diffuse, x, y, 0, 0, 0, 0);
                        storeHalf4(gbuffer->specular, x, y, 0, 0, 0, 0);
                        storeHalf4(gbuffer->normal, x, y, 0, 0, 0, 0);
                        storeHalf4(gbuffer->occlusionRoughnessMetallic, x, y, 0, 0, 0, 0);
                        pos[0] = pos[1] = pos[2] = pos[3] = 0.0f;
                        reinterpret_cast<float*>(rowp(scene->depth, y))[x] = 0.0f;
                        continue;
                    }
                    Primitive const& t = *winner;
                    Varyings const in = interpolate(t, px, py);
                    // fine derivatives over the 2x2 quad (raster.h "derivatives")
                    float const qx = (float)(x & ~1u) + 0.5f, qy = (float)(gy & ~1u) + 0.5f;
                    Varyings const xl = interpolate(t, qx, py), xr = interpolate(t, qx + 1.0f, py);
                    Varyings const yt = interpolate(t, px, qy), yb = interpolate(t, px, qy + 1.0f);
                    vec3 const N = perturbNormal(*t.material, in.normal, xr.world - xl.world, yb.world - yt.world, xr.uv - xl.uv,
                                                 yb.uv - yt.uv, in.uv);
                    vec3 const color = sampleTexture(t.material->color, in.uv);
                    vec3 const orm = sampleTexture(t.material->orm, in.uv);
                    storeHalf4(gbuffer->diffuse, x, y, color.x, color.y, color.z, 1.0f);                       // frag :72
                    storeHalf4(gbuffer->specular, x, y, color.x, color.y, color.z, 1.0f);                      // :75
                    storeHalf4(gbuffer->normal, x, y, N.x, N.y, N.z, 0.0f);                                    // :68
                    storeHalf4(gbuffer->occlusionRoughnessMetallic, x, y, orm.x, orm.y, orm.z, 1.0f);          // :79
                    pos[0] = in.world.x;                                                                       // :63
                    pos[1] = in.world.y;
                    pos[2] = in.world.z;
                    pos[3] = 1.0f;
                    reinterpret_cast<float*>(rowp(scene->depth, y))[x] = best;
                }
            }
        }
    });
}

// The two programmable stages alone, for tests/test_spirv_pin.py (the same expressions assembleScene and the shading loop
// above evaluate). out: clip xyzw, world xyz, normal xyz, uv xy = 12 floats (depth pass: clip only).
void oracle_vertex_stage(const szg_vertex_packed* vertex, const szg_mat4* model, const szg_mat4* modelInverseTranspose,
                         const szg_mat4* projection, const szg_mat4* view, int shadow, float out[12])
{
    vec4 const local{vertex->position[0], vertex->position[1], vertex->position[2], 1.0f};
    for (int i = 0; i < 12; i++)
    {
        out[i] = 0.0f;
    }
    vec4 clip;
    if (shadow != 0)
    {
        clip = (load(*projection) * load(*model)) * local; // depthpass.vert:37
    }
    else
    {
        mat4 const projView = load(*projection) * load(*view);
        vec4 const position = load(*model) * local;
        clip = projView * position;
        vec4 const n = load(*modelInverseTranspose) * vec4{vertex->normal[0], vertex->normal[1], vertex->normal[2], 0.0f};
        vec3 const normal = normalize(vec3{n.x, n.y, n.z});
        out[4] = position.x;
        out[5] = position.y;
        out[6] = position.z;
        out[7] = normal.x;
        out[8] = normal.y;
        out[9] = normal.z;
        out[10] = vertex->uv_x;
        out[11] = vertex->uv_y;
    }
    out[0] = clip.x;
    out[1] = clip.y;
    out[2] = clip.z;
    out[3] = clip.w;
}
// offscreen.frag on one fragment: interpolated inputs and the screen-space differences the fixed function hands to dFdx /
// dFdy. out: worldPosition, normal, diffuse, specular, ORM (vec4 each, before the attachment's format conversion).
void oracle_fragment_stage(const szg_material* material, const float world[3], const float normal[3], const float uv[2],
                           const float dWorldDx[3], const float dWorldDy[3], const float dUvDx[2], const float dUvDy[2], float out[20])
{
    vec2 const st{uv[0], uv[1]};
    vec3 const N = perturbNormal(*material, vec3{normal[0], normal[1], normal[2]}, vec3{dWorldDx[0], dWorldDx[1], dWorldDx[2]},
                                 vec3{dWorldDy[0], dWorldDy[1], dWorldDy[2]}, vec2{dUvDx[0], dUvDx[1]}, vec2{dUvDy[0], dUvDy[1]}, st);
    vec3 const color = sampleTexture(material->color, st);
    vec3 const orm = sampleTexture(material->orm, st);
    float const v[20] = {world[0], world[1], world[2], 1.0f, N.x, N.y, N.z, 0.0f, color.x, color.y, color.z, 1.0f,
                         color.x, color.y, color.z, 1.0f, orm.x, orm.y, orm.z, 1.0f};
    std::memcpy(out, v, sizeof v);
}

// One shadow map (pipelines.cpp:674-806): depth only, front faces culled, GREATER_OR_EQUAL against a 0 clear.
void oracle_shadow_raster(const szg_image* map, const szg_mat4* projView, float depthBiasConstant, float depthBiasSlope,
                          const szg_mesh_instanced* meshes, uint32_t meshCount, int threads)
{
    float const W = (float)map->width, H = (float)map->height;
    mat4 identity{};
    identity.m[0] = identity.m[5] = identity.m[10] = identity.m[15] = 1.0f;
    std::vector<Primitive> const prims = assembleScene(meshes, meshCount, load(*projView), identity, true, W, H);
    bool const biased = depthBiasConstant != 0.0f || depthBiasSlope != 0.0f;
    parallel_rows(map->height, threads, [&](uint32_t r0, uint32_t r1) {
        for (uint32_t y = r0; y < r1; y++)
        {
            for (uint32_t x = 0; x < map->width; x++)
            {
                float const px = (float)x + 0.5f, py = (float)y + 0.5f;
                float best = 0.0f;
                for (Primitive const& t : prims)
                {
                    float e[3];
                    edges(t, px, py, e);
                    float depth;
                    if (!covers(t, e) || !fragmentDepth(t, e, depth))
                    {
                        continue;
                    }
                    if (biased)
                    {
                        // Vulkan depth bias: o = m * slope + r * constant, m = max(|dz/dx|, |dz/dy|) from the
                        // primitive's own depth interpolant at the neighbouring pixel centres, r = 2^(exponent(z) - 23)
                        float ex[3], ey[3], dzx = depth, dzy = depth;
                        edges(t, px + 1.0f, py, ex);
                        edges(t, px, py + 1.0f, ey);
                        float const zx = ((ex[0] * t.z[0] + ex[1] * t.z[1]) + ex[2] * t.z[2]) /
                                         ((ex[0] * t.w[0] + ex[1] * t.w[1]) + ex[2] * t.w[2]);
                        float const zy = ((ey[0] * t.z[0] + ey[1] * t.z[1]) + ey[2] * t.z[2]) /
                                         ((ey[0] * t.w[0] + ey[1] * t.w[1]) + ey[2] * t.w[2]);
                        dzx = fabsf(zx - depth);
                        dzy = fabsf(zy - depth);
                        float const m = fmaxf(dzx, dzy);
                        int exponent = 0;
                        (void)frexpf(depth, &exponent); // depth = f * 2^exponent, f in [.5, 1)
                        float const r = ldexpf(1.0f, (exponent - 1) - 23);
                        float const o = m * depthBiasSlope + r * depthBiasConstant;
                        depth = fminf(fmaxf(depth + o, 0.0f), 1.0f);
                    }
                    if (depth >= best)
                    {
                        best = depth;
                    }
                }
                reinterpret_cast<float*>(rowp(*map, y))[x] = best;
            }
        }
    });
}

} // extern "C"
