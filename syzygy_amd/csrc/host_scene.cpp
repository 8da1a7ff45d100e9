// host_scene.cpp — CPU-only input preparation behind szg/host.h.
//
// Restates the reference's scene -> packed-struct code path:
//   geometry/geometryhelpers.cpp:83-204, renderer/lights.cpp:9-46,
//   renderer/scene.cpp:52-91, :532-574, :584-623, :689-794
// on top of a minimal restatement of the glm 1.0.1 routines they call
// (perspectiveLH_ZO, orthoLH_ZO, yawPitchRoll/orientate, translate, inverse).
// All arithmetic is fp32 like glm's default precision.

#include "szg/host.h"

#include <cfloat>
#include <cmath>
#include <cstring>

namespace
{
struct V3
{
    float x, y, z;
};
struct V4
{
    float x, y, z, w;
};

inline V3 v3(const float* p) { return {p[0], p[1], p[2]}; }
inline void store(float* p, V3 v)
{
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// glm::dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// glm::normalize: v * inversesqrt(dot(v, v))
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline V3 vmin(V3 a, V3 b) { return {b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z}; }
inline V3 vmax(V3 a, V3 b) { return {a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z}; }

constexpr V3 WORLD_FORWARD{0.0f, 0.0f, 1.0f}; // geometrystatics.hpp:7
constexpr V3 WORLD_UP{0.0f, -1.0f, 0.0f};     // geometrystatics.hpp:8
constexpr V3 WORLD_RIGHT{1.0f, 0.0f, 0.0f};   // geometrystatics.hpp:9

constexpr float PI_F = 3.14159265358979323846264338327950288f;
constexpr float HALF_PI_F = 1.57079632679489661923132169163975144f;
constexpr float TWO_PI_F = 6.28318530717958647692528676655900576f;

inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

inline float& at(szg_mat4& m, int col, int row) { return m.m[col * 4 + row]; }
inline float at(const szg_mat4& m, int col, int row) { return m.m[col * 4 + row]; }

szg_mat4 zero4()
{
    szg_mat4 m;
    std::memset(&m, 0, sizeof m);
    return m;
}
szg_mat4 identity4()
{
    szg_mat4 m = zero4();
    at(m, 0, 0) = at(m, 1, 1) = at(m, 2, 2) = at(m, 3, 3) = 1.0f;
    return m;
}

// glm mat4 * vec4: m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]*v.w (pairwise: (a+b)+(c+d))
V4 mul(const szg_mat4& m, V4 v)
{
    V4 r;
    float* o = &r.x;
    for (int row = 0; row < 4; row++)
    {
        float const a = at(m, 0, row) * v.x;
        float const b = at(m, 1, row) * v.y;
        float const c = at(m, 2, row) * v.z;
        float const d = at(m, 3, row) * v.w;
        o[row] = (a + b) + (c + d);
    }
    return r;
}

// glm mat4 * mat4: column j of the result is A*B[j] accumulated as
// A[0]*b0 + A[1]*b1 + A[2]*b2 + A[3]*b3 (left to right)
szg_mat4 mul(const szg_mat4& a, const szg_mat4& b)
{
    szg_mat4 r;
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++)
        {
            float acc = at(a, 0, row) * at(b, col, 0);
            acc = acc + at(a, 1, row) * at(b, col, 1);
            acc = acc + at(a, 2, row) * at(b, col, 2);
            acc = acc + at(a, 3, row) * at(b, col, 3);
            at(r, col, row) = acc;
        }
    return r;
}

// glm::inverse(mat4) — cofactor expansion, glm/detail/func_matrix.inl compute_inverse<4,4>
szg_mat4 inverse(const szg_mat4& m)
{
    auto M = [&](int c, int r) { return at(m, c, r); };
    float const Coef00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
    float const Coef02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
    float const Coef03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    float const Coef04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    float const Coef06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
    float const Coef07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    float const Coef08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
    float const Coef10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
    float const Coef11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    float const Coef12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    float const Coef14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
    float const Coef15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    float const Coef16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
    float const Coef18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
    float const Coef19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    float const Coef20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    float const Coef22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
    float const Coef23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);

    float const Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
    float const Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    float const Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
    float const Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    float const Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
    float const Fac5[4] = {Coef20, Coef20, Coef22, Coef23};

    float const Vec0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)};
    float const Vec1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)};
    float const Vec2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)};
    float const Vec3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};

    float const SignA[4] = {+1.0f, -1.0f, +1.0f, -1.0f};
    float const SignB[4] = {-1.0f, +1.0f, -1.0f, +1.0f};

    szg_mat4 inv;
    for (int i = 0; i < 4; i++)
    {
        float const Inv0 = Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i] + Vec3[i] * Fac2[i];
        float const Inv1 = Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i] + Vec3[i] * Fac4[i];
        float const Inv2 = Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i] + Vec3[i] * Fac5[i];
        float const Inv3 = Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i] + Vec2[i] * Fac5[i];
        at(inv, 0, i) = Inv0 * SignA[i];
        at(inv, 1, i) = Inv1 * SignB[i];
        at(inv, 2, i) = Inv2 * SignA[i];
        at(inv, 3, i) = Inv3 * SignB[i];
    }

    float const Dot0x = M(0, 0) * at(inv, 0, 0);
    float const Dot0y = M(0, 1) * at(inv, 1, 0);
    float const Dot0z = M(0, 2) * at(inv, 2, 0);
    float const Dot0w = M(0, 3) * at(inv, 3, 0);
    float const Dot1 = (Dot0x + Dot0y) + (Dot0z + Dot0w);
    float const OneOverDeterminant = 1.0f / Dot1;

    for (float& f : inv.m)
        f = f * OneOverDeterminant;
    return inv;
}

szg_mat4 transpose(const szg_mat4& m)
{
    szg_mat4 r;
    for (int c = 0; c < 4; c++)
        for (int w = 0; w < 4; w++)
            at(r, c, w) = at(m, w, c);
    return r;
}

// glm::yawPitchRoll(yaw, pitch, roll), glm/gtx/euler_angles.inl
szg_mat4 yawPitchRoll(float yaw, float pitch, float roll)
{
    float const ch = std::cos(yaw), sh = std::sin(yaw);
    float const cp = std::cos(pitch), sp = std::sin(pitch);
    float const cb = std::cos(roll), sb = std::sin(roll);
    szg_mat4 r = zero4();
    at(r, 0, 0) = ch * cb + sh * sp * sb;
    at(r, 0, 1) = sb * cp;
    at(r, 0, 2) = -sh * cb + ch * sp * sb;
    at(r, 1, 0) = -ch * sb + sh * sp * cb;
    at(r, 1, 1) = cb * cp;
    at(r, 1, 2) = sb * sh + ch * sp * cb;
    at(r, 2, 0) = sh * cp;
    at(r, 2, 1) = -sp;
    at(r, 2, 2) = ch * cp;
    at(r, 3, 3) = 1.0f;
    return r;
}

// glm::orientate4(angles) = yawPitchRoll(angles.z, angles.x, angles.y)
szg_mat4 orientate4(V3 e) { return yawPitchRoll(e.z, e.x, e.y); }

// glm::perspectiveLH_ZO(fovy, aspect, zNear, zFar)
szg_mat4 perspectiveLH_ZO(float fovy, float aspect, float zNear, float zFar)
{
    float const tanHalfFovy = std::tan(fovy / 2.0f);
    szg_mat4 r = zero4();
    at(r, 0, 0) = 1.0f / (aspect * tanHalfFovy);
    at(r, 1, 1) = 1.0f / tanHalfFovy;
    at(r, 2, 2) = zFar / (zFar - zNear);
    at(r, 2, 3) = 1.0f;
    at(r, 3, 2) = -(zFar * zNear) / (zFar - zNear);
    return r;
}

// glm::orthoLH_ZO(left, right, bottom, top, zNear, zFar)
szg_mat4 orthoLH_ZO(float l, float r_, float b, float t, float zNear, float zFar)
{
    szg_mat4 r = identity4();
    at(r, 0, 0) = 2.0f / (r_ - l);
    at(r, 1, 1) = 2.0f / (t - b);
    at(r, 2, 2) = 1.0f / (zFar - zNear);
    at(r, 3, 0) = -(r_ + l) / (r_ - l);
    at(r, 3, 1) = -(t + b) / (t - b);
    at(r, 3, 2) = -zNear / (zFar - zNear);
    return r;
}

szg_mat4 translate(V3 p)
{
    szg_mat4 r = identity4();
    at(r, 3, 0) = p.x;
    at(r, 3, 1) = p.y;
    at(r, 3, 2) = p.z;
    return r;
}

// geometryhelpers.cpp:102-105: orientate3(e) * WORLD_FORWARD = third column
V3 forwardFromEulers(V3 e)
{
    szg_mat4 const o = orientate4(e);
    // mat3 * vec3 = m[0]*v.x + m[1]*v.y + m[2]*v.z
    V3 r;
    r.x = at(o, 0, 0) * WORLD_FORWARD.x + at(o, 1, 0) * WORLD_FORWARD.y + at(o, 2, 0) * WORLD_FORWARD.z;
    r.y = at(o, 0, 1) * WORLD_FORWARD.x + at(o, 1, 1) * WORLD_FORWARD.y + at(o, 2, 1) * WORLD_FORWARD.z;
    r.z = at(o, 0, 2) * WORLD_FORWARD.x + at(o, 1, 2) * WORLD_FORWARD.y + at(o, 2, 2) * WORLD_FORWARD.z;
    return r;
}

// geometryhelpers.cpp:107-145
V3 eulersFromForward(V3 forward)
{
    float const len2 = dot(forward, forward);
    // glm::epsilonEqual(length2, 0, epsilon<float>()) : abs(a - b) < epsilon
    if (std::fabs(len2 - 0.0f) < FLT_EPSILON)
    {
        return {0.0f, 0.0f, 0.0f};
    }
    V3 const f = normalize(forward);
    float const dot_forward = dot(f, WORLD_FORWARD);
    float const dot_right = dot(f, WORLD_RIGHT);
    float const dot_up = dot(f, WORLD_UP);
    float const roll = 0.0f;
    float const pitch = std::asin(dot_up);
    float const yaw = std::atan2(dot_right, dot_forward);
    return {pitch, roll, yaw};
}

// geometryhelpers.cpp:147-157
szg_mat4 transformVk(V3 position, V3 eulers) { return mul(translate(position), orientate4(eulers)); }
szg_mat4 viewVk(V3 position, V3 eulers) { return inverse(transformVk(position, eulers)); }

// geometryhelpers.cpp:83-95: near and far swapped on purpose (reverse-Z)
szg_mat4 projectionVk(float fov_y_degrees, float aspect, float near_plane, float far_plane)
{
    float const swappedNear = far_plane;
    float const swappedFar = near_plane;
    return perspectiveLH_ZO(radians(fov_y_degrees), aspect, swappedNear, swappedFar);
}

// geometryhelpers.cpp:97-100
szg_mat4 projectionOrthoVk(V3 mn, V3 mx) { return orthoLH_ZO(mn.x, mx.x, mn.y, mx.y, mx.z, mn.z); }

// geometryhelpers.cpp:55-61 (returns projection + point, as written there)
V3 projectPointOnPlane(V3 planePoint, V3 planeNormal, V3 point)
{
    V3 const toPoint = point - planePoint;
    V3 const projection = dot(toPoint, planeNormal) * planeNormal;
    return projection + point;
}

// geometryhelpers.cpp:171-204 with AABB::collectVertices (geometrytypes.cpp:20-32)
szg_mat4 projectionOrthoAABBVk(const szg_mat4& view, const szg_aabb& bounds)
{
    V3 const c = v3(bounds.center);
    V3 const h = v3(bounds.half_extent);
    V3 const verts[8] = {
        c + V3{h.x, h.y, h.z},  c + V3{h.x, h.y, -h.z},  c + V3{h.x, -h.y, h.z},  c + V3{h.x, -h.y, -h.z},
        c + V3{-h.x, h.y, h.z}, c + V3{-h.x, h.y, -h.z}, c + V3{-h.x, -h.y, h.z}, c + V3{-h.x, -h.y, -h.z},
    };
    V4 const cv = mul(view, V4{c.x, c.y, c.z, 1.0f});
    V3 const centerViewSpace{cv.x, cv.y, cv.z};
    V3 const forwardViewSpace = WORLD_FORWARD;

    V3 viewMax{-FLT_MAX, -FLT_MAX, -FLT_MAX};
    V3 viewMin{FLT_MAX, FLT_MAX, FLT_MAX};
    for (V3 const vertex : verts)
    {
        V4 const vv = mul(view, V4{vertex.x, vertex.y, vertex.z, 1.0f});
        V3 const projected = projectPointOnPlane(centerViewSpace, forwardViewSpace, V3{vv.x, vv.y, vv.z});
        viewMax = vmax(projected, viewMax);
        viewMin = vmin(projected, viewMin);
    }
    return projectionOrthoVk(viewMin, viewMax);
}

void makeDirectional(const float color[4], float strength, V3 eulers, const szg_aabb& bounds,
                     szg_directional_light_packed* out)
{
    std::memset(out, 0, sizeof *out);
    szg_mat4 const view = viewVk(V3{0.0f, 0.0f, 0.0f}, eulers);
    szg_mat4 const projection = projectionOrthoAABBVk(view, bounds);
    std::memcpy(out->color, color, sizeof(float) * 4);
    V3 const f = forwardFromEulers(eulers);
    out->forward[0] = f.x;
    out->forward[1] = f.y;
    out->forward[2] = f.z;
    out->forward[3] = 0.0f;
    out->projection = projection;
    out->view = view;
    out->strength = strength;
}
} // namespace

extern "C" {

void szg_forward_from_eulers(const float eulers[3], float out_forward[3]) { store(out_forward, forwardFromEulers(v3(eulers))); }
void szg_eulers_from_forward(const float forward[3], float out_eulers[3]) { store(out_eulers, eulersFromForward(v3(forward))); }

void szg_projection_vk(float fov_y_degrees, float aspect, float near_plane, float far_plane, szg_mat4* out)
{
    *out = projectionVk(fov_y_degrees, aspect, near_plane, far_plane);
}
void szg_projection_ortho_vk(const float mn[3], const float mx[3], szg_mat4* out) { *out = projectionOrthoVk(v3(mn), v3(mx)); }
void szg_transform_vk(const float position[3], const float eulers[3], szg_mat4* out) { *out = transformVk(v3(position), v3(eulers)); }
void szg_view_vk(const float position[3], const float eulers[3], szg_mat4* out) { *out = viewVk(v3(position), v3(eulers)); }
// geometry/transform.cpp:11-15 Transform::toMatrix: glm::translate(t) * glm::orientate4(eulers) * glm::scale(s)
void szg_transform_matrix(const float translation[3], const float eulers[3], const float scale[3], szg_mat4* out)
{
    szg_mat4 sc = zero4();
    at(sc, 0, 0) = scale[0];
    at(sc, 1, 1) = scale[1];
    at(sc, 2, 2) = scale[2];
    at(sc, 3, 3) = 1.0f;
    *out = mul(mul(translate(v3(translation)), orientate4(v3(eulers))), sc);
}
// geometry/transform.cpp:17-28 Transform::lookAt over Ray::create(from, to) (direction = to - from, possibly unnormalised)
void szg_transform_look_at(const float from[3], const float to[3], const float scale[3], szg_transform* out)
{
    V3 const forward = normalize(v3(to) - v3(from));
    store(out->translation, v3(from));
    store(out->eulerAnglesRadians, eulersFromForward(forward));
    store(out->scale, v3(scale));
}
void szg_projection_ortho_aabb_vk(const szg_mat4* view, const szg_aabb* bounds, szg_mat4* out)
{
    *out = projectionOrthoAABBVk(*view, *bounds);
}
void szg_mat4_inverse(const szg_mat4* m, szg_mat4* out) { *out = inverse(*m); }
void szg_mat4_inverse_transpose(const szg_mat4* m, szg_mat4* out) { *out = transpose(inverse(*m)); }
void szg_mat4_mul(const szg_mat4* a, const szg_mat4* b, szg_mat4* out) { *out = mul(*a, *b); }

// geometrytypes.cpp:11-19
void szg_aabb_create(const float mn[3], const float mx[3], szg_aabb* out)
{
    for (int k = 0; k < 3; k++)
    {
        float const safeMin = std::fmin(mn[k], mx[k]);
        float const safeMax = std::fmax(mn[k], mx[k]);
        float const center = 0.5f * (safeMax + safeMin);
        out->center[k] = center;
        out->half_extent[k] = safeMax - center;
    }
}

// Scene::calculateShadowBounds, scene.cpp:95-148
int szg_calculate_shadow_bounds(const szg_shadow_caster* casters, uint32_t caster_count, szg_aabb* out)
{
    std::memset(out, 0, sizeof *out);
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t c = 0; c < caster_count; c++)
    {
        szg_shadow_caster const& caster = casters[c];
        if (caster.casts_shadow == 0u || caster.render == 0u)
        {
            continue;
        }
        // AABB::collectVertices, geometrytypes.cpp:20-32
        float corners[8][3];
        for (int v = 0; v < 8; v++)
        {
            corners[v][0] = caster.vertex_bounds.center[0] + ((v & 4) ? -caster.vertex_bounds.half_extent[0] : caster.vertex_bounds.half_extent[0]);
            corners[v][1] = caster.vertex_bounds.center[1] + ((v & 2) ? -caster.vertex_bounds.half_extent[1] : caster.vertex_bounds.half_extent[1]);
            corners[v][2] = caster.vertex_bounds.center[2] + ((v & 1) ? -caster.vertex_bounds.half_extent[2] : caster.vertex_bounds.half_extent[2]);
        }
        for (uint32_t t = 0; t < caster.transform_count; t++)
        {
            szg_mat4 m;
            szg_transform_matrix(caster.transforms[t].translation, caster.transforms[t].eulerAnglesRadians, caster.transforms[t].scale, &m);
            for (int v = 0; v < 8; v++)
            {
                // mat4 * vec4(vertex, 1): columns scaled by the components, summed left to right (glm)
                for (int k = 0; k < 3; k++)
                {
                    float const w = m.m[0 * 4 + k] * corners[v][0] + m.m[1 * 4 + k] * corners[v][1] + m.m[2 * 4 + k] * corners[v][2] +
                                    m.m[3 * 4 + k] * 1.0f;
                    mn[k] = std::fmin(w, mn[k]);
                    mx[k] = std::fmax(w, mx[k]);
                }
            }
        }
    }
    if (mn[0] > mx[0] || mn[1] > mx[1] || mn[2] > mx[2])
    {
        return 0; // not a single valid vertex
    }
    szg_aabb_create(mn, mx, out);
    return 1;
}

// tickMeshInstance, scene.cpp:461-523
void szg_tick_mesh_instance(uint32_t animation, const szg_transform* originals, szg_transform* transforms, uint32_t count,
                            double time_elapsed_seconds, double delta_time_seconds, szg_mat4* out_models,
                            szg_mat4* out_model_inverse_transposes)
{
    for (uint32_t i = 0; i < count; i++)
    {
        szg_transform const& original = originals[i];
        szg_transform& current = transforms[i];
        if (animation == SZG_INSTANCE_ANIMATION_DIAGONAL_WAVE)
        {
            // scene.cpp:490-500: the sum is float arithmetic, the division and the sine are double
            float const diagonal = original.translation[0] - (-10.0f) + original.translation[2] - (-10.0f);
            double const timeOffset = diagonal / 3.1415;
            double const y = std::sin(time_elapsed_seconds + timeOffset);
            current.translation[0] = original.translation[0] + 0.0f;
            current.translation[1] = original.translation[1] + static_cast<float>(y);
            current.translation[2] = original.translation[2] + 0.0f;
        }
        else if (animation == SZG_INSTANCE_ANIMATION_SPIN_ALONG_WORLD_UP)
        {
            current.eulerAnglesRadians[2] += static_cast<float>(delta_time_seconds); // scene.cpp:508-509
        }
    }
    for (uint32_t i = 0; i < count; i++) // scene.cpp:515-522
    {
        szg_transform_matrix(transforms[i].translation, transforms[i].eulerAnglesRadians, transforms[i].scale, &out_models[i]);
        szg_mat4_inverse_transpose(&out_models[i], &out_model_inverse_transposes[i]);
    }
}

// scene.cpp:52-75
void szg_atmosphere_default_earth(szg_atmosphere* a)
{
    std::memset(a, 0, sizeof *a);
    float constexpr KILOMETERS_PER_MEGAMETER = 1000.0f;
    a->sunEulerAngles[0] = 1.0f;
    a->planetRadiusMegameters = 6.360f;
    a->atmosphereRadiusMegameters = 6.420f;
    a->groundColor[0] = a->groundColor[1] = a->groundColor[2] = 1.0f;
    a->scatteringRayleighPerMegameter[0] = 5.802f;
    a->scatteringRayleighPerMegameter[1] = 13.558f;
    a->scatteringRayleighPerMegameter[2] = 33.1f;
    a->altitudeDecayRayleighMegameters = 8.0f / KILOMETERS_PER_MEGAMETER;
    a->scatteringMiePerMegameter[0] = a->scatteringMiePerMegameter[1] = a->scatteringMiePerMegameter[2] = 3.996f;
    a->absorptionMiePerMegameter[0] = a->absorptionMiePerMegameter[1] = a->absorptionMiePerMegameter[2] = 4.40f;
    a->altitudeDecayMieMegameters = 1.2f / KILOMETERS_PER_MEGAMETER;
    a->absorptionOzonePerMegameter[0] = 0.650f;
    a->absorptionOzonePerMegameter[1] = 1.881f;
    a->absorptionOzonePerMegameter[2] = 0.085f;
    a->sunIntensitySpectrum[0] = a->sunIntensitySpectrum[1] = a->sunIntensitySpectrum[2] = 1.0f;
    a->sunAngularRadius = radians(32.0f / 60.0f);
}

// scene.cpp:77-83
void szg_camera_default(szg_camera* c)
{
    std::memset(c, 0, sizeof *c);
    c->cameraPosition[0] = 0.0f;
    c->cameraPosition[1] = -10.0f;
    c->cameraPosition[2] = -13.0f;
    c->fovDegrees = 70.0f;
    c->near_plane = 0.1f;
    c->far_plane = 10000.0f;
    c->orthographic = 0;
}

// scene.cpp:87-89
void szg_sun_animation_default(szg_sun_animation* s)
{
    s->frozen = 0;
    s->time = 0.5f;
    s->speed = 100.0f;
    s->skipNight = 0;
}

// scene.cpp:689-692
void szg_atmosphere_direction_to_sun(const szg_atmosphere* a, float out[3]) { store(out, -forwardFromEulers(v3(a->sunEulerAngles))); }

// scene.cpp:694-716
void szg_atmosphere_to_device_equivalent(const szg_atmosphere* a, szg_atmosphere_packed* out)
{
    std::memset(out, 0, sizeof *out);
    V3 sunDirection = normalize(-forwardFromEulers(v3(a->sunEulerAngles)));
    sunDirection.y *= -1.0f; // sky-view shaders use +y as up
    std::memcpy(out->scatteringRayleighPerMm, a->scatteringRayleighPerMegameter, 12);
    out->densityScaleRayleighMm = a->altitudeDecayRayleighMegameters;
    std::memcpy(out->absorptionRayleighPerMm, a->absorptionRayleighPerMegameter, 12);
    out->planetRadiusMm = a->planetRadiusMegameters;
    std::memcpy(out->scatteringMiePerMm, a->scatteringMiePerMegameter, 12);
    out->densityScaleMieMm = a->altitudeDecayMieMegameters;
    std::memcpy(out->absorptionMiePerMm, a->absorptionMiePerMegameter, 12);
    out->atmosphereRadiusMm = a->atmosphereRadiusMegameters;
    store(out->incidentDirectionSun, -sunDirection);
    std::memcpy(out->scatteringOzonePerMm, a->scatteringOzonePerMegameter, 12);
    std::memcpy(out->absorptionOzonePerMm, a->absorptionOzonePerMegameter, 12);
    std::memcpy(out->sunIntensitySpectrum, a->sunIntensitySpectrum, 12);
    out->sunAngularRadius = a->sunAngularRadius;
}

// lights.cpp:9-27
void szg_make_directional(const float color[4], float strength, const float eulers[3], const szg_aabb* captured_bounds,
                          szg_directional_light_packed* out)
{
    makeDirectional(color, strength, v3(eulers), *captured_bounds, out);
}

// scene.cpp:718-737 with createSunlight (:584-598) and createMoonlight (:599-623)
void szg_atmosphere_baked(const szg_atmosphere* a, const szg_aabb* scene_bounds, szg_atmosphere_packed* out_atmosphere,
                          szg_directional_light_packed* out_sunlight, szg_directional_light_packed* out_moonlight)
{
    V3 const directionToSun = -forwardFromEulers(v3(a->sunEulerAngles));
    float const sunCosine = dot(WORLD_UP, directionToSun);
    float constexpr SUNSET_COSINE = 0.06f;

    if (out_sunlight != nullptr)
    {
        float constexpr SUNLIGHT_STRENGTH = 4.0f;
        float const color[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        makeDirectional(color, SUNLIGHT_STRENGTH, v3(a->sunEulerAngles), *scene_bounds, out_sunlight);
    }
    if (out_moonlight != nullptr)
    {
        float constexpr MOONRISE_LENGTH = 0.12f;
        // glm::clamp(0.0F, 1.0F, v) as written in the reference: min(max(0, 1), v)
        float const v = std::fabs(sunCosine - SUNSET_COSINE) / MOONRISE_LENGTH;
        float const lo = 1.0f;
        float const x0 = 0.0f;
        float const clamped = std::fmin(std::fmax(x0, lo), v);
        float const moonlightStrength = 0.02f * clamped;
        float const color[4] = {0.3f, 0.4f, 0.6f, 1.0f};
        makeDirectional(color, moonlightStrength, V3{-HALF_PI_F, 0.0f, 0.0f}, *scene_bounds, out_moonlight);
    }
    if (out_atmosphere != nullptr)
    {
        szg_atmosphere_to_device_equivalent(a, out_atmosphere);
    }
}

// scene.cpp:739-794
void szg_camera_to_device_equivalent(const szg_camera* c, float aspect_ratio, szg_camera_packed* out)
{
    std::memset(out, 0, sizeof *out);
    V3 const pos = v3(c->cameraPosition);
    V3 const eul = v3(c->eulerAngles);

    szg_mat4 proj;
    if (c->orthographic != 0)
    {
        float const height = std::tan(radians(c->fovDegrees) / 2.0f);
        V3 const mn{-aspect_ratio * height, -height, c->near_plane};
        V3 const mx{aspect_ratio * height, height, c->far_plane};
        proj = projectionOrthoVk(mn, mx);
    }
    else
    {
        proj = projectionVk(c->fovDegrees, aspect_ratio, c->near_plane, c->far_plane);
    }
    szg_mat4 const view = viewVk(pos, eul);
    szg_mat4 const rotation = orientate4(eul);

    out->projection = proj;
    out->inverseProjection = inverse(proj);
    out->view = view;
    out->viewInverseTranspose = transpose(inverse(view));
    out->rotation = rotation;
    out->projViewInverse = inverse(mul(proj, view));
    V4 const fw = mul(rotation, V4{WORLD_FORWARD.x, WORLD_FORWARD.y, WORLD_FORWARD.z, 0.0f});
    out->forwardWorld[0] = fw.x;
    out->forwardWorld[1] = fw.y;
    out->forwardWorld[2] = fw.z;
    out->forwardWorld[3] = fw.w;
    out->position[0] = pos.x;
    out->position[1] = pos.y;
    out->position[2] = pos.z;
    out->position[3] = 1.0f;
}

// lights.cpp:29-46
void szg_make_spot(const szg_spotlight_params* p, szg_spot_light_packed* out)
{
    std::memset(out, 0, sizeof *out);
    std::memcpy(out->color, p->color, 16);
    V3 const f = forwardFromEulers(v3(p->eulerAngles));
    out->forward[0] = f.x;
    out->forward[1] = f.y;
    out->forward[2] = f.z;
    out->forward[3] = 0.0f;
    out->projection = projectionVk(p->verticalFOVDegrees, p->horizontalScale, p->near_plane, p->far_plane);
    out->view = viewVk(v3(p->position), v3(p->eulerAngles));
    out->position[0] = p->position[0];
    out->position[1] = p->position[1];
    out->position[2] = p->position[2];
    out->position[3] = 1.0f;
    out->strength = p->strength;
    out->falloffFactor = p->falloffFactor;
    out->falloffDistance = p->falloffDistance;
}

// scene.cpp:218-229
void szg_spotlight_params_default(const float color_rgb[3], const float position[3], const float eulers[3],
                                  szg_spotlight_params* out)
{
    std::memset(out, 0, sizeof *out);
    out->color[0] = color_rgb[0];
    out->color[1] = color_rgb[1];
    out->color[2] = color_rgb[2];
    out->color[3] = 1.0f;
    out->strength = 1000.0f;
    out->falloffFactor = 1.0f;
    out->falloffDistance = 1.0f;
    out->verticalFOVDegrees = 30.0f;
    out->horizontalScale = 1.0f;
    std::memcpy(out->eulerAngles, eulers, 12);
    std::memcpy(out->position, position, 12);
    out->near_plane = 0.1f;
    out->far_plane = 1000.0f;
}

// scene.cpp:532-574
void szg_scene_tick_sun(szg_sun_animation* anim, szg_atmosphere* atmosphere, double delta_time_seconds)
{
    float constexpr DAY_LENGTH_SECONDS = 60.0f * 60.0f * 24.0f;
    if (anim->frozen == 0)
    {
        float const t = anim->time + anim->speed * static_cast<float>(delta_time_seconds) / DAY_LENGTH_SECONDS;
        anim->time = t - std::floor(t); // glm::fract
    }
    if (anim->skipNight != 0 && anim->frozen == 0)
    {
        float constexpr SUNSET_LENGTH_TIME = 0.015f;
        float constexpr HORIZON_A_TIME = 0.25f - SUNSET_LENGTH_TIME;
        float constexpr HORIZON_B_TIME = 0.75f + SUNSET_LENGTH_TIME;
        bool const isNight = anim->time < HORIZON_A_TIME || anim->time > HORIZON_B_TIME;
        if (isNight)
        {
            bool const sunRisesAtA = anim->speed > 0.0f;
            anim->time = sunRisesAtA ? HORIZON_A_TIME : HORIZON_B_TIME;
        }
    }
    float constexpr SUN_START_RADIANS = HALF_PI_F;
    float constexpr SUN_END_RADIANS = SUN_START_RADIANS + TWO_PI_F;
    // glm::lerp(x, y, a) = x * (1 - a) + y * a
    atmosphere->sunEulerAngles[0] = SUN_START_RADIANS * (1.0f - anim->time) + SUN_END_RADIANS * anim->time;
    (void)PI_F;
}

} // extern "C"
