/*
 * szg/host.h — host-side input preparation (CPU only): scene parameters ->
 * the packed blocks of szg/abi.h. Restates the glm-based helpers of the
 * reference (glm 1.0.1, cmake/dependencies.cmake:39-46, is not vendored; its
 * published formulas are restated, compile definitions
 * GLM_FORCE_DEPTH_ZERO_TO_ONE / GLM_FORCE_RADIANS: syzygy/CMakeLists.txt:81-94).
 *
 * Pinned by the reference's own 22 euler known-answer cases
 * (geometry/geometrytests.cpp:120-186) for the euler<->forward convention;
 * projection / inverse are unpinned by the reference and are checked against an
 * independent numpy restatement in tests/.
 */
#ifndef SZG_HOST_H
#define SZG_HOST_H

#include "szg/abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* geometry/geometrystatics.hpp:7-9 */
#define SZG_WORLD_FORWARD_X 0.0f
#define SZG_WORLD_FORWARD_Y 0.0f
#define SZG_WORLD_FORWARD_Z 1.0f
#define SZG_WORLD_UP_X 0.0f
#define SZG_WORLD_UP_Y (-1.0f)
#define SZG_WORLD_UP_Z 0.0f

/* geometry/geometrytypes.hpp:22-38 */
typedef struct szg_aabb
{
    float center[3];
    float half_extent[3];
} szg_aabb;

/* renderer/scene.hpp Atmosphere (defaults scene.cpp:52-75) */
typedef struct szg_atmosphere
{
    float sunEulerAngles[3]; /* (pitch, roll, yaw) */
    float planetRadiusMegameters;
    float atmosphereRadiusMegameters;
    float groundColor[3]; /* not sent to the device (scene.cpp:701-715) */
    float scatteringRayleighPerMegameter[3];
    float absorptionRayleighPerMegameter[3];
    float altitudeDecayRayleighMegameters;
    float scatteringMiePerMegameter[3];
    float absorptionMiePerMegameter[3];
    float altitudeDecayMieMegameters;
    float scatteringOzonePerMegameter[3];
    float absorptionOzonePerMegameter[3];
    float sunIntensitySpectrum[3];
    float sunAngularRadius;
} szg_atmosphere;

/* renderer/scene.hpp Camera (defaults scene.cpp:77-83) */
typedef struct szg_camera
{
    float cameraPosition[3];
    float eulerAngles[3];
    float fovDegrees;
    float near_plane;
    float far_plane;
    uint32_t orthographic;
} szg_camera;

/* renderer/lights.hpp:18-30 */
typedef struct szg_spotlight_params
{
    float color[4];
    float strength;
    float falloffFactor;
    float falloffDistance;
    float verticalFOVDegrees;
    float horizontalScale;
    float eulerAngles[3];
    float position[3];
    float near_plane;
    float far_plane;
} szg_spotlight_params;

/* renderer/scene.hpp SunAnimation (defaults scene.cpp:87-91) */
typedef struct szg_sun_animation
{
    uint32_t frozen;
    float time;
    float speed;
    uint32_t skipNight;
} szg_sun_animation;

/* geometryhelpers.cpp:102-105 / :107-145 */
void szg_forward_from_eulers(const float eulers[3], float out_forward[3]);
void szg_eulers_from_forward(const float forward[3], float out_eulers[3]);
/* geometryhelpers.cpp:83-95 (perspectiveLH_ZO with near/far swapped: reverse-Z) */
void szg_projection_vk(float fov_y_degrees, float aspect, float near_plane, float far_plane, szg_mat4* out);
/* geometryhelpers.cpp:97-100 */
void szg_projection_ortho_vk(const float min[3], const float max[3], szg_mat4* out);
/* geometryhelpers.cpp:147-157 */
void szg_transform_vk(const float position[3], const float eulers[3], szg_mat4* out);
void szg_view_vk(const float position[3], const float eulers[3], szg_mat4* out);
/* geometry/transform.cpp:11-15 Transform::toMatrix (model matrix of a mesh instance, scene.cpp:205-211) */
void szg_transform_matrix(const float translation[3], const float eulers[3], const float scale[3], szg_mat4* out);

/* geometry/transform.hpp Transform */
typedef struct szg_transform
{
    float translation[3];
    float eulerAnglesRadians[3];
    float scale[3];
} szg_transform;
/* renderer/scene.hpp:96-105 InstanceAnimation */
#define SZG_INSTANCE_ANIMATION_NONE 0u
#define SZG_INSTANCE_ANIMATION_DIAGONAL_WAVE 1u
#define SZG_INSTANCE_ANIMATION_SPIN_ALONG_WORLD_UP 2u
/* One shadow-casting MeshInstanced for Scene::calculateShadowBounds: Mesh::vertexBounds + the instance transforms. */
typedef struct szg_shadow_caster
{
    szg_aabb vertex_bounds;          /* assets.hpp:40 Mesh::vertexBounds (AABB::create(min, max) of the vertex positions) */
    const szg_transform* transforms; /* MeshInstanced::transforms */
    uint32_t transform_count;
    uint32_t render;                 /* MeshInstanced::render */
    uint32_t casts_shadow;           /* MeshInstanced::castsShadow */
    uint32_t padding;
} szg_shadow_caster;
/* geometrytypes.cpp:11-19 AABB::create */
void szg_aabb_create(const float min[3], const float max[3], szg_aabb* out);
/* Transform::lookAt(Ray::create(from, to), scale) (geometry/transform.cpp:17-28): translation = from, eulers =
 * eulersFromForward(normalize(to - from)). */
void szg_transform_look_at(const float from[3], const float to[3], const float scale[3], szg_transform* out);

/* Scene::calculateShadowBounds (scene.cpp:95-148): the world AABB of every rendered shadow caster's transformed
 * vertex-bounds corners; the `captured_bounds` of szg_atmosphere_baked / szg_make_directional. Returns 0 and leaves
 * *out zeroed when no caster contributes (scene.cpp:138-143), 1 otherwise. */
int szg_calculate_shadow_bounds(const szg_shadow_caster* casters, uint32_t caster_count, szg_aabb* out);

/* tickMeshInstance (scene.cpp:461-523): advance `transforms` of one MeshInstanced by its animation and refill the
 * staged `models` / `modelInverseTransposes` arrays (count entries each) that the raster passes read. */
void szg_tick_mesh_instance(uint32_t animation, const szg_transform* originals, szg_transform* transforms, uint32_t count,
                            double time_elapsed_seconds, double delta_time_seconds, szg_mat4* out_models,
                            szg_mat4* out_model_inverse_transposes);
/* geometryhelpers.cpp:171-204 */
void szg_projection_ortho_aabb_vk(const szg_mat4* view, const szg_aabb* bounds, szg_mat4* out);
/* glm::inverse / glm::inverseTranspose / operator* on mat4 */
void szg_mat4_inverse(const szg_mat4* m, szg_mat4* out);
void szg_mat4_inverse_transpose(const szg_mat4* m, szg_mat4* out);
void szg_mat4_mul(const szg_mat4* a, const szg_mat4* b, szg_mat4* out);

/* scene.cpp:52-75, :77-83, :87-91 */
void szg_atmosphere_default_earth(szg_atmosphere* out);
void szg_camera_default(szg_camera* out);
void szg_sun_animation_default(szg_sun_animation* out);
/* scene.cpp:689-692, :694-716, :718-737 (index 0 = sun, 1 = moon: renderer.cpp:312-329) */
void szg_atmosphere_direction_to_sun(const szg_atmosphere* a, float out[3]);
void szg_atmosphere_to_device_equivalent(const szg_atmosphere* a, szg_atmosphere_packed* out);
void szg_atmosphere_baked(const szg_atmosphere* a, const szg_aabb* scene_bounds, szg_atmosphere_packed* out_atmosphere,
                          szg_directional_light_packed* out_sunlight, szg_directional_light_packed* out_moonlight);
/* scene.cpp:739-794 */
void szg_camera_to_device_equivalent(const szg_camera* c, float aspect_ratio, szg_camera_packed* out);
/* lights.cpp:9-27, :29-46 */
void szg_make_directional(const float color[4], float strength, const float eulers[3], const szg_aabb* captured_bounds,
                          szg_directional_light_packed* out);
void szg_make_spot(const szg_spotlight_params* params, szg_spot_light_packed* out);
/* scene.cpp:218-229 addSpotlight defaults (strength 1000, falloff 1/1, fov 30, near .1, far 1000) */
void szg_spotlight_params_default(const float color_rgb[3], const float position[3], const float eulers[3],
                                  szg_spotlight_params* out);
/* scene.cpp:532-574: advance the sun animation by dt seconds and set sunEulerAngles.x */
void szg_scene_tick_sun(szg_sun_animation* anim, szg_atmosphere* atmosphere, double delta_time_seconds);

#ifdef __cplusplus
}
#endif
#endif /* SZG_HOST_H */
