/*
 * szg/raster.h — C-ABI of the real-mesh G-buffer and shadow-map raster passes:
 * the two fixed-function passes of DeferredShadingPipeline::recordDrawCommands that come BEFORE the
 * compute path, rebuilt as a HIP compute rasteriser.
 *
 *   G-buffer pass   renderer/pipelines/deferred.cpp:493-713, raster state :342-392 + :509
 *                   (viewport = draw rect, depth 0..1; triangle list; CULL_BACK, front face CLOCKWISE;
 *                   depth test GREATER, write on, clear 0; 5 colour attachments cleared to 0)
 *                   shaders deferred/offscreen.vert:41-56, deferred/offscreen.frag:25-79
 *   shadow pass     renderer/pipelines.cpp:593-806 (depth only; CULL_FRONT, front face CLOCKWISE;
 *                   GREATER_OR_EQUAL; clear 0; depth bias constant/slope from ShadowPassParameters,
 *                   shadowpass.hpp:24-28, both 0 by default), offscreenpass/depthpass.vert:30-38
 *
 * The fixed-function rasteriser is implementation-defined at the bit level (sub-pixel snapping, derivative
 * evaluation, interpolation precision), so this pass has no bit-level reference: "parity unpinned" (DESIGN.md).
 * The rules below are the ones Vulkan specifies, stated exactly so that the CPU oracle and the kernels agree
 * bit for bit:
 *
 *   vertex       world = model * (position, 1); clip = (projection * view) * world   (offscreen.vert:46-51:
 *                GLSL `P * V * p` multiplies the matrices first); normal = normalize((MIT * (n, 0)).xyz).
 *                Shadow: clip = (projView * model) * (position, 1)   (depthpass.vert:37).
 *   clipping     none is performed: coverage and interpolation are evaluated in homogeneous clip space
 *                (Olano & Greer 1997), which is what clipping + rasterising the pieces computes. A fragment
 *                exists only where 0 <= z_clip <= w_clip (Vulkan's depth clip volume).
 *   coverage     pixel centre (x + .5, y + .5). With h = ((x_c + w_c) * W/2, (y_c + w_c) * H/2, w_c) per vertex,
 *                edge function E_i = cross(h_j, h_k) . (px, py, 1) for (i, j, k) cyclic. A pixel is covered
 *                when s*E_i > 0 for the three edges, s = sign of det(h_0, h_1, h_2); a pixel centre exactly on
 *                an edge belongs to the triangle whose edge is a left edge (s*a_i > 0) or a top edge
 *                (a_i == 0 and s*b_i > 0) — the top-left rule. Two triangles that share an edge get exactly
 *                negated coefficients, so the raster is watertight and never double-hits.
 *   facing       det > 0 is clockwise in framebuffer space = front-facing (deferred.cpp:380).
 *   depth        z = (sum E_i z_i) / (sum E_i w_i); attributes = (sum E_i a_i) / (sum E_i) (perspective correct).
 *   order        primitives are numbered in submission order (mesh, surface, instance, triangle:
 *                vkCmdDrawIndexed with instanceCount, deferred.cpp:691-698); with GREATER the earliest
 *                primitive wins a depth tie.
 *   derivatives  dFdx / dFdy are fine derivatives of the 2x2 pixel quad: the covering triangle's interpolant
 *                evaluated at the quad's two pixels of this row (column), right minus left (bottom minus top),
 *                helper pixels included.
 *   textures     RGBA8 (UNORM or SRGB), one mip level (image.cpp:86), LINEAR / REPEAT sampler
 *                (material.cpp:115-120): u*W - .5, floor, positive modulo, fp32 weights.
 */
#ifndef SZG_RASTER_H
#define SZG_RASTER_H

#include "szg/abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* renderer/gputypes.hpp:117-126 VertexPacked / shaders/types/vertex.glsl */
typedef struct szg_vertex_packed
{
    float position[3];
    float uv_x;
    float normal[3];
    float uv_y;
    float color[4];
} szg_vertex_packed;

/* One RGBA8 texture in DEVICE memory (assets.cpp:271: R8G8B8A8_UNORM or _SRGB), rows top to bottom. */
typedef struct szg_texture
{
    const void* data;
    uint32_t width, height;
    uint32_t pitch_bytes;
    uint32_t srgb; /* 1: texels are sRGB-encoded, decoded before filtering */
} szg_texture;

/* renderer/material.hpp MaterialData: the three maps bound at set 3 (offscreen.frag:19-21) */
typedef struct szg_material
{
    szg_texture color;
    szg_texture normal;
    szg_texture orm;
} szg_material;

/* assets/assets.hpp:30-35 GeometrySurface */
typedef struct szg_surface
{
    uint32_t first_index;
    uint32_t index_count;
    szg_material material;
} szg_surface;

/* renderer/scene.hpp:109-147 MeshInstanced + assets.hpp:37-44 Mesh, flattened.
 * d_* point to DEVICE memory; `surfaces` is a HOST array. */
typedef struct szg_mesh_instanced
{
    const szg_vertex_packed* d_vertices;
    uint32_t vertex_count; /* indices >= vertex_count drop their triangle */
    uint32_t index_count;  /* size of d_indices; surfaces reaching past it are truncated */
    const uint32_t* d_indices;
    const szg_surface* surfaces;
    uint32_t surface_count;
    uint32_t instance_count;                     /* models.deviceSize() */
    const szg_mat4* d_models;                    /* MeshInstanced::models */
    const szg_mat4* d_model_inverse_transposes;  /* MeshInstanced::modelInverseTransposes (unused by the shadow pass) */
    uint32_t render;                             /* MeshInstanced::render */
    uint32_t casts_shadow;                       /* MeshInstanced::castsShadow */
} szg_mesh_instanced;

/* The G-buffer pass of DeferredShadingPipeline::recordDrawCommands (deferred.cpp:493-713): clears the five
 * G-buffer planes and scene_texture->depth over the draw rect and rasterises every rendered mesh into them.
 * Row tiles as in abi.h (a rank rasterises only its rows). */
int szg_deferred_record_gbuffer_raster(szg_deferred_t* p, void* stream, szg_rect draw_rect, const szg_rowtile* tile,
                                       const szg_scene_texture* scene_texture, uint32_t view_camera_index,
                                       const szg_camera_packed* d_cameras, const szg_mesh_instanced* meshes,
                                       uint32_t mesh_count);

/* The shadow passes (ShadowPassArray::recordInitialize + recordDrawCommands, shadowpass.cpp:188-270): one depth-only
 * raster per light into the maps the pipeline owns, slots = directional lights then spot lights, capped at
 * max_shadow_maps. Depth bias from szg_deferred_set_configuration. */
int szg_deferred_record_shadow_raster(szg_deferred_t* p, void* stream,
                                      const szg_directional_light_packed* d_directional_lights,
                                      uint32_t directional_light_count, const szg_spot_light_packed* h_spot_lights,
                                      uint32_t spot_light_count, const szg_mesh_instanced* meshes, uint32_t mesh_count);

/* DeferredShadingPipeline::recordDrawCommands (deferred.hpp:60-70) with real scene geometry: shadow raster
 * (when the pipeline owns maps), G-buffer raster, then the lights pass. */
int szg_deferred_record_draw_commands_meshes(
    szg_deferred_t* p, void* stream, szg_rect draw_rect, const szg_rowtile* tile, const szg_scene_texture* scene_texture,
    uint32_t atmospheric_directional_lights_count, const szg_directional_light_packed* d_directional_lights,
    uint32_t directional_light_count, const szg_spot_light_packed* h_spot_lights, uint32_t spot_light_count,
    uint32_t view_camera_index, const szg_camera_packed* d_cameras, const szg_mesh_instanced* meshes, uint32_t mesh_count);

#ifdef __cplusplus
} /* extern "C" */
static_assert(sizeof(szg_vertex_packed) == 48, "gputypes.hpp:126");
#endif

#endif /* SZG_RASTER_H */
