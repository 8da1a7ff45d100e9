/*
 * szg/abi.h — data ABI + C entry points of the MI355X-native deferred-shading +
 * Hillaire-atmosphere path.
 *
 * This header is the drop-in boundary. Everything here is plain C: PODs, device
 * pointers, sizes, int status codes. No HIP, torch or C++ types cross it
 * (`void* stream` is a hipStream_t; NULL = the default stream).
 *
 * Every item cites the reference interface it replaces, as
 * `path:line` relative to the reference checkout (syzygy/source/syzygy/... or
 * shaders/...).
 *
 * Conventions (reference geometryhelpers.hpp:16-29, skyview_LUT.comp:106-112):
 *   engine/world space : +x right, +y DOWN, +z forward, metres
 *   atmosphere space   : +y UP, megametres, origin at the planet centre
 *   matrices           : column-major, M*v
 *   all arithmetic     : fp32
 */
#ifndef SZG_ABI_H
#define SZG_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SZG_ABI_VERSION 1

/* ------------------------------------------------------------------------- */
/* Status codes (reference: construction returns optional/nullptr + log,      */
/* record* never fail — deferred.cpp:153-163, skyview.cpp:713-740).           */
/* ------------------------------------------------------------------------- */
#define SZG_OK 0
#define SZG_ERR_INVALID_ARGUMENT (-1)
#define SZG_ERR_NO_DEVICE (-2)
#define SZG_ERR_OUT_OF_MEMORY (-3)
#define SZG_ERR_HIP (-4)
#define SZG_ERR_CAPACITY (-5)
#define SZG_ERR_TIMEOUT (-8) /* a collective rendezvous did not complete within its deadline (szg_rowtile_comm_create) */

/* ------------------------------------------------------------------------- */
/* Packed parameter blocks — byte-identical to renderer/gputypes.hpp:17-126   */
/* (std430 mirrors of shaders/types/{camera.glsl,atmosphere.glinl,lights.glsl}) */
/* ------------------------------------------------------------------------- */
typedef struct szg_mat4
{
    float m[16]; /* column-major: m[col*4 + row] */
} szg_mat4;

/* gputypes.hpp:17-36, shaders/types/camera.glsl:1-18 */
typedef struct szg_camera_packed
{
    szg_mat4 projection;
    szg_mat4 inverseProjection;
    szg_mat4 view;
    szg_mat4 viewInverseTranspose;
    szg_mat4 rotation;
    szg_mat4 projViewInverse;
    float forwardWorld[4];
    float position[4];
} szg_camera_packed;

/* gputypes.hpp:39-72, shaders/types/atmosphere.glinl:3-32 */
typedef struct szg_atmosphere_packed
{
    float scatteringRayleighPerMm[3];
    float densityScaleRayleighMm;
    float absorptionRayleighPerMm[3];
    float planetRadiusMm;
    float scatteringMiePerMm[3];
    float densityScaleMieMm;
    float absorptionMiePerMm[3];
    float atmosphereRadiusMm;
    float incidentDirectionSun[3]; /* atmosphere space, direction light travels */
    uint32_t padding0;
    float scatteringOzonePerMm[3];
    uint32_t padding1;
    float absorptionOzonePerMm[3];
    uint32_t padding2;
    float sunIntensitySpectrum[3];
    float sunAngularRadius;
} szg_atmosphere_packed;

/* gputypes.hpp:74-90, shaders/types/lights.glsl:1-12 */
typedef struct szg_directional_light_packed
{
    float color[4];
    float forward[4];
    szg_mat4 projection;
    szg_mat4 view;
    float strength;
    uint32_t padding0[3];
} szg_directional_light_packed;

/* gputypes.hpp:92-115, shaders/types/lights.glsl:14-29 */
typedef struct szg_spot_light_packed
{
    float color[4];
    float forward[4];
    szg_mat4 projection;
    szg_mat4 view;
    float position[4];
    float strength;
    float falloffFactor;
    float falloffDistance;
    uint32_t padding0;
} szg_spot_light_packed;

/* ------------------------------------------------------------------------- */
/* Push-constant blocks, kept byte-identical so a caller that already fills    */
/* the reference's structs can hand them over unchanged. Device addresses are  */
/* plain 64-bit HIP device pointers.                                           */
/* ------------------------------------------------------------------------- */
/* skyview.hpp:64-72 (transmittance_LUT.comp:15-20), 16 B */
typedef struct szg_pc_transmittance
{
    uint64_t atmosphereBuffer;
    uint32_t atmosphereIndex;
    uint32_t padding;
} szg_pc_transmittance;

/* skyview.hpp:89-96 (skyview_LUT.comp:22-29), GLSL block size 32 B */
typedef struct szg_pc_skyview
{
    uint64_t atmosphereBuffer;
    uint64_t cameraBuffer;
    uint32_t atmosphereIndex;
    uint32_t cameraIndex;
    uint32_t padding[2];
} szg_pc_skyview;

/* skyview.hpp:125-144 (camera.comp:50-67), 64 B */
typedef struct szg_pc_composite
{
    uint64_t atmosphereBuffer;
    uint64_t cameraBuffer;
    uint32_t atmosphereIndex;
    uint32_t cameraIndex;
    uint32_t drawExtent[2];
    uint32_t sunShadowMapIndex;
    uint32_t padding0;
    uint32_t gbufferExtent[2];
    uint64_t directionalLights;
    uint32_t sunLightIndex;
    uint32_t padding1;
} szg_pc_composite;

/* deferred.hpp:82-98 (lights.comp:41-57), 64 B */
typedef struct szg_pc_lights
{
    uint64_t cameraBuffer;
    uint32_t padding0;
    uint32_t padding1;
    uint64_t directionalLightsBuffer;
    uint64_t spotLightsBuffer;
    uint32_t directionalLightCount;
    uint32_t spotLightCount;
    uint32_t directionalLightSkipCount;
    uint32_t cameraIndex;
    float gbufferOffset[2];
    float gbufferExtent[2];
} szg_pc_lights;

/* ------------------------------------------------------------------------- */
/* Images: Vulkan optimal-tiled images become linear row-major device buffers */
/* with an explicit pitch. Formats from gbuffer.cpp:27-40, skyview.cpp:78-88,  */
/* 175-185, editor/uilayer.cpp:285-308, shadowpass.cpp:22-60.                  */
/* ------------------------------------------------------------------------- */
typedef enum szg_format
{
    SZG_FORMAT_UNDEFINED = 0,
    SZG_FORMAT_RGBA16_SFLOAT = 1, /* 8 B/texel  : G-buffer diffuse/specular/normal/ORM */
    SZG_FORMAT_RGBA32_SFLOAT = 2, /* 16 B/texel : G-buffer worldPosition, both LUTs, debug colour */
    SZG_FORMAT_RGBA16_UNORM = 3,  /* 8 B/texel  : scene colour */
    SZG_FORMAT_D32_SFLOAT = 4     /* 4 B/texel  : scene depth, shadow maps (reverse-Z, 0 = far) */
} szg_format;

typedef struct szg_image
{
    void* data;           /* device pointer; NULL = absent */
    uint32_t width;       /* ALLOCATED extent in texels (imageSize() of the reference) */
    uint32_t height;
    uint32_t pitch_bytes; /* row pitch, >= width * texel size */
    uint32_t format;      /* szg_format */
} szg_image;

/* VkRect2D of the reference API. The reference dispatches every pass over the extent only and pushes
 * gbufferOffset = 0 (lights.comp:112 with deferred.cpp:764, :778-787; camera.comp has no offset at all,
 * skyview.cpp:658-665): whatever the offset says, it shades the top-left width x height texels. A non-zero x or y
 * is therefore a request this path does not serve: every record_* entry point refuses it with
 * SZG_ERR_INVALID_ARGUMENT instead of silently rendering somewhere else. */
typedef struct szg_rect
{
    int32_t x, y;
    uint32_t width, height;
} szg_rect;

/* renderer/gbuffer.hpp:17-48: five separate planes, NEAREST clamp-to-edge. */
typedef struct szg_gbuffer
{
    szg_image diffuse;       /* RGBA16F, a = 1 geometry / 0 background (offscreen.frag:71) */
    szg_image specular;      /* RGBA16F */
    szg_image normal;        /* RGBA16F, w = 0 */
    szg_image worldPosition; /* RGBA32F, w = 1 */
    szg_image occlusionRoughnessMetallic; /* RGBA16F */
} szg_gbuffer;

/* renderer/scenetexture.hpp:11-81: colour + depth render target. `debug_color`
 * has no reference counterpart: when non-NULL each pass also writes its
 * pre-quantisation fp32 RGBA result there, for parity tests. */
typedef struct szg_scene_texture
{
    szg_image color;       /* RGBA16_UNORM */
    szg_image depth;       /* D32_SFLOAT, NEAREST clamp-to-border(0) */
    szg_image debug_color; /* RGBA32_SFLOAT or data == NULL */
} szg_scene_texture;

/* renderer/shadowpass.hpp:33-93 (consumer side). maps[i].data == NULL means
 * "no shadow map for this light": shadow factor 1.0, identical to sampling a
 * cleared map (shadowmap.glinl:56). `maps` is a HOST array. */
typedef struct szg_shadowmaps
{
    uint32_t count;
    uint32_t padding;
    const szg_image* maps;
} szg_shadowmaps;

/* Row tiling for multi-GPU (no reference counterpart). The frame's
 * rows are cut into blocks of `block_rows`; block b belongs to rank b % nranks.
 * A rank's images hold only its own rows, packed: local row l maps to global
 * row ((l / block_rows) * nranks + rank) * block_rows + l % block_rows.
 * nranks == 1 (or a NULL pointer) is the whole frame. */
typedef struct szg_rowtile
{
    uint32_t block_rows;
    uint32_t rank;
    uint32_t nranks;
    uint32_t local_rows; /* number of rows this rank holds */
} szg_rowtile;

/* ------------------------------------------------------------------------- */
/* Synthetic geometry for the G-buffer fill (the rasteriser is   */
/* replaced by an analytic ray-cast that reproduces the output conventions of  */
/* deferred/offscreen.frag:61-79 and the raster state deferred.cpp:342-392).   */
/* ------------------------------------------------------------------------- */
typedef struct szg_fill_box
{
    float center[3];
    float half_extent[3];
    float metallic;
    float roughness;
} szg_fill_box;

typedef struct szg_fill_scene
{
    float ground_y;           /* ground plane y (world, +y down), e.g. -1 (editor.cpp:541) */
    float ground_half_extent; /* plane covers |x|,|z| <= this */
    float checker_cell;       /* checkerboard cell in metres (assets.cpp:1343-1348 colours) */
    float ground_roughness;   /* default 60/255 (assets.cpp:1311) */
    uint32_t box_count;
    uint32_t padding;
    const szg_fill_box* boxes; /* HOST array */
} szg_fill_scene;

/* ------------------------------------------------------------------------- */
/* Library                                                                     */
/* ------------------------------------------------------------------------- */
int szg_abi_version(void);
/* What this binary was built from: the hash of its sources that the build stamped into it (__graft_entry__.source_hash("hip"),
 * passed as -DSZG_SOURCE_HASH), and the contraction mask (szg/contraction.h) it was compiled with: "<hash> contract=0x3616".
 * "unknown" for a build made by plain `make`. bench.py and smoke() print it, so that a record names the binary that ran,
 * not the tree that happened to be on disk. */
const char* szg_build_id(void);
/* Last error text of the calling thread ("" if none). */
const char* szg_last_error(void);
/* Number of visible HIP devices, or a negative status. */
int szg_device_count(void);

/* ------------------------------------------------------------------------- */
/* SkyViewComputePipeline (renderer/pipelines/skyview.hpp:24-51)               */
/* ------------------------------------------------------------------------- */
/* STREAM RULE (both pipelines). A pipeline object owns per-pass device state that its record_* calls rewrite in stream
 * order and its kernels read back: the per-frame constant blocks (one for the LUT passes, one for the composite), the LUT
 * status words, the LUT-reuse key, the light records of the deferred pipeline. Therefore all record_* calls of ONE
 * pipeline object must be ordered on ONE stream (or ordered against each other by events), like the reference records
 * them into one command buffer - with the single exception stated at the calls themselves: the LUT passes
 * (szg_skyview_record_transmittance / _skyview_lut / _skyview_lut_rows, szg_skyview_allgather_lut_rows) may run on a
 * second stream than szg_skyview_record_composite, because they use their own constant block; the caller then orders
 * "LUTs complete" before "composite" with an event. Two composites of the same pipeline recorded on two unordered
 * streams race on the composite's block: use two pipeline objects for that. Calls from several host threads on one
 * object need external locking. */
typedef struct szg_skyview szg_skyview_t;

typedef struct szg_skyview_desc
{
    uint32_t transmittance_width;  /* reference: 512 (skyview.cpp:78, common.glinl:13) */
    uint32_t transmittance_height; /* reference: 128 */
    uint32_t skyview_width;        /* reference: 2048 (skyview.cpp:175) */
    uint32_t skyview_height;       /* reference: 1024 */
    uint32_t flags;                /* must be 0 */
    uint32_t padding;
} szg_skyview_desc;

/* No flags are defined yet. */

/* skyview.hpp:36-37 `create(device, allocator)`; returns SZG_OK and *out, or a
 * negative status and *out = NULL (reference: nullptr, skyview.cpp:713-740). */
int szg_skyview_create(szg_skyview_t** out, const szg_skyview_desc* desc, int device);
/* skyview.hpp:27-34 destructor / destroy(). NULL is allowed. */
void szg_skyview_destroy(szg_skyview_t* p);

/* skyview.hpp:39-51 recordDrawCommands: transmittance LUT -> sky-view LUT ->
 * camera composite, enqueued on `stream`, returns immediately
 * (skyview.cpp:751-911). All `d_*` are device pointers to arrays of the packed
 * structs (the reference's TStagedBuffer::deviceAddress()). */
int szg_skyview_record_draw_commands(
    szg_skyview_t* p, void* stream, const szg_scene_texture* scene_texture, szg_rect draw_rect,
    const szg_rowtile* tile, const szg_gbuffer* gbuffer, const szg_shadowmaps* shadow_maps,
    uint32_t atmosphere_index, const szg_atmosphere_packed* d_atmospheres, uint32_t view_camera_index,
    const szg_camera_packed* d_cameras, uint32_t sun_light_index,
    const szg_directional_light_packed* d_lights);

/* The three dispatches of recordDrawCommands, individually (skyview.cpp:795-845,
 * :847-893, :581-666). Tests and the bench time them separately. */
int szg_skyview_record_transmittance(szg_skyview_t* p, void* stream, uint32_t atmosphere_index,
                                     const szg_atmosphere_packed* d_atmospheres);
int szg_skyview_record_skyview_lut(szg_skyview_t* p, void* stream, uint32_t atmosphere_index,
                                   const szg_atmosphere_packed* d_atmospheres, uint32_t view_camera_index,
                                   const szg_camera_packed* d_cameras);
/* Multi-GPU extension (no reference counterpart): compute only texel rows [row_begin, row_end) of the
 * sky-view LUT, so that N ranks each produce 1/N of it and exchange the slices with one all-gather
 * (texels are independent: skyview_LUT.comp:91-128 reads only the transmittance LUT). The full
 * transmittance LUT must have been recorded on this pipeline first. */
int szg_skyview_record_skyview_lut_rows(szg_skyview_t* p, void* stream, uint32_t atmosphere_index,
                                        const szg_atmosphere_packed* d_atmospheres, uint32_t view_camera_index,
                                        const szg_camera_packed* d_cameras, uint32_t row_begin, uint32_t row_end);
/* The rows rank `rank` of `nranks` computes (an even split; the LUT height must divide), and the exchange that completes
 * the LUT on every rank: one in-place all-gather of the slices on `stream` (szg_rowtile_allgather on the pipeline's own
 * LUT memory). The sky-view LUT is the Amdahl term of the row-tiled frame (identical on every rank), hence this second,
 * optional collective of SURVEY 8e. The LUT's status word ("every texel is a finite number", see
 * szg_skyview_invalidate_luts) travels with it: each rank contributes the status of the rows it recorded with
 * szg_skyview_record_skyview_lut_rows (or "unknown" when its last slice launch was not exactly its share), the words are
 * all-gathered beside the slices (at most 64 ranks) and OR-ed, so no rank re-scans texels another rank wrote. */
struct szg_rowtile_comm;
int szg_skyview_lut_row_slice(const szg_skyview_t* p, uint32_t rank, uint32_t nranks, uint32_t* row_begin, uint32_t* row_end);
int szg_skyview_allgather_lut_rows(szg_skyview_t* p, struct szg_rowtile_comm* comm, void* stream);
int szg_skyview_record_composite(szg_skyview_t* p, void* stream, const szg_scene_texture* scene_texture,
                                 szg_rect draw_rect, const szg_rowtile* tile, const szg_gbuffer* gbuffer,
                                 const szg_shadowmaps* shadow_maps, uint32_t atmosphere_index,
                                 const szg_atmosphere_packed* d_atmospheres, uint32_t view_camera_index,
                                 const szg_camera_packed* d_cameras, uint32_t sun_light_index,
                                 const szg_directional_light_packed* d_lights);

/* ---- Aerial-perspective froxel LUT + fast composite (north_star item) ------------------------
 * NO reference counterpart: the reference marches the aerial perspective inline per geometry pixel
 * (camera.comp:273-275). This opt-in extension precomputes it on a SZG_AERIAL_W x _H x _D froxel grid over the
 * view frustum. The TEXEL VALUES are pinned by the reference's math: froxel (i, j, k) holds
 *   luminance     = computeLuminanceScatteringIntegral(position, direction(i, j), d_k)   (common.glinl:364-424)
 *   transmittance = sampleTransmittanceLUT_Segment(position, position + d_k * direction)  (common.glinl:114-136)
 * with direction(i, j) the camera.comp:324-328 view ray through the froxel centre ((i + .5) / W, (j + .5) / H) and
 * d_k = (k + .5) / D * max_distance_mm; they are parity-checked against the oracle. Only their trilinear USE by
 * szg_skyview_record_composite_fast() is approximate, so that mode is never part of the parity frame. */
#define SZG_AERIAL_W 32u
#define SZG_AERIAL_H 32u
#define SZG_AERIAL_D 32u
int szg_skyview_record_aerial_lut(szg_skyview_t* p, void* stream, uint32_t atmosphere_index,
                                  const szg_atmosphere_packed* d_atmospheres, uint32_t view_camera_index,
                                  const szg_camera_packed* d_cameras, float max_distance_mm);
/* The two volumes as images of SZG_AERIAL_W x (SZG_AERIAL_H * SZG_AERIAL_D) RGBA32F texels (slice k = rows
 * [k * H, (k + 1) * H)). */
int szg_skyview_aerial_lut(const szg_skyview_t* p, szg_image* out_luminance, szg_image* out_transmittance);
/* szg_skyview_record_composite() with the geometry pixels' inline 32-step march replaced by one trilinear fetch
 * of the aerial LUT recorded last (APPROXIMATE; sky pixels and every other term are unchanged). */
int szg_skyview_record_composite_fast(szg_skyview_t* p, void* stream, const szg_scene_texture* scene_texture,
                                      szg_rect draw_rect, const szg_rowtile* tile, const szg_gbuffer* gbuffer,
                                      const szg_shadowmaps* shadow_maps, uint32_t atmosphere_index,
                                      const szg_atmosphere_packed* d_atmospheres, uint32_t view_camera_index,
                                      const szg_camera_packed* d_cameras, uint32_t sun_light_index,
                                      const szg_directional_light_packed* d_lights);

/* ---- Multi-scattering LUT (north_star item) ---------------------------------------------------
 * NO reference counterpart (the only trace is the unused constant phaseIsotropic, common.glinl:281-282), so there
 * is no reference parity: "parity unpinned". Build-defined after Hillaire 2020 section 5.5: SZG_MULTISCATTER_DIM^2
 * texels over (u = 0.5 + 0.5 cos(sun zenith), v = altitude / atmosphere thickness); per texel 64 sphere directions
 * (8 x 8 stratified), each a 20-step march accumulating the 2nd-order in-scattered luminance L2 and the transfer
 * factor f_ms with the reference's own extinction / sun-transmittance functions (common.glinl:145-216) and the
 * ground term of camera.comp:203-227 (albedo 0.4 / pi); texel = mean(L2) / (1 - mean(f_ms)). One 64-lane wavefront
 * per texel, lane = direction, butterfly reduction across the wave. Opt-in and NOT consumed by any parity pass:
 * adding it to the scattering integral would change results. Validated against the scalar oracle
 * (same reduction tree) and by energy sanity (0 <= f_ms < 1). */
#define SZG_MULTISCATTER_DIM 32u
int szg_skyview_record_multiscatter_lut(szg_skyview_t* p, void* stream, uint32_t atmosphere_index,
                                        const szg_atmosphere_packed* d_atmospheres);
int szg_skyview_multiscatter_lut(const szg_skyview_t* p, szg_image* out);

/* Accessors to the LUT images the pipeline owns (skyview.hpp:52-97 `map`).
 * A caller that WRITES the texels of either LUT itself (tests upload a LUT; the multi-GPU path all-gathers sky-view row
 * slices into the image) must fetch the image through the accessor for each such write, or record the pass that produces
 * the slices (szg_skyview_record_skyview_lut_rows) before it: either marks the LUT as externally written, and the next
 * pass that consumes it first re-scans the texels (the kernels keep a status word next to each LUT: the transmittance
 * LUT's value range, the sky-view LUT's finiteness). */
int szg_skyview_transmittance_lut(const szg_skyview_t* p, szg_image* out);
int szg_skyview_skyview_lut(const szg_skyview_t* p, szg_image* out);
/* The same notice without fetching the image again: the caller has written (or is about to write) texels of the LUTs named
 * in `which` through a pointer it kept. The next consumer re-scans them, and with LUT reuse (below) the next record of that
 * LUT recomputes it. A written transmittance LUT also invalidates the sky-view LUT, which is a function of it. */
#define SZG_LUT_TRANSMITTANCE 1u
#define SZG_LUT_SKYVIEW 2u
int szg_skyview_invalidate_luts(szg_skyview_t* p, uint32_t which);

/* LUT reuse across frames (no reference counterpart: the reference records both LUT dispatches every frame,
 * skyview.cpp:799-893, although the transmittance LUT depends on the atmosphere block alone and the sky-view LUT on that
 * block - which carries the sun direction - and the camera position; SURVEY 8e asks for it because the results are
 * identical). Off by default. When enabled, szg_skyview_record_transmittance / _record_skyview_lut / _record_draw_commands
 * first compare, ON THE DEVICE and in stream order, the parameter blocks they are given with the ones the current texels
 * were computed from, bit for bit (the blocks are device memory: the host never reads them), and the LUT kernel behind the
 * comparison returns at once when nothing changed: a frame with an unchanged atmosphere, sun and camera position costs two
 * one-wave launches and two empty ones instead of ~1 ms. Changing any dword of the atmosphere block recomputes both LUTs,
 * moving the camera recomputes the sky-view LUT. Texels written by the caller (the accessors above,
 * szg_skyview_invalidate_luts) and row-slice launches are always followed by a recompute. */
int szg_skyview_set_lut_reuse(szg_skyview_t* p, int enable);

/* ------------------------------------------------------------------------- */
/* DeferredShadingPipeline (renderer/pipelines/deferred.hpp:23-119)            */
/* ------------------------------------------------------------------------- */
typedef struct szg_deferred szg_deferred_t;

typedef struct szg_deferred_desc
{
    uint32_t capacity_width;  /* dimensionCapacity (deferred.hpp:31); reference 4096 (renderer.hpp:96) */
    uint32_t capacity_height;
    uint32_t max_spot_lights; /* reference 16 (deferred.cpp:166) */
    uint32_t max_shadow_maps; /* reference 10 (deferred.cpp:179) */
    uint32_t shadow_map_dim;  /* reference 8192 (deferred.cpp:180); 0 = allocate none */
    uint32_t padding;
} szg_deferred_desc;

/* deferred.hpp:109-115 + shadowpass.hpp:24-28 */
typedef struct szg_deferred_configuration
{
    float depthBiasConstant;
    float depthBiasSlope;
} szg_deferred_configuration;

/* deferred.hpp:26-32 constructor. */
int szg_deferred_create(szg_deferred_t** out, const szg_deferred_desc* desc, int device);
/* deferred.hpp:49 cleanup(). NULL is allowed. */
void szg_deferred_destroy(szg_deferred_t* p);

/* deferred.hpp:34-44 recordDrawCommands (deferred.cpp:435-792): upload spot
 * lights -> (shadow maps: inputs) -> G-buffer fill -> clear colour
 * to opaque black -> lights dispatch. `h_spot_lights` is a HOST span exactly as
 * in the reference (std::span<SpotLightPacked const>); `geometry` replaces
 * std::span<MeshInstanced const>: NULL keeps the G-buffer contents the caller
 * wrote through szg_deferred_gbuffer(). */
int szg_deferred_record_draw_commands(
    szg_deferred_t* p, void* stream, szg_rect draw_rect, const szg_rowtile* tile,
    const szg_scene_texture* scene_texture, uint32_t atmospheric_directional_lights_count,
    const szg_directional_light_packed* d_directional_lights, uint32_t directional_light_count,
    const szg_spot_light_packed* h_spot_lights, uint32_t spot_light_count, uint32_t view_camera_index,
    const szg_camera_packed* d_cameras, const szg_fill_scene* geometry);

/* The two device passes of recordDrawCommands, individually. */
int szg_deferred_record_gbuffer_fill(szg_deferred_t* p, void* stream, szg_rect draw_rect, const szg_rowtile* tile,
                                     const szg_scene_texture* scene_texture, uint32_t view_camera_index,
                                     const szg_camera_packed* d_cameras, const szg_fill_scene* geometry);
int szg_deferred_record_lights(szg_deferred_t* p, void* stream, szg_rect draw_rect, const szg_rowtile* tile,
                               const szg_scene_texture* scene_texture,
                               uint32_t atmospheric_directional_lights_count,
                               const szg_directional_light_packed* d_directional_lights,
                               uint32_t directional_light_count, const szg_spot_light_packed* h_spot_lights,
                               uint32_t spot_light_count, uint32_t view_camera_index,
                               const szg_camera_packed* d_cameras);

/* Shadow-map generation: the depth-only pass of shadowpass.cpp:188-270 +
 * offscreenpass/depthpass.vert:30-38 + pipelines.cpp:640-663 (front-face culling, reverse-Z, GREATER_OR_EQUAL,
 * clear to 0 = far) for the ANALYTIC scene: per texel centre the light's ray is cast against the boxes of
 * `geometry` and the depth of the nearest BACK face (what front-face culling leaves) is stored; the ground plane,
 * single-sided and facing the lights, is culled exactly as in the raster pass. Slots run directional lights [0, n)
 * then spot lights, capped at max_shadow_maps (shadowpass.cpp:219-225). Only maps the pipeline owns
 * (shadow_map_dim > 0) are written; the depth bias of szg_deferred_configuration is stored but not applied
 * (rasteriser-specific units). Called by szg_deferred_record_draw_commands when geometry != NULL. */
int szg_deferred_record_shadow_maps(szg_deferred_t* p, void* stream, const szg_directional_light_packed* d_directional_lights,
                                    uint32_t directional_light_count, const szg_spot_light_packed* h_spot_lights,
                                    uint32_t spot_light_count, const szg_fill_scene* geometry);

/* deferred.hpp:46-47 gbuffer() / shadowMaps(). Pointers stay valid until destroy. */
const szg_gbuffer* szg_deferred_gbuffer(szg_deferred_t* p);
const szg_shadowmaps* szg_deferred_shadow_maps(szg_deferred_t* p);
/* Attach/detach a caller-owned D32F shadow map for light slot `index` (slots
 * run directional lights first, then spot lights: lights.comp:138-161).
 * map == NULL or map->data == NULL detaches (shadow factor 1). */
int szg_deferred_set_shadow_map(szg_deferred_t* p, uint32_t index, const szg_image* map);

/* deferred.hpp:114-115 */
int szg_deferred_get_configuration(const szg_deferred_t* p, szg_deferred_configuration* out);
int szg_deferred_set_configuration(szg_deferred_t* p, const szg_deferred_configuration* cfg);

/* ------------------------------------------------------------------------- */
/* OETF: the pass right after the path                      */
/* ------------------------------------------------------------------------- */
/* editor/editorconfig.hpp:5-10 GammaTransferFunction */
#define SZG_OETF_PURE_GAMMA 0u /* shaders/transfer/oetf_pure_gamma.comp:9  pow(x, 1/2.2) */
#define SZG_OETF_SRGB 1u       /* shaders/transfer/oetf_srgb.comp:9-19 */

/* In-place linear -> display encoding of the top-left width x height region of `image`
 * (editor/editor.cpp:303-340 dispatches it over the swapchain extent on the UI output texture, an RGBA16
 * UNORM image: editor/uilayer.cpp:285-291; the shader declares `rgba16f`, the resource is UNORM16 — loads and
 * stores follow the resource). Alpha is passed through. 16 B/px of HBM traffic: HBM-bound. */
int szg_record_oetf(void* stream, const szg_image* image, uint32_t width, uint32_t height, uint32_t transfer_function);

/* ------------------------------------------------------------------------- */
/* Multi-GPU composition (no reference counterpart; BASELINE north_star)       */
/* ------------------------------------------------------------------------- */
/* After a gather, rank 0 holds `nranks` packed row-tiles back to back in
 * `gathered` (rank r at r * tile_pitch_bytes). Scatter their rows into the
 * full-frame image `dst` following szg_rowtile's cyclic map. One HBM-bound
 * copy kernel. */
int szg_compose_rowtiles(void* stream, const void* gathered, size_t tile_stride_bytes, uint32_t nranks,
                         uint32_t block_rows, const szg_image* dst, uint32_t width, uint32_t height);

/* Number of local rows rank `rank` holds for a frame of `height` rows. */
uint32_t szg_rowtile_local_rows(uint32_t height, uint32_t block_rows, uint32_t rank, uint32_t nranks);

/* ------------------------------------------------------------------------- */
/* Multi-GPU collectives (no reference counterpart; BASELINE north_star:       */
/* "a single RCCL gather over xGMI for the composed image")                    */
/* ------------------------------------------------------------------------- */
/* One process per GPU. A communicator binds this process (rank r of n) to the others through RCCL, which is loaded at
 * run time (librccl.so.1; SZG_RCCL_LIBRARY overrides the name): the library itself does not link against it. Rank 0
 * obtains SZG_ROWTILE_COMM_ID_BYTES opaque bytes from szg_rowtile_comm_unique_id() and hands them to every rank by any
 * means (the caller's launcher: a file, a socket, MPI, torch.distributed); every rank then calls
 * szg_rowtile_comm_create(rank, nranks, id, device) - a collective call, like ncclCommInitRank. */
typedef struct szg_rowtile_comm szg_rowtile_comm_t;
#define SZG_ROWTILE_COMM_ID_BYTES 256
int szg_rowtile_comm_unique_id(void* out_id);
int szg_rowtile_comm_create(szg_rowtile_comm_t** out, int rank, int nranks, const void* unique_id, int device);
/* The same with an explicit deadline for the rendezvous (milliseconds; 0 = wait for ever). szg_rowtile_comm_create uses
 * SZG_COMM_TIMEOUT_S seconds (default 300). When the other ranks do not join in time the call returns SZG_ERR_TIMEOUT:
 * RCCL's blocking initialisation cannot be cancelled, so the process must then EXIT with a non-zero status (which is what
 * lets a launcher tear the other ranks down) - do not retry in the same process, and never re-exec a process that has
 * touched the GPU. */
int szg_rowtile_comm_create_deadline(szg_rowtile_comm_t** out, int rank, int nranks, const void* unique_id, int device, int timeout_ms);
/* Which RCCL the collectives are bound to: "<path of the shared object> (RCCL version code N; how it was found; gather: ...)",
 * or the reason why none could be bound. Binding order: SZG_RCCL_LIBRARY, then an RCCL already mapped into the process
 * (inside PyTorch: the one torch.distributed uses - never a second copy), then librccl.so.1 by name. SZG_LOG=1 prints it. */
const char* szg_rowtile_comm_backend(void);
void szg_rowtile_comm_destroy(szg_rowtile_comm_t* comm);
int szg_rowtile_comm_rank(const szg_rowtile_comm_t* comm);
int szg_rowtile_comm_size(const szg_rowtile_comm_t* comm); /* the rank count RCCL reports for the communicator */
/* THE collective of the frame: every rank's packed row tile (`tile_bytes` bytes at `tile`, the same size on every rank:
 * szg_rowtile stride rows x pitch) lands at gathered + rank * tile_bytes on `root`; `gathered` is ignored elsewhere.
 * N - 1 point-to-point streams into the root (ncclGather, or grouped ncclSend / ncclRecv), enqueued on `stream`; returns
 * immediately. Follow it with szg_compose_rowtiles() on the root. */
int szg_rowtile_gather(szg_rowtile_comm_t* comm, void* stream, const void* tile, size_t tile_bytes, void* gathered, int root);
/* In-place all-gather: rank r has filled bytes [r * bytes_per_rank, (r + 1) * bytes_per_rank) of `buffer` (the same
 * layout on every rank); afterwards every rank holds all of it. Uses a communicator of its own, so it may be in flight
 * (on another stream) while a gather is. */
int szg_rowtile_allgather(szg_rowtile_comm_t* comm, void* stream, void* buffer, size_t bytes_per_rank);

/* ------------------------------------------------------------------------- */
/* Host input prep (CPU only; renderer/scene.cpp, renderer/lights.cpp,         */
/* geometry/geometryhelpers.cpp). See szg/host.h.                              */
/* ------------------------------------------------------------------------- */

#ifdef __cplusplus
} /* extern "C" */

static_assert(sizeof(szg_camera_packed) == 416, "gputypes.hpp:36");
static_assert(sizeof(szg_atmosphere_packed) == 128, "gputypes.hpp:72");
static_assert(sizeof(szg_directional_light_packed) == 176, "gputypes.hpp:90");
static_assert(sizeof(szg_spot_light_packed) == 192, "gputypes.hpp:115");
static_assert(sizeof(szg_pc_transmittance) == 16, "skyview.hpp:64-72");
static_assert(sizeof(szg_pc_skyview) == 32, "skyview.hpp:89-96");
static_assert(sizeof(szg_pc_composite) == 64, "skyview.hpp:125-144");
static_assert(sizeof(szg_pc_lights) == 64, "deferred.hpp:82-98");
static_assert(offsetof(szg_pc_composite, drawExtent) == 24 && offsetof(szg_pc_composite, gbufferExtent) == 40 &&
                  offsetof(szg_pc_composite, directionalLights) == 48 && offsetof(szg_pc_composite, sunLightIndex) == 56,
              "camera.comp:50-67 std430 offsets");
static_assert(offsetof(szg_pc_lights, directionalLightsBuffer) == 16 && offsetof(szg_pc_lights, directionalLightCount) == 32 &&
                  offsetof(szg_pc_lights, gbufferOffset) == 48 && offsetof(szg_pc_lights, gbufferExtent) == 56,
              "lights.comp:41-57 std430 offsets");
#endif

#endif /* SZG_ABI_H */
