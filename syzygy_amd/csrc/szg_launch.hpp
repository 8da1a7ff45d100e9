// szg_launch.hpp — internal launch interface between the C-ABI layer (szg_api.cpp)
// and the kernel translation units. Not part of the public boundary.
#pragma once

#include <hip/hip_runtime_api.h>

#include "szg/abi.h"
#include "szg/raster.h"

namespace szg
{
// Per-light record produced by the prep kernel, consumed by the lights kernel
// through scalar (wave-uniform) loads. 36 dwords.
struct LightRec
{
    float shadowRows[16];   // TO_TEX_COORD_MAT * projection * view (shadowmap.glinl:19), ROW-major: rows[r*4+c]
    float dir[3];           // normalize(-forward) (lights.comp:67/78)
    float falloffFactor;
    float colorStrength[3]; // color.rgb * strength (lights.comp:68/88)
    float falloffDistance;
    float position[3];
    unsigned isSpot;
    const float* map; // D32F shadow map or nullptr
    unsigned mapWidth, mapHeight;
    unsigned mapPitchFloats;
    unsigned leanOK; // bit 0: falloffDistance / falloffFactor / colour*strength of moderate magnitude (lean exact ops allowed)
    // (bit 0 also says: the light's own factors are finite and non-zero where they divide, so a culled pixel's term is an exact 0)
    // bit 1 (implies bit 0), "tight": |position| <= 2^12, |shadow rows| <= 2^14 and falloffBound in [2^-20, 2^20]. Together with
    // pixel positions <= 2^12 this bounds every operand of the lean ops from ABOVE by construction (|clip| < 2^28, d^2 < 2^28,
    // falloff < 2^48), so that k_lights' optimistic pass only has to track their LOWER bounds (kernels_deferred.hip).
    float falloffBound;       // falloffFactor / falloffDistance^2 when leanOK: falloffBound * d^2 ~ the light's falloff at distance d
    float rcpFalloffDistance; // rcpN(falloffDistance) when leanOK: the shared reciprocal of dist / falloffDistance (once per light, not per wave)
};
static_assert(sizeof(LightRec) == 144, "LightRec layout");

// Device-visible shadow-map slot table entry (uploaded by the host).
struct ShadowSlot
{
    const float* map;
    unsigned width, height, pitchFloats, pad;
};
static_assert(sizeof(ShadowSlot) == 24, "ShadowSlot layout");

struct AerialLut
{
    const float* luminance; // nullptr = exact composite
    unsigned W, H, D;
    float maxDistance;
};

struct TileArgs
{
    unsigned block_rows, rank, nranks, local_rows;
};

// Transmittance LUT block: W*H RGBA32F texels followed by ONE status dword (16 bytes reserved). The dword is 0 when
// every texel's rgb lies in [2^-50, 2] — the operand domain in which the march may use the lean exact division for
// transmittance ratios (szg_device.hpp TLut::moderate). launch_transmittance maintains it; launch_lut_range recomputes
// it for texels written by the caller. Every kernel that takes `tlut` reads the dword behind the texels.
inline size_t tlut_block_bytes(unsigned W, unsigned H) { return (size_t)W * H * 16u + 16u; }
hipError_t launch_lut_range(hipStream_t s, float* lut, unsigned W, unsigned H);
// `d_dirty` (may be nullptr): LUT reuse flag written by launch_lut_key; the kernel returns at once when it reads 0.
hipError_t launch_transmittance(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, float* lut, unsigned W,
                                unsigned H, const unsigned* d_dirty = nullptr);
// LUT reuse across frames ("cache the LUTs across frames when atmosphere / sun / camera altitude are unchanged
// (allowed because results are identical)"; the reference recomputes both every frame, skyview.cpp:799-893). The parameter
// blocks live in device memory, so the comparison runs on the device, in stream order, with no host round trip: k_lut_key
// (one wave) compares the atmosphere block (and, for the sky-view LUT, the camera position and the generation of the
// transmittance LUT) bit for bit with what the texels were computed from, records the verdict in state[69 + which] and, when
// dirty, stores the new key and clears the LUT's status dword. The LUT kernel launched behind it exits at once on a clean
// verdict. LUT_KEY_DWORDS dwords of state, zeroed at creation; `force` = the host knows the texels are stale (first use,
// texels handed out for writing, a row-slice launch, an explicit invalidate).
constexpr unsigned LUT_KEY_DWORDS = 72u;
constexpr unsigned SLUT_STATUS_RANKS = 64u; // most ranks szg_skyview_allgather_lut_rows exchanges slice status words of
hipError_t launch_lut_key(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, const szg_camera_packed* d_cam,
                          unsigned camIndex, unsigned* d_state, unsigned which, bool force, float* lutBlock, unsigned W, unsigned H);
// Sky-view LUT block: W*H RGBA32F texels followed by ONE status dword (16 bytes reserved): 0 when every texel's rgb is a
// finite number of moderate size (|x| <= 2^100). The composite consults it before it leaves a sample of this LUT unevaluated
// (the reflection term of a non-metal pixel is an exact zero only if the sample is finite). launch_skyview over all rows
// maintains it; after a partial launch (row slices of the multi-GPU path, completed by an all-gather) or a write by the
// caller, launch_slut_check recomputes it from the texels.
inline size_t slut_block_bytes(unsigned W, unsigned H) { return (size_t)W * H * 16u + 16u; }
hipError_t launch_slut_check(hipStream_t s, float* lut, unsigned W, unsigned H);
// The N-rank form of the same: the status of this rank's slice goes into d_all[rank] (1 when `known` is false), the caller
// all-gathers d_all (4 bytes per rank), and the LUT's status dword becomes the OR of the nranks words.
hipError_t launch_slut_status_stage(hipStream_t s, unsigned* d_all, unsigned rank, const float* lut, unsigned W, unsigned H, bool known);
hipError_t launch_slut_status_reduce(hipStream_t s, const unsigned* d_all, unsigned nranks, float* lut, unsigned W, unsigned H);
hipError_t launch_skyview(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, const szg_camera_packed* d_cam,
                          unsigned camIndex, const float* tlut, unsigned tW, unsigned tH, float* lut, unsigned W, unsigned H,
                          unsigned rowBegin, unsigned rowEnd, const unsigned* d_dirty, const void* d_prep);
// Per-frame constants (szg_device.hpp FramePrep): one-lane kernel in front of the sky-view LUT kernel and the composite, which
// read `d_prep` (frame_prep_bytes() bytes of device memory) instead of deriving the same ~300 instructions' worth in every wave.
size_t frame_prep_bytes();
// `d_sun` (may be null): the directional light whose shadow map the composite samples; its TO_TEX * projection * view goes into
// the block (FramePrep::sunShadow).
hipError_t launch_frame_prep(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, unsigned tW, unsigned tH, void* d_prep,
                             const szg_directional_light_packed* d_sun = nullptr);
hipError_t launch_light_prep(hipStream_t s, const szg_directional_light_packed* d_dir, unsigned dirCount, unsigned dirSkip,
                             const szg_spot_light_packed* d_spot, unsigned spotCount, const ShadowSlot* d_slots,
                             unsigned slotCount, LightRec* d_out);
hipError_t launch_lights(hipStream_t s, const szg_scene_texture& scene, unsigned drawW, unsigned drawH, TileArgs tile,
                         const szg_gbuffer& g, const szg_camera_packed* d_cam, unsigned camIndex, const LightRec* d_lights,
                         unsigned lightCount);
hipError_t launch_gbuffer_fill(hipStream_t s, const szg_scene_texture& scene, unsigned drawW, unsigned drawH, TileArgs tile,
                               const szg_gbuffer& g, const szg_camera_packed* d_cam, unsigned camIndex, float ground_y,
                               float ground_half_extent, float checker_cell, float ground_roughness,
                               const szg_fill_box* d_boxes, unsigned boxCount);
hipError_t launch_composite(hipStream_t s, const szg_scene_texture& scene, unsigned drawW, unsigned drawH, TileArgs tile,
                            const szg_gbuffer& g, ShadowSlot sunSlot, const szg_atmosphere_packed* d_atm, unsigned atmIndex, const szg_camera_packed* d_cam,
                            unsigned camIndex, const szg_directional_light_packed* d_dir, unsigned sunIndex, const float* tlut,
                            unsigned tW, unsigned tH, const float* slut, unsigned sW, unsigned sH, AerialLut aerial, const void* d_prep);
hipError_t launch_aerial_lut(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, const szg_camera_packed* d_cam,
                             unsigned camIndex, const float* tlut, unsigned tW, unsigned tH, float* luminance, float* transmittance,
                             unsigned W, unsigned H, unsigned D, float maxDistance);
hipError_t launch_multiscatter(hipStream_t s, const szg_atmosphere_packed* d_atm, unsigned atmIndex, const float* tlut, unsigned tW,
                               unsigned tH, float* out, unsigned dim);
// Per-slot state for shadow-map generation, built on the device by k_shadow_prep.
struct ShadowGen
{
    float projView[16];
    float invProjView[16];
    float* map; // nullptr = skip
    unsigned dim, pitchFloats, pad;
};
hipError_t launch_shadow_maps(hipStream_t s, const szg_directional_light_packed* d_dir, unsigned dirCount,
                              const szg_spot_light_packed* d_spot, unsigned spotCount, const ShadowSlot* d_ownedSlots,
                              unsigned slotCount, ShadowGen* d_gen, const szg_fill_box* d_boxes, unsigned boxCount, unsigned maxDim);
hipError_t launch_shadow_prep(hipStream_t s, const szg_directional_light_packed* d_dir, unsigned dirCount,
                              const szg_spot_light_packed* d_spot, unsigned spotCount, const ShadowSlot* d_ownedSlots,
                              unsigned slotCount, ShadowGen* d_gen);

// ---- compute rasteriser (kernels_raster.hip, include/szg/raster.h) ----
// One vkCmdDrawIndexed of the reference (one surface of one mesh, all its instances), uploaded by the host.
struct RasterDraw
{
    const szg_vertex_packed* vertices;
    const uint32_t* indices;
    const szg_mat4* models;
    const szg_mat4* mits;
    uint32_t vertexCount, firstIndex, triCount, instanceCount;
    uint32_t firstPrim; // submission-order number of this draw's first primitive (instance-major, then triangle)
    uint32_t pad;
    szg_texture tex[3]; // color, normal, ORM (offscreen.frag:19-21)
};
// One assembled primitive: signed homogeneous edge functions + clip z, w per vertex (raster.h "coverage", "depth").
struct PrimRec
{
    float a[3], b[3], c[3];
    float z[3], w[3];
    uint32_t draw, instance, tri;
    uint32_t pad[2];
};
static_assert(sizeof(PrimRec) == 80, "PrimRec layout");
// Device buffers of one raster pass, owned by the pipeline and grown on demand (szg_api.cpp).
struct RasterBuffers
{
    PrimRec* prims = nullptr;       // [capacity] submission order
    uint2* boxes = nullptr;         // [capacity] pixel boxes, submission order (x: min | max << 16, y likewise)
    unsigned* keysA = nullptr;      // [capacity] sort keys / scratch
    unsigned* keysB = nullptr;
    unsigned* valsA = nullptr;      // [capacity] identity order
    unsigned* valsB = nullptr;      // [capacity] sorted order
    uint2* orderedBoxes = nullptr;  // [capacity] boxes in walk order
    uint2* chunkBoxes = nullptr;    // [capacity / 64]
    uint2* superBoxes = nullptr;    // [capacity / 4096 + 1]
    void* sortTemp = nullptr;
    size_t sortTempBytes = 0;
    const unsigned* order = nullptr; // valsA or valsB: set by launch_raster_setup
    size_t capacity = 0;
};
// above this many primitives the boxes are radix-sorted (size class, Morton code) before the hierarchy is built
constexpr unsigned RASTER_SORT_THRESHOLD = 4096u;
// kernels_raster_sort.hip (rocPRIM): temp-storage size for `n` pairs, and the sort itself
hipError_t raster_sort_temp_bytes(unsigned n, size_t& bytes);
hipError_t raster_sort_pairs(hipStream_t s, void* temp, size_t tempBytes, const unsigned* keysIn, unsigned* keysOut,
                             const unsigned* valsIn, unsigned* valsOut, unsigned n);
hipError_t launch_raster_setup(hipStream_t s, bool shadow, const RasterDraw* d_draws, unsigned drawCount, unsigned primCount,
                               const szg_camera_packed* d_cam, unsigned camIndex, const ShadowGen* d_gen, unsigned W, unsigned H,
                               RasterBuffers& b);
hipError_t launch_raster_tile(hipStream_t s, const szg_scene_texture& scene, unsigned drawW, unsigned drawH, TileArgs tile,
                              const szg_gbuffer& g, const RasterDraw* d_draws, const RasterBuffers& b, unsigned primCount,
                              const szg_camera_packed* d_cam, unsigned camIndex);
hipError_t launch_shadow_tile(hipStream_t s, const ShadowGen* d_gen, unsigned dim, const RasterBuffers& b, unsigned primCount,
                              float biasConstant, float biasSlope);
// `table`: 65536 x u16, the OETF of every UNORM16 channel value (launch_oetf_table fills it)
hipError_t launch_oetf_table(hipStream_t s, unsigned short* table, unsigned function);
hipError_t launch_oetf(hipStream_t s, const szg_image& image, unsigned width, unsigned height, const unsigned short* table);
hipError_t launch_compose_rowtiles(hipStream_t s, const void* gathered, size_t tileStrideBytes, unsigned nranks,
                                   unsigned blockRows, const szg_image& dst, unsigned width, unsigned height);
} // namespace szg
