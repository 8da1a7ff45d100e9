// szg_api.cpp — the extern "C" boundary declared in include/szg/abi.h.
//
// Host-side mirror of renderer/pipelines/skyview.cpp:713-965 and
// renderer/pipelines/deferred.cpp:145-337, :435-792 with Vulkan, VMA and
// descriptor plumbing replaced by HIP device pointers and one stream. Every
// record_* call validates shapes on the host, enqueues, and returns; in-stream
// order replaces the reference's full barriers (imageoperations.cpp:18-33).

#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "szg/abi.h"
#ifdef SZG_LITERAL
#define SZG_CONTRACT SZG_CONTRACT_NONE
#endif
#include "szg/contraction.h"
#include "szg_internal.hpp"
#include "szg_launch.hpp"

namespace
{
thread_local char g_error[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    fprintf(stderr, "[szg] error: %s\n", g_error);
    return code;
}
int fail_hip(hipError_t e, const char* what)
{
    return fail(e == hipErrorOutOfMemory ? SZG_ERR_OUT_OF_MEMORY : SZG_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}
#define SZG_HIP(expr)                                                                                                          \
    do                                                                                                                         \
    {                                                                                                                          \
        hipError_t const _e = (expr);                                                                                          \
        if (_e != hipSuccess)                                                                                                  \
        {                                                                                                                      \
            return fail_hip(_e, #expr);                                                                                        \
        }                                                                                                                      \
    } while (0)

unsigned texel_bytes(unsigned fmt)
{
    switch (fmt)
    {
    case SZG_FORMAT_RGBA16_SFLOAT:
    case SZG_FORMAT_RGBA16_UNORM:
        return 8;
    case SZG_FORMAT_RGBA32_SFLOAT:
        return 16;
    case SZG_FORMAT_D32_SFLOAT:
        return 4;
    default:
        return 0;
    }
}

// An image must be present, of the expected format, at least w x h, with a
// pitch that covers its rows and keeps texels naturally aligned.
bool check_image(const szg_image& im, unsigned fmt, unsigned w, unsigned h, const char* name)
{
    unsigned const tb = texel_bytes(fmt);
    if (im.data == nullptr)
    {
        fail(SZG_ERR_INVALID_ARGUMENT, "%s: data is NULL", name);
        return false;
    }
    if (im.format != fmt)
    {
        fail(SZG_ERR_INVALID_ARGUMENT, "%s: format %u, expected %u", name, im.format, fmt);
        return false;
    }
    if (im.width < w || im.height < h)
    {
        fail(SZG_ERR_INVALID_ARGUMENT, "%s: %ux%u smaller than the required %ux%u", name, im.width, im.height, w, h);
        return false;
    }
    if ((size_t)im.pitch_bytes < (size_t)im.width * tb || im.pitch_bytes % tb != 0 ||
        (reinterpret_cast<uintptr_t>(im.data) % tb) != 0)
    {
        fail(SZG_ERR_INVALID_ARGUMENT, "%s: pitch %u / alignment invalid for %u-byte texels", name, im.pitch_bytes, tb);
        return false;
    }
    return true;
}

bool check_gbuffer(const szg_gbuffer* g, unsigned w, unsigned h)
{
    if (g == nullptr)
    {
        fail(SZG_ERR_INVALID_ARGUMENT, "gbuffer is NULL");
        return false;
    }
    return check_image(g->diffuse, SZG_FORMAT_RGBA16_SFLOAT, w, h, "gbuffer.diffuse") &&
           check_image(g->specular, SZG_FORMAT_RGBA16_SFLOAT, w, h, "gbuffer.specular") &&
           check_image(g->normal, SZG_FORMAT_RGBA16_SFLOAT, w, h, "gbuffer.normal") &&
           check_image(g->worldPosition, SZG_FORMAT_RGBA32_SFLOAT, w, h, "gbuffer.worldPosition") &&
           check_image(g->occlusionRoughnessMetallic, SZG_FORMAT_RGBA16_SFLOAT, w, h, "gbuffer.occlusionRoughnessMetallic");
}

bool check_scene(const szg_scene_texture* s, unsigned w, unsigned h, bool needDepth)
{
    if (s == nullptr)
    {
        fail(SZG_ERR_INVALID_ARGUMENT, "scene_texture is NULL");
        return false;
    }
    if (!check_image(s->color, SZG_FORMAT_RGBA16_UNORM, w, h, "scene_texture.color"))
    {
        return false;
    }
    if (needDepth && !check_image(s->depth, SZG_FORMAT_D32_SFLOAT, w, h, "scene_texture.depth"))
    {
        return false;
    }
    if (s->debug_color.data != nullptr && !check_image(s->debug_color, SZG_FORMAT_RGBA32_SFLOAT, w, h, "scene_texture.debug_color"))
    {
        return false;
    }
    return true;
}

// Resolve the tile: returns false (with error) if inconsistent.
bool resolve_tile(const szg_rowtile* tile, unsigned drawH, szg::TileArgs& out)
{
    if (tile == nullptr || tile->nranks <= 1u)
    {
        out = szg::TileArgs{1u, 0u, 1u, drawH};
        return true;
    }
    if (tile->block_rows == 0u || tile->rank >= tile->nranks)
    {
        fail(SZG_ERR_INVALID_ARGUMENT, "rowtile: block_rows %u rank %u nranks %u", tile->block_rows, tile->rank, tile->nranks);
        return false;
    }
    unsigned const expect = szg_rowtile_local_rows(drawH, tile->block_rows, tile->rank, tile->nranks);
    if (tile->local_rows != expect)
    {
        fail(SZG_ERR_INVALID_ARGUMENT, "rowtile: local_rows %u, expected %u for a %u-row frame", tile->local_rows, expect, drawH);
        return false;
    }
    out = szg::TileArgs{tile->block_rows, tile->rank, tile->nranks, tile->local_rows};
    return true;
}

// Small ring of pinned staging buffers for host -> device parameter uploads
// (the reference's TStagedBuffer staging half, buffers.hpp:209-299). A slot is
// reused only after the copy that read it has completed.
struct StagingRing
{
    static constexpr int SLOTS = 8;
    void* host[SLOTS] = {};
    hipEvent_t done[SLOTS] = {};
    bool used[SLOTS] = {};
    size_t bytes = 0;
    int next = 0;

    int init(size_t n)
    {
        bytes = n;
        for (int i = 0; i < SLOTS; i++)
        {
            SZG_HIP(hipHostMalloc(&host[i], n, hipHostMallocDefault));
            SZG_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        }
        return SZG_OK;
    }
    void destroy()
    {
        for (int i = 0; i < SLOTS; i++)
        {
            if (done[i] != nullptr)
            {
                (void)hipEventDestroy(done[i]);
            }
            if (host[i] != nullptr)
            {
                (void)hipHostFree(host[i]);
            }
            host[i] = nullptr;
            done[i] = nullptr;
        }
    }
    // Copy `n` bytes from `src` (host) to `dst` (device) on `stream`.
    int upload(hipStream_t stream, void* dst, const void* src, size_t n)
    {
        if (n == 0)
        {
            return SZG_OK;
        }
        if (n > bytes)
        {
            return fail(SZG_ERR_CAPACITY, "staging upload of %zu bytes exceeds %zu", n, bytes);
        }
        int const s = next;
        next = (next + 1) % SLOTS;
        if (used[s])
        {
            SZG_HIP(hipEventSynchronize(done[s]));
        }
        std::memcpy(host[s], src, n);
        SZG_HIP(hipMemcpyAsync(dst, host[s], n, hipMemcpyHostToDevice, stream));
        SZG_HIP(hipEventRecord(done[s], stream));
        used[s] = true;
        return SZG_OK;
    }
};

int select_device(int device)
{
    int count = 0;
    hipError_t const e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
    {
        return fail(SZG_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= count)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "device %d out of range [0, %d)", device, count);
    }
    SZG_HIP(hipSetDevice(device));
    return SZG_OK;
}

// VkRect2D offset: the reference dispatches every pass over the extent only and pushes gbufferOffset = 0
// (deferred.cpp:764, :778-787; skyview.cpp:658-665), i.e. it renders into the top-left sub-rectangle whatever the offset
// says. A caller passing a non-zero offset expects something this path (like the reference's) does not do: refuse it.
bool check_rect(const szg_rect& r, const char* what)
{
    if (r.x != 0 || r.y != 0)
    {
        fail(SZG_ERR_INVALID_ARGUMENT, "%s: draw_rect offset (%d, %d) must be (0, 0): passes cover the top-left extent only", what, r.x, r.y);
        return false;
    }
    return true;
}

// A pipeline lives on the device it was created on; record_* may be called while another device is current
// (one process driving several GPUs): switch for the duration of the call and switch back.
struct DeviceGuard
{
    int previous = -1;
    bool switched = false;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&previous) == hipSuccess && previous != device)
        {
            switched = hipSetDevice(device) == hipSuccess;
        }
    }
    ~DeviceGuard()
    {
        if (switched)
        {
            (void)hipSetDevice(previous);
        }
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

szg_image make_image(void* data, unsigned w, unsigned h, unsigned fmt)
{
    szg_image im;
    im.data = data;
    im.width = w;
    im.height = h;
    im.pitch_bytes = w * texel_bytes(fmt);
    im.format = fmt;
    return im;
}
} // namespace

// ---------------------------------------------------------------------------
struct szg_skyview
{
    int device = 0;
    szg_skyview_desc desc{};
    float* d_transmittance = nullptr;
    float* d_skyview = nullptr;
    float* d_multiscatter = nullptr;
    float* d_aerialLuminance = nullptr;
    float* d_aerialTransmittance = nullptr;
    float aerialMaxDistance = 0.0f; // 0 = never recorded
    bool haveTransmittance = false, haveSkyview = false;
    // false while the caller may have written the transmittance texels through szg_skyview_transmittance_lut():
    // the next consumer recomputes the block's status dword first (szg_launch.hpp "transmittance LUT block")
    mutable bool tlutStatusValid = false;
    // the same for the sky-view LUT's status dword (szg_launch.hpp "sky-view LUT block"): false after a partial (row-slice)
    // launch or after szg_skyview_skyview_lut() handed the texels out
    mutable bool slutStatusValid = false;
    // the status dword behind the sky-view LUT currently describes exactly the rows of this pipeline's last slice launch
    mutable bool sliceStatusKnown = false;
    uint32_t sliceRowBegin = 0, sliceRowEnd = 0; // ... which were these rows
    unsigned* d_slutStatusAll = nullptr; // szg::SLUT_STATUS_RANKS dwords: the ranks' slice status words (szg_skyview_allgather_lut_rows)
    // LUT reuse across frames (szg_launch.hpp "LUT reuse"; off by default = the reference's recompute-every-frame)
    bool lutReuse = false;
    unsigned* d_lutKey = nullptr;                      // LUT_KEY_DWORDS dwords of device state
    // szg::frame_prep_bytes() each: per-frame constants (k_frame_prep), one block per pass, because a caller may record the
    // LUT pass of frame k+1 on another stream than the composite of frame k (rowtile.py does)
    void* d_framePrep = nullptr;     // sky-view LUT pass
    void* d_framePrepDraw = nullptr; // composite
    mutable bool forceTransmittance = true, forceSkyview = true; // the host knows the texels are stale
};

static hipError_t ensure_tlut_status(szg_skyview* p, hipStream_t s)
{
    if (p->tlutStatusValid)
    {
        return hipSuccess;
    }
    hipError_t const e = szg::launch_lut_range(s, p->d_transmittance, p->desc.transmittance_width, p->desc.transmittance_height);
    p->tlutStatusValid = (e == hipSuccess);
    return e;
}

static hipError_t ensure_slut_status(szg_skyview* p, hipStream_t s)
{
    if (p->slutStatusValid)
    {
        return hipSuccess;
    }
    hipError_t const e = szg::launch_slut_check(s, p->d_skyview, p->desc.skyview_width, p->desc.skyview_height);
    p->slutStatusValid = (e == hipSuccess);
    return e;
}

struct szg_deferred
{
    int device = 0;
    szg_deferred_desc desc{};
    szg_deferred_configuration config{};
    szg_gbuffer gbuffer{};
    void* d_gbufferPlanes[5] = {};
    std::vector<szg_image> shadowImages; // host table returned by szg_deferred_shadow_maps
    szg_shadowmaps shadowMaps{};
    void* d_ownedShadowMaps = nullptr;
    szg_spot_light_packed* d_spots = nullptr;
    szg::ShadowSlot* d_slots = nullptr;
    szg::LightRec* d_lightRecs = nullptr;
    szg::ShadowSlot* d_ownedSlots = nullptr; // the maps the pipeline allocated itself (never changes after create)
    szg::ShadowGen* d_shadowGen = nullptr;
    szg_fill_box* d_boxes = nullptr;
    unsigned maxBoxes = 1024;
    unsigned maxDirectional = 16;
    StagingRing staging;
    // compute rasteriser (szg/raster.h): buffers grow on demand and are kept
    szg::RasterDraw* d_rasterDraws = nullptr;
    size_t rasterDrawCapacity = 0;
    szg::RasterBuffers raster;
};

void szg::set_last_error(const char* message)
{
    snprintf(g_error, sizeof g_error, "%s", message != nullptr ? message : "");
    fprintf(stderr, "[szg] error: %s\n", g_error);
}

extern "C" {

int szg_abi_version(void) { return SZG_ABI_VERSION; }

#ifndef SZG_SOURCE_HASH
#define SZG_SOURCE_HASH "unknown"
#endif
const char* szg_build_id(void)
{
    static char text[64];
    static std::once_flag once;
    std::call_once(once, [] { snprintf(text, sizeof text, "%s contract=0x%04x", SZG_SOURCE_HASH, (unsigned)(SZG_CONTRACT)); });
    return text;
}
const char* szg_last_error(void) { return g_error; }

int szg_device_count(void)
{
    int count = 0;
    hipError_t const e = hipGetDeviceCount(&count);
    if (e != hipSuccess)
    {
        return fail(SZG_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return count;
}

uint32_t szg_rowtile_local_rows(uint32_t height, uint32_t block_rows, uint32_t rank, uint32_t nranks)
{
    if (nranks <= 1u)
    {
        return height;
    }
    if (block_rows == 0u || rank >= nranks)
    {
        return 0u;
    }
    uint32_t const nblocks = (height + block_rows - 1u) / block_rows;
    uint32_t rows = 0;
    for (uint32_t b = rank; b < nblocks; b += nranks)
    {
        uint32_t const begin = b * block_rows;
        uint32_t const end = begin + block_rows < height ? begin + block_rows : height;
        rows += end - begin;
    }
    return rows;
}

// ---------------------------------------------------------------------------
// SkyViewComputePipeline
// ---------------------------------------------------------------------------
int szg_skyview_create(szg_skyview_t** out, const szg_skyview_desc* desc, int device)
{
    if (out == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_create: out is NULL");
    }
    *out = nullptr;
    szg_skyview_desc d{512u, 128u, 2048u, 1024u, 0u, 0u};
    if (desc != nullptr)
    {
        d = *desc;
    }
    if (d.transmittance_width < 2u || d.transmittance_height < 2u || d.skyview_width < 2u || d.skyview_height < 2u ||
        d.transmittance_width > 16384u || d.transmittance_height > 16384u || d.skyview_width > 16384u ||
        d.skyview_height > 16384u)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_create: LUT extents out of range");
    }
    if (d.flags != 0u)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_create: unknown flags 0x%x", d.flags);
    }
    int const rc = select_device(device);
    if (rc != SZG_OK)
    {
        return rc;
    }
    szg_skyview* p = new (std::nothrow) szg_skyview();
    if (p == nullptr)
    {
        return fail(SZG_ERR_OUT_OF_MEMORY, "szg_skyview_create: host allocation failed");
    }
    p->device = device;
    p->desc = d;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->d_transmittance),
                             szg::tlut_block_bytes(d.transmittance_width, d.transmittance_height));
    if (e == hipSuccess)
    {
        e = hipMalloc(reinterpret_cast<void**>(&p->d_skyview), szg::slut_block_bytes(d.skyview_width, d.skyview_height));
    }
    size_t const aerialBytes = (size_t)SZG_AERIAL_W * SZG_AERIAL_H * SZG_AERIAL_D * 16u;
    if (e == hipSuccess)
    {
        e = hipMalloc(reinterpret_cast<void**>(&p->d_multiscatter), (size_t)SZG_MULTISCATTER_DIM * SZG_MULTISCATTER_DIM * 16u);
    }
    if (e == hipSuccess)
    {
        e = hipMalloc(reinterpret_cast<void**>(&p->d_aerialLuminance), aerialBytes);
    }
    if (e == hipSuccess)
    {
        e = hipMalloc(reinterpret_cast<void**>(&p->d_aerialTransmittance), aerialBytes);
    }
    if (e == hipSuccess)
    {
        e = hipMalloc(reinterpret_cast<void**>(&p->d_lutKey), szg::LUT_KEY_DWORDS * sizeof(unsigned));
    }
    if (e == hipSuccess)
    {
        e = hipMemset(p->d_lutKey, 0, szg::LUT_KEY_DWORDS * sizeof(unsigned));
    }
    if (e == hipSuccess)
    {
        e = hipMalloc(&p->d_framePrep, szg::frame_prep_bytes());
    }
    if (e == hipSuccess)
    {
        e = hipMalloc(reinterpret_cast<void**>(&p->d_slutStatusAll), szg::SLUT_STATUS_RANKS * sizeof(unsigned));
    }
    if (e == hipSuccess)
    {
        e = hipMalloc(&p->d_framePrepDraw, szg::frame_prep_bytes());
    }
    if (e != hipSuccess)
    {
        szg_skyview_destroy(p);
        return fail_hip(e, "szg_skyview_create: hipMalloc");
    }
    *out = p;
    return SZG_OK;
}

int szg_skyview_set_lut_reuse(szg_skyview_t* p, int enable)
{
    if (p == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_set_lut_reuse: NULL argument");
    }
    p->lutReuse = enable != 0;
    // whatever was computed while reuse was off has no key on the device: the first frame after a switch recomputes
    p->forceTransmittance = true;
    p->forceSkyview = true;
    return SZG_OK;
}

int szg_skyview_invalidate_luts(szg_skyview_t* p, uint32_t which)
{
    if (p == nullptr || (which & ~(SZG_LUT_TRANSMITTANCE | SZG_LUT_SKYVIEW)) != 0u)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_invalidate_luts: NULL pipeline or unknown LUT bits 0x%x", which);
    }
    if ((which & SZG_LUT_TRANSMITTANCE) != 0u)
    {
        p->tlutStatusValid = false;
        p->forceTransmittance = true;
        p->forceSkyview = true; // every sky-view texel is a function of the transmittance texels
    }
    if ((which & SZG_LUT_SKYVIEW) != 0u)
    {
        p->slutStatusValid = false;
        p->sliceStatusKnown = false;
        p->forceSkyview = true;
    }
    return SZG_OK;
}

void szg_skyview_destroy(szg_skyview_t* p)
{
    if (p == nullptr)
    {
        return;
    }
    (void)hipSetDevice(p->device);
    if (p->d_transmittance != nullptr)
    {
        (void)hipFree(p->d_transmittance);
    }
    if (p->d_skyview != nullptr)
    {
        (void)hipFree(p->d_skyview);
    }
    if (p->d_multiscatter != nullptr)
    {
        (void)hipFree(p->d_multiscatter);
    }
    if (p->d_aerialLuminance != nullptr)
    {
        (void)hipFree(p->d_aerialLuminance);
    }
    if (p->d_aerialTransmittance != nullptr)
    {
        (void)hipFree(p->d_aerialTransmittance);
    }
    if (p->d_lutKey != nullptr)
    {
        (void)hipFree(p->d_lutKey);
    }
    if (p->d_framePrep != nullptr)
    {
        (void)hipFree(p->d_framePrep);
    }
    if (p->d_slutStatusAll != nullptr)
    {
        (void)hipFree(p->d_slutStatusAll);
    }
    if (p->d_framePrepDraw != nullptr)
    {
        (void)hipFree(p->d_framePrepDraw);
    }
    delete p;
}

int szg_skyview_transmittance_lut(const szg_skyview_t* p, szg_image* out)
{
    if (p == nullptr || out == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_transmittance_lut: NULL argument");
    }
    *out = make_image(p->d_transmittance, p->desc.transmittance_width, p->desc.transmittance_height, SZG_FORMAT_RGBA32_SFLOAT);
    p->tlutStatusValid = false; // the caller may write the texels through this view
    p->forceTransmittance = true;
    p->forceSkyview = true;
    return SZG_OK;
}

int szg_skyview_skyview_lut(const szg_skyview_t* p, szg_image* out)
{
    if (p == nullptr || out == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_skyview_lut: NULL argument");
    }
    *out = make_image(p->d_skyview, p->desc.skyview_width, p->desc.skyview_height, SZG_FORMAT_RGBA32_SFLOAT);
    p->slutStatusValid = false; // the caller may write the texels through this view (row slices gathered from other ranks)
    p->sliceStatusKnown = false;
    p->forceSkyview = true;
    return SZG_OK;
}

int szg_skyview_record_transmittance(szg_skyview_t* p, void* stream, uint32_t atmosphere_index,
                                     const szg_atmosphere_packed* d_atmospheres)
{
    if (p == nullptr || d_atmospheres == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_record_transmittance: NULL argument");
    }
    DeviceGuard const guard(p->device);
    hipStream_t const s = static_cast<hipStream_t>(stream);
    const unsigned* dirty = nullptr;
    if (p->lutReuse)
    {
        // the status dword of texels written behind our back must be settled before a clean verdict may keep them
        SZG_HIP(szg::launch_lut_key(s, d_atmospheres, atmosphere_index, nullptr, 0u, p->d_lutKey, 0u, p->forceTransmittance,
                                    p->d_transmittance, p->desc.transmittance_width, p->desc.transmittance_height));
        dirty = p->d_lutKey + 69;
    }
    SZG_HIP(szg::launch_transmittance(s, d_atmospheres, atmosphere_index, p->d_transmittance, p->desc.transmittance_width,
                                      p->desc.transmittance_height, dirty));
    p->haveTransmittance = true;
    p->tlutStatusValid = true;
    p->forceTransmittance = false;
    return SZG_OK;
}

int szg_skyview_record_skyview_lut_rows(szg_skyview_t* p, void* stream, uint32_t atmosphere_index,
                                        const szg_atmosphere_packed* d_atmospheres, uint32_t view_camera_index,
                                        const szg_camera_packed* d_cameras, uint32_t row_begin, uint32_t row_end)
{
    if (p == nullptr || d_atmospheres == nullptr || d_cameras == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_record_skyview_lut_rows: NULL argument");
    }
    if (row_begin > row_end || row_end > p->desc.skyview_height)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_record_skyview_lut_rows: rows [%u, %u) outside the %u-row LUT", row_begin,
                    row_end, p->desc.skyview_height);
    }
    DeviceGuard const guard(p->device);
    hipStream_t const s = static_cast<hipStream_t>(stream);
    bool const whole = (row_begin == 0u && row_end == p->desc.skyview_height);
    SZG_HIP(ensure_tlut_status(p, s));
    const unsigned* dirty = nullptr;
    if (p->lutReuse && whole)
    {
        SZG_HIP(szg::launch_lut_key(s, d_atmospheres, atmosphere_index, d_cameras, view_camera_index, p->d_lutKey, 1u, p->forceSkyview,
                                    p->d_skyview, p->desc.skyview_width, p->desc.skyview_height));
        dirty = p->d_lutKey + 70;
    }
    SZG_HIP(szg::launch_frame_prep(s, d_atmospheres, atmosphere_index, p->desc.transmittance_width, p->desc.transmittance_height,
                                   p->d_framePrep));
    SZG_HIP(szg::launch_skyview(s, d_atmospheres, atmosphere_index, d_cameras, view_camera_index, p->d_transmittance,
                                p->desc.transmittance_width, p->desc.transmittance_height, p->d_skyview, p->desc.skyview_width,
                                p->desc.skyview_height, row_begin, row_end, dirty, p->d_framePrep));
    p->haveSkyview = true;
    // a launch over all rows leaves the status dword behind the texels right; a slice does not know the other rows
    p->slutStatusValid = whole;
    p->sliceStatusKnown = !whole && dirty == nullptr; // (launch_skyview cleared the dword, the slice's waves set it)
    p->sliceRowBegin = row_begin;
    p->sliceRowEnd = row_end;
    p->forceSkyview = !whole; // a slice leaves the other rows to someone else: no key describes the whole LUT
    return SZG_OK;
}

int szg_skyview_lut_row_slice(const szg_skyview_t* p, uint32_t rank, uint32_t nranks, uint32_t* row_begin, uint32_t* row_end)
{
    if (p == nullptr || row_begin == nullptr || row_end == nullptr || nranks == 0u || rank >= nranks)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_lut_row_slice: NULL argument or rank outside [0, nranks)");
    }
    if (p->desc.skyview_height % nranks != 0u)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_lut_row_slice: %u LUT rows do not divide over %u ranks", p->desc.skyview_height,
                    nranks);
    }
    uint32_t const n = p->desc.skyview_height / nranks;
    *row_begin = rank * n;
    *row_end = (rank + 1u) * n;
    return SZG_OK;
}

int szg_skyview_allgather_lut_rows(szg_skyview_t* p, szg_rowtile_comm_t* comm, void* stream)
{
    if (p == nullptr || comm == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_allgather_lut_rows: NULL argument");
    }
    int const nranks = szg_rowtile_comm_size(comm);
    if (nranks < 1 || p->desc.skyview_height % (uint32_t)nranks != 0u)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_allgather_lut_rows: %u LUT rows do not divide over %d ranks",
                    p->desc.skyview_height, nranks);
    }
    if (nranks > (int)szg::SLUT_STATUS_RANKS)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_allgather_lut_rows: more than %u ranks", szg::SLUT_STATUS_RANKS);
    }
    DeviceGuard const guard(p->device);
    hipStream_t const s = static_cast<hipStream_t>(stream);
    size_t const slice = (size_t)(p->desc.skyview_height / (uint32_t)nranks) * p->desc.skyview_width * 16u;
    // this rank's status word first (the dword behind the texels is overwritten by nobody: the slices end in front of it).
    // It is known when the dword describes the whole LUT, or exactly the rows this rank contributes.
    unsigned const rank = (unsigned)szg_rowtile_comm_rank(comm);
    uint32_t const rowsPerRank = p->desc.skyview_height / (uint32_t)nranks;
    bool const known = p->slutStatusValid ||
                       (p->sliceStatusKnown && p->sliceRowBegin == rank * rowsPerRank && p->sliceRowEnd == (rank + 1u) * rowsPerRank);
    SZG_HIP(szg::launch_slut_status_stage(s, p->d_slutStatusAll, rank, p->d_skyview, p->desc.skyview_width,
                                          p->desc.skyview_height, known));
    int rc = szg_rowtile_allgather(comm, stream, p->d_skyview, slice);
    if (rc == SZG_OK)
    {
        // ... and the ranks' status words beside the slices: the LUT's status dword becomes their OR, so the composite needs
        // no 32 MiB re-scan of texels other ranks wrote. Every rank makes this second, 4-byte exchange unconditionally.
        rc = szg_rowtile_allgather(comm, stream, p->d_slutStatusAll, sizeof(unsigned));
    }
    if (rc == SZG_OK)
    {
        SZG_HIP(szg::launch_slut_status_reduce(s, p->d_slutStatusAll, (unsigned)nranks, p->d_skyview, p->desc.skyview_width,
                                               p->desc.skyview_height));
    }
    p->slutStatusValid = (rc == SZG_OK);
    p->sliceStatusKnown = false;
    p->forceSkyview = true; // no reuse key describes texels other ranks wrote
    return rc;
}

int szg_skyview_record_skyview_lut(szg_skyview_t* p, void* stream, uint32_t atmosphere_index,
                                   const szg_atmosphere_packed* d_atmospheres, uint32_t view_camera_index,
                                   const szg_camera_packed* d_cameras)
{
    if (p == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_record_skyview_lut: NULL argument");
    }
    return szg_skyview_record_skyview_lut_rows(p, stream, atmosphere_index, d_atmospheres, view_camera_index, d_cameras, 0u,
                                               p->desc.skyview_height);
}

static int record_composite(szg_skyview_t* p, void* stream, const szg_scene_texture* scene_texture, szg_rect draw_rect,
                            const szg_rowtile* tile, const szg_gbuffer* gbuffer, const szg_shadowmaps* shadow_maps,
                            uint32_t atmosphere_index, const szg_atmosphere_packed* d_atmospheres, uint32_t view_camera_index,
                            const szg_camera_packed* d_cameras, uint32_t sun_light_index,
                            const szg_directional_light_packed* d_lights, bool fast)
{
    if (p == nullptr || d_atmospheres == nullptr || d_cameras == nullptr || d_lights == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_record_composite: NULL argument");
    }
    if (!check_rect(draw_rect, "szg_skyview_record_composite"))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    DeviceGuard const guard(p->device);
    if (draw_rect.width == 0u || draw_rect.height == 0u)
    {
        return SZG_OK; // empty extent: nothing to dispatch (computeDispatchCount(0) == 0)
    }
    szg::TileArgs t{};
    if (!resolve_tile(tile, draw_rect.height, t))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    if (!check_scene(scene_texture, draw_rect.width, t.local_rows, true) || !check_gbuffer(gbuffer, draw_rect.width, t.local_rows))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    // camera.comp:369 indexes shadowMaps[sunLightIndex]; only that slot is needed.
    szg::ShadowSlot sun{nullptr, 0u, 0u, 0u, 0u};
    unsigned slotCount = 0;
    if (shadow_maps != nullptr && shadow_maps->maps != nullptr && sun_light_index < shadow_maps->count &&
        shadow_maps->maps[sun_light_index].data != nullptr)
    {
        const szg_image& m = shadow_maps->maps[sun_light_index];
        if (!check_image(m, SZG_FORMAT_D32_SFLOAT, 1u, 1u, "shadow map"))
        {
            return SZG_ERR_INVALID_ARGUMENT;
        }
        sun = szg::ShadowSlot{static_cast<const float*>(m.data), m.width, m.height, m.pitch_bytes / 4u, 0u};
        slotCount = sun_light_index + 1u;
    }
    (void)slotCount;
    szg::AerialLut aerial{nullptr, SZG_AERIAL_W, SZG_AERIAL_H, SZG_AERIAL_D, 0.0f};
    if (fast)
    {
        if (!(p->aerialMaxDistance > 0.0f))
        {
            return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_record_composite_fast: no aerial LUT has been recorded");
        }
        aerial.luminance = p->d_aerialLuminance;
        aerial.maxDistance = p->aerialMaxDistance;
    }
    SZG_HIP(ensure_tlut_status(p, static_cast<hipStream_t>(stream)));
    SZG_HIP(ensure_slut_status(p, static_cast<hipStream_t>(stream)));
    SZG_HIP(szg::launch_frame_prep(static_cast<hipStream_t>(stream), d_atmospheres, atmosphere_index, p->desc.transmittance_width,
                                   p->desc.transmittance_height, p->d_framePrepDraw,
                                   sun.map != nullptr ? d_lights + sun_light_index : nullptr));
    SZG_HIP(szg::launch_composite(static_cast<hipStream_t>(stream), *scene_texture, draw_rect.width, draw_rect.height, t, *gbuffer,
                                  sun, d_atmospheres, atmosphere_index, d_cameras, view_camera_index, d_lights, sun_light_index,
                                  p->d_transmittance, p->desc.transmittance_width, p->desc.transmittance_height, p->d_skyview,
                                  p->desc.skyview_width, p->desc.skyview_height, aerial, p->d_framePrepDraw));
    return SZG_OK;
}

int szg_skyview_record_composite(szg_skyview_t* p, void* stream, const szg_scene_texture* scene_texture, szg_rect draw_rect,
                                 const szg_rowtile* tile, const szg_gbuffer* gbuffer, const szg_shadowmaps* shadow_maps,
                                 uint32_t atmosphere_index, const szg_atmosphere_packed* d_atmospheres,
                                 uint32_t view_camera_index, const szg_camera_packed* d_cameras, uint32_t sun_light_index,
                                 const szg_directional_light_packed* d_lights)
{
    return record_composite(p, stream, scene_texture, draw_rect, tile, gbuffer, shadow_maps, atmosphere_index, d_atmospheres,
                            view_camera_index, d_cameras, sun_light_index, d_lights, false);
}

int szg_skyview_record_composite_fast(szg_skyview_t* p, void* stream, const szg_scene_texture* scene_texture, szg_rect draw_rect,
                                      const szg_rowtile* tile, const szg_gbuffer* gbuffer, const szg_shadowmaps* shadow_maps,
                                      uint32_t atmosphere_index, const szg_atmosphere_packed* d_atmospheres,
                                      uint32_t view_camera_index, const szg_camera_packed* d_cameras, uint32_t sun_light_index,
                                      const szg_directional_light_packed* d_lights)
{
    return record_composite(p, stream, scene_texture, draw_rect, tile, gbuffer, shadow_maps, atmosphere_index, d_atmospheres,
                            view_camera_index, d_cameras, sun_light_index, d_lights, true);
}

int szg_skyview_record_multiscatter_lut(szg_skyview_t* p, void* stream, uint32_t atmosphere_index,
                                        const szg_atmosphere_packed* d_atmospheres)
{
    if (p == nullptr || d_atmospheres == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_record_multiscatter_lut: NULL argument");
    }
    DeviceGuard const guard(p->device);
    SZG_HIP(ensure_tlut_status(p, static_cast<hipStream_t>(stream)));
    SZG_HIP(szg::launch_multiscatter(static_cast<hipStream_t>(stream), d_atmospheres, atmosphere_index, p->d_transmittance,
                                     p->desc.transmittance_width, p->desc.transmittance_height, p->d_multiscatter,
                                     SZG_MULTISCATTER_DIM));
    return SZG_OK;
}

int szg_skyview_multiscatter_lut(const szg_skyview_t* p, szg_image* out)
{
    if (p == nullptr || out == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_multiscatter_lut: NULL argument");
    }
    *out = make_image(p->d_multiscatter, SZG_MULTISCATTER_DIM, SZG_MULTISCATTER_DIM, SZG_FORMAT_RGBA32_SFLOAT);
    return SZG_OK;
}

int szg_skyview_record_aerial_lut(szg_skyview_t* p, void* stream, uint32_t atmosphere_index,
                                  const szg_atmosphere_packed* d_atmospheres, uint32_t view_camera_index,
                                  const szg_camera_packed* d_cameras, float max_distance_mm)
{
    if (p == nullptr || d_atmospheres == nullptr || d_cameras == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_record_aerial_lut: NULL argument");
    }
    if (!(max_distance_mm > 0.0f) || !(max_distance_mm < 1.0e6f))
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_record_aerial_lut: max_distance_mm must be in (0, 1e6)");
    }
    DeviceGuard const guard(p->device);
    SZG_HIP(ensure_tlut_status(p, static_cast<hipStream_t>(stream)));
    SZG_HIP(szg::launch_aerial_lut(static_cast<hipStream_t>(stream), d_atmospheres, atmosphere_index, d_cameras, view_camera_index,
                                   p->d_transmittance, p->desc.transmittance_width, p->desc.transmittance_height,
                                   p->d_aerialLuminance, p->d_aerialTransmittance, SZG_AERIAL_W, SZG_AERIAL_H, SZG_AERIAL_D,
                                   max_distance_mm));
    p->aerialMaxDistance = max_distance_mm;
    return SZG_OK;
}

int szg_skyview_aerial_lut(const szg_skyview_t* p, szg_image* out_luminance, szg_image* out_transmittance)
{
    if (p == nullptr || out_luminance == nullptr || out_transmittance == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_skyview_aerial_lut: NULL argument");
    }
    *out_luminance = make_image(p->d_aerialLuminance, SZG_AERIAL_W, SZG_AERIAL_H * SZG_AERIAL_D, SZG_FORMAT_RGBA32_SFLOAT);
    *out_transmittance = make_image(p->d_aerialTransmittance, SZG_AERIAL_W, SZG_AERIAL_H * SZG_AERIAL_D, SZG_FORMAT_RGBA32_SFLOAT);
    return SZG_OK;
}

int szg_skyview_record_draw_commands(szg_skyview_t* p, void* stream, const szg_scene_texture* scene_texture, szg_rect draw_rect,
                                     const szg_rowtile* tile, const szg_gbuffer* gbuffer, const szg_shadowmaps* shadow_maps,
                                     uint32_t atmosphere_index, const szg_atmosphere_packed* d_atmospheres,
                                     uint32_t view_camera_index, const szg_camera_packed* d_cameras, uint32_t sun_light_index,
                                     const szg_directional_light_packed* d_lights)
{
    // skyview.cpp:795-845, :847-893, :895-910: three dispatches in this order, every frame.
    int rc = szg_skyview_record_transmittance(p, stream, atmosphere_index, d_atmospheres);
    if (rc != SZG_OK)
    {
        return rc;
    }
    rc = szg_skyview_record_skyview_lut(p, stream, atmosphere_index, d_atmospheres, view_camera_index, d_cameras);
    if (rc != SZG_OK)
    {
        return rc;
    }
    return szg_skyview_record_composite(p, stream, scene_texture, draw_rect, tile, gbuffer, shadow_maps, atmosphere_index,
                                        d_atmospheres, view_camera_index, d_cameras, sun_light_index, d_lights);
}

// ---------------------------------------------------------------------------
// DeferredShadingPipeline
// ---------------------------------------------------------------------------
int szg_deferred_create(szg_deferred_t** out, const szg_deferred_desc* desc, int device)
{
    if (out == nullptr || desc == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_create: NULL argument");
    }
    *out = nullptr;
    if (desc->capacity_width == 0u || desc->capacity_height == 0u || desc->capacity_width > 32768u ||
        desc->capacity_height > 32768u)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_create: capacity %ux%u out of range", desc->capacity_width,
                    desc->capacity_height);
    }
    if (desc->max_spot_lights > 65536u || desc->max_shadow_maps > 65536u)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_create: light/shadow capacity out of range");
    }
    int const rc = select_device(device);
    if (rc != SZG_OK)
    {
        return rc;
    }
    szg_deferred* p = new (std::nothrow) szg_deferred();
    if (p == nullptr)
    {
        return fail(SZG_ERR_OUT_OF_MEMORY, "szg_deferred_create: host allocation failed");
    }
    p->device = device;
    p->desc = *desc;
    unsigned const W = desc->capacity_width, H = desc->capacity_height;
    unsigned const fmts[5] = {SZG_FORMAT_RGBA16_SFLOAT, SZG_FORMAT_RGBA16_SFLOAT, SZG_FORMAT_RGBA16_SFLOAT,
                              SZG_FORMAT_RGBA32_SFLOAT, SZG_FORMAT_RGBA16_SFLOAT};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 5 && e == hipSuccess; i++)
    {
        size_t const bytes = (size_t)W * H * texel_bytes(fmts[i]);
        e = hipMalloc(&p->d_gbufferPlanes[i], bytes);
        if (e == hipSuccess)
        {
            // attachments are cleared to 0 at the start of every G-buffer pass (deferred.cpp:493-560)
            e = hipMemset(p->d_gbufferPlanes[i], 0, bytes);
        }
    }
    unsigned const nSpots = desc->max_spot_lights > 0u ? desc->max_spot_lights : 1u;
    unsigned const nSlots = desc->max_shadow_maps > 0u ? desc->max_shadow_maps : 1u;
    if (e == hipSuccess)
    {
        e = hipMalloc(reinterpret_cast<void**>(&p->d_spots), (size_t)nSpots * sizeof(szg_spot_light_packed));
    }
    if (e == hipSuccess)
    {
        e = hipMalloc(reinterpret_cast<void**>(&p->d_slots), (size_t)nSlots * sizeof(szg::ShadowSlot));
    }
    if (e == hipSuccess)
    {
        e = hipMemset(p->d_slots, 0, (size_t)nSlots * sizeof(szg::ShadowSlot));
    }
    if (e == hipSuccess)
    {
        e = hipMalloc(reinterpret_cast<void**>(&p->d_lightRecs), (size_t)(nSpots + p->maxDirectional) * sizeof(szg::LightRec));
    }
    if (e == hipSuccess)
    {
        e = hipMalloc(reinterpret_cast<void**>(&p->d_boxes), (size_t)p->maxBoxes * sizeof(szg_fill_box));
    }
    if (e == hipSuccess)
    {
        e = hipMalloc(reinterpret_cast<void**>(&p->d_ownedSlots), (size_t)nSlots * sizeof(szg::ShadowSlot));
    }
    if (e == hipSuccess)
    {
        e = hipMemset(p->d_ownedSlots, 0, (size_t)nSlots * sizeof(szg::ShadowSlot));
    }
    if (e == hipSuccess)
    {
        e = hipMalloc(reinterpret_cast<void**>(&p->d_shadowGen), (size_t)nSlots * sizeof(szg::ShadowGen));
    }
    p->shadowImages.assign(desc->max_shadow_maps, szg_image{nullptr, 0u, 0u, 0u, SZG_FORMAT_D32_SFLOAT});
    if (e == hipSuccess && desc->shadow_map_dim > 0u && desc->max_shadow_maps > 0u)
    {
        // D32F array, cleared to 0 = far = unoccluded (shadowpass.cpp:188-248). Producing the
        // depth (triangle raster) is outside this path.
        size_t const one = (size_t)desc->shadow_map_dim * desc->shadow_map_dim * 4u;
        e = hipMalloc(&p->d_ownedShadowMaps, one * desc->max_shadow_maps);
        if (e == hipSuccess)
        {
            e = hipMemset(p->d_ownedShadowMaps, 0, one * desc->max_shadow_maps);
        }
        if (e == hipSuccess)
        {
            std::vector<szg::ShadowSlot> owned(desc->max_shadow_maps);
            for (unsigned i = 0; i < desc->max_shadow_maps; i++)
            {
                p->shadowImages[i] = make_image(static_cast<unsigned char*>(p->d_ownedShadowMaps) + one * i, desc->shadow_map_dim,
                                                desc->shadow_map_dim, SZG_FORMAT_D32_SFLOAT);
                owned[i] = szg::ShadowSlot{static_cast<const float*>(p->shadowImages[i].data), desc->shadow_map_dim,
                                           desc->shadow_map_dim, desc->shadow_map_dim, 0u};
            }
            e = hipMemcpy(p->d_ownedSlots, owned.data(), owned.size() * sizeof(szg::ShadowSlot), hipMemcpyHostToDevice);
        }
    }
    if (e != hipSuccess)
    {
        szg_deferred_destroy(p);
        return fail_hip(e, "szg_deferred_create: device allocation");
    }
    size_t stagingBytes = (size_t)nSpots * sizeof(szg_spot_light_packed);
    if ((size_t)nSlots * sizeof(szg::ShadowSlot) > stagingBytes)
    {
        stagingBytes = (size_t)nSlots * sizeof(szg::ShadowSlot);
    }
    if ((size_t)p->maxBoxes * sizeof(szg_fill_box) > stagingBytes)
    {
        stagingBytes = (size_t)p->maxBoxes * sizeof(szg_fill_box);
    }
    int const src = p->staging.init(stagingBytes);
    if (src != SZG_OK)
    {
        szg_deferred_destroy(p);
        return src;
    }
    p->gbuffer.diffuse = make_image(p->d_gbufferPlanes[0], W, H, SZG_FORMAT_RGBA16_SFLOAT);
    p->gbuffer.specular = make_image(p->d_gbufferPlanes[1], W, H, SZG_FORMAT_RGBA16_SFLOAT);
    p->gbuffer.normal = make_image(p->d_gbufferPlanes[2], W, H, SZG_FORMAT_RGBA16_SFLOAT);
    p->gbuffer.worldPosition = make_image(p->d_gbufferPlanes[3], W, H, SZG_FORMAT_RGBA32_SFLOAT);
    p->gbuffer.occlusionRoughnessMetallic = make_image(p->d_gbufferPlanes[4], W, H, SZG_FORMAT_RGBA16_SFLOAT);
    p->shadowMaps.count = desc->max_shadow_maps;
    p->shadowMaps.padding = 0;
    p->shadowMaps.maps = p->shadowImages.empty() ? nullptr : p->shadowImages.data();
    *out = p;
    return SZG_OK;
}

void szg_deferred_destroy(szg_deferred_t* p)
{
    if (p == nullptr)
    {
        return;
    }
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (void* plane : p->d_gbufferPlanes)
    {
        if (plane != nullptr)
        {
            (void)hipFree(plane);
        }
    }
    szg::RasterBuffers const& rb = p->raster;
    void* const rest[] = {p->d_ownedShadowMaps, p->d_spots,  p->d_slots, p->d_lightRecs, p->d_boxes,     p->d_ownedSlots,  p->d_shadowGen,
                          p->d_rasterDraws,     rb.prims,    rb.boxes,   rb.keysA,       rb.keysB,       rb.valsA,         rb.valsB,
                          rb.orderedBoxes,      rb.chunkBoxes, rb.superBoxes, rb.sortTemp};
    for (void* r : rest)
    {
        if (r != nullptr)
        {
            (void)hipFree(r);
        }
    }
    p->staging.destroy();
    delete p;
}

const szg_gbuffer* szg_deferred_gbuffer(szg_deferred_t* p) { return p != nullptr ? &p->gbuffer : nullptr; }
const szg_shadowmaps* szg_deferred_shadow_maps(szg_deferred_t* p) { return p != nullptr ? &p->shadowMaps : nullptr; }

int szg_deferred_set_shadow_map(szg_deferred_t* p, uint32_t index, const szg_image* map)
{
    if (p == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_set_shadow_map: NULL pipeline");
    }
    if (index >= p->shadowImages.size())
    {
        return fail(SZG_ERR_CAPACITY, "szg_deferred_set_shadow_map: slot %u >= capacity %zu", index, p->shadowImages.size());
    }
    if (map == nullptr || map->data == nullptr)
    {
        p->shadowImages[index] = szg_image{nullptr, 0u, 0u, 0u, SZG_FORMAT_D32_SFLOAT};
        return SZG_OK;
    }
    if (!check_image(*map, SZG_FORMAT_D32_SFLOAT, 1u, 1u, "shadow map"))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    p->shadowImages[index] = *map;
    return SZG_OK;
}

int szg_deferred_get_configuration(const szg_deferred_t* p, szg_deferred_configuration* out)
{
    if (p == nullptr || out == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_get_configuration: NULL argument");
    }
    *out = p->config;
    return SZG_OK;
}

int szg_deferred_set_configuration(szg_deferred_t* p, const szg_deferred_configuration* cfg)
{
    if (p == nullptr || cfg == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_set_configuration: NULL argument");
    }
    p->config = *cfg;
    return SZG_OK;
}

int szg_deferred_record_gbuffer_fill(szg_deferred_t* p, void* stream, szg_rect draw_rect, const szg_rowtile* tile,
                                     const szg_scene_texture* scene_texture, uint32_t view_camera_index,
                                     const szg_camera_packed* d_cameras, const szg_fill_scene* geometry)
{
    if (p == nullptr || d_cameras == nullptr || geometry == nullptr || scene_texture == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_record_gbuffer_fill: NULL argument");
    }
    if (!check_rect(draw_rect, "szg_deferred_record_gbuffer_fill"))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    DeviceGuard const guard(p->device);
    if (draw_rect.width == 0u || draw_rect.height == 0u)
    {
        return SZG_OK;
    }
    szg::TileArgs t{};
    if (!resolve_tile(tile, draw_rect.height, t))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    if (!check_gbuffer(&p->gbuffer, draw_rect.width, t.local_rows) ||
        !check_image(scene_texture->depth, SZG_FORMAT_D32_SFLOAT, draw_rect.width, t.local_rows, "scene_texture.depth"))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    if (geometry->box_count > p->maxBoxes || (geometry->box_count > 0u && geometry->boxes == nullptr))
    {
        return fail(SZG_ERR_CAPACITY, "szg_deferred_record_gbuffer_fill: %u boxes (capacity %u)", geometry->box_count, p->maxBoxes);
    }
    if (!(geometry->checker_cell > 0.0f))
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_record_gbuffer_fill: checker_cell must be > 0");
    }
    hipStream_t const s = static_cast<hipStream_t>(stream);
    int const rc = p->staging.upload(s, p->d_boxes, geometry->boxes, (size_t)geometry->box_count * sizeof(szg_fill_box));
    if (rc != SZG_OK)
    {
        return rc;
    }
    SZG_HIP(szg::launch_gbuffer_fill(s, *scene_texture, draw_rect.width, draw_rect.height, t, p->gbuffer, d_cameras,
                                     view_camera_index, geometry->ground_y, geometry->ground_half_extent, geometry->checker_cell,
                                     geometry->ground_roughness, p->d_boxes, geometry->box_count));
    return SZG_OK;
}

int szg_deferred_record_lights(szg_deferred_t* p, void* stream, szg_rect draw_rect, const szg_rowtile* tile,
                               const szg_scene_texture* scene_texture, uint32_t atmospheric_directional_lights_count,
                               const szg_directional_light_packed* d_directional_lights, uint32_t directional_light_count,
                               const szg_spot_light_packed* h_spot_lights, uint32_t spot_light_count, uint32_t view_camera_index,
                               const szg_camera_packed* d_cameras)
{
    if (p == nullptr || d_cameras == nullptr || scene_texture == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_record_lights: NULL argument");
    }
    if (!check_rect(draw_rect, "szg_deferred_record_lights"))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    DeviceGuard const guard(p->device);
    if (directional_light_count > 0u && d_directional_lights == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_record_lights: directional lights NULL");
    }
    if (spot_light_count > 0u && h_spot_lights == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_record_lights: spot lights NULL");
    }
    if (spot_light_count > p->desc.max_spot_lights)
    {
        return fail(SZG_ERR_CAPACITY, "szg_deferred_record_lights: %u spot lights exceed the capacity %u", spot_light_count,
                    p->desc.max_spot_lights);
    }
    unsigned const skip = atmospheric_directional_lights_count;
    unsigned const nDir = directional_light_count > skip ? directional_light_count - skip : 0u;
    if (nDir > p->maxDirectional)
    {
        return fail(SZG_ERR_CAPACITY, "szg_deferred_record_lights: %u directional lights exceed the capacity %u", nDir,
                    p->maxDirectional);
    }
    if (draw_rect.width == 0u || draw_rect.height == 0u)
    {
        return SZG_OK;
    }
    szg::TileArgs t{};
    if (!resolve_tile(tile, draw_rect.height, t))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    if (!check_scene(scene_texture, draw_rect.width, t.local_rows, false) ||
        !check_gbuffer(&p->gbuffer, draw_rect.width, t.local_rows))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    hipStream_t const s = static_cast<hipStream_t>(stream);

    // deferred.cpp:458-474: upload the spot lights to the pipeline's own buffer
    int rc = p->staging.upload(s, p->d_spots, h_spot_lights, (size_t)spot_light_count * sizeof(szg_spot_light_packed));
    if (rc != SZG_OK)
    {
        return rc;
    }
    // shadow-map slot table (descriptor array of shadowpass.cpp:300-340)
    unsigned const slotCount = (unsigned)p->shadowImages.size();
    if (slotCount > 0u)
    {
        std::vector<szg::ShadowSlot> slots(slotCount);
        for (unsigned i = 0; i < slotCount; i++)
        {
            const szg_image& m = p->shadowImages[i];
            slots[i] = szg::ShadowSlot{static_cast<const float*>(m.data), m.width, m.height, m.pitch_bytes / 4u, 0u};
        }
        rc = p->staging.upload(s, p->d_slots, slots.data(), slots.size() * sizeof(szg::ShadowSlot));
        if (rc != SZG_OK)
        {
            return rc;
        }
    }
    SZG_HIP(szg::launch_light_prep(s, d_directional_lights, directional_light_count, skip, p->d_spots, spot_light_count, p->d_slots,
                                   slotCount, p->d_lightRecs));
    SZG_HIP(szg::launch_lights(s, *scene_texture, draw_rect.width, draw_rect.height, t, p->gbuffer, d_cameras, view_camera_index,
                               p->d_lightRecs, nDir + spot_light_count));
    return SZG_OK;
}

int szg_deferred_record_shadow_maps(szg_deferred_t* p, void* stream, const szg_directional_light_packed* d_directional_lights,
                                    uint32_t directional_light_count, const szg_spot_light_packed* h_spot_lights,
                                    uint32_t spot_light_count, const szg_fill_scene* geometry)
{
    if (p == nullptr || geometry == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_record_shadow_maps: NULL argument");
    }
    DeviceGuard const guard(p->device);
    if ((directional_light_count > 0u && d_directional_lights == nullptr) || (spot_light_count > 0u && h_spot_lights == nullptr))
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_record_shadow_maps: light array NULL");
    }
    if (spot_light_count > p->desc.max_spot_lights)
    {
        return fail(SZG_ERR_CAPACITY, "szg_deferred_record_shadow_maps: %u spot lights exceed the capacity %u", spot_light_count,
                    p->desc.max_spot_lights);
    }
    if (geometry->box_count > p->maxBoxes || (geometry->box_count > 0u && geometry->boxes == nullptr))
    {
        return fail(SZG_ERR_CAPACITY, "szg_deferred_record_shadow_maps: %u boxes (capacity %u)", geometry->box_count, p->maxBoxes);
    }
    if (p->d_ownedShadowMaps == nullptr)
    {
        return SZG_OK; // the pipeline owns no shadow maps (shadow_map_dim == 0): nothing to render into
    }
    unsigned const lights = directional_light_count + spot_light_count;
    unsigned const slots = lights < p->desc.max_shadow_maps ? lights : p->desc.max_shadow_maps; // shadowpass.cpp:219-225
    if (slots == 0u)
    {
        return SZG_OK;
    }
    hipStream_t const s = static_cast<hipStream_t>(stream);
    int rc = p->staging.upload(s, p->d_spots, h_spot_lights, (size_t)spot_light_count * sizeof(szg_spot_light_packed));
    if (rc != SZG_OK)
    {
        return rc;
    }
    rc = p->staging.upload(s, p->d_boxes, geometry->boxes, (size_t)geometry->box_count * sizeof(szg_fill_box));
    if (rc != SZG_OK)
    {
        return rc;
    }
    SZG_HIP(szg::launch_shadow_maps(s, d_directional_lights, directional_light_count, p->d_spots, spot_light_count, p->d_ownedSlots,
                                    slots, p->d_shadowGen, p->d_boxes, geometry->box_count, p->desc.shadow_map_dim));
    return SZG_OK;
}

int szg_deferred_record_draw_commands(szg_deferred_t* p, void* stream, szg_rect draw_rect, const szg_rowtile* tile,
                                      const szg_scene_texture* scene_texture, uint32_t atmospheric_directional_lights_count,
                                      const szg_directional_light_packed* d_directional_lights,
                                      uint32_t directional_light_count, const szg_spot_light_packed* h_spot_lights,
                                      uint32_t spot_light_count, uint32_t view_camera_index, const szg_camera_packed* d_cameras,
                                      const szg_fill_scene* geometry)
{
    if (geometry != nullptr)
    {
        // deferred.cpp:480-490 shadow maps, then :493-713 the G-buffer pass
        int rc = szg_deferred_record_shadow_maps(p, stream, d_directional_lights, directional_light_count, h_spot_lights,
                                                 spot_light_count, geometry);
        if (rc != SZG_OK)
        {
            return rc;
        }
        rc = szg_deferred_record_gbuffer_fill(p, stream, draw_rect, tile, scene_texture, view_camera_index, d_cameras, geometry);
        if (rc != SZG_OK)
        {
            return rc;
        }
    }
    return szg_deferred_record_lights(p, stream, draw_rect, tile, scene_texture, atmospheric_directional_lights_count,
                                      d_directional_lights, directional_light_count, h_spot_lights, spot_light_count,
                                      view_camera_index, d_cameras);
}

} // extern "C"

namespace
{
// Per-device OETF tables (one per transfer function), built on first use by k_oetf_table and kept for the life of
// the process. `ready` orders later uses on other streams behind the build.
struct OetfTables
{
    std::mutex lock;
    unsigned short* table[16][2] = {};
    hipEvent_t ready[16][2] = {};
};
OetfTables g_oetf;

int oetf_table(hipStream_t s, unsigned function, const unsigned short** out)
{
    int device = 0;
    SZG_HIP(hipGetDevice(&device));
    if (device < 0 || device >= 16)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_record_oetf: device %d outside the table cache", device);
    }
    std::lock_guard<std::mutex> guard(g_oetf.lock);
    if (g_oetf.table[device][function] == nullptr)
    {
        unsigned short* t = nullptr;
        SZG_HIP(hipMalloc(reinterpret_cast<void**>(&t), 65536u * sizeof(unsigned short)));
        hipEvent_t e = nullptr;
        hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (err == hipSuccess)
        {
            err = szg::launch_oetf_table(s, t, function);
        }
        if (err == hipSuccess)
        {
            err = hipEventRecord(e, s);
        }
        if (err != hipSuccess)
        {
            (void)hipFree(t);
            if (e != nullptr)
            {
                (void)hipEventDestroy(e);
            }
            return fail_hip(err, "szg_record_oetf: building the transfer table");
        }
        g_oetf.table[device][function] = t;
        g_oetf.ready[device][function] = e;
    }
    else
    {
        SZG_HIP(hipStreamWaitEvent(s, g_oetf.ready[device][function], 0));
    }
    *out = g_oetf.table[device][function];
    return SZG_OK;
}
} // namespace

extern "C" {

int szg_record_oetf(void* stream, const szg_image* image, uint32_t width, uint32_t height, uint32_t transfer_function)
{
    if (image == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_record_oetf: image is NULL");
    }
    if (transfer_function != SZG_OETF_PURE_GAMMA && transfer_function != SZG_OETF_SRGB)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_record_oetf: unknown transfer function %u", transfer_function);
    }
    if (width == 0u || height == 0u)
    {
        return SZG_OK;
    }
    if (!check_image(*image, SZG_FORMAT_RGBA16_UNORM, width, height, "oetf image"))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    if (image->pitch_bytes % 16u != 0u || reinterpret_cast<uintptr_t>(image->data) % 16u != 0u)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_record_oetf: image rows must be 16-byte aligned");
    }
    const unsigned short* table = nullptr;
    int const rc = oetf_table(static_cast<hipStream_t>(stream), transfer_function, &table);
    if (rc != SZG_OK)
    {
        return rc;
    }
    SZG_HIP(szg::launch_oetf(static_cast<hipStream_t>(stream), *image, width, height, table));
    return SZG_OK;
}

int szg_compose_rowtiles(void* stream, const void* gathered, size_t tile_stride_bytes, uint32_t nranks, uint32_t block_rows,
                         const szg_image* dst, uint32_t width, uint32_t height)
{
    if (gathered == nullptr || dst == nullptr)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_compose_rowtiles: NULL argument");
    }
    if (nranks == 0u || block_rows == 0u)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_compose_rowtiles: nranks/block_rows must be > 0");
    }
    if (!check_image(*dst, SZG_FORMAT_RGBA16_UNORM, width, height, "compose dst"))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    if ((width * 8u) % 16u != 0u || tile_stride_bytes % 16u != 0u || dst->pitch_bytes % 16u != 0u ||
        reinterpret_cast<uintptr_t>(gathered) % 16u != 0u || reinterpret_cast<uintptr_t>(dst->data) % 16u != 0u)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_compose_rowtiles: rows must be 16-byte multiples and 16-byte aligned");
    }
    size_t maxRows = 0;
    for (uint32_t r = 0; r < nranks; r++)
    {
        size_t const rows = szg_rowtile_local_rows(height, block_rows, r, nranks);
        if (rows > maxRows)
        {
            maxRows = rows;
        }
    }
    if (maxRows * width * 8u > tile_stride_bytes)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_compose_rowtiles: tile stride %zu smaller than the largest tile %zu",
                    tile_stride_bytes, maxRows * width * 8u);
    }
    SZG_HIP(szg::launch_compose_rowtiles(static_cast<hipStream_t>(stream), gathered, tile_stride_bytes, nranks, block_rows, *dst,
                                         width, height));
    return SZG_OK;
}

// ---------------------------------------------------------------------------
// Compute rasteriser (szg/raster.h)
// ---------------------------------------------------------------------------
} // extern "C"

namespace
{
// Flatten the rendered meshes into the reference's draw calls (deferred.cpp:624-699 / pipelines.cpp:738-800):
// one draw per (mesh, surface), primitives numbered instance-major.
int collect_draws(const char* who, const szg_mesh_instanced* meshes, uint32_t meshCount, bool shadow,
                  std::vector<szg::RasterDraw>& draws, uint32_t& primCount)
{
    draws.clear();
    uint64_t prims = 0;
    for (uint32_t i = 0; i < meshCount; i++)
    {
        szg_mesh_instanced const& m = meshes[i];
        // collectGeometryCullFlags (deferred.cpp:394-428): render flag, mesh and both model buffers present
        if (m.render == 0u || m.d_vertices == nullptr || m.d_indices == nullptr || m.d_models == nullptr ||
            m.d_model_inverse_transposes == nullptr || m.instance_count == 0u)
        {
            continue;
        }
        if (shadow && m.casts_shadow == 0u) // pipelines.cpp:742
        {
            continue;
        }
        if (m.surface_count > 0u && m.surfaces == nullptr)
        {
            return fail(SZG_ERR_INVALID_ARGUMENT, "%s: mesh %u has %u surfaces but a NULL surface array", who, i, m.surface_count);
        }
        for (uint32_t k = 0; k < m.surface_count; k++)
        {
            szg_surface const& surf = m.surfaces[k];
            uint32_t const avail = surf.first_index >= m.index_count ? 0u : m.index_count - surf.first_index;
            uint32_t const tris = (surf.index_count < avail ? surf.index_count : avail) / 3u;
            if (tris == 0u)
            {
                continue;
            }
            szg::RasterDraw d{};
            d.vertices = m.d_vertices;
            d.indices = m.d_indices;
            d.models = m.d_models;
            d.mits = m.d_model_inverse_transposes;
            d.vertexCount = m.vertex_count;
            d.firstIndex = surf.first_index;
            d.triCount = tris;
            d.instanceCount = m.instance_count;
            d.firstPrim = (uint32_t)prims;
            d.tex[0] = surf.material.color;
            d.tex[1] = surf.material.normal;
            d.tex[2] = surf.material.orm;
            for (szg_texture const& t : d.tex)
            {
                if (t.data != nullptr && (t.width == 0u || t.height == 0u || t.width > 32768u || t.height > 32768u ||
                                          t.pitch_bytes < t.width * 4u || t.pitch_bytes % 4u != 0u))
                {
                    return fail(SZG_ERR_INVALID_ARGUMENT, "%s: mesh %u surface %u has a malformed texture", who, i, k);
                }
            }
            prims += (uint64_t)tris * m.instance_count;
            if (prims > 0x40000000ull)
            {
                return fail(SZG_ERR_CAPACITY, "%s: more than 2^30 primitives", who);
            }
            draws.push_back(d);
        }
    }
    primCount = (uint32_t)prims;
    return SZG_OK;
}

int ensure_raster_capacity(szg_deferred* p, hipStream_t s, size_t draws, size_t prims)
{
    if (draws > p->rasterDrawCapacity)
    {
        SZG_HIP(hipStreamSynchronize(s));
        if (p->d_rasterDraws != nullptr)
        {
            (void)hipFree(p->d_rasterDraws);
            p->d_rasterDraws = nullptr;
            p->rasterDrawCapacity = 0; // (a failing hipMalloc below must not leave the old capacity behind a NULL pointer)
        }
        size_t const n = draws * 2u;
        SZG_HIP(hipMalloc(reinterpret_cast<void**>(&p->d_rasterDraws), n * sizeof(szg::RasterDraw)));
        p->rasterDrawCapacity = n;
    }
    szg::RasterBuffers& rb = p->raster;
    if (prims > rb.capacity)
    {
        SZG_HIP(hipStreamSynchronize(s));
        void* const old[] = {rb.prims, rb.boxes, rb.keysA, rb.keysB, rb.valsA, rb.valsB, rb.orderedBoxes, rb.chunkBoxes, rb.superBoxes,
                             rb.sortTemp};
        for (void* o : old)
        {
            if (o != nullptr)
            {
                (void)hipFree(o);
            }
        }
        rb = szg::RasterBuffers{};
        size_t const n = ((prims * 3u / 2u) + 4095u) / 4096u * 4096u;
        SZG_HIP(hipMalloc(reinterpret_cast<void**>(&rb.prims), n * sizeof(szg::PrimRec)));
        SZG_HIP(hipMalloc(reinterpret_cast<void**>(&rb.boxes), n * sizeof(uint2)));
        SZG_HIP(hipMalloc(reinterpret_cast<void**>(&rb.keysA), n * sizeof(unsigned)));
        SZG_HIP(hipMalloc(reinterpret_cast<void**>(&rb.keysB), n * sizeof(unsigned)));
        SZG_HIP(hipMalloc(reinterpret_cast<void**>(&rb.valsA), n * sizeof(unsigned)));
        SZG_HIP(hipMalloc(reinterpret_cast<void**>(&rb.valsB), n * sizeof(unsigned)));
        SZG_HIP(hipMalloc(reinterpret_cast<void**>(&rb.orderedBoxes), n * sizeof(uint2)));
        SZG_HIP(hipMalloc(reinterpret_cast<void**>(&rb.chunkBoxes), (n / 64u) * sizeof(uint2)));
        SZG_HIP(hipMalloc(reinterpret_cast<void**>(&rb.superBoxes), (n / 4096u) * sizeof(uint2)));
        // radix-sort temp storage for up to n pairs (a function of the count only)
        SZG_HIP(szg::raster_sort_temp_bytes((unsigned)n, rb.sortTempBytes));
        SZG_HIP(hipMalloc(&rb.sortTemp, rb.sortTempBytes > 0u ? rb.sortTempBytes : 16u));
        rb.capacity = n;
    }
    return SZG_OK;
}

int upload_draws(szg_deferred* p, hipStream_t s, const std::vector<szg::RasterDraw>& draws)
{
    size_t const per = p->staging.bytes / sizeof(szg::RasterDraw);
    if (per == 0u)
    {
        return fail(SZG_ERR_CAPACITY, "staging ring smaller than one draw record");
    }
    for (size_t i = 0; i < draws.size(); i += per)
    {
        size_t const n = draws.size() - i < per ? draws.size() - i : per;
        int const rc = p->staging.upload(s, p->d_rasterDraws + i, draws.data() + i, n * sizeof(szg::RasterDraw));
        if (rc != SZG_OK)
        {
            return rc;
        }
    }
    return SZG_OK;
}
} // namespace

extern "C" {

int szg_deferred_record_gbuffer_raster(szg_deferred_t* p, void* stream, szg_rect draw_rect, const szg_rowtile* tile,
                                       const szg_scene_texture* scene_texture, uint32_t view_camera_index,
                                       const szg_camera_packed* d_cameras, const szg_mesh_instanced* meshes, uint32_t mesh_count)
{
    if (p == nullptr || d_cameras == nullptr || scene_texture == nullptr || (mesh_count > 0u && meshes == nullptr))
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_record_gbuffer_raster: NULL argument");
    }
    if (!check_rect(draw_rect, "szg_deferred_record_gbuffer_raster"))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    DeviceGuard const guard(p->device);
    if (draw_rect.width == 0u || draw_rect.height == 0u)
    {
        return SZG_OK;
    }
    if (draw_rect.width > 32768u || draw_rect.height > 32768u)
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_record_gbuffer_raster: draw extent above 32768");
    }
    szg::TileArgs t{};
    if (!resolve_tile(tile, draw_rect.height, t))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    if (!check_gbuffer(&p->gbuffer, draw_rect.width, t.local_rows) ||
        !check_image(scene_texture->depth, SZG_FORMAT_D32_SFLOAT, draw_rect.width, t.local_rows, "scene_texture.depth"))
    {
        return SZG_ERR_INVALID_ARGUMENT;
    }
    std::vector<szg::RasterDraw> draws;
    uint32_t primCount = 0;
    int rc = collect_draws("szg_deferred_record_gbuffer_raster", meshes, mesh_count, false, draws, primCount);
    if (rc != SZG_OK)
    {
        return rc;
    }
    hipStream_t const s = static_cast<hipStream_t>(stream);
    rc = ensure_raster_capacity(p, s, draws.size(), primCount);
    if (rc != SZG_OK)
    {
        return rc;
    }
    rc = upload_draws(p, s, draws);
    if (rc != SZG_OK)
    {
        return rc;
    }
    SZG_HIP(szg::launch_raster_setup(s, false, p->d_rasterDraws, (unsigned)draws.size(), primCount, d_cameras, view_camera_index, nullptr,
                                     draw_rect.width, draw_rect.height, p->raster));
    SZG_HIP(szg::launch_raster_tile(s, *scene_texture, draw_rect.width, draw_rect.height, t, p->gbuffer, p->d_rasterDraws, p->raster,
                                    primCount, d_cameras, view_camera_index));
    return SZG_OK;
}

int szg_deferred_record_shadow_raster(szg_deferred_t* p, void* stream, const szg_directional_light_packed* d_directional_lights,
                                      uint32_t directional_light_count, const szg_spot_light_packed* h_spot_lights,
                                      uint32_t spot_light_count, const szg_mesh_instanced* meshes, uint32_t mesh_count)
{
    if (p == nullptr || (mesh_count > 0u && meshes == nullptr))
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_record_shadow_raster: NULL argument");
    }
    DeviceGuard const guard(p->device);
    if ((directional_light_count > 0u && d_directional_lights == nullptr) || (spot_light_count > 0u && h_spot_lights == nullptr))
    {
        return fail(SZG_ERR_INVALID_ARGUMENT, "szg_deferred_record_shadow_raster: light array NULL");
    }
    if (spot_light_count > p->desc.max_spot_lights)
    {
        return fail(SZG_ERR_CAPACITY, "szg_deferred_record_shadow_raster: %u spot lights exceed the capacity %u", spot_light_count,
                    p->desc.max_spot_lights);
    }
    if (p->d_ownedShadowMaps == nullptr)
    {
        return SZG_OK; // the pipeline owns no shadow maps
    }
    unsigned const lights = directional_light_count + spot_light_count;
    unsigned const slots = lights < p->desc.max_shadow_maps ? lights : p->desc.max_shadow_maps; // shadowpass.cpp:219-225
    if (slots == 0u)
    {
        return SZG_OK;
    }
    std::vector<szg::RasterDraw> draws;
    uint32_t primCount = 0;
    int rc = collect_draws("szg_deferred_record_shadow_raster", meshes, mesh_count, true, draws, primCount);
    if (rc != SZG_OK)
    {
        return rc;
    }
    hipStream_t const s = static_cast<hipStream_t>(stream);
    rc = ensure_raster_capacity(p, s, draws.size(), primCount);
    if (rc != SZG_OK)
    {
        return rc;
    }
    rc = p->staging.upload(s, p->d_spots, h_spot_lights, (size_t)spot_light_count * sizeof(szg_spot_light_packed));
    if (rc != SZG_OK)
    {
        return rc;
    }
    rc = upload_draws(p, s, draws);
    if (rc != SZG_OK)
    {
        return rc;
    }
    SZG_HIP(szg::launch_shadow_prep(s, d_directional_lights, directional_light_count, p->d_spots, spot_light_count, p->d_ownedSlots, slots,
                                    p->d_shadowGen));
    unsigned const dim = p->desc.shadow_map_dim;
    for (unsigned slot = 0; slot < slots; slot++)
    {
        // the primitive buffers are reused slot after slot: stream order keeps setup(k+1) behind tile(k)
        SZG_HIP(szg::launch_raster_setup(s, true, p->d_rasterDraws, (unsigned)draws.size(), primCount, nullptr, 0u, p->d_shadowGen + slot, dim,
                                         dim, p->raster));
        SZG_HIP(szg::launch_shadow_tile(s, p->d_shadowGen + slot, dim, p->raster, primCount, p->config.depthBiasConstant,
                                        p->config.depthBiasSlope));
    }
    return SZG_OK;
}

int szg_deferred_record_draw_commands_meshes(szg_deferred_t* p, void* stream, szg_rect draw_rect, const szg_rowtile* tile,
                                             const szg_scene_texture* scene_texture, uint32_t atmospheric_directional_lights_count,
                                             const szg_directional_light_packed* d_directional_lights,
                                             uint32_t directional_light_count, const szg_spot_light_packed* h_spot_lights,
                                             uint32_t spot_light_count, uint32_t view_camera_index, const szg_camera_packed* d_cameras,
                                             const szg_mesh_instanced* meshes, uint32_t mesh_count)
{
    // deferred.cpp:480-490 shadow maps, :493-713 G-buffer pass, :715-787 lights
    int rc = szg_deferred_record_shadow_raster(p, stream, d_directional_lights, directional_light_count, h_spot_lights, spot_light_count,
                                               meshes, mesh_count);
    if (rc != SZG_OK)
    {
        return rc;
    }
    rc = szg_deferred_record_gbuffer_raster(p, stream, draw_rect, tile, scene_texture, view_camera_index, d_cameras, meshes, mesh_count);
    if (rc != SZG_OK)
    {
        return rc;
    }
    return szg_deferred_record_lights(p, stream, draw_rect, tile, scene_texture, atmospheric_directional_lights_count,
                                      d_directional_lights, directional_light_count, h_spot_lights, spot_light_count, view_camera_index,
                                      d_cameras);
}

} // extern "C"

