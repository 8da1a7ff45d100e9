// szg/pipelines.hpp — header-only C++ mirror of the reference's render-pass classes for
// the deferred-shading + atmosphere path, over the C-ABI of szg/abi.h.
//
// A caller written against the reference (Renderer::recordDraw, renderer.cpp:278-443)
// ports one to one:
//
//   reference                                              this header
//   ------------------------------------------------------ -----------------------------
//   VkCommandBuffer cmd                                    hipStream_t cmd
//   TStagedBuffer<T>            buffers.hpp:209-299        szg::TStagedBuffer<T>
//   SceneTexture                scenetexture.hpp:11-81     szg::SceneTexture
//   GBuffer / ShadowPassArray   gbuffer.hpp, shadowpass.hpp szg_gbuffer / szg_shadowmaps
//   DeferredShadingPipeline     pipelines/deferred.hpp:23  szg::DeferredShadingPipeline
//   SkyViewComputePipeline      pipelines/skyview.hpp:24   szg::SkyViewComputePipeline
//   std::span<MeshInstanced const> sceneGeometry           szg_fill_scene const* (synthetic)
//
// Error behaviour follows the reference: construction failures give an invalid object /
// nullptr plus a log line; record* return void and never throw.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <memory>
#include <span>
#include <vector>

#include "szg/abi.h"
#include "szg/raster.h"
#include "szg/host.h"

namespace szg
{
using CameraPacked = szg_camera_packed;
using AtmospherePacked = szg_atmosphere_packed;
using DirectionalLightPacked = szg_directional_light_packed;
using SpotLightPacked = szg_spot_light_packed;

// buffers.hpp:209-299 — host staging + device copy, recordCopyToDevice on a stream.
// recordCopyToDevice is asynchronous: the copy runs in stream order, possibly behind a whole frame of kernels, while the
// caller goes on to stage the next frame. The reference guards its staging memory with the frame fences (two frames in
// flight, framebuffer.cpp:134); here the staged bytes are handed to the stream through a small ring of pinned slots, and a
// slot is rewritten only after the event recorded behind its last copy has completed.
template <typename T> struct TStagedBuffer
{
    static constexpr int RING_SLOTS = 3;
    TStagedBuffer() = default;
    TStagedBuffer(TStagedBuffer const&) = delete;
    auto operator=(TStagedBuffer const&) -> TStagedBuffer& = delete;
    TStagedBuffer(TStagedBuffer&& o) noexcept { *this = std::move(o); }
    auto operator=(TStagedBuffer&& o) noexcept -> TStagedBuffer&
    {
        destroy();
        m_host = o.m_host;
        m_device = o.m_device;
        m_capacity = o.m_capacity;
        m_staged = o.m_staged;
        m_deviceSize = o.m_deviceSize;
        m_dirty = o.m_dirty;
        m_next = o.m_next;
        for (int i = 0; i < RING_SLOTS; i++)
        {
            m_ring[i] = o.m_ring[i];
            m_done[i] = o.m_done[i];
            m_used[i] = o.m_used[i];
            o.m_ring[i] = nullptr;
            o.m_done[i] = nullptr;
            o.m_used[i] = false;
        }
        o.m_host = nullptr;
        o.m_device = nullptr;
        o.m_capacity = 0;
        return *this;
    }
    ~TStagedBuffer() { destroy(); }

    static auto allocate(size_t capacity) -> TStagedBuffer<T>
    {
        TStagedBuffer<T> b;
        bool ok = hipHostMalloc(reinterpret_cast<void**>(&b.m_host), capacity * sizeof(T), hipHostMallocDefault) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&b.m_device), capacity * sizeof(T)) == hipSuccess;
        for (int i = 0; ok && i < RING_SLOTS; i++)
        {
            ok = hipHostMalloc(reinterpret_cast<void**>(&b.m_ring[i]), capacity * sizeof(T), hipHostMallocDefault) == hipSuccess &&
                 hipEventCreateWithFlags(&b.m_done[i], hipEventDisableTiming) == hipSuccess;
        }
        if (!ok)
        {
            std::fprintf(stderr, "[szg] TStagedBuffer::allocate(%zu) failed\n", capacity);
            b.destroy();
            return b;
        }
        b.m_capacity = capacity;
        return b;
    }
    void clearStaged()
    {
        m_staged = 0;
        m_dirty = true;
    }
    void push(T const& value) { push(std::span<T const>{&value, 1}); }
    void push(std::span<T const> data)
    {
        if (m_staged + data.size() > m_capacity)
        {
            std::fprintf(stderr, "[szg] TStagedBuffer::push exceeds capacity\n");
            return;
        }
        std::memcpy(m_host + m_staged, data.data(), data.size_bytes());
        m_staged += data.size();
        m_dirty = true;
    }
    void stage(std::span<T const> data)
    {
        clearStaged();
        push(data);
    }
    void pop(size_t count) { m_staged = count > m_staged ? 0 : m_staged - count; }
    void recordCopyToDevice(hipStream_t cmd)
    {
        if (m_staged > 0)
        {
            int const slot = m_next;
            m_next = (m_next + 1) % RING_SLOTS;
            if (m_used[slot])
            {
                (void)hipEventSynchronize(m_done[slot]); // the copy that last read this slot has run
            }
            std::memcpy(m_ring[slot], m_host, m_staged * sizeof(T));
            (void)hipMemcpyAsync(m_device, m_ring[slot], m_staged * sizeof(T), hipMemcpyHostToDevice, cmd);
            (void)hipEventRecord(m_done[slot], cmd);
            m_used[slot] = true;
        }
        m_deviceSize = m_staged;
        m_dirty = false;
    }
    [[nodiscard]] auto deviceAddress() const -> T const* { return m_device; }
    [[nodiscard]] auto deviceSize() const -> size_t { return m_deviceSize; }
    [[nodiscard]] auto stagedSize() const -> size_t { return m_staged; }
    [[nodiscard]] auto stagingCapacity() const -> size_t { return m_capacity; }
    [[nodiscard]] auto isDirty() const -> bool { return m_dirty; }
    [[nodiscard]] auto readValidStaged() const -> std::span<T const>
    {
        if (m_dirty)
        {
            std::fprintf(stderr, "[szg] Dirty buffer was accessed with a read\n"); // buffers.hpp:252-260
        }
        return {m_host, m_staged};
    }
    [[nodiscard]] auto valid() const -> bool { return m_capacity > 0; }

private:
    void destroy()
    {
        for (int i = 0; i < RING_SLOTS; i++)
        {
            if (m_used[i])
            {
                (void)hipEventSynchronize(m_done[i]);
            }
            if (m_done[i] != nullptr)
            {
                (void)hipEventDestroy(m_done[i]);
            }
            if (m_ring[i] != nullptr)
            {
                (void)hipHostFree(m_ring[i]);
            }
            m_ring[i] = nullptr;
            m_done[i] = nullptr;
            m_used[i] = false;
        }
        if (m_host != nullptr)
        {
            (void)hipHostFree(m_host);
        }
        if (m_device != nullptr)
        {
            (void)hipFree(m_device);
        }
        m_host = nullptr;
        m_device = nullptr;
        m_capacity = 0;
    }
    T* m_host{nullptr};
    T* m_device{nullptr};
    T* m_ring[RING_SLOTS]{};
    hipEvent_t m_done[RING_SLOTS]{};
    bool m_used[RING_SLOTS]{};
    int m_next{0};
    size_t m_capacity{0};
    size_t m_staged{0};
    size_t m_deviceSize{0};
    bool m_dirty{false};
};

// scenetexture.hpp:11-81 — colour RGBA16 UNORM + depth D32F at a capacity extent.
struct SceneTexture
{
    SceneTexture() = default;
    SceneTexture(SceneTexture const&) = delete;
    auto operator=(SceneTexture const&) -> SceneTexture& = delete;
    ~SceneTexture()
    {
        (void)hipFree(m_texture.color.data);
        (void)hipFree(m_texture.depth.data);
        (void)hipFree(m_texture.debug_color.data);
    }
    static auto create(uint32_t width, uint32_t height, bool debugColor = false) -> std::unique_ptr<SceneTexture>
    {
        auto t = std::make_unique<SceneTexture>();
        auto image = [&](szg_image& im, uint32_t fmt, uint32_t texel) {
            im = szg_image{nullptr, width, height, width * texel, fmt};
            return hipMalloc(&im.data, (size_t)width * height * texel) == hipSuccess &&
                   hipMemset(im.data, 0, (size_t)width * height * texel) == hipSuccess;
        };
        bool ok = image(t->m_texture.color, SZG_FORMAT_RGBA16_UNORM, 8) && image(t->m_texture.depth, SZG_FORMAT_D32_SFLOAT, 4);
        if (ok && debugColor)
        {
            ok = image(t->m_texture.debug_color, SZG_FORMAT_RGBA32_SFLOAT, 16);
        }
        if (!ok)
        {
            std::fprintf(stderr, "[szg] SceneTexture::create(%u, %u) failed\n", width, height);
            return nullptr;
        }
        return t;
    }
    [[nodiscard]] auto texture() const -> szg_scene_texture const& { return m_texture; }
    [[nodiscard]] auto color() const -> szg_image const& { return m_texture.color; }
    [[nodiscard]] auto depth() const -> szg_image const& { return m_texture.depth; }

private:
    szg_scene_texture m_texture{};
};

// The reference's record calls return void and log what goes wrong (SZG_ERROR); so do these wrappers: a negative status of
// the C-ABI call is printed once per call site to stderr with szg_last_error()'s text and kept in the pipeline's
// lastStatus(), so that a refused pass (e.g. a draw rect with a non-zero offset, szg/abi.h) is never a silently
// unrendered frame.
namespace detail
{
inline auto note(int status, char const* what, int& last) -> int
{
    last = status;
    if (status < 0)
    {
        std::fprintf(stderr, "[szg] %s failed with status %d: %s\n", what, status, szg_last_error());
    }
    return status;
}
} // namespace detail

// pipelines/deferred.hpp:23-119
struct DeferredShadingPipeline
{
    using Configuration = szg_deferred_configuration;

    // deferred.hpp:26-32. Invalid (valid() == false) on failure, like the reference's
    // shaders that stay "invalid" (deferred.cpp:153-163).
    DeferredShadingPipeline(uint32_t capacityWidth, uint32_t capacityHeight, uint32_t maxSpotLights = 16,
                            uint32_t maxShadowMaps = 10, uint32_t shadowMapDimension = 0, int device = 0)
    {
        szg_deferred_desc const desc{capacityWidth, capacityHeight, maxSpotLights, maxShadowMaps, shadowMapDimension, 0};
        if (szg_deferred_create(&m_handle, &desc, device) != SZG_OK)
        {
            m_handle = nullptr;
        }
    }
    DeferredShadingPipeline(DeferredShadingPipeline const&) = delete;
    auto operator=(DeferredShadingPipeline const&) -> DeferredShadingPipeline& = delete;
    ~DeferredShadingPipeline() { cleanup(); }

    // deferred.hpp:34-44
    void recordDrawCommands(hipStream_t cmd, szg_rect drawRect, SceneTexture& sceneTexture,
                            uint32_t atmosphericDirectionalLightsCount,
                            TStagedBuffer<DirectionalLightPacked> const& directionalLights,
                            std::span<SpotLightPacked const> spotLights, uint32_t viewCameraIndex,
                            TStagedBuffer<CameraPacked> const& cameras, szg_fill_scene const* sceneGeometry,
                            szg_rowtile const* tile = nullptr)
    {
        detail::note(szg_deferred_record_draw_commands(m_handle, cmd, drawRect, tile, &sceneTexture.texture(),
                                                atmosphericDirectionalLightsCount, directionalLights.deviceAddress(),
                                                static_cast<uint32_t>(directionalLights.deviceSize()), spotLights.data(),
                                                static_cast<uint32_t>(spotLights.size()), viewCameraIndex,
                                                cameras.deviceAddress(), sceneGeometry), "szg_deferred_record_draw_commands", m_lastStatus);
    }
    // deferred.hpp:34-44 with the reference's own last argument, std::span<MeshInstanced const> sceneGeometry
    // (szg/raster.h): shadow raster, G-buffer raster, lights.
    void recordDrawCommands(hipStream_t cmd, szg_rect drawRect, SceneTexture& sceneTexture,
                            uint32_t atmosphericDirectionalLightsCount,
                            TStagedBuffer<DirectionalLightPacked> const& directionalLights,
                            std::span<SpotLightPacked const> spotLights, uint32_t viewCameraIndex,
                            TStagedBuffer<CameraPacked> const& cameras, std::span<szg_mesh_instanced const> sceneGeometry,
                            szg_rowtile const* tile = nullptr)
    {
        detail::note(szg_deferred_record_draw_commands_meshes(m_handle, cmd, drawRect, tile, &sceneTexture.texture(),
                                                       atmosphericDirectionalLightsCount, directionalLights.deviceAddress(),
                                                       static_cast<uint32_t>(directionalLights.deviceSize()), spotLights.data(),
                                                       static_cast<uint32_t>(spotLights.size()), viewCameraIndex,
                                                       cameras.deviceAddress(), sceneGeometry.data(),
                                                       static_cast<uint32_t>(sceneGeometry.size())), "szg_deferred_record_draw_commands_meshes", m_lastStatus);
    }
    [[nodiscard]] auto lastStatus() const -> int { return m_lastStatus; } // status of the last record call (SZG_OK or negative)
    [[nodiscard]] auto gbuffer() -> szg_gbuffer const& { return *szg_deferred_gbuffer(m_handle); }          // deferred.hpp:46
    [[nodiscard]] auto shadowMaps() -> szg_shadowmaps const& { return *szg_deferred_shadow_maps(m_handle); } // deferred.hpp:47
    void cleanup() // deferred.hpp:49
    {
        szg_deferred_destroy(m_handle);
        m_handle = nullptr;
    }
    [[nodiscard]] auto getConfiguration() const -> Configuration // deferred.hpp:114
    {
        Configuration c{};
        (void)szg_deferred_get_configuration(m_handle, &c);
        return c;
    }
    void setConfiguration(Configuration c) { (void)szg_deferred_set_configuration(m_handle, &c); } // deferred.hpp:115
    [[nodiscard]] auto valid() const -> bool { return m_handle != nullptr; }

private:
    szg_deferred_t* m_handle{nullptr};
    int m_lastStatus{SZG_OK};
};

// pipelines/skyview.hpp:24-51
struct SkyViewComputePipeline
{
    auto operator=(SkyViewComputePipeline&&) -> SkyViewComputePipeline& = delete;
    SkyViewComputePipeline(SkyViewComputePipeline const&) = delete;
    auto operator=(SkyViewComputePipeline const&) -> SkyViewComputePipeline& = delete;
    SkyViewComputePipeline(SkyViewComputePipeline&& o) noexcept : m_handle(o.m_handle) { o.m_handle = nullptr; }
    ~SkyViewComputePipeline() { szg_skyview_destroy(m_handle); }

    // skyview.hpp:36-37; nullptr on any failure (skyview.cpp:713-740)
    [[nodiscard]] static auto create(int device = 0, szg_skyview_desc const* desc = nullptr)
        -> std::unique_ptr<SkyViewComputePipeline>
    {
        szg_skyview_t* h = nullptr;
        if (szg_skyview_create(&h, desc, device) != SZG_OK)
        {
            return nullptr;
        }
        return std::unique_ptr<SkyViewComputePipeline>(new SkyViewComputePipeline(h));
    }

    // skyview.hpp:39-51
    void recordDrawCommands(hipStream_t cmd, SceneTexture& sceneTexture, szg_rect drawRect, szg_gbuffer const& gbuffer,
                            szg_shadowmaps const& shadowMaps, uint32_t atmosphereIndex,
                            TStagedBuffer<AtmospherePacked> const& atmospheres, uint32_t viewCameraIndex,
                            TStagedBuffer<CameraPacked> const& cameras, uint32_t sunLightIndex,
                            TStagedBuffer<DirectionalLightPacked> const& lights, szg_rowtile const* tile = nullptr)
    {
        detail::note(szg_skyview_record_draw_commands(m_handle, cmd, &sceneTexture.texture(), drawRect, tile, &gbuffer, &shadowMaps,
                                               atmosphereIndex, atmospheres.deviceAddress(), viewCameraIndex,
                                               cameras.deviceAddress(), sunLightIndex, lights.deviceAddress()), "szg_skyview_record_draw_commands", m_lastStatus);
    }
    // ---- extensions without a reference counterpart (szg/abi.h) ----
    // LUT reuse across frames whose atmosphere / sun / camera position are unchanged (identical results)
    [[nodiscard]] auto lastStatus() const -> int { return m_lastStatus; } // status of the last record call (SZG_OK or negative)
    void setLUTReuse(bool enable) { (void)szg_skyview_set_lut_reuse(m_handle, enable ? 1 : 0); }
    void invalidateLUTs(uint32_t which = SZG_LUT_TRANSMITTANCE | SZG_LUT_SKYVIEW) { (void)szg_skyview_invalidate_luts(m_handle, which); }
    // Row-tiled multi-GPU frame: transmittance LUT, this rank's slice of the sky-view LUT, the all-gather of the slices on
    // `lutStream` (ordered behind `cmd` by an event), then the composite on `cmd` once the LUT is complete.
    void recordDrawCommandsTiled(hipStream_t cmd, hipStream_t lutStream, hipEvent_t scratchEvent, szg_rowtile_comm_t* comm,
                                 SceneTexture& sceneTexture, szg_rect drawRect, szg_gbuffer const& gbuffer,
                                 szg_shadowmaps const& shadowMaps, uint32_t atmosphereIndex,
                                 TStagedBuffer<AtmospherePacked> const& atmospheres, uint32_t viewCameraIndex,
                                 TStagedBuffer<CameraPacked> const& cameras, uint32_t sunLightIndex,
                                 TStagedBuffer<DirectionalLightPacked> const& lights, szg_rowtile const& tile)
    {
        uint32_t begin = 0, end = 0;
        detail::note(szg_skyview_record_transmittance(m_handle, cmd, atmosphereIndex, atmospheres.deviceAddress()), "szg_skyview_record_transmittance", m_lastStatus);
        if (szg_skyview_lut_row_slice(m_handle, tile.rank, tile.nranks, &begin, &end) == SZG_OK)
        {
            detail::note(szg_skyview_record_skyview_lut_rows(m_handle, cmd, atmosphereIndex, atmospheres.deviceAddress(), viewCameraIndex,
                                                      cameras.deviceAddress(), begin, end), "szg_skyview_record_skyview_lut_rows", m_lastStatus);
            (void)hipEventRecord(scratchEvent, cmd);
            (void)hipStreamWaitEvent(lutStream, scratchEvent, 0);
            detail::note(szg_skyview_allgather_lut_rows(m_handle, comm, lutStream), "szg_skyview_allgather_lut_rows", m_lastStatus);
            (void)hipEventRecord(scratchEvent, lutStream);
            (void)hipStreamWaitEvent(cmd, scratchEvent, 0);
        }
        else
        {
            // the LUT's rows do not divide over the ranks (every rank sees that alike): each rank computes the whole LUT,
            // as the single-GPU frame does, and the optional second collective is skipped
            detail::note(szg_skyview_record_skyview_lut(m_handle, cmd, atmosphereIndex, atmospheres.deviceAddress(), viewCameraIndex,
                                                 cameras.deviceAddress()), "szg_skyview_record_skyview_lut", m_lastStatus);
        }
        detail::note(szg_skyview_record_composite(m_handle, cmd, &sceneTexture.texture(), drawRect, &tile, &gbuffer, &shadowMaps,
                                           atmosphereIndex, atmospheres.deviceAddress(), viewCameraIndex, cameras.deviceAddress(),
                                           sunLightIndex, lights.deviceAddress()), "szg_skyview_record_composite", m_lastStatus);
    }
    [[nodiscard]] auto transmittanceLUT() const -> szg_image
    {
        szg_image im{};
        (void)szg_skyview_transmittance_lut(m_handle, &im);
        return im;
    }
    [[nodiscard]] auto skyviewLUT() const -> szg_image
    {
        szg_image im{};
        (void)szg_skyview_skyview_lut(m_handle, &im);
        return im;
    }

private:
    explicit SkyViewComputePipeline(szg_skyview_t* h) : m_handle(h) {}
    szg_skyview_t* m_handle{nullptr};
    int m_lastStatus{SZG_OK};
};
} // namespace szg
