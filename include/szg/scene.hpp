// szg/scene.hpp — header-only C++ mirror of the reference's Scene and of the part of Renderer that records the
// deferred + atmosphere frame, over szg/host.h, szg/assets.hpp and szg/pipelines.hpp: the caller either side of the hot path
// (SURVEY 8 f1), with the reference's names.
//
//   reference                                              this header
//   ------------------------------------------------------ ------------------------------------------------------------
//   TickTiming            core/timing.hpp:5-9              szg::TickTiming
//   Scene                 renderer/scene.hpp:160-222        szg::Scene: sunAnimation, atmosphere, camera, spotlights(+Render),
//                         renderer/scene.cpp:95-574         calculateShadowBounds, shadowBounds, geometry, addMeshInstance,
//                                                           addSpotlight, defaultScene, tick
//   Renderer::recordDraw  renderer/renderer.cpp:278-443     szg::Renderer::recordDraw (deferred pipeline branch; the debug
//                                                           lines and the generic compute collection are editor features)
//
// Not mirrored: Scene::handleInput (window input), Scene::diagonalWaveScene (its instance rotations come from the
// reference's random quaternion source and are not reproducible).
#pragma once

#include <optional>
#include <span>
#include <string>
#include <vector>

#include "szg/assets.hpp"
#include "szg/host.h"
#include "szg/pipelines.hpp"

namespace szg
{
struct TickTiming
{
    double timeElapsedSeconds;
    double deltaTimeSeconds;
};

struct Scene
{
    Scene()
    {
        szg_sun_animation_default(&sunAnimation);
        szg_atmosphere_default_earth(&atmosphere);
        szg_camera_default(&camera);
    }

    szg_sun_animation sunAnimation{};
    szg_atmosphere atmosphere{};
    szg_camera camera{};
    bool spotlightsRender{false};
    std::vector<SpotLightPacked> spotlights{};

    // scene.cpp:95-148
    void calculateShadowBounds()
    {
        std::vector<szg_shadow_caster> casters;
        for (MeshInstanced const& instance : m_geometry)
        {
            auto const mesh = instance.getMesh();
            if (mesh == nullptr)
            {
                continue;
            }
            szg_shadow_caster c{};
            c.vertex_bounds = mesh->vertexBounds;
            c.transforms = instance.transforms.data();
            c.transform_count = static_cast<uint32_t>(instance.transforms.size());
            c.render = instance.render ? 1u : 0u;
            c.casts_shadow = instance.castsShadow ? 1u : 0u;
            casters.push_back(c);
        }
        (void)szg_calculate_shadow_bounds(casters.data(), static_cast<uint32_t>(casters.size()), &m_shadowBounds);
    }
    [[nodiscard]] auto shadowBounds() const -> szg_aabb { return m_shadowBounds; }

    [[nodiscard]] auto geometry() const -> std::span<MeshInstanced const> { return m_geometry; }
    [[nodiscard]] auto geometry() -> std::span<MeshInstanced> { return m_geometry; }

    // scene.cpp:157-216
    void addMeshInstance(std::optional<std::shared_ptr<Mesh const>> const& mesh, uint32_t animation, std::string const& name,
                         std::span<szg_transform const> transforms, bool castsShadow = true)
    {
        MeshInstanced instance{};
        instance.render = true;
        instance.castsShadow = castsShadow;
        instance.name = "meshInstanced_" + name;
        if (mesh.has_value())
        {
            instance.setMesh(mesh.value());
        }
        instance.animation = animation;
        instance.setInstances(transforms);
        m_geometry.push_back(std::move(instance));
    }

    // scene.cpp:218-236
    void addSpotlight(float const (&color)[3], szg_transform const& transform)
    {
        szg_spotlight_params params{};
        szg_spotlight_params_default(color, transform.translation, transform.eulerAnglesRadians, &params);
        SpotLightPacked packed{};
        szg_make_spot(&params, &packed);
        spotlights.push_back(packed);
        spotlightsRender = true;
    }

    // scene.cpp:238-333: the floor (scale 400 x 1 x 400, casts no shadow), one floating mesh 4 m up, and a green and a red
    // spot light of strength 30 and 60 degrees looking at it from 8 m above and 8 m to either side
    static auto defaultScene(std::optional<std::shared_ptr<Mesh const>> const& initialMesh) -> Scene
    {
        Scene scene{};
        szg_transform const floor[1] = {{{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {400.0f, 1.0f, 400.0f}}};
        scene.addMeshInstance(initialMesh, SZG_INSTANCE_ANIMATION_NONE, "Floor", floor, false);

        float const up[3] = {SZG_WORLD_UP_X, SZG_WORLD_UP_Y, SZG_WORLD_UP_Z};
        float const floating[3] = {4.0f * up[0], 4.0f * up[1], 4.0f * up[2]};
        szg_transform const floatingTransform[1] = {{{floating[0], floating[1], floating[2]}, {0.0f, 0.0f, 0.0f}, {1.0f, 1.0f, 1.0f}}};
        scene.addMeshInstance(initialMesh, SZG_INSTANCE_ANIMATION_NONE, "Floating", floatingTransform);

        // lightsOffset = 8 * (WORLD_FORWARD + WORLD_RIGHT), lightsHeight = 8 * WORLD_UP
        float const offset[3] = {8.0f * (SZG_WORLD_FORWARD_X + 1.0f), 8.0f * (SZG_WORLD_FORWARD_Y + 0.0f), 8.0f * (SZG_WORLD_FORWARD_Z + 0.0f)};
        float const height[3] = {8.0f * up[0], 8.0f * up[1], 8.0f * up[2]};
        float const one[3] = {1.0f, 1.0f, 1.0f};
        float const colors[2][4] = {{0.0f, 1.0f, 0.0f, 1.0f}, {1.0f, 0.0f, 0.0f, 1.0f}};
        for (int k = 0; k < 2; k++)
        {
            float const sign = k == 0 ? 1.0f : -1.0f;
            float const from[3] = {(floating[0] + height[0]) + sign * offset[0], (floating[1] + height[1]) + sign * offset[1],
                                   (floating[2] + height[2]) + sign * offset[2]};
            szg_transform lightTransform{};
            szg_transform_look_at(from, floating, one, &lightTransform);
            szg_spotlight_params params{};
            params.strength = 30.0f;
            params.falloffFactor = 1.0f;
            params.falloffDistance = 1.0f;
            params.verticalFOVDegrees = 60.0f;
            params.horizontalScale = 1.0f;
            params.near_plane = 0.1f;
            params.far_plane = 1000.0f;
            for (int c = 0; c < 4; c++)
            {
                params.color[c] = colors[k][c];
            }
            for (int c = 0; c < 3; c++)
            {
                params.eulerAngles[c] = lightTransform.eulerAnglesRadians[c];
                params.position[c] = lightTransform.translation[c];
            }
            SpotLightPacked packed{};
            szg_make_spot(&params, &packed);
            scene.spotlights.push_back(packed);
        }
        scene.spotlightsRender = true;
        return scene;
    }

    // scene.cpp:532-580: sun animation, then every mesh instance's animation; the instances' staged matrices are refreshed
    // here and copied by Renderer::recordDraw, as in the reference
    void tick(TickTiming lastFrame)
    {
        szg_scene_tick_sun(&sunAnimation, &atmosphere, lastFrame.deltaTimeSeconds);
        for (MeshInstanced& instance : m_geometry)
        {
            instance.tick(lastFrame.timeElapsedSeconds, lastFrame.deltaTimeSeconds);
        }
    }

  private:
    szg_aabb m_shadowBounds{};
    std::vector<MeshInstanced> m_geometry{};
};

// The part of Renderer (renderer.hpp / renderer.cpp:114-124, :278-443) that owns the staged parameter buffers and the two
// pipelines and records one frame of the deferred + atmosphere path.
struct Renderer
{
    static auto create(uint32_t capacityWidth, uint32_t capacityHeight, uint32_t shadowMapDimension = 8192) -> std::optional<Renderer>
    {
        Renderer r{};
        r.m_camerasBuffer = TStagedBuffer<CameraPacked>::allocate(1);         // renderer.hpp:120-122
        r.m_atmospheresBuffer = TStagedBuffer<AtmospherePacked>::allocate(1);
        r.m_directionalLightsBuffer = TStagedBuffer<DirectionalLightPacked>::allocate(2);
        r.m_deferredShadingPipeline = std::make_unique<DeferredShadingPipeline>(capacityWidth, capacityHeight, 16, 10, shadowMapDimension);
        r.m_skyViewComputePipeline = SkyViewComputePipeline::create();
        if (!r.m_camerasBuffer.valid() || !r.m_atmospheresBuffer.valid() || !r.m_directionalLightsBuffer.valid() ||
            !r.m_deferredShadingPipeline->valid() || r.m_skyViewComputePipeline == nullptr)
        {
            return std::nullopt;
        }
        return r;
    }

    void setRenderAtmosphere(bool render) { m_renderAtmosphere = render; }

    void recordDraw(hipStream_t cmd, Scene const& scene, SceneTexture& sceneTexture, szg_rect sceneSubregion)
    {
        if (sceneSubregion.width == 0 || sceneSubregion.height == 0)
        {
            return; // aspectRatio(extent) has no value, renderer.cpp:291-298
        }
        double const aspectRatio = static_cast<double>(sceneSubregion.width) / static_cast<double>(sceneSubregion.height);
        CameraPacked mainCamera{};
        szg_camera_to_device_equivalent(&scene.camera, static_cast<float>(aspectRatio), &mainCamera);
        m_camerasBuffer.clearStaged();
        m_camerasBuffer.push(mainCamera);
        m_camerasBuffer.recordCopyToDevice(cmd);

        szg_aabb const bounds = scene.shadowBounds();
        AtmospherePacked atmosphere{};
        DirectionalLightPacked lights[2];
        szg_atmosphere_baked(&scene.atmosphere, &bounds, &atmosphere, &lights[0], &lights[1]); // index 0 = sun, 1 = moon
        m_atmospheresBuffer.clearStaged();
        m_atmospheresBuffer.push(atmosphere);
        m_atmospheresBuffer.recordCopyToDevice(cmd);
        m_directionalLightsBuffer.clearStaged();
        m_directionalLightsBuffer.push(lights);
        m_directionalLightsBuffer.recordCopyToDevice(cmd);

        std::vector<szg_mesh_instanced> geometry;
        for (MeshInstanced const& instance : scene.geometry())
        {
            instance.recordCopyToDevice(cmd); // renderer.cpp:344-353
            geometry.push_back(instance.view());
        }
        std::vector<SpotLightPacked> const none{};
        m_deferredShadingPipeline->recordDrawCommands(cmd, sceneSubregion, sceneTexture, m_renderAtmosphere ? 1u : 0u,
                                                      m_directionalLightsBuffer, scene.spotlightsRender ? scene.spotlights : none, 0,
                                                      m_camerasBuffer, std::span<szg_mesh_instanced const>{geometry});
        if (m_renderAtmosphere)
        {
            m_skyViewComputePipeline->recordDrawCommands(cmd, sceneTexture, sceneSubregion, m_deferredShadingPipeline->gbuffer(),
                                                         m_deferredShadingPipeline->shadowMaps(), 0, m_atmospheresBuffer, 0,
                                                         m_camerasBuffer, 0, m_directionalLightsBuffer);
        }
    }

    [[nodiscard]] auto deferredShadingPipeline() -> DeferredShadingPipeline& { return *m_deferredShadingPipeline; }

  private:
    bool m_renderAtmosphere{true};
    TStagedBuffer<CameraPacked> m_camerasBuffer{};
    TStagedBuffer<AtmospherePacked> m_atmospheresBuffer{};
    TStagedBuffer<DirectionalLightPacked> m_directionalLightsBuffer{};
    std::unique_ptr<DeferredShadingPipeline> m_deferredShadingPipeline{};
    std::unique_ptr<SkyViewComputePipeline> m_skyViewComputePipeline{};
};
} // namespace szg
