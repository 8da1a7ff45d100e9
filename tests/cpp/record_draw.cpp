// record_draw.cpp — a Renderer::recordDraw-style caller (reference renderer.cpp:278-443)
// written against include/szg/pipelines.hpp. Renders one frame and writes the RGBA16 scene
// colour to argv[1]; tests/test_gpu_cpp_shim.py compares it with the Python path / oracle.
#include <cstdio>
#include <vector>

#include "szg/pipelines.hpp"

int main(int argc, char** argv)
{
    if (argc < 4)
    {
        std::fprintf(stderr, "usage: record_draw out.bin width height\n");
        return 2;
    }
    uint32_t const W = (uint32_t)std::atoi(argv[2]), H = (uint32_t)std::atoi(argv[3]);

    // scene -> packed blocks (renderer.cpp:302-342)
    szg_camera camera;
    szg_camera_default(&camera);
    szg_camera_packed cameraPacked;
    szg_camera_to_device_equivalent(&camera, (float)W / (float)H, &cameraPacked);

    szg_atmosphere atmosphere;
    szg_atmosphere_default_earth(&atmosphere);
    atmosphere.sunEulerAngles[0] = 3.14159265358979f + 35.0f * 3.14159265358979f / 180.0f;
    szg_aabb const bounds{{0.0f, -7.0f, 39.0f}, {64.0f, 8.0f, 46.0f}};
    szg_atmosphere_packed atmospherePacked;
    szg_directional_light_packed sun, moon;
    szg_atmosphere_baked(&atmosphere, &bounds, &atmospherePacked, &sun, &moon);

    float const red[3] = {1.0f, 0.0f, 0.0f};
    float const position[3] = {-20.0f, -28.0f, -20.0f};
    float const toTarget[3] = {20.0f, 20.0f, 20.0f};
    float eulers[3];
    szg_eulers_from_forward(toTarget, eulers);
    szg_spotlight_params params;
    szg_spotlight_params_default(red, position, eulers, &params);
    std::vector<szg::SpotLightPacked> spotlights(1);
    szg_make_spot(&params, &spotlights[0]);

    auto cameras = szg::TStagedBuffer<szg::CameraPacked>::allocate(1);
    auto atmospheres = szg::TStagedBuffer<szg::AtmospherePacked>::allocate(1);
    auto lights = szg::TStagedBuffer<szg::DirectionalLightPacked>::allocate(2);
    auto sceneTexture = szg::SceneTexture::create(W, H);
    szg::DeferredShadingPipeline deferred(W, H, 16, 10, 0);
    auto skyView = szg::SkyViewComputePipeline::create();
    if (!cameras.valid() || !atmospheres.valid() || !lights.valid() || !sceneTexture || !deferred.valid() || !skyView)
    {
        std::fprintf(stderr, "setup failed: %s\n", szg_last_error());
        return 1;
    }

    hipStream_t cmd = nullptr;
    (void)hipStreamCreate(&cmd);
    cameras.push(cameraPacked);
    cameras.recordCopyToDevice(cmd);
    atmospheres.push(atmospherePacked);
    atmospheres.recordCopyToDevice(cmd);
    lights.push(sun);  // index 0 = sun, 1 = moon (renderer.cpp:312-329)
    lights.push(moon);
    lights.recordCopyToDevice(cmd);

    szg_fill_box boxes[2] = {{{0.0f, -8.0f, 6.0f}, {5.0f, 5.0f, 5.0f}, 0.0f, 60.0f / 255.0f},
                             {{0.0f, -8.0f, -6.0f}, {5.0f, 5.0f, 5.0f}, 1.0f, 60.0f / 255.0f}}; // editor.cpp:510-539
    szg_fill_scene const geometry{-1.0f, 4000.0f, 4.0f, 60.0f / 255.0f, 2, 0, boxes};
    szg_rect const sceneSubregion{0, 0, W, H};

    // renderer.cpp:383-415
    deferred.recordDrawCommands(cmd, sceneSubregion, *sceneTexture, 1, lights, spotlights, 0, cameras, &geometry);
    skyView->recordDrawCommands(cmd, *sceneTexture, sceneSubregion, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0,
                                cameras, 0, lights);
    if (hipStreamSynchronize(cmd) != hipSuccess)
    {
        std::fprintf(stderr, "stream failed\n");
        return 1;
    }
    std::vector<uint16_t> host((size_t)W * H * 4);
    (void)hipMemcpy2D(host.data(), (size_t)W * 8, sceneTexture->color().data, sceneTexture->color().pitch_bytes, (size_t)W * 8, H,
                      hipMemcpyDeviceToHost);
    FILE* f = std::fopen(argv[1], "wb");
    std::fwrite(host.data(), 2, host.size(), f);
    std::fclose(f);
    std::printf("ok %ux%u\n", W, H);
    (void)hipStreamDestroy(cmd);
    return 0;
}
