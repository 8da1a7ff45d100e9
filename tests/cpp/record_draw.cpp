// record_draw.cpp — a Renderer::recordDraw-style caller (reference renderer.cpp:278-443)
// written against include/szg/pipelines.hpp. Renders one frame and writes the RGBA16 scene
// colour to argv[1]; tests/test_gpu_cpp_shim.py compares it with the Python path / oracle.
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "szg/assets.hpp"
#include "szg/pipelines.hpp"
#include "szg/scene.hpp"

namespace
{
// Device copy of a host array (the test's stand-in for the engine's GPU mesh / texture buffers).
template <typename T> T* upload(std::vector<T> const& host)
{
    T* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), host.size() * sizeof(T)) != hipSuccess ||
        hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
    {
        return nullptr;
    }
    return d;
}

// The editor's start-up scene (editor.cpp:500-545) from the asset library's built-in cube / plane and default
// material maps (assets.cpp:1294-1610), as szg_mesh_instanced records.
struct DefaultScene
{
    std::vector<szg_surface> cubeSurfaces, planeSurfaces;
    std::vector<szg_mesh_instanced> meshes;

    bool build()
    {
        auto vertex = [](float x, float y, float z, float u, float v, float nx, float ny, float nz) {
            return szg_vertex_packed{{x, y, z}, u, {nx, ny, nz}, v, {1.0f, 1.0f, 1.0f, 1.0f}};
        };
        std::vector<szg_vertex_packed> plane{vertex(-1, 0, 1, 0, 0, 0, -1, 0), vertex(1, 0, 1, 1, 0, 0, -1, 0),
                                             vertex(1, 0, -1, 1, 1, 0, -1, 0), vertex(-1, 0, -1, 0, 1, 0, -1, 0)};
        std::vector<uint32_t> planeIndices{0, 1, 3, 1, 2, 3};
        struct Face
        {
            float o[3], ex[3], ey[3], n[3];
        };
        Face const faces[6] = {{{-1, -1, 1}, {2, 0, 0}, {0, 0, -2}, {0, -1, 0}}, {{-1, 1, -1}, {2, 0, 0}, {0, 0, 2}, {0, 1, 0}},
                               {{1, -1, -1}, {0, 0, 2}, {0, 2, 0}, {1, 0, 0}},   {{-1, -1, 1}, {0, 0, -2}, {0, 2, 0}, {-1, 0, 0}},
                               {{-1, -1, -1}, {2, 0, 0}, {0, 2, 0}, {0, 0, -1}}, {{1, -1, 1}, {-2, 0, 0}, {0, 2, 0}, {0, 0, 1}}};
        std::vector<szg_vertex_packed> cube;
        std::vector<uint32_t> cubeIndices;
        for (Face const& f : faces)
        {
            uint32_t const base = (uint32_t)cube.size();
            float const uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
            for (auto const& c : uv)
            {
                cube.push_back(vertex(f.o[0] + c[0] * f.ex[0] + c[1] * f.ey[0], f.o[1] + c[0] * f.ex[1] + c[1] * f.ey[1],
                                      f.o[2] + c[0] * f.ex[2] + c[1] * f.ey[2], c[0], c[1], f.n[0], f.n[1], f.n[2]));
            }
            for (uint32_t i : {0u, 1u, 2u, 0u, 2u, 3u})
            {
                cubeIndices.push_back(base + i);
            }
        }
        // default maps: 64x64 RGBA8
        uint32_t const N = 64;
        std::vector<uint8_t> color(N * N * 4), normal(N * N * 4), orm(N * N * 4);
        for (uint32_t y = 0; y < N; y++)
        {
            for (uint32_t x = 0; x < N; x++)
            {
                uint8_t const grey = ((x / 4 + y / 4) % 2 == 0) ? 200 : 100;
                uint8_t* c = &color[(y * N + x) * 4];
                c[0] = c[1] = c[2] = grey;
                c[3] = 255;
                uint8_t* n = &normal[(y * N + x) * 4];
                n[0] = n[1] = 127;
                n[2] = 255;
                n[3] = 0;
                uint8_t* o = &orm[(y * N + x) * 4];
                o[0] = 255;
                o[1] = 60;
                o[2] = o[3] = 0;
            }
        }
        szg_material material{};
        material.color = szg_texture{upload(color), N, N, N * 4, 0};
        material.normal = szg_texture{upload(normal), N, N, N * 4, 0};
        material.orm = szg_texture{upload(orm), N, N, N * 4, 0};
        cubeSurfaces = {szg_surface{0, (uint32_t)cubeIndices.size(), material}};
        planeSurfaces = {szg_surface{0, (uint32_t)planeIndices.size(), material}};

        auto instance = [&](std::vector<szg_vertex_packed> const& v, std::vector<uint32_t> const& idx, std::vector<szg_surface> const& surfaces,
                            float tx, float ty, float tz, float sx, float sy, float sz) {
            float const t[3] = {tx, ty, tz}, e[3] = {0, 0, 0}, sc[3] = {sx, sy, sz};
            std::vector<szg_mat4> model(1), mit(1);
            szg_transform_matrix(t, e, sc, &model[0]);         // Transform::toMatrix, scene.cpp:207
            szg_mat4_inverse_transpose(&model[0], &mit[0]);    // scene.cpp:210
            szg_mesh_instanced m{};
            m.d_vertices = upload(v);
            m.vertex_count = (uint32_t)v.size();
            m.d_indices = upload(idx);
            m.index_count = (uint32_t)idx.size();
            m.surfaces = surfaces.data();
            m.surface_count = (uint32_t)surfaces.size();
            m.d_models = upload(model);
            m.d_model_inverse_transposes = upload(mit);
            m.instance_count = 1;
            m.render = 1;
            m.casts_shadow = 1;
            meshes.push_back(m);
            return m.d_vertices != nullptr && m.d_indices != nullptr && m.d_models != nullptr && m.d_model_inverse_transposes != nullptr;
        };
        return instance(cube, cubeIndices, cubeSurfaces, 0, -8, 6, 5, 5, 5) && instance(cube, cubeIndices, cubeSurfaces, 0, -8, -6, 5, 5, 5) &&
               instance(plane, planeIndices, planeSurfaces, 0, -1, 0, 20, 1, 20) && material.color.data != nullptr;
    }
};
} // namespace

// The engine's own frame, written the way Editor::run + Renderer::recordDraw write it (editor.cpp:500-545, renderer.cpp:278-443):
// asset library -> Scene::defaultScene -> ticks -> shadow bounds -> Renderer::recordDraw. Everything here is the
// header-only mirrors of include/szg/{assets,scene,pipelines}.hpp.
int engineFrame(const char* outPath, uint32_t W, uint32_t H, int ticks)
{
    auto library = szg::AssetLibrary::loadDefaultAssets();
    auto renderer = szg::Renderer::create(W, H, 512);
    auto sceneTexture = szg::SceneTexture::create(W, H);
    if (!library.has_value() || !renderer.has_value() || !sceneTexture)
    {
        std::fprintf(stderr, "setup failed: %s\n", szg_last_error());
        return 1;
    }
    szg::Scene scene = szg::Scene::defaultScene(library->defaultMesh(szg::AssetLibrary::DefaultMeshAssets::Cube));
    scene.sunAnimation.time = 0.6f; // afternoon
    scene.geometry()[1].animation = SZG_INSTANCE_ANIMATION_SPIN_ALONG_WORLD_UP;
    hipStream_t cmd = nullptr;
    (void)hipStreamCreate(&cmd);
    szg_rect const sceneSubregion{0, 0, W, H};
    double elapsed = 0.0;
    for (int k = 0; k < ticks; k++)
    {
        double const dt = 1.0 / 60.0;
        scene.tick(szg::TickTiming{elapsed, dt});
        elapsed += dt;
        scene.calculateShadowBounds();
        renderer->recordDraw(cmd, scene, *sceneTexture, sceneSubregion);
    }
    if (hipStreamSynchronize(cmd) != hipSuccess)
    {
        std::fprintf(stderr, "stream failed\n");
        return 1;
    }
    std::vector<uint16_t> host((size_t)W * H * 4);
    (void)hipMemcpy2D(host.data(), (size_t)W * 8, sceneTexture->color().data, sceneTexture->color().pitch_bytes, (size_t)W * 8, H,
                      hipMemcpyDeviceToHost);
    FILE* f = std::fopen(outPath, "wb");
    std::fwrite(host.data(), 2, host.size(), f);
    std::fclose(f);
    szg_aabb const b = scene.shadowBounds();
    std::printf("ok %ux%u sun %.9g bounds %.9g %.9g %.9g / %.9g %.9g %.9g\n", W, H, scene.atmosphere.sunEulerAngles[0], b.center[0], b.center[1],
                b.center[2], b.half_extent[0], b.half_extent[1], b.half_extent[2]);
    (void)hipStreamDestroy(cmd);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 4)
    {
        std::fprintf(stderr, "usage: record_draw out.bin width height [meshes | tiled [rank nranks idfile] | gltf <file> [loader flags] | scene [ticks]]\n");
        return 2;
    }
    uint32_t const W = (uint32_t)std::atoi(argv[2]), H = (uint32_t)std::atoi(argv[3]);
    if (argc > 4 && std::strcmp(argv[4], "scene") == 0)
    {
        return engineFrame(argv[1], W, H, argc > 5 ? std::atoi(argv[5]) : 3);
    }
    bool const gltf = argc > 5 && std::strcmp(argv[4], "gltf") == 0;
    bool const realMeshes = gltf || (argc > 4 && std::strcmp(argv[4], "meshes") == 0);
    // "tiled": the row-tiled multi-GPU frame as ONE rank of a world of one - the C-ABI's communicator (RCCL), the LUT row
    // slice + all-gather, the tile gather to the root and the compose kernel, all from C++ (no Python, no torch)
    bool const tiled = argc > 4 && std::strcmp(argv[4], "tiled") == 0;
    // "tiled rank nranks idfile": one of nranks processes; rank 0 creates the communicator id and leaves it in `idfile`, the
    // others pick it up there (a launcher's job in an engine), rank 0 gathers, composes and writes the frame
    uint32_t const tileRank = tiled && argc > 7 ? (uint32_t)std::atoi(argv[5]) : 0u;
    uint32_t const tileRanks = tiled && argc > 7 ? (uint32_t)std::atoi(argv[6]) : 1u;
    const char* const idFile = tiled && argc > 7 ? argv[7] : nullptr;

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
    szg::DeferredShadingPipeline deferred(W, H, 16, 10, realMeshes ? 512 : 0);
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
    DefaultScene scene;
    if (gltf)
    {
        // An engine-style caller: asset library -> mesh instances -> recordDrawCommands (editor.cpp:500-545 does this with
        // the library's cube and plane). Every mesh of the file gets two instances; the floor is the built-in plane.
        auto library = szg::AssetLibrary::loadDefaultAssets();
        if (!library.has_value())
        {
            std::fprintf(stderr, "default assets failed: %s\n", szg_last_error());
            return 1;
        }
        size_t const builtins = library->meshes().size();
        if (library->loadGLTFFromPath(argv[5], argc > 6 ? (uint32_t)std::atoi(argv[6]) : 0u) == 0)
        {
            std::fprintf(stderr, "no mesh loaded\n");
            return 1;
        }
        std::vector<szg::MeshInstanced> instances(library->meshes().size() - builtins + 1);
        for (size_t k = 0; k + 1 < instances.size(); k++)
        {
            float const fk = (float)k;
            szg_transform const two[2] = {{{-3.0f + 8.0f * fk, -6.0f, 2.0f}, {0.3f, 0.2f, 0.1f}, {4.0f, 4.0f, 4.0f}},
                                          {{6.0f, -4.0f, 8.0f + 4.0f * fk}, {0.0f, 1.0f, 0.0f}, {3.0f, 5.0f, 3.0f}}};
            instances[k].setMesh(library->meshes()[builtins + k]);
            instances[k].setInstances(two);
            instances[k].render = true;
            instances[k].name = library->meshes()[builtins + k]->name;
        }
        szg_transform const floor[1] = {{{0.0f, -1.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {20.0f, 1.0f, 20.0f}}};
        instances.back().setMesh(library->defaultMesh(szg::AssetLibrary::DefaultMeshAssets::Plane));
        instances.back().setInstances(floor);
        instances.back().render = true;
        std::vector<szg_mesh_instanced> views;
        for (szg::MeshInstanced& instance : instances)
        {
            instance.prepareForRendering(cmd);
            views.push_back(instance.view());
        }
        deferred.recordDrawCommands(cmd, sceneSubregion, *sceneTexture, 1, lights, spotlights, 0, cameras, std::span<szg_mesh_instanced const>{views});
        skyView->recordDrawCommands(cmd, *sceneTexture, sceneSubregion, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0,
                                    cameras, 0, lights);
        if (hipStreamSynchronize(cmd) != hipSuccess) // before the instances and the library (their device memory) go away
        {
            std::fprintf(stderr, "stream failed\n");
            return 1;
        }
        for (auto const& mesh : library->meshes())
        {
            std::printf("%s %u vertices %zu surfaces\n", mesh->name.c_str(), mesh->meshBuffers->vertexCount, mesh->surfaces.size());
        }
    }
    else if (realMeshes)
    {
        if (!scene.build())
        {
            std::fprintf(stderr, "mesh upload failed\n");
            return 1;
        }
        // the reference's own call shape: std::span<MeshInstanced const> sceneGeometry (shadow maps are rendered too)
        deferred.recordDrawCommands(cmd, sceneSubregion, *sceneTexture, 1, lights, spotlights, 0, cameras,
                                    std::span<szg_mesh_instanced const>{scene.meshes});
    }
    else if (tiled)
    {
        unsigned char id[SZG_ROWTILE_COMM_ID_BYTES];
        szg_rowtile_comm_t* comm = nullptr;
        if (tileRank == 0u)
        {
            if (szg_rowtile_comm_unique_id(id) != SZG_OK)
            {
                std::fprintf(stderr, "communicator id: %s\n", szg_last_error());
                return 1;
            }
            if (idFile != nullptr)
            {
                std::string const tmp = std::string(idFile) + ".tmp";
                FILE* f = std::fopen(tmp.c_str(), "wb");
                if (f == nullptr || std::fwrite(id, 1, sizeof id, f) != sizeof id || std::fclose(f) != 0 || std::rename(tmp.c_str(), idFile) != 0)
                {
                    std::fprintf(stderr, "cannot publish the communicator id\n");
                    return 1;
                }
            }
        }
        else
        {
            bool have = false;
            for (int spin = 0; spin < 6000 && !have; spin++)
            {
                FILE* f = std::fopen(idFile, "rb");
                if (f != nullptr)
                {
                    have = std::fread(id, 1, sizeof id, f) == sizeof id;
                    std::fclose(f);
                }
                if (!have)
                {
                    usleep(10000);
                }
            }
            if (!have)
            {
                std::fprintf(stderr, "rank %u: no communicator id in %s\n", tileRank, idFile);
                return 1;
            }
        }
        if (szg_rowtile_comm_create(&comm, (int)tileRank, (int)tileRanks, id, 0) != SZG_OK)
        {
            std::fprintf(stderr, "communicator: %s\n", szg_last_error());
            return 1;
        }
        szg_rowtile const tile{8u, tileRank, tileRanks, szg_rowtile_local_rows(H, 8u, tileRank, tileRanks)};
        // every rank sends the same number of bytes: the largest share of rows any rank holds
        uint32_t strideRows = 0;
        for (uint32_t r = 0; r < tileRanks; r++)
        {
            strideRows = std::max(strideRows, szg_rowtile_local_rows(H, 8u, r, tileRanks));
        }
        hipStream_t side = nullptr;
        hipEvent_t ev = nullptr;
        (void)hipStreamCreate(&side);
        (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        deferred.recordDrawCommands(cmd, sceneSubregion, *sceneTexture, 1, lights, spotlights, 0, cameras, &geometry, &tile);
        skyView->recordDrawCommandsTiled(cmd, side, ev, comm, *sceneTexture, sceneSubregion, deferred.gbuffer(), deferred.shadowMaps(), 0,
                                         atmospheres, 0, cameras, 0, lights, tile);
        // THE collective: every rank's packed rows to rank 0, then the row scatter into the frame
        size_t const tileBytes = (size_t)sceneTexture->color().pitch_bytes * strideRows;
        void *gathered = nullptr, *composed = nullptr;
        (void)hipMalloc(&gathered, tileBytes * tileRanks);
        (void)hipMalloc(&composed, (size_t)W * H * 8);
        szg_image const frame{composed, W, H, W * 8u, SZG_FORMAT_RGBA16_UNORM};
        int rc = szg_rowtile_gather(comm, cmd, sceneTexture->color().data, tileBytes, gathered, 0);
        if (rc == SZG_OK && tileRank == 0u)
        {
            rc = szg_compose_rowtiles(cmd, gathered, tileBytes, tileRanks, tile.block_rows, &frame, W, H);
        }
        if (rc != SZG_OK || hipStreamSynchronize(cmd) != hipSuccess)
        {
            std::fprintf(stderr, "tiled frame: %s\n", szg_last_error());
            return 1;
        }
        std::printf("ranks %d\n", szg_rowtile_comm_size(comm));
        if (tileRank != 0u)
        {
            szg_rowtile_comm_destroy(comm);
            return 0; // the frame leaves through rank 0
        }
        (void)hipMemcpy2DAsync(sceneTexture->color().data, sceneTexture->color().pitch_bytes, composed, (size_t)W * 8, (size_t)W * 8, H,
                               hipMemcpyDeviceToDevice, cmd);
        (void)hipStreamSynchronize(cmd);
        szg_rowtile_comm_destroy(comm);
        (void)hipFree(gathered);
        (void)hipFree(composed);
        (void)hipEventDestroy(ev);
        (void)hipStreamDestroy(side);
    }
    else
    {
        deferred.recordDrawCommands(cmd, sceneSubregion, *sceneTexture, 1, lights, spotlights, 0, cameras, &geometry);
    }
    if (!gltf && !tiled)
    {
        skyView->recordDrawCommands(cmd, *sceneTexture, sceneSubregion, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0,
                                    cameras, 0, lights);
    }
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
