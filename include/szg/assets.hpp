// szg/assets.hpp — header-only C++ mirror of the reference's asset and scene-geometry classes over szg/assets.h, so that
// a caller written against the engine loads a glTF file and hands the result to
// DeferredShadingPipeline::recordDrawCommands(..., std::span<MeshInstanced const>) the way Renderer::recordDraw does.
//
//   reference                                                   this header
//   ----------------------------------------------------------- -----------------------------------------------------
//   ImageView (renderer/imageview.hpp), R8G8B8A8 UNORM / SRGB   szg::ImageView (RGBA8 rows in device memory)
//   MaterialData        renderer/material.hpp                   szg::MaterialData {ORM, normal, color}
//   GeometrySurface     assets/assets.hpp:30-36                 szg::GeometrySurface
//   GPUMeshBuffers      renderer/buffers.hpp                    szg::GPUMeshBuffers (vertex + index arrays in device memory)
//   Mesh                assets/assets.hpp:38-44                 szg::Mesh
//   AssetLibrary        assets/assets.hpp:95-244                szg::AssetLibrary: loadDefaultAssets, loadGLTFFromPath,
//                                                               loadTextureFromPath, defaultMesh, deduplicated names
//   MeshInstanced       renderer/scene.hpp:109-147              szg::MeshInstanced: originals / transforms, models +
//                                                               modelInverseTransposes staged buffers, setMesh,
//                                                               material overrides
//   tickMeshInstance    renderer/scene.cpp:461-523              MeshInstanced::tick (+ recordCopyToDevice, renderer.cpp:344-353)
//
// Error behaviour follows the reference: a failed load logs and leaves the library unchanged; nothing throws.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <filesystem>
#include <memory>
#include <optional>
#include <span>
#include <string>
#include <unordered_map>
#include <vector>

#include "szg/assets.h"
#include "szg/pipelines.hpp"

namespace szg
{
namespace detail
{
inline auto uploadBytes(void const* host, size_t bytes) -> void*
{
    void* device = nullptr;
    if (bytes == 0 || hipMalloc(&device, bytes) != hipSuccess)
    {
        return nullptr;
    }
    if (hipMemcpy(device, host, bytes, hipMemcpyHostToDevice) != hipSuccess)
    {
        (void)hipFree(device);
        return nullptr;
    }
    return device;
}
} // namespace detail

// One RGBA8 texture in device memory.
struct ImageView
{
    ImageView() = default;
    ImageView(ImageView const&) = delete;
    auto operator=(ImageView const&) -> ImageView& = delete;
    ~ImageView() { (void)hipFree(data); }

    // detail::uploadImageToGPU + ImageView::allocate (assets.cpp:60-159, :257-313)
    static auto upload(uint8_t const* rgba, uint32_t width, uint32_t height, bool srgb, std::string name) -> std::shared_ptr<ImageView const>
    {
        auto view = std::make_shared<ImageView>();
        view->data = detail::uploadBytes(rgba, size_t{width} * height * 4);
        if (view->data == nullptr)
        {
            std::fprintf(stderr, "[szg] Failed to upload image to GPU.\n");
            return nullptr;
        }
        view->width = width;
        view->height = height;
        view->srgb = srgb;
        view->name = std::move(name);
        return view;
    }
    [[nodiscard]] auto texture() const -> szg_texture { return szg_texture{data, width, height, width * 4, srgb ? 1u : 0u}; }

    void* data{nullptr};
    uint32_t width{0}, height{0};
    bool srgb{false};
    std::string name{};
};

struct MaterialData
{
    std::shared_ptr<ImageView const> ORM{};
    std::shared_ptr<ImageView const> normal{};
    std::shared_ptr<ImageView const> color{};
};

struct GeometrySurface
{
    uint32_t firstIndex{0};
    uint32_t indexCount{0};
    MaterialData material{};
};

struct GPUMeshBuffers
{
    GPUMeshBuffers() = default;
    GPUMeshBuffers(GPUMeshBuffers const&) = delete;
    auto operator=(GPUMeshBuffers const&) -> GPUMeshBuffers& = delete;
    ~GPUMeshBuffers()
    {
        (void)hipFree(vertices);
        (void)hipFree(indices);
    }
    szg_vertex_packed* vertices{nullptr};
    uint32_t* indices{nullptr};
    uint32_t vertexCount{0}, indexCount{0};
};

struct Mesh
{
    std::string name{};
    std::vector<GeometrySurface> surfaces{};
    szg_aabb vertexBounds{};
    std::unique_ptr<GPUMeshBuffers> meshBuffers{};
};

class AssetLibrary
{
  public:
    enum class DefaultMeshAssets
    {
        Cube,
        Plane
    };

    // assets.cpp:1286-1613: the three default maps and the two built-in meshes
    static auto loadDefaultAssets() -> std::optional<AssetLibrary>
    {
        AssetLibrary library{};
        std::vector<uint8_t> map(size_t{SZG_DEFAULT_MAP_DIMENSIONS} * SZG_DEFAULT_MAP_DIMENSIONS * 4);
        std::shared_ptr<ImageView const>* const targets[3] = {&library.m_defaultColorMap, &library.m_defaultNormalMap, &library.m_defaultORMMap};
        char const* const names[3] = {"texture_defaultColor", "texture_defaultNormal", "texture_defaultORM"};
        for (int kind = SZG_MAP_COLOR; kind <= SZG_MAP_ORM; kind++)
        {
            if (szg_default_material_map(kind, map.data()) != SZG_OK)
            {
                return std::nullopt;
            }
            *targets[kind] = ImageView::upload(map.data(), SZG_DEFAULT_MAP_DIMENSIONS, SZG_DEFAULT_MAP_DIMENSIONS, false,
                                               library.deduplicateAssetName(names[kind]));
            if (*targets[kind] == nullptr)
            {
                return std::nullopt;
            }
            library.m_textures.push_back(*targets[kind]);
        }
        for (int kind : {SZG_DEFAULT_MESH_PLANE, SZG_DEFAULT_MESH_CUBE})
        {
            szg_asset_mesh m{};
            if (szg_default_mesh(kind, &m) != SZG_OK)
            {
                return std::nullopt;
            }
            std::shared_ptr<Mesh const> mesh = library.registerMesh(m, {});
            if (mesh == nullptr)
            {
                return std::nullopt;
            }
            (kind == SZG_DEFAULT_MESH_PLANE ? library.m_meshPlane : library.m_meshCube) = mesh;
        }
        return library;
    }

    // assets.cpp:1192-1266; returns the number of meshes registered (the reference logs it)
    auto loadGLTFFromPath(std::filesystem::path const& filePath, uint32_t flags = 0) -> size_t
    {
        std::fprintf(stderr, "[szg] Loading glTF from %s\n", filePath.string().c_str());
        szg_gltf* asset = nullptr;
        if (szg_gltf_load_file(filePath.string().c_str(), flags, &asset) != SZG_OK)
        {
            std::fprintf(stderr, "[szg] Failed to load glTF: %s\n", szg_last_error());
            return 0;
        }
        if (char const* warnings = szg_gltf_warnings(asset); warnings[0] != '\0')
        {
            std::fprintf(stderr, "[szg] glTF warnings:\n%s", warnings);
        }
        // uploadMaterialDataAsAssets: every map a material brings becomes a texture asset; the rest stay the defaults
        std::vector<MaterialData> materials;
        for (uint32_t i = 0; i < szg_gltf_material_count(asset); i++)
        {
            szg_asset_material m{};
            (void)szg_gltf_material(asset, i, &m);
            MaterialData data{m_defaultORMMap, m_defaultNormalMap, m_defaultColorMap};
            auto bring = [&](szg_asset_texture const& t, std::shared_ptr<ImageView const>& slot) {
                if (t.rgba == nullptr)
                {
                    return;
                }
                if (auto view = ImageView::upload(t.rgba, t.width, t.height, t.srgb != 0, deduplicateAssetName(t.name)); view != nullptr)
                {
                    m_textures.push_back(view);
                    slot = view;
                }
            };
            bring(m.orm, data.ORM);
            bring(m.color, data.color);
            bring(m.normal, data.normal);
            materials.push_back(std::move(data));
        }
        size_t loaded = 0;
        for (uint32_t i = 0; i < szg_gltf_mesh_count(asset); i++)
        {
            szg_asset_mesh m{};
            (void)szg_gltf_mesh(asset, i, &m);
            if (registerMesh(m, materials) != nullptr)
            {
                loaded++;
            }
        }
        szg_gltf_destroy(asset);
        std::fprintf(stderr, "[szg] Loaded %zu meshes from glTF\n", loaded);
        return loaded;
    }

    // assets.cpp:1131-1168 (fileFormat: sRGB or UNORM interpretation of the 8-bit texels)
    auto loadTextureFromPath(bool srgb, std::filesystem::path const& filePath) -> std::shared_ptr<ImageView const>
    {
        uint32_t width = 0, height = 0;
        uint8_t* rgba = nullptr;
        if (int const status = szg_load_image_file_rgba(filePath.string().c_str(), &width, &height, &rgba); status != SZG_OK)
        {
            std::fprintf(stderr, status == SZG_ERR_IO ? "[szg] Failed to open file for texture.\n"
                                                     : "[szg] Failed to convert file to 32 bit RGBA image.\n");
            return nullptr;
        }
        auto view = ImageView::upload(rgba, width, height, srgb, deduplicateAssetName("texture_" + filePath.stem().string()));
        szg_free_rgba(rgba);
        if (view != nullptr)
        {
            m_textures.push_back(view);
        }
        return view;
    }

    [[nodiscard]] auto defaultMesh(DefaultMeshAssets asset) const -> std::shared_ptr<Mesh const>
    {
        return asset == DefaultMeshAssets::Cube ? m_meshCube : m_meshPlane;
    }
    [[nodiscard]] auto meshes() const -> std::span<std::shared_ptr<Mesh const> const> { return m_meshes; }
    [[nodiscard]] auto textures() const -> std::span<std::shared_ptr<ImageView const> const> { return m_textures; }
    [[nodiscard]] auto defaultMaterial() const -> MaterialData { return MaterialData{m_defaultORMMap, m_defaultNormalMap, m_defaultColorMap}; }

  private:
    AssetLibrary() = default;

    // assets.cpp:1678-1692: mesh_Cube, mesh_Cube_2, mesh_Cube_3, ...
    auto deduplicateAssetName(std::string const& name) -> std::string
    {
        size_t const count = ++m_nameDuplicationCounters[name];
        return count == 1 ? name : name + "_" + std::to_string(count);
    }
    auto registerMesh(szg_asset_mesh const& m, std::span<MaterialData const> materials) -> std::shared_ptr<Mesh const>
    {
        auto mesh = std::make_shared<Mesh>();
        mesh->name = deduplicateAssetName(m.name);
        mesh->vertexBounds = m.vertex_bounds;
        for (uint32_t k = 0; k < m.surface_count; k++)
        {
            szg_asset_surface const& s = m.surfaces[k];
            MaterialData const material = (s.material >= 0 && static_cast<size_t>(s.material) < materials.size())
                                              ? materials[static_cast<size_t>(s.material)]
                                              : defaultMaterial();
            mesh->surfaces.push_back(GeometrySurface{s.first_index, s.index_count, material});
        }
        // detail::uploadMeshToGPU, assets.cpp:161-255
        mesh->meshBuffers = std::make_unique<GPUMeshBuffers>();
        mesh->meshBuffers->vertices = static_cast<szg_vertex_packed*>(detail::uploadBytes(m.vertices, size_t{m.vertex_count} * sizeof(szg_vertex_packed)));
        mesh->meshBuffers->indices = static_cast<uint32_t*>(detail::uploadBytes(m.indices, size_t{m.index_count} * sizeof(uint32_t)));
        mesh->meshBuffers->vertexCount = m.vertex_count;
        mesh->meshBuffers->indexCount = m.index_count;
        if (mesh->meshBuffers->vertices == nullptr || mesh->meshBuffers->indices == nullptr)
        {
            std::fprintf(stderr, "[szg] mesh upload failed for %s\n", mesh->name.c_str());
            return nullptr;
        }
        m_meshes.push_back(mesh);
        return mesh;
    }

    std::unordered_map<std::string, size_t> m_nameDuplicationCounters{};
    std::shared_ptr<ImageView const> m_defaultColorMap{}, m_defaultNormalMap{}, m_defaultORMMap{};
    std::vector<std::shared_ptr<ImageView const>> m_textures{};
    std::shared_ptr<Mesh const> m_meshPlane{}, m_meshCube{};
    std::vector<std::shared_ptr<Mesh const>> m_meshes{};
};

// renderer/scene.hpp:109-147
struct MeshInstanced
{
    bool render{false};
    bool castsShadow{true};
    std::string name{};
    uint32_t animation{SZG_INSTANCE_ANIMATION_NONE};

    std::vector<szg_transform> originals{};
    std::vector<szg_transform> transforms{};
    std::unique_ptr<TStagedBuffer<szg_mat4>> models{};
    std::unique_ptr<TStagedBuffer<szg_mat4>> modelInverseTransposes{};

    // The instance part of Scene::addMeshInstance (scene.cpp:181-213): originals = transforms = the given ones, both staged
    // buffers sized for them and filled with Transform::toMatrix() and its inverse transpose.
    void setInstances(std::span<szg_transform const> instances)
    {
        originals.assign(instances.begin(), instances.end());
        transforms = originals;
        models = std::make_unique<TStagedBuffer<szg_mat4>>(TStagedBuffer<szg_mat4>::allocate(instances.size()));
        modelInverseTransposes = std::make_unique<TStagedBuffer<szg_mat4>>(TStagedBuffer<szg_mat4>::allocate(instances.size()));
        for (szg_transform const& t : originals)
        {
            szg_mat4 model{}, inverseTranspose{};
            szg_transform_matrix(t.translation, t.eulerAnglesRadians, t.scale, &model);
            szg_mat4_inverse_transpose(&model, &inverseTranspose);
            models->push(model);
            modelInverseTransposes->push(inverseTranspose);
        }
    }
    void setMesh(std::shared_ptr<Mesh const> mesh)
    {
        m_mesh = std::move(mesh);
        m_surfaceMaterialOverrides.clear();
        m_view.clear();
    }
    [[nodiscard]] auto getMesh() const -> std::shared_ptr<Mesh const> { return m_mesh; }
    void setMaterialOverrides(size_t surface, MaterialData const& material)
    {
        if (m_mesh == nullptr || surface >= m_mesh->surfaces.size())
        {
            return;
        }
        m_surfaceMaterialOverrides.resize(m_mesh->surfaces.size());
        m_surfaceMaterialOverrides[surface] = material;
        m_view.clear();
    }

    // tickMeshInstance (scene.cpp:461-523): the animation advances `transforms`; the staged matrices are rebuilt
    void tick(double elapsedSeconds, double deltaSeconds)
    {
        if (models == nullptr || modelInverseTransposes == nullptr)
        {
            return;
        }
        std::vector<szg_mat4> m(transforms.size()), mit(transforms.size());
        szg_tick_mesh_instance(animation, originals.data(), transforms.data(), static_cast<uint32_t>(transforms.size()), elapsedSeconds,
                               deltaSeconds, m.data(), mit.data());
        models->clearStaged();
        models->push(m);
        modelInverseTransposes->clearStaged();
        modelInverseTransposes->push(mit);
    }
    // renderer.cpp:344-353
    void recordCopyToDevice(hipStream_t cmd) const
    {
        if (models != nullptr)
        {
            models->recordCopyToDevice(cmd);
        }
        if (modelInverseTransposes != nullptr)
        {
            modelInverseTransposes->recordCopyToDevice(cmd);
        }
    }
    void prepareForRendering(hipStream_t cmd, double elapsedSeconds = 0.0, double deltaSeconds = 0.0)
    {
        tick(elapsedSeconds, deltaSeconds);
        recordCopyToDevice(cmd);
    }

    // The record szg/raster.h consumes; valid while this object and its mesh live and until the next override.
    [[nodiscard]] auto view() const -> szg_mesh_instanced
    {
        szg_mesh_instanced out{};
        if (m_mesh == nullptr || m_mesh->meshBuffers == nullptr || models == nullptr || modelInverseTransposes == nullptr)
        {
            return out;
        }
        if (m_view.empty())
        {
            for (size_t k = 0; k < m_mesh->surfaces.size(); k++)
            {
                GeometrySurface const& s = m_mesh->surfaces[k];
                // the instance's overrides first, then the asset's materials (scene.hpp:141-143)
                MaterialData material = s.material;
                if (k < m_surfaceMaterialOverrides.size())
                {
                    MaterialData const& o = m_surfaceMaterialOverrides[k];
                    material.ORM = o.ORM != nullptr ? o.ORM : material.ORM;
                    material.normal = o.normal != nullptr ? o.normal : material.normal;
                    material.color = o.color != nullptr ? o.color : material.color;
                }
                szg_surface surface{};
                surface.first_index = s.firstIndex;
                surface.index_count = s.indexCount;
                surface.material.color = material.color != nullptr ? material.color->texture() : szg_texture{};
                surface.material.normal = material.normal != nullptr ? material.normal->texture() : szg_texture{};
                surface.material.orm = material.ORM != nullptr ? material.ORM->texture() : szg_texture{};
                m_view.push_back(surface);
            }
        }
        out.d_vertices = m_mesh->meshBuffers->vertices;
        out.vertex_count = m_mesh->meshBuffers->vertexCount;
        out.d_indices = m_mesh->meshBuffers->indices;
        out.index_count = m_mesh->meshBuffers->indexCount;
        out.surfaces = m_view.data();
        out.surface_count = static_cast<uint32_t>(m_view.size());
        out.d_models = models->deviceAddress();
        out.d_model_inverse_transposes = modelInverseTransposes->deviceAddress();
        out.instance_count = static_cast<uint32_t>(models->deviceSize());
        out.render = render ? 1u : 0u;
        out.casts_shadow = castsShadow ? 1u : 0u;
        return out;
    }

  private:
    std::shared_ptr<Mesh const> m_mesh{};
    std::vector<MaterialData> m_surfaceMaterialOverrides{};
    mutable std::vector<szg_surface> m_view{};
};
} // namespace szg
