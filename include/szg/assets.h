/*
 * szg/assets.h — the data format in front of the real-mesh raster passes ("glTF via assets/"):
 * glTF 2.0 / GLB files -> the vertex, index, surface and material-map arrays szg/raster.h consumes. CPU only.
 *
 * Mirrors AssetLibrary::loadGLTFFromPath (assets/assets.cpp:1192-1266) and its helpers:
 *
 *   loadGLTFAsset            assets.cpp:406-431   ".gltf" -> JSON, anything else -> GLB; GLB and external buffers loaded
 *   getTextureSources        assets.cpp:434-468   texture -> image indirection
 *   convertGLTFImage...      assets.cpp:470-575   image bytes -> 8-bit RGBA + per-channel overrides
 *   parseMaterialIndices     assets.cpp:579-645   baseColor / normal / occlusion / metallicRoughness texture indices
 *   uploadMaterialDataAs...  assets.cpp:735-879   ORM map = metallicRoughness image with R := 255, else occlusion image
 *                                                 with G := B := 0; colour maps are R8G8B8A8_SRGB, the others UNORM
 *                                                 (assets.cpp:706-715); a map that is absent or fails stays the default
 *   loadMeshes               assets.cpp:887-1092  one Mesh per glTF mesh, one GeometrySurface per primitive, indices
 *                                                 rebased onto the mesh's vertex array, defaults normal (1,0,0) uv 0
 *                                                 colour 1, then position.y and normal.y negated (FLIP_Y, :1046-1054),
 *                                                 bounds = AABB::create(min, max) of the positions
 *   default maps             assets.cpp:1294-1398 64x64 checkerboard colour, flat normal, (255, 60, 0) ORM
 *
 * Asset IO is OUT OF SCOPE of the hot path; this loader exists only so that the rasteriser's
 * tests can be fed meshes and maps from files, and it is frozen. The glTF parsing is third-party code in the reference
 * (fastgltf, a FetchContent download that is not in the checkout); image decoding is stb_image, vendored under
 * thirdparty/stb. glTF 2.0 accessors (all component types, `normalized`, byteStride, sparse), GLB containers and data:
 * URIs are read from the glTF 2.0 specification. Of the two image encodings glTF 2.0 allows only PNG is decoded
 * (RFC 1951 / the PNG specification: bit depths 1-16, all colour types, tRNS, Adam7; 16-bit samples keep their high
 * byte and gamma chunks are ignored, as a 4-channel 8-bit request to the reference's decoder answers; checksums not
 * verified). JPEG is NOT decoded.
 * An image that is not decoded fails like any undecodable image does in the reference (warning, default map kept).
 * The reference holds no usable asset for this path (assets/sphere.glb is a 132-byte LFS pointer): parity unpinned;
 * tests write glTF/GLB/PNG files with an independent Python encoder and compare array by array, and decode PNG files
 * written by Pillow (libpng): identical to Pillow's own decode (tests/golden/images).
 *
 * Where the reference would read out of bounds or trips an assert (accessor past its buffer, attribute longer than
 * POSITION, wrong accessor type) this loader skips the item with a warning instead; every such case is listed in
 * szg_gltf_warnings().
 */
#ifndef SZG_ASSETS_H
#define SZG_ASSETS_H

#include <stddef.h>

#include "szg/host.h"
#include "szg/raster.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SZG_ERR_IO (-6)
#define SZG_ERR_PARSE (-7)

/* The reference handles image sources held as bytes (data: URIs) or as files (assets.cpp:482-548) and warns
 * "Unsupported glTF image source found." for images stored in a bufferView — which is how GLB files embed them.
 * Default: the same (the map stays the default one). With this flag such images are decoded too. */
#define SZG_GLTF_DECODE_BUFFER_VIEW_IMAGES 1u

typedef struct szg_gltf szg_gltf; /* opaque; owns every array it hands out */

/* One material map in HOST memory, tightly packed RGBA8 rows top to bottom. rgba == NULL: the map is the
 * AssetLibrary default (szg_default_material_map). */
typedef struct szg_asset_texture
{
    const uint8_t* rgba;
    uint32_t width, height;
    uint32_t srgb;    /* 1 for colour maps (VK_FORMAT_R8G8B8A8_SRGB), 0 otherwise */
    const char* name; /* "texture_<image name | material_index_kind>" (assets.cpp:717-724, :310), "" for a default */
} szg_asset_texture;

typedef struct szg_asset_material
{
    const char* name;
    szg_asset_texture color, normal, orm;
} szg_asset_material;

typedef struct szg_asset_surface
{
    uint32_t first_index, index_count;
    int32_t material; /* index for szg_gltf_material, or -1: the default material (assets.cpp:935-962) */
} szg_asset_surface;

typedef struct szg_asset_mesh
{
    const char* name;                  /* "mesh_<glTF name>" (assets.cpp:1249) */
    const szg_vertex_packed* vertices; /* HOST memory */
    uint32_t vertex_count;
    const uint32_t* indices;
    uint32_t index_count;
    const szg_asset_surface* surfaces;
    uint32_t surface_count;
    szg_aabb vertex_bounds; /* Mesh::vertexBounds (assets.cpp:1061-1073) */
    int32_t gltf_mesh_index;
} szg_asset_mesh;

/* path ending in ".gltf": JSON, otherwise GLB (assets.cpp:422-430); external buffers and images are resolved
 * against the file's directory. SZG_OK and *out, or a negative code and szg_last_error(). */
int szg_gltf_load_file(const char* path, uint32_t flags, szg_gltf** out);
/* The same from memory; `asset_root` (may be NULL: no external files) is the directory relative URIs refer to. */
int szg_gltf_load_memory(const void* bytes, size_t size, int is_glb, const char* asset_root, uint32_t flags, szg_gltf** out);
void szg_gltf_destroy(szg_gltf* asset);

/* Meshes that loaded (glTF meshes without a usable primitive are dropped, assets.cpp:1056-1059), in glTF order. */
uint32_t szg_gltf_mesh_count(const szg_gltf* asset);
int szg_gltf_mesh(const szg_gltf* asset, uint32_t index, szg_asset_mesh* out);
/* One entry per glTF material, glTF indexing. */
uint32_t szg_gltf_material_count(const szg_gltf* asset);
int szg_gltf_material(const szg_gltf* asset, uint32_t index, szg_asset_material* out);
/* The SZG_WARNING lines the reference would log while loading, newline separated ("" if none). */
const char* szg_gltf_warnings(const szg_gltf* asset);

/* AssetLibrary's default maps (assets.cpp:1294-1398), 64 x 64 RGBA8 = 16384 bytes written to `rgba`. */
#define SZG_MAP_COLOR 0
#define SZG_MAP_NORMAL 1
#define SZG_MAP_ORM 2
#define SZG_DEFAULT_MAP_DIMENSIONS 64
int szg_default_material_map(int kind, uint8_t* rgba);

/* AssetLibrary's built-in meshes (assets.cpp:1400-1472 "mesh_Plane": a 2 x 2 quad in the xz plane facing -y; :1474-1610
 * "mesh_Cube": 2 x 2 x 2, four vertices per face, every face with the full uv square), one surface each with the default
 * material (-1). The cube's vertices are built without a colour (assets.cpp:1484-1507): value-initialised, (0, 0, 0, 0);
 * no shader reads the vertex colour. The arrays live in the library. */
#define SZG_DEFAULT_MESH_CUBE 0
#define SZG_DEFAULT_MESH_PLANE 1
int szg_default_mesh(int kind, szg_asset_mesh* out);

/* detail_stbi::loadRGBA (assets.cpp:319-364): PNG bytes -> RGBA8 (a JPEG stream is SZG_ERR_PARSE). The caller frees *out_rgba with szg_free_rgba. */
int szg_decode_image_rgba(const void* bytes, size_t size, uint32_t* out_width, uint32_t* out_height, uint8_t** out_rgba);
/* The file part of AssetLibrary::loadTextureFromPath (assets.cpp:1131-1168): read the file, decode it as above.
 * SZG_ERR_IO if it cannot be read ("Failed to open file for texture."), SZG_ERR_PARSE if it cannot be decoded. */
int szg_load_image_file_rgba(const char* path, uint32_t* out_width, uint32_t* out_height, uint8_t** out_rgba);
void szg_free_rgba(uint8_t* rgba);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* SZG_ASSETS_H */
