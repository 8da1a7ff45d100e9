"""ctypes mirrors of include/szg/abi.h and include/szg/host.h.

Field order and types follow the C headers one to one; sizes are asserted at
import time against the values the reference asserts (renderer/gputypes.hpp:36,
:72, :90, :115).
"""
import ctypes as C

import numpy as np

SZG_ABI_VERSION = 1

SZG_OK = 0
SZG_ERR_INVALID_ARGUMENT = -1
SZG_ERR_NO_DEVICE = -2
SZG_ERR_OUT_OF_MEMORY = -3
SZG_ERR_HIP = -4
SZG_ERR_CAPACITY = -5

SZG_AERIAL_W = SZG_AERIAL_H = SZG_AERIAL_D = 32
SZG_OETF_PURE_GAMMA = 0
SZG_OETF_SRGB = 1

SZG_ROWTILE_COMM_ID_BYTES = 256
SZG_LUT_TRANSMITTANCE = 1
SZG_LUT_SKYVIEW = 2
SZG_FORMAT_UNDEFINED = 0
SZG_FORMAT_RGBA16_SFLOAT = 1
SZG_FORMAT_RGBA32_SFLOAT = 2
SZG_FORMAT_RGBA16_UNORM = 3
SZG_FORMAT_D32_SFLOAT = 4

TEXEL_BYTES = {
    SZG_FORMAT_RGBA16_SFLOAT: 8,
    SZG_FORMAT_RGBA32_SFLOAT: 16,
    SZG_FORMAT_RGBA16_UNORM: 8,
    SZG_FORMAT_D32_SFLOAT: 4,
}


class Mat4(C.Structure):
    _fields_ = [("m", C.c_float * 16)]

    def to_numpy(self):
        """4x4 array indexed [row, col] (the struct itself is column-major)."""
        return np.array(self.m, dtype=np.float32).reshape(4, 4).T.copy()

    @classmethod
    def from_numpy(cls, a):
        out = cls()
        flat = np.asarray(a, dtype=np.float32).T.reshape(16)
        for i in range(16):
            out.m[i] = float(flat[i])
        return out


class CameraPacked(C.Structure):
    _fields_ = [
        ("projection", Mat4),
        ("inverseProjection", Mat4),
        ("view", Mat4),
        ("viewInverseTranspose", Mat4),
        ("rotation", Mat4),
        ("projViewInverse", Mat4),
        ("forwardWorld", C.c_float * 4),
        ("position", C.c_float * 4),
    ]


class AtmospherePacked(C.Structure):
    _fields_ = [
        ("scatteringRayleighPerMm", C.c_float * 3),
        ("densityScaleRayleighMm", C.c_float),
        ("absorptionRayleighPerMm", C.c_float * 3),
        ("planetRadiusMm", C.c_float),
        ("scatteringMiePerMm", C.c_float * 3),
        ("densityScaleMieMm", C.c_float),
        ("absorptionMiePerMm", C.c_float * 3),
        ("atmosphereRadiusMm", C.c_float),
        ("incidentDirectionSun", C.c_float * 3),
        ("padding0", C.c_uint32),
        ("scatteringOzonePerMm", C.c_float * 3),
        ("padding1", C.c_uint32),
        ("absorptionOzonePerMm", C.c_float * 3),
        ("padding2", C.c_uint32),
        ("sunIntensitySpectrum", C.c_float * 3),
        ("sunAngularRadius", C.c_float),
    ]


class DirectionalLightPacked(C.Structure):
    _fields_ = [
        ("color", C.c_float * 4),
        ("forward", C.c_float * 4),
        ("projection", Mat4),
        ("view", Mat4),
        ("strength", C.c_float),
        ("padding0", C.c_uint32 * 3),
    ]


class SpotLightPacked(C.Structure):
    _fields_ = [
        ("color", C.c_float * 4),
        ("forward", C.c_float * 4),
        ("projection", Mat4),
        ("view", Mat4),
        ("position", C.c_float * 4),
        ("strength", C.c_float),
        ("falloffFactor", C.c_float),
        ("falloffDistance", C.c_float),
        ("padding0", C.c_uint32),
    ]


class Image(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("pitch_bytes", C.c_uint32),
        ("format", C.c_uint32),
    ]


class Rect(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32)]


class GBuffer(C.Structure):
    _fields_ = [
        ("diffuse", Image),
        ("specular", Image),
        ("normal", Image),
        ("worldPosition", Image),
        ("occlusionRoughnessMetallic", Image),
    ]


class SceneTexture(C.Structure):
    _fields_ = [("color", Image), ("depth", Image), ("debug_color", Image)]


class ShadowMaps(C.Structure):
    _fields_ = [("count", C.c_uint32), ("padding", C.c_uint32), ("maps", C.POINTER(Image))]


class RowTile(C.Structure):
    _fields_ = [
        ("block_rows", C.c_uint32),
        ("rank", C.c_uint32),
        ("nranks", C.c_uint32),
        ("local_rows", C.c_uint32),
    ]


class FillBox(C.Structure):
    _fields_ = [
        ("center", C.c_float * 3),
        ("half_extent", C.c_float * 3),
        ("metallic", C.c_float),
        ("roughness", C.c_float),
    ]


class FillScene(C.Structure):
    _fields_ = [
        ("ground_y", C.c_float),
        ("ground_half_extent", C.c_float),
        ("checker_cell", C.c_float),
        ("ground_roughness", C.c_float),
        ("box_count", C.c_uint32),
        ("padding", C.c_uint32),
        ("boxes", C.POINTER(FillBox)),
    ]


class SkyviewDesc(C.Structure):
    _fields_ = [
        ("transmittance_width", C.c_uint32),
        ("transmittance_height", C.c_uint32),
        ("skyview_width", C.c_uint32),
        ("skyview_height", C.c_uint32),
        ("flags", C.c_uint32),
        ("padding", C.c_uint32),
    ]


class DeferredDesc(C.Structure):
    _fields_ = [
        ("capacity_width", C.c_uint32),
        ("capacity_height", C.c_uint32),
        ("max_spot_lights", C.c_uint32),
        ("max_shadow_maps", C.c_uint32),
        ("shadow_map_dim", C.c_uint32),
        ("padding", C.c_uint32),
    ]


class DeferredConfiguration(C.Structure):
    _fields_ = [("depthBiasConstant", C.c_float), ("depthBiasSlope", C.c_float)]


# ---- szg/raster.h ---------------------------------------------------------
class VertexPacked(C.Structure):
    """renderer/gputypes.hpp:117-126"""

    _fields_ = [
        ("position", C.c_float * 3),
        ("uv_x", C.c_float),
        ("normal", C.c_float * 3),
        ("uv_y", C.c_float),
        ("color", C.c_float * 4),
    ]


class Texture(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("pitch_bytes", C.c_uint32),
        ("srgb", C.c_uint32),
    ]


class Material(C.Structure):
    _fields_ = [("color", Texture), ("normal", Texture), ("orm", Texture)]


class Surface(C.Structure):
    _fields_ = [("first_index", C.c_uint32), ("index_count", C.c_uint32), ("material", Material)]


class MeshInstanced(C.Structure):
    _fields_ = [
        ("d_vertices", C.c_void_p),
        ("vertex_count", C.c_uint32),
        ("index_count", C.c_uint32),
        ("d_indices", C.c_void_p),
        ("surfaces", C.POINTER(Surface)),
        ("surface_count", C.c_uint32),
        ("instance_count", C.c_uint32),
        ("d_models", C.c_void_p),
        ("d_model_inverse_transposes", C.c_void_p),
        ("render", C.c_uint32),
        ("casts_shadow", C.c_uint32),
    ]


assert C.sizeof(VertexPacked) == 48

VERTEX_DTYPE = np.dtype(
    [("position", np.float32, 3), ("uv_x", np.float32), ("normal", np.float32, 3), ("uv_y", np.float32), ("color", np.float32, 4)]
)
assert VERTEX_DTYPE.itemsize == 48


# ---- szg/host.h -----------------------------------------------------------
class AABB(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("half_extent", C.c_float * 3)]


class Atmosphere(C.Structure):
    _fields_ = [
        ("sunEulerAngles", C.c_float * 3),
        ("planetRadiusMegameters", C.c_float),
        ("atmosphereRadiusMegameters", C.c_float),
        ("groundColor", C.c_float * 3),
        ("scatteringRayleighPerMegameter", C.c_float * 3),
        ("absorptionRayleighPerMegameter", C.c_float * 3),
        ("altitudeDecayRayleighMegameters", C.c_float),
        ("scatteringMiePerMegameter", C.c_float * 3),
        ("absorptionMiePerMegameter", C.c_float * 3),
        ("altitudeDecayMieMegameters", C.c_float),
        ("scatteringOzonePerMegameter", C.c_float * 3),
        ("absorptionOzonePerMegameter", C.c_float * 3),
        ("sunIntensitySpectrum", C.c_float * 3),
        ("sunAngularRadius", C.c_float),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("cameraPosition", C.c_float * 3),
        ("eulerAngles", C.c_float * 3),
        ("fovDegrees", C.c_float),
        ("near_plane", C.c_float),
        ("far_plane", C.c_float),
        ("orthographic", C.c_uint32),
    ]


class SpotlightParams(C.Structure):
    _fields_ = [
        ("color", C.c_float * 4),
        ("strength", C.c_float),
        ("falloffFactor", C.c_float),
        ("falloffDistance", C.c_float),
        ("verticalFOVDegrees", C.c_float),
        ("horizontalScale", C.c_float),
        ("eulerAngles", C.c_float * 3),
        ("position", C.c_float * 3),
        ("near_plane", C.c_float),
        ("far_plane", C.c_float),
    ]


class Transform(C.Structure):
    _fields_ = [("translation", C.c_float * 3), ("eulerAnglesRadians", C.c_float * 3), ("scale", C.c_float * 3)]


class ShadowCaster(C.Structure):
    _fields_ = [
        ("vertex_bounds", AABB),
        ("transforms", C.POINTER(Transform)),
        ("transform_count", C.c_uint32),
        ("render", C.c_uint32),
        ("casts_shadow", C.c_uint32),
        ("padding", C.c_uint32),
    ]


SZG_INSTANCE_ANIMATION_NONE = 0
SZG_INSTANCE_ANIMATION_DIAGONAL_WAVE = 1
SZG_INSTANCE_ANIMATION_SPIN_ALONG_WORLD_UP = 2


class SunAnimation(C.Structure):
    _fields_ = [("frozen", C.c_uint32), ("time", C.c_float), ("speed", C.c_float), ("skipNight", C.c_uint32)]


assert C.sizeof(CameraPacked) == 416
assert C.sizeof(AtmospherePacked) == 128
assert C.sizeof(DirectionalLightPacked) == 176
assert C.sizeof(SpotLightPacked) == 192
assert C.sizeof(Image) == 24
assert C.sizeof(RowTile) == 16


def f3(x, y, z):
    return (C.c_float * 3)(float(x), float(y), float(z))


def f4(x, y, z, w):
    return (C.c_float * 4)(float(x), float(y), float(z), float(w))


# Every symbol include/szg/abi.h and include/szg/host.h declare, with ctypes
# signatures. tests/test_abi.py checks the built library exports each of them.
VP = C.c_void_p
U32 = C.c_uint32
P = C.POINTER

ABI_FUNCTIONS = {
    "szg_abi_version": (C.c_int, []),
    "szg_build_id": (C.c_char_p, []),
    "szg_last_error": (C.c_char_p, []),
    "szg_device_count": (C.c_int, []),
    "szg_skyview_create": (C.c_int, [P(VP), P(SkyviewDesc), C.c_int]),
    "szg_skyview_destroy": (None, [VP]),
    "szg_skyview_record_draw_commands": (
        C.c_int,
        [VP, VP, P(SceneTexture), Rect, P(RowTile), P(GBuffer), P(ShadowMaps), U32, VP, U32, VP, U32, VP],
    ),
    "szg_skyview_record_transmittance": (C.c_int, [VP, VP, U32, VP]),
    "szg_skyview_record_skyview_lut": (C.c_int, [VP, VP, U32, VP, U32, VP]),
    "szg_skyview_record_skyview_lut_rows": (C.c_int, [VP, VP, U32, VP, U32, VP, U32, U32]),
    "szg_skyview_record_composite": (
        C.c_int,
        [VP, VP, P(SceneTexture), Rect, P(RowTile), P(GBuffer), P(ShadowMaps), U32, VP, U32, VP, U32, VP],
    ),
    "szg_skyview_record_multiscatter_lut": (C.c_int, [VP, VP, U32, VP]),
    "szg_skyview_multiscatter_lut": (C.c_int, [VP, P(Image)]),
    "szg_skyview_record_aerial_lut": (C.c_int, [VP, VP, U32, VP, U32, VP, C.c_float]),
    "szg_skyview_aerial_lut": (C.c_int, [VP, P(Image), P(Image)]),
    "szg_skyview_record_composite_fast": (
        C.c_int,
        [VP, VP, P(SceneTexture), Rect, P(RowTile), P(GBuffer), P(ShadowMaps), U32, VP, U32, VP, U32, VP],
    ),
    "szg_skyview_transmittance_lut": (C.c_int, [VP, P(Image)]),
    "szg_skyview_skyview_lut": (C.c_int, [VP, P(Image)]),
    "szg_skyview_invalidate_luts": (C.c_int, [VP, C.c_uint32]),
    "szg_skyview_lut_row_slice": (C.c_int, [VP, C.c_uint32, C.c_uint32, P(C.c_uint32), P(C.c_uint32)]),
    "szg_skyview_allgather_lut_rows": (C.c_int, [VP, VP, VP]),
    "szg_rowtile_comm_unique_id": (C.c_int, [VP]),
    "szg_rowtile_comm_create": (C.c_int, [P(VP), C.c_int, C.c_int, VP, C.c_int]),
    "szg_rowtile_comm_create_deadline": (C.c_int, [P(VP), C.c_int, C.c_int, VP, C.c_int, C.c_int]),
    "szg_rowtile_comm_backend": (C.c_char_p, []),
    "szg_rowtile_comm_destroy": (None, [VP]),
    "szg_rowtile_comm_rank": (C.c_int, [VP]),
    "szg_rowtile_comm_size": (C.c_int, [VP]),
    "szg_rowtile_gather": (C.c_int, [VP, VP, VP, C.c_size_t, VP, C.c_int]),
    "szg_rowtile_allgather": (C.c_int, [VP, VP, VP, C.c_size_t]),
    "szg_skyview_set_lut_reuse": (C.c_int, [VP, C.c_int]),
    "szg_deferred_create": (C.c_int, [P(VP), P(DeferredDesc), C.c_int]),
    "szg_deferred_destroy": (None, [VP]),
    "szg_deferred_record_draw_commands": (
        C.c_int,
        [VP, VP, Rect, P(RowTile), P(SceneTexture), U32, VP, U32, P(SpotLightPacked), U32, U32, VP, P(FillScene)],
    ),
    "szg_deferred_record_gbuffer_fill": (C.c_int, [VP, VP, Rect, P(RowTile), P(SceneTexture), U32, VP, P(FillScene)]),
    "szg_deferred_record_lights": (
        C.c_int,
        [VP, VP, Rect, P(RowTile), P(SceneTexture), U32, VP, U32, P(SpotLightPacked), U32, U32, VP],
    ),
    "szg_deferred_record_shadow_maps": (C.c_int, [VP, VP, VP, U32, P(SpotLightPacked), U32, P(FillScene)]),
    "szg_deferred_gbuffer": (P(GBuffer), [VP]),
    "szg_deferred_shadow_maps": (P(ShadowMaps), [VP]),
    "szg_deferred_set_shadow_map": (C.c_int, [VP, U32, P(Image)]),
    "szg_deferred_get_configuration": (C.c_int, [VP, P(DeferredConfiguration)]),
    "szg_deferred_set_configuration": (C.c_int, [VP, P(DeferredConfiguration)]),
    "szg_record_oetf": (C.c_int, [VP, P(Image), U32, U32, U32]),
    "szg_compose_rowtiles": (C.c_int, [VP, VP, C.c_size_t, U32, U32, P(Image), U32, U32]),
    "szg_rowtile_local_rows": (U32, [U32, U32, U32, U32]),
}

# include/szg/raster.h
RASTER_FUNCTIONS = {
    "szg_deferred_record_gbuffer_raster": (C.c_int, [VP, VP, Rect, P(RowTile), P(SceneTexture), U32, VP, P(MeshInstanced), U32]),
    "szg_deferred_record_shadow_raster": (C.c_int, [VP, VP, VP, U32, P(SpotLightPacked), U32, P(MeshInstanced), U32]),
    "szg_deferred_record_draw_commands_meshes": (
        C.c_int,
        [VP, VP, Rect, P(RowTile), P(SceneTexture), U32, VP, U32, P(SpotLightPacked), U32, U32, VP, P(MeshInstanced), U32],
    ),
}

# include/szg/assets.h
SZG_ERR_IO = -6
SZG_ERR_PARSE = -7
SZG_ERR_TIMEOUT = -8
SZG_GLTF_DECODE_BUFFER_VIEW_IMAGES = 1
SZG_MAP_COLOR, SZG_MAP_NORMAL, SZG_MAP_ORM = 0, 1, 2
SZG_DEFAULT_MAP_DIMENSIONS = 64
SZG_DEFAULT_MESH_CUBE, SZG_DEFAULT_MESH_PLANE = 0, 1


class AssetTexture(C.Structure):
    _fields_ = [("rgba", P(C.c_uint8)), ("width", U32), ("height", U32), ("srgb", U32), ("name", C.c_char_p)]


class AssetMaterial(C.Structure):
    _fields_ = [("name", C.c_char_p), ("color", AssetTexture), ("normal", AssetTexture), ("orm", AssetTexture)]


class AssetSurface(C.Structure):
    _fields_ = [("first_index", U32), ("index_count", U32), ("material", C.c_int32)]


class AssetMesh(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("vertices", VP),
        ("vertex_count", U32),
        ("indices", P(U32)),
        ("index_count", U32),
        ("surfaces", P(AssetSurface)),
        ("surface_count", U32),
        ("vertex_bounds", AABB),
        ("gltf_mesh_index", C.c_int32),
    ]


ASSET_FUNCTIONS = {
    "szg_gltf_load_file": (C.c_int, [C.c_char_p, U32, P(VP)]),
    "szg_gltf_load_memory": (C.c_int, [VP, C.c_size_t, C.c_int, C.c_char_p, U32, P(VP)]),
    "szg_gltf_destroy": (None, [VP]),
    "szg_gltf_mesh_count": (U32, [VP]),
    "szg_gltf_mesh": (C.c_int, [VP, U32, P(AssetMesh)]),
    "szg_gltf_material_count": (U32, [VP]),
    "szg_gltf_material": (C.c_int, [VP, U32, P(AssetMaterial)]),
    "szg_gltf_warnings": (C.c_char_p, [VP]),
    "szg_default_material_map": (C.c_int, [C.c_int, P(C.c_uint8)]),
    "szg_default_mesh": (C.c_int, [C.c_int, P(AssetMesh)]),
    "szg_decode_image_rgba": (C.c_int, [VP, C.c_size_t, P(U32), P(U32), P(P(C.c_uint8))]),
    "szg_load_image_file_rgba": (C.c_int, [C.c_char_p, P(U32), P(U32), P(P(C.c_uint8))]),
    "szg_free_rgba": (None, [P(C.c_uint8)]),
}

HOST_FUNCTIONS = {
    "szg_forward_from_eulers": (None, [P(C.c_float), P(C.c_float)]),
    "szg_eulers_from_forward": (None, [P(C.c_float), P(C.c_float)]),
    "szg_projection_vk": (None, [C.c_float, C.c_float, C.c_float, C.c_float, P(Mat4)]),
    "szg_projection_ortho_vk": (None, [P(C.c_float), P(C.c_float), P(Mat4)]),
    "szg_transform_vk": (None, [P(C.c_float), P(C.c_float), P(Mat4)]),
    "szg_view_vk": (None, [P(C.c_float), P(C.c_float), P(Mat4)]),
    "szg_transform_matrix": (None, [P(C.c_float), P(C.c_float), P(C.c_float), P(Mat4)]),
    "szg_aabb_create": (None, [P(C.c_float), P(C.c_float), P(AABB)]),
    "szg_transform_look_at": (None, [P(C.c_float), P(C.c_float), P(C.c_float), P(Transform)]),
    "szg_calculate_shadow_bounds": (C.c_int, [P(ShadowCaster), U32, P(AABB)]),
    "szg_tick_mesh_instance": (None, [U32, P(Transform), P(Transform), U32, C.c_double, C.c_double, P(Mat4), P(Mat4)]),
    "szg_projection_ortho_aabb_vk": (None, [P(Mat4), P(AABB), P(Mat4)]),
    "szg_mat4_inverse": (None, [P(Mat4), P(Mat4)]),
    "szg_mat4_inverse_transpose": (None, [P(Mat4), P(Mat4)]),
    "szg_mat4_mul": (None, [P(Mat4), P(Mat4), P(Mat4)]),
    "szg_atmosphere_default_earth": (None, [P(Atmosphere)]),
    "szg_camera_default": (None, [P(Camera)]),
    "szg_sun_animation_default": (None, [P(SunAnimation)]),
    "szg_atmosphere_direction_to_sun": (None, [P(Atmosphere), P(C.c_float)]),
    "szg_atmosphere_to_device_equivalent": (None, [P(Atmosphere), P(AtmospherePacked)]),
    "szg_atmosphere_baked": (
        None,
        [P(Atmosphere), P(AABB), P(AtmospherePacked), P(DirectionalLightPacked), P(DirectionalLightPacked)],
    ),
    "szg_camera_to_device_equivalent": (None, [P(Camera), C.c_float, P(CameraPacked)]),
    "szg_make_directional": (None, [P(C.c_float), C.c_float, P(C.c_float), P(AABB), P(DirectionalLightPacked)]),
    "szg_make_spot": (None, [P(SpotlightParams), P(SpotLightPacked)]),
    "szg_spotlight_params_default": (None, [P(C.c_float), P(C.c_float), P(C.c_float), P(SpotlightParams)]),
    "szg_scene_tick_sun": (None, [P(SunAnimation), P(Atmosphere), C.c_double]),
}


def bind(lib, table):
    """Attach restype/argtypes from `table` to `lib`; raises AttributeError on a missing export."""
    for name, (restype, argtypes) in table.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib
