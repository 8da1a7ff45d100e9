"""glTF 2.0 / GLB assets for the real-mesh raster passes: the host-side mirror of AssetLibrary::loadGLTFFromPath
(assets/assets.cpp:1192-1266) over include/szg/assets.h. The C++ library does the parsing and the conversion
(FLIP_Y, index rebasing, ORM channel overrides, default maps); this module only copies the results into numpy
arrays and builds `meshes.MeshInstanced` objects from them.
"""
import ctypes as C
import os

import numpy as np

from . import abi, meshes
from ._lib import lib


class AssetError(RuntimeError):
    pass


def _last_error():
    return lib().szg_last_error().decode(errors="replace")


def default_material_map(kind):
    """AssetLibrary's default colour / normal / ORM map (assets.cpp:1294-1398) as a uint8 [64, 64, 4] array."""
    n = abi.SZG_DEFAULT_MAP_DIMENSIONS
    out = np.empty((n, n, 4), np.uint8)
    if lib().szg_default_material_map(int(kind), out.ctypes.data_as(C.POINTER(C.c_uint8))) != abi.SZG_OK:
        raise AssetError(_last_error())
    return out


def decode_image_rgba(data):
    """detail_stbi::loadRGBA (assets.cpp:319-364): PNG bytes -> uint8 [h, w, 4] (JPEG is refused)."""
    data = bytes(data)
    w, h, ptr = abi.U32(), abi.U32(), C.POINTER(C.c_uint8)()
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    status = lib().szg_decode_image_rgba(C.cast(buf, C.c_void_p), len(data), C.byref(w), C.byref(h), C.byref(ptr))
    if status != abi.SZG_OK:
        raise AssetError(_last_error())
    try:
        return np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 4)).copy()
    finally:
        lib().szg_free_rgba(ptr)


def load_image_rgba(path):
    """The file part of AssetLibrary::loadTextureFromPath (assets.cpp:1131-1168): an image file -> uint8 [h, w, 4]."""
    w, h, ptr = abi.U32(), abi.U32(), C.POINTER(C.c_uint8)()
    status = lib().szg_load_image_file_rgba(os.fsencode(path), C.byref(w), C.byref(h), C.byref(ptr))
    if status != abi.SZG_OK:
        raise AssetError(f"{_last_error()} (status {status})")
    try:
        return np.frombuffer(C.string_at(ptr, h.value * w.value * 4), np.uint8).reshape(h.value, w.value, 4).copy()
    finally:
        lib().szg_free_rgba(ptr)


class GltfMesh:
    """One loaded Mesh (assets.hpp:30-44): packed vertices, rebased indices, surfaces, bounds."""

    def __init__(self, name, vertices, indices, surfaces, bounds, gltf_mesh_index):
        self.name, self.vertices, self.indices = name, vertices, indices
        self.surfaces = surfaces  # [(first_index, index_count, material index or -1)]
        self.bounds = bounds  # (center[3], half_extent[3])
        self.gltf_mesh_index = gltf_mesh_index


class GltfAsset:
    def __init__(self, meshes_, materials, warnings):
        self.meshes, self.materials, self.warnings = meshes_, materials, warnings
        self._default = None

    def default_material(self):
        if self._default is None:
            self._default = {
                "color": (default_material_map(abi.SZG_MAP_COLOR), False),
                "normal": (default_material_map(abi.SZG_MAP_NORMAL), False),
                "orm": (default_material_map(abi.SZG_MAP_ORM), False),
            }
        return self._default

    def material(self, index):
        """The three maps of a surface's material as meshes.MeshInstanced wants them; a map the file does not
        provide (or that failed to load) is the library default (assets.cpp:756 fallbackMaterialData)."""
        default = self.default_material()
        if index < 0:
            return default
        m = self.materials[index]
        return {key: (m[key] if m[key] is not None else default[key]) for key in ("color", "normal", "orm")}

    def instanced(self, mesh_index, models, **kwargs):
        """A meshes.MeshInstanced of loaded mesh `mesh_index` with the given model matrices."""
        m = self.meshes[mesh_index]
        surfaces = [(first, count, self.material(material)) for first, count, material in m.surfaces]
        return meshes.MeshInstanced(m.vertices, m.indices, surfaces, models, name=m.name, **kwargs)


def default_mesh(kind):
    """AssetLibrary::defaultMesh (assets.cpp:1668-1676): the built-in cube (abi.SZG_DEFAULT_MESH_CUBE) or plane as a GltfMesh."""
    m = abi.AssetMesh()
    if lib().szg_default_mesh(int(kind), C.byref(m)) != abi.SZG_OK:
        raise AssetError(_last_error())
    return _mesh(m)


def _mesh(m):
    vertices = np.zeros(m.vertex_count, abi.VERTEX_DTYPE)
    if m.vertex_count:
        C.memmove(vertices.ctypes.data, m.vertices, vertices.nbytes)
    indices = np.zeros(m.index_count, np.uint32)
    if m.index_count:
        C.memmove(indices.ctypes.data, m.indices, indices.nbytes)
    surfaces = [(m.surfaces[k].first_index, m.surfaces[k].index_count, m.surfaces[k].material) for k in range(m.surface_count)]
    bounds = (np.array(m.vertex_bounds.center, np.float32), np.array(m.vertex_bounds.half_extent, np.float32))
    return GltfMesh((m.name or b"").decode(errors="replace"), vertices, indices, surfaces, bounds, m.gltf_mesh_index)


def _texture(t):
    if not t.rgba:
        return None
    return (np.ctypeslib.as_array(t.rgba, shape=(t.height, t.width, 4)).copy(), bool(t.srgb)), t.name.decode(errors="replace")


def _collect(handle):
    L = lib()
    out_meshes, out_materials = [], []
    for i in range(L.szg_gltf_material_count(handle)):
        m = abi.AssetMaterial()
        if L.szg_gltf_material(handle, i, C.byref(m)) != abi.SZG_OK:
            raise AssetError(_last_error())
        record = {"name": (m.name or b"").decode(errors="replace"), "texture_names": {}}
        for key in ("color", "normal", "orm"):
            tex = _texture(getattr(m, key))
            record[key] = tex[0] if tex else None
            record["texture_names"][key] = tex[1] if tex else ""
        out_materials.append(record)
    for i in range(L.szg_gltf_mesh_count(handle)):
        m = abi.AssetMesh()
        if L.szg_gltf_mesh(handle, i, C.byref(m)) != abi.SZG_OK:
            raise AssetError(_last_error())
        out_meshes.append(_mesh(m))
    warnings = L.szg_gltf_warnings(handle).decode(errors="replace").splitlines()
    return GltfAsset(out_meshes, out_materials, warnings)


def load_gltf(path, flags=0):
    """AssetLibrary::loadGLTFFromPath: '.gltf' is read as JSON, anything else as GLB (assets.cpp:422-430)."""
    handle = C.c_void_p()
    status = lib().szg_gltf_load_file(os.fsencode(path), int(flags), C.byref(handle))
    if status != abi.SZG_OK:
        raise AssetError(f"{_last_error()} (status {status})")
    try:
        return _collect(handle)
    finally:
        lib().szg_gltf_destroy(handle)


def load_gltf_bytes(data, is_glb, asset_root=None, flags=0):
    data = bytes(data)
    handle = C.c_void_p()
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    root = os.fsencode(asset_root) if asset_root is not None else None
    status = lib().szg_gltf_load_memory(C.cast(buf, C.c_void_p), len(data), int(bool(is_glb)), root, int(flags), C.byref(handle))
    if status != abi.SZG_OK:
        raise AssetError(f"{_last_error()} (status {status})")
    try:
        return _collect(handle)
    finally:
        lib().szg_gltf_destroy(handle)
