"""Scene geometry for the real-mesh raster passes (include/szg/raster.h): the host-side mirror of
MeshInstanced / Mesh / GeometrySurface / MaterialData (renderer/scene.hpp:109-147, assets/assets.hpp:30-44) plus the
reference's built-in assets, regenerated here from their definitions:

  plane_mesh, cube_mesh   AssetLibrary default meshes (assets.cpp:1400-1472, :1474-1610): positions, uvs, normals, indices
  default_material        the three 64x64 default maps (assets.cpp:1294-1398): grey checkerboard of 4-texel cells,
                          flat normal (127,127,255), non-occluded dielectric ORM (255,60,0)
  reference_default_scene the scene the editor starts with (editor.cpp:500-545): two cubes of scale 5 and a 20x20 floor

torch only owns the device copies; the same object also yields host-pointer structs for the CPU oracle.
"""
import ctypes as C

import numpy as np

from . import abi
from ._lib import lib

DEFAULT_IMAGE_DIMENSIONS = 64  # assets.cpp:1294


def constant_texture(rgba, size=DEFAULT_IMAGE_DIMENSIONS):
    t = np.empty((size, size, 4), np.uint8)
    t[...] = np.asarray(rgba, np.uint8)
    return t


def default_color_map():
    """assets.cpp:1330-1356: light (200) / dark (100) grey squares of 4x4 texels, alpha 255."""
    n = DEFAULT_IMAGE_DIMENSIONS
    y, x = np.mgrid[0:n, 0:n]
    light = ((x // 4) + (y // 4)) % 2 == 0
    t = np.empty((n, n, 4), np.uint8)
    t[..., :3] = np.where(light, 200, 100)[..., None]
    t[..., 3] = 255
    return t


def default_material():
    return {
        "color": (default_color_map(), False),
        "normal": (constant_texture((127, 127, 255, 0)), False),  # assets.cpp:1375-1379
        "orm": (constant_texture((255, 60, 0, 0)), False),  # assets.cpp:1309-1313
    }


def _vertices(rows, color=(1.0, 1.0, 1.0, 1.0)):
    v = np.zeros(len(rows), abi.VERTEX_DTYPE)
    for i, (pos, uv, nrm) in enumerate(rows):
        v[i]["position"] = pos
        v[i]["uv_x"], v[i]["uv_y"] = uv
        v[i]["normal"] = nrm
        v[i]["color"] = color
    return v


def plane_mesh():
    """A 2x2 quad in the xz plane facing -y (up), two triangles (assets.cpp:1400-1432)."""
    up = (0.0, -1.0, 0.0)
    v = _vertices([((-1, 0, 1), (0, 0), up), ((1, 0, 1), (1, 0), up), ((1, 0, -1), (1, 1), up), ((-1, 0, -1), (0, 1), up)])
    return v, np.array([0, 1, 3, 1, 2, 3], np.uint32)


def cube_mesh():
    """A 2x2x2 cube, four vertices per face, every face with the full uv square (assets.cpp:1474-1570). The reference builds
    these vertices without a colour (value-initialised to 0; no shader reads it)."""
    faces = [  # (uv origin, uv x edge, uv y edge, normal)
        ((-1, -1, 1), (2, 0, 0), (0, 0, -2), (0, -1, 0)),
        ((-1, 1, -1), (2, 0, 0), (0, 0, 2), (0, 1, 0)),
        ((1, -1, -1), (0, 0, 2), (0, 2, 0), (1, 0, 0)),
        ((-1, -1, 1), (0, 0, -2), (0, 2, 0), (-1, 0, 0)),
        ((-1, -1, -1), (2, 0, 0), (0, 2, 0), (0, 0, -1)),
        ((1, -1, 1), (-2, 0, 0), (0, 2, 0), (0, 0, 1)),
    ]
    rows, indices = [], []
    for origin, ex, ey, normal in faces:
        o, ex, ey = np.array(origin, np.float32), np.array(ex, np.float32), np.array(ey, np.float32)
        base = len(rows)
        rows += [(o, (0, 0), normal), (o + ex, (1, 0), normal), (o + ex + ey, (1, 1), normal), (o + ey, (0, 1), normal)]
        indices += [base, base + 1, base + 2, base, base + 2, base + 3]
    return _vertices(rows, color=(0.0, 0.0, 0.0, 0.0)), np.array(indices, np.uint32)


def transform_matrix(translation=(0, 0, 0), eulers=(0, 0, 0), scale=(1, 1, 1)):
    """Transform::toMatrix (geometry/transform.cpp:11-15) as a ctypes Mat4."""
    out = abi.Mat4()
    lib().szg_transform_matrix(abi.f3(*translation), abi.f3(*eulers), abi.f3(*scale), C.byref(out))
    return out


def inverse_transpose(m):
    out = abi.Mat4()
    lib().szg_mat4_inverse_transpose(C.byref(m), C.byref(out))
    return out


class MeshInstanced:
    """One mesh with its surfaces and instance transforms. `surfaces` = [(first_index, index_count, material)], a
    material being {"color"|"normal"|"orm": (uint8 [h, w, 4] array, srgb flag)}."""

    def __init__(self, vertices, indices, surfaces, models, render=True, casts_shadow=True, name=""):
        self.name = name
        self.vertices = np.ascontiguousarray(vertices)
        self.indices = np.ascontiguousarray(indices, np.uint32)
        self.surfaces = list(surfaces)
        self.models = list(models)
        self.mits = [inverse_transpose(m) for m in self.models]  # scene.cpp:205-211
        self.render, self.casts_shadow = bool(render), bool(casts_shadow)
        self._models_np = np.array([np.array(m.m, np.float32) for m in self.models], np.float32).reshape(-1, 16)
        self._mits_np = np.array([np.array(m.m, np.float32) for m in self.mits], np.float32).reshape(-1, 16)
        self._device = None
        self._keep = []

    # -- device copies (torch owns the memory)
    def _to_device(self, device):
        import torch

        if self._device is not None:
            return self._device

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(device)

        d = {"vertices": up(self.vertices), "indices": up(self.indices), "models": up(self._models_np), "mits": up(self._mits_np),
             "textures": {}}
        for _, _, material in self.surfaces:
            for key, (tex, _) in material.items():
                if id(tex) not in d["textures"]:
                    d["textures"][id(tex)] = up(tex)
        self._device = d
        return d

    def _struct(self, ptr_of, tex_ptr_of):
        n = len(self.surfaces)
        surf = (abi.Surface * max(n, 1))()
        for i, (first, count, material) in enumerate(self.surfaces):
            surf[i].first_index, surf[i].index_count = int(first), int(count)
            for key in ("color", "normal", "orm"):
                if key not in material:
                    continue
                tex, srgb = material[key]
                t = getattr(surf[i].material, key)
                t.data, t.width, t.height, t.pitch_bytes, t.srgb = tex_ptr_of(tex), tex.shape[1], tex.shape[0], tex.shape[1] * 4, int(srgb)
        m = abi.MeshInstanced()
        m.d_vertices, m.vertex_count = ptr_of("vertices"), len(self.vertices)
        m.d_indices, m.index_count = ptr_of("indices"), len(self.indices)
        m.surfaces, m.surface_count = C.cast(surf, C.POINTER(abi.Surface)), n
        m.d_models, m.d_model_inverse_transposes, m.instance_count = ptr_of("models"), ptr_of("mits"), len(self.models)
        m.render, m.casts_shadow = int(self.render), int(self.casts_shadow)
        self._keep = (self._keep + [surf])[-8:]  # the struct points into `surf`
        return m

    def device_struct(self, device="cuda"):
        d = self._to_device(device)
        return self._struct(lambda k: d[k].data_ptr(), lambda tex: d["textures"][id(tex)].data_ptr())

    def host_struct(self):
        host = {"vertices": self.vertices, "indices": self.indices, "models": self._models_np, "mits": self._mits_np}
        return self._struct(lambda k: host[k].ctypes.data, lambda tex: tex.ctypes.data)


def mesh_array(meshes, device=None):
    """ctypes array of abi.MeshInstanced with device pointers (`device` given) or host pointers (oracle)."""
    arr = (abi.MeshInstanced * max(len(meshes), 1))()
    for i, m in enumerate(meshes):
        arr[i] = m.device_struct(device) if device is not None else m.host_struct()
    return arr


def reference_default_scene():
    """editor.cpp:500-545: two cubes of scale 5 at (0, -8, +-6) and a floor plane of scale (20, 1, 20) at y = -1."""
    material = default_material()
    cv, ci = cube_mesh()
    pv, pi = plane_mesh()
    cube_surfaces = [(0, len(ci), material)]
    return [
        MeshInstanced(cv, ci, cube_surfaces, [transform_matrix((0.0, -8.0, 6.0), (0, 0, 0), (5, 5, 5))], name="Model_1"),
        MeshInstanced(cv, ci, cube_surfaces, [transform_matrix((0.0, -8.0, -6.0), (0, 0, 0), (5, 5, 5))], name="Model_2"),
        MeshInstanced(pv, pi, [(0, len(pi), material)], [transform_matrix((0.0, -1.0, 0.0), (0, 0, 0), (20, 1, 20))], name="Floor"),
    ]


def meshes_of_fill_scene(fill):
    """The analytic scene of abi.FillScene (ground rectangle + boxes) as meshes, for cross-checking the rasteriser
    against the ray-cast fill: one instanced cube mesh per (roughness, metallic) class and the ground plane."""
    cv, ci = cube_mesh()
    pv, pi = plane_mesh()
    normal = constant_texture((127, 127, 255, 0), 1)
    color = default_color_map()
    out = []
    classes = {}
    for b in range(fill.box_count):
        box = fill.boxes[b]
        key = (round(box.roughness * 255.0), round(box.metallic * 255.0))
        classes.setdefault(key, []).append(transform_matrix(tuple(box.center), (0, 0, 0), tuple(box.half_extent)))
    for (rough, metal), models in sorted(classes.items()):
        material = {"color": (color, False), "normal": (normal, False), "orm": (constant_texture((255, rough, metal, 0), 1), False)}
        out.append(MeshInstanced(cv, ci, [(0, len(ci), material)], models, name=f"boxes_r{rough}_m{metal}"))
    ground = {"color": (color, False), "normal": (normal, False),
              "orm": (constant_texture((255, round(fill.ground_roughness * 255.0), 0, 0), 1), False)}
    h = fill.ground_half_extent
    out.append(MeshInstanced(pv, pi, [(0, len(pi), ground)], [transform_matrix((0.0, fill.ground_y, 0.0), (0, 0, 0), (h, 1.0, h))], name="ground"))
    return out
