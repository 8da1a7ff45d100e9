"""ctypes binding of oracle/libszg_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
All pointers handed to the oracle are HOST memory (numpy arrays).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from syzygy_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIBM = None
_LITERAL = None

VP = C.c_void_p
U32 = C.c_uint32
P = C.POINTER
FP = P(C.c_float)


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def _load(name):
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        h = C.CDLL(path)
        h.oracle_builtin_eval.argtypes = [C.c_int, FP, FP, FP, C.c_size_t]
        h.oracle_abi_version.restype = C.c_int
        h.oracle_transmittance_lut.argtypes = [P(abi.AtmospherePacked), U32, U32, U32, FP, C.c_int]
        h.oracle_transmittance_texel.argtypes = [P(abi.AtmospherePacked), U32, U32, U32, U32, FP]
        h.oracle_rmu_to_uv.argtypes = [P(abi.AtmospherePacked), U32, U32, C.c_float, C.c_float, FP]
        h.oracle_uv_to_rmu.argtypes = [P(abi.AtmospherePacked), U32, U32, C.c_float, C.c_float, FP]
        h.oracle_scattering_integral.argtypes = [P(abi.AtmospherePacked), FP, U32, U32, FP, FP, C.c_float, FP]
        h.oracle_skyview_lut.argtypes = [P(abi.AtmospherePacked), U32, P(abi.CameraPacked), U32, FP, U32, U32, U32, U32, FP,
                                         U32, U32, C.c_int]
        h.oracle_lights.argtypes = [P(abi.SceneTexture), abi.Rect, P(abi.RowTile), P(abi.GBuffer), P(abi.ShadowMaps),
                                    P(abi.CameraPacked), U32, P(abi.DirectionalLightPacked), U32, U32, P(abi.SpotLightPacked),
                                    U32, C.c_int]
        h.oracle_composite.argtypes = [P(abi.SceneTexture), abi.Rect, P(abi.RowTile), P(abi.GBuffer), P(abi.ShadowMaps),
                                       P(abi.AtmospherePacked), U32, P(abi.CameraPacked), U32, P(abi.DirectionalLightPacked),
                                       U32, FP, U32, U32, FP, U32, U32, C.c_int]
        h.oracle_gbuffer_fill.argtypes = [P(abi.SceneTexture), abi.Rect, P(abi.RowTile), P(abi.GBuffer), P(abi.CameraPacked),
                                          U32, P(abi.FillScene), C.c_int]
        h.oracle_shadow_map.argtypes = [P(abi.Mat4), P(abi.Mat4), U32, P(abi.FillScene), FP, C.c_int]
        h.oracle_gbuffer_raster.argtypes = [P(abi.SceneTexture), abi.Rect, P(abi.RowTile), P(abi.GBuffer), P(abi.CameraPacked), U32,
                                            P(abi.MeshInstanced), U32, C.c_int]
        h.oracle_shadow_raster.argtypes = [P(abi.Image), P(abi.Mat4), C.c_float, C.c_float, P(abi.MeshInstanced), U32, C.c_int]
        h.oracle_multiscatter_lut.argtypes = [P(abi.AtmospherePacked), U32, FP, U32, U32, U32, FP, FP]
        h.oracle_aerial_lut.argtypes = [P(abi.AtmospherePacked), U32, P(abi.CameraPacked), U32, FP, U32, U32, U32, U32, U32,
                                        C.c_float, FP, FP, C.c_int]
        h.oracle_oetf.argtypes = [P(abi.Image), U32, U32, U32]
        h.oracle_float_to_half.restype = C.c_uint16
        h.oracle_float_to_half.argtypes = [C.c_float]
        h.oracle_half_to_float.restype = C.c_float
        h.oracle_half_to_float.argtypes = [C.c_uint16]
        h.oracle_unorm16_store.restype = C.c_uint16
        h.oracle_unorm16_store.argtypes = [C.c_float]
        assert h.oracle_abi_version() == abi.SZG_ABI_VERSION
        return h


def lib():
    """The oracle with the GLSL built-ins pinned to include/szg/fpmath.h (the parity checker)."""
    global _LIB
    if _LIB is None:
        # SZG_ORACLE_LITERAL=1 (together with SZG_HIP_LIBRARY=.../libszg_hip_literal.so): the whole test suite and every sweep
        # tool then compare the two LITERAL builds - the pair that is pinned against the reference's SPIR-V
        _LIB = lib_literal() if os.environ.get("SZG_ORACLE_LITERAL") == "1" else _load("libszg_oracle.so")
    return _LIB


def lib_libm():
    """The same oracle with libm built-ins: an independent cross-check of the pinned one."""
    global _LIBM
    if _LIBM is None:
        _LIBM = _load("libszg_oracle_libm.so")
    return _LIBM


def lib_literal():
    """The oracle with its contraction rule switched off (-DSZG_ORACLE_LITERAL): what tests/test_spirv_pin.py compares with
    the literal execution of the reference's SPIR-V."""
    global _LITERAL
    if _LITERAL is None:
        _LITERAL = _load("libszg_oracle_literal.so")
    return _LITERAL


class use_literal:
    """Context manager: route the module-level helpers to the literal (uncontracted) build."""

    def __enter__(self):
        global _LIB
        self._saved = lib()
        _LIB = lib_literal()

    def __exit__(self, *a):
        global _LIB
        _LIB = self._saved
        return False


class use_libm:
    """Context manager: route the module-level helpers to the libm build."""

    def __enter__(self):
        global _LIB
        self._saved = lib()
        _LIB = lib_libm()

    def __exit__(self, *a):
        global _LIB
        _LIB = self._saved
        return False


def builtin_eval(fn, x, y=None, libm=False):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
    out = np.empty_like(x)
    (lib_libm() if libm else lib()).oracle_builtin_eval(fn, fptr(x), fptr(y), fptr(out), x.size)
    return out


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(FP)


def host_image(array, fmt):
    """abi.Image over a host numpy array [h, w(, c)]."""
    im = abi.Image()
    im.data = array.ctypes.data
    im.height, im.width = array.shape[0], array.shape[1]
    im.pitch_bytes = array.strides[0]
    im.format = fmt
    return im


class HostFrame:
    """Host-memory G-buffer + scene texture of `width` x `rows` (rows = local rows)."""

    def __init__(self, width, rows, debug=True):
        self.width, self.rows = width, rows
        self.diffuse = np.zeros((rows, width, 4), np.float16)
        self.specular = np.zeros((rows, width, 4), np.float16)
        self.normal = np.zeros((rows, width, 4), np.float16)
        self.position = np.zeros((rows, width, 4), np.float32)
        self.orm = np.zeros((rows, width, 4), np.float16)
        self.color = np.zeros((rows, width, 4), np.uint16)
        self.depth = np.zeros((rows, width), np.float32)
        self.debug = np.zeros((rows, width, 4), np.float32) if debug else None

    def gbuffer(self):
        g = abi.GBuffer()
        g.diffuse = host_image(self.diffuse, abi.SZG_FORMAT_RGBA16_SFLOAT)
        g.specular = host_image(self.specular, abi.SZG_FORMAT_RGBA16_SFLOAT)
        g.normal = host_image(self.normal, abi.SZG_FORMAT_RGBA16_SFLOAT)
        g.worldPosition = host_image(self.position, abi.SZG_FORMAT_RGBA32_SFLOAT)
        g.occlusionRoughnessMetallic = host_image(self.orm, abi.SZG_FORMAT_RGBA16_SFLOAT)
        return g

    def scene(self):
        st = abi.SceneTexture()
        st.color = host_image(self.color, abi.SZG_FORMAT_RGBA16_UNORM)
        st.depth = host_image(self.depth, abi.SZG_FORMAT_D32_SFLOAT)
        if self.debug is not None:
            st.debug_color = host_image(self.debug, abi.SZG_FORMAT_RGBA32_SFLOAT)
        return st

    def planes(self):
        return {"diffuse": self.diffuse, "specular": self.specular, "normal": self.normal, "worldPosition": self.position,
                "occlusionRoughnessMetallic": self.orm}


def transmittance_lut(atm_packed, width, height, threads=1):
    out = np.zeros((height, width, 4), np.float32)
    lib().oracle_transmittance_lut(C.byref(atm_packed), 0, width, height, fptr(out), threads)
    return out


def skyview_lut(atm_packed, cam_packed, tlut, width, height, row_begin=0, row_end=None, threads=1, out=None):
    if out is None:
        out = np.zeros((height, width, 4), np.float32)
    row_end = height if row_end is None else row_end
    lib().oracle_skyview_lut(C.byref(atm_packed), 0, C.byref(cam_packed), 0, fptr(tlut), tlut.shape[1], tlut.shape[0], width,
                             height, fptr(out), row_begin, row_end, threads)
    return out


def gbuffer_fill(frame, draw_rect, tile, cam_packed, fill_scene, threads=1):
    g, st = frame.gbuffer(), frame.scene()
    lib().oracle_gbuffer_fill(C.byref(st), draw_rect, C.byref(tile) if tile is not None else None, C.byref(g),
                              C.byref(cam_packed), 0, C.byref(fill_scene), threads)


def gbuffer_raster(frame, draw_rect, tile, cam_packed, meshes, threads=1):
    """oracle_gbuffer_raster over syzygy_amd.meshes.MeshInstanced objects (host pointers)."""
    from syzygy_amd.meshes import mesh_array

    g, st = frame.gbuffer(), frame.scene()
    arr = mesh_array(meshes, None)
    lib().oracle_gbuffer_raster(C.byref(st), draw_rect, C.byref(tile) if tile is not None else None, C.byref(g),
                                C.byref(cam_packed), 0, arr, len(meshes), threads)


def shadow_raster(proj_view, dim, meshes, bias_constant=0.0, bias_slope=0.0, threads=1):
    """One depth-only shadow raster of `meshes` with the light matrix `proj_view` (abi.Mat4)."""
    from syzygy_amd.meshes import mesh_array

    out = np.zeros((dim, dim), np.float32)
    im = host_image(out, abi.SZG_FORMAT_D32_SFLOAT)
    arr = mesh_array(meshes, None)
    lib().oracle_shadow_raster(C.byref(im), C.byref(proj_view), float(bias_constant), float(bias_slope), arr, len(meshes), threads)
    return out


def lights(frame, draw_rect, tile, shadow_maps, cam_packed, dir_lights, dir_count, dir_skip, spot_lights, spot_count, threads=1):
    g, st = frame.gbuffer(), frame.scene()
    lib().oracle_lights(C.byref(st), draw_rect, C.byref(tile) if tile is not None else None, C.byref(g),
                        C.byref(shadow_maps) if shadow_maps is not None else None, C.byref(cam_packed), 0,
                        C.cast(dir_lights, P(abi.DirectionalLightPacked)), dir_count, dir_skip,
                        C.cast(spot_lights, P(abi.SpotLightPacked)) if spot_count else None, spot_count, threads)


def composite(frame, draw_rect, tile, shadow_maps, atm_packed, cam_packed, dir_lights, sun_index, tlut, slut, threads=1):
    g, st = frame.gbuffer(), frame.scene()
    lib().oracle_composite(C.byref(st), draw_rect, C.byref(tile) if tile is not None else None, C.byref(g),
                           C.byref(shadow_maps) if shadow_maps is not None else None, C.byref(atm_packed), 0,
                           C.byref(cam_packed), 0, C.cast(dir_lights, P(abi.DirectionalLightPacked)), sun_index, fptr(tlut),
                           tlut.shape[1], tlut.shape[0], fptr(slut), slut.shape[1], slut.shape[0], threads)


def oetf(color_u16, function):
    """In-place OETF on a host [h, w, 4] uint16 array."""
    im = host_image(color_u16, abi.SZG_FORMAT_RGBA16_UNORM)
    lib().oracle_oetf(C.byref(im), color_u16.shape[1], color_u16.shape[0], function)
    return color_u16


def aerial_lut(atm_packed, cam_packed, tlut, max_distance, dims=(32, 32, 32), threads=1):
    W, H, D = dims
    lum = np.zeros((D * H, W, 4), np.float32)
    tr = np.zeros((D * H, W, 4), np.float32)
    lib().oracle_aerial_lut(C.byref(atm_packed), 0, C.byref(cam_packed), 0, fptr(tlut), tlut.shape[1], tlut.shape[0], W, H, D,
                            max_distance, fptr(lum), fptr(tr), threads)
    return lum, tr


def multiscatter_lut(atm_packed, tlut, dim=32):
    out = np.zeros((dim, dim, 4), np.float32)
    fms = np.zeros((dim, dim, 4), np.float32)
    lib().oracle_multiscatter_lut(C.byref(atm_packed), 0, fptr(tlut), tlut.shape[1], tlut.shape[0], dim, fptr(out), fptr(fms))
    return out, fms


def shadow_map(light, dim, fill_scene, threads=1):
    """Depth map of the analytic scene from `light` (a packed directional or spot light)."""
    out = np.zeros((dim, dim), np.float32)
    lib().oracle_shadow_map(C.byref(light.projection), C.byref(light.view), dim, C.byref(fill_scene), fptr(out), threads)
    return out
