"""Host-side mirror of the reference's render-pass API for the deferred-shading +
atmosphere path, over the C-ABI library.

Names, argument order and meaning follow the reference:
  TStagedBuffer             renderer/buffers.hpp:209-299, buffers.cpp:180-255
  SceneTexture              renderer/scenetexture.hpp:11-81
  DeferredShadingPipeline   renderer/pipelines/deferred.hpp:23-119
  SkyViewComputePipeline    renderer/pipelines/skyview.hpp:24-51
with `cmd` (VkCommandBuffer) replaced by a HIP stream handle and Vulkan images by
linear device buffers. torch is used only to own device memory and streams.
"""
import ctypes as C

import numpy as np
import torch

from . import abi
from ._lib import check, lib


def _stream_handle(cmd):
    """`cmd` may be None (current torch stream), a torch.cuda.Stream or a raw handle."""
    if cmd is None:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if isinstance(cmd, torch.cuda.Stream):
        return C.c_void_p(cmd.cuda_stream)
    return C.c_void_p(int(cmd))


def _image(tensor, width, height, fmt):
    im = abi.Image()
    im.data = tensor.data_ptr()
    im.width = width
    im.height = height
    im.pitch_bytes = width * abi.TEXEL_BYTES[fmt]
    im.format = fmt
    return im


def rect(width, height):
    return abi.Rect(0, 0, int(width), int(height))


class TStagedBuffer:
    """buffers.hpp:209-299: host staging + device copy of an array of packed structs.

    recordCopyToDevice() is asynchronous: the copy runs in stream order, possibly behind a whole frame of kernels, so the
    pinned memory it reads must stay untouched until it has run. The reference guards its staging memory with the frame
    fences (two frames in flight, framebuffer.cpp:134); here every copy reads its own slot of a small ring of pinned
    buffers, and a slot is rewritten only after the event recorded behind its last copy has completed."""

    SLOTS = 3

    def __init__(self, struct_type, capacity, device="cuda:0"):
        self.struct_type = struct_type
        self.capacity = int(capacity)
        self._staged = []
        self._device_size = 0
        self._dirty = False
        nbytes = C.sizeof(struct_type) * self.capacity
        self._ring = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(self.SLOTS)]
        self._ring_done = [None] * self.SLOTS
        self._ring_next = 0
        self._device = torch.zeros(nbytes, dtype=torch.uint8, device=device)

    @classmethod
    def allocate(cls, struct_type, capacity, device="cuda:0"):
        return cls(struct_type, capacity, device)

    def clearStaged(self):
        self._staged = []
        self._dirty = True

    def push(self, value):
        values = value if isinstance(value, (list, tuple)) else [value]
        if len(self._staged) + len(values) > self.capacity:
            raise ValueError("TStagedBuffer: staged size exceeds capacity")
        self._staged.extend(values)
        self._dirty = True

    def stage(self, values):
        self.clearStaged()
        self.push(list(values))

    def pop(self, count):
        del self._staged[len(self._staged) - count:]
        self._dirty = True

    def recordCopyToDevice(self, cmd=None):
        n = len(self._staged)
        size = C.sizeof(self.struct_type)
        if n:
            raw = b"".join(bytes(v) for v in self._staged)
            slot = self._ring_next
            self._ring_next = (slot + 1) % self.SLOTS
            if self._ring_done[slot] is not None:
                self._ring_done[slot].synchronize()  # the copy that last read this slot has run
            host = self._ring[slot]
            host[: n * size] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
            stream = torch.cuda.current_stream() if cmd is None else cmd
            with torch.cuda.stream(stream) if isinstance(stream, torch.cuda.Stream) else _NullCtx():
                self._device[: n * size].copy_(host[: n * size], non_blocking=True)
                done = torch.cuda.Event()
                done.record()
            self._ring_done[slot] = done
        self._device_size = n
        self._dirty = False

    def deviceAddress(self):
        return self._device.data_ptr()

    def deviceSize(self):
        return self._device_size

    def stagedSize(self):
        return len(self._staged)

    def stagingCapacity(self):
        return self.capacity

    def isDirty(self):
        return self._dirty

    def readValidStaged(self):
        return list(self._staged)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class SceneTexture:
    """scenetexture.hpp:11-81: colour (RGBA16 UNORM) + depth (D32F), allocated at a
    capacity extent and rendered into a sub-rect at offset (0, 0). `debug=True` adds
    the fp32 pre-quantisation colour plane used by the parity tests."""

    def __init__(self, width, height, device="cuda:0", debug=False):
        self.width, self.height = int(width), int(height)
        self.color = torch.zeros((self.height, self.width, 4), dtype=torch.int16, device=device)
        self.depth = torch.zeros((self.height, self.width), dtype=torch.float32, device=device)
        self.debug = torch.zeros((self.height, self.width, 4), dtype=torch.float32, device=device) if debug else None

    def abi(self):
        st = abi.SceneTexture()
        st.color = _image(self.color, self.width, self.height, abi.SZG_FORMAT_RGBA16_UNORM)
        st.depth = _image(self.depth, self.width, self.height, abi.SZG_FORMAT_D32_SFLOAT)
        if self.debug is not None:
            st.debug_color = _image(self.debug, self.width, self.height, abi.SZG_FORMAT_RGBA32_SFLOAT)
        return st

    def color_numpy(self):
        return self.color.cpu().numpy().view(np.uint16)


def recordOETF(cmd, sceneTexture, width, height, transferFunction=abi.SZG_OETF_SRGB):
    """editor.cpp:303-340: in-place linear -> display encoding of the colour image (sRGB by default,
    editorconfig.hpp:13)."""
    im = sceneTexture.abi().color
    check(lib().szg_record_oetf(_stream_handle(cmd), C.byref(im), int(width), int(height), int(transferFunction)))


class DeferredShadingPipeline:
    """deferred.hpp:23-119."""

    def __init__(self, dimensionCapacity, max_spot_lights=16, max_shadow_maps=10, shadow_map_dim=0, device_index=0):
        desc = abi.DeferredDesc(int(dimensionCapacity[0]), int(dimensionCapacity[1]), int(max_spot_lights),
                                int(max_shadow_maps), int(shadow_map_dim), 0)
        handle = C.c_void_p()
        check(lib().szg_deferred_create(C.byref(handle), C.byref(desc), int(device_index)))
        self._h = handle
        self.capacity = (int(dimensionCapacity[0]), int(dimensionCapacity[1]))

    def recordDrawCommands(self, cmd, drawRect, sceneTexture, atmosphericDirectionalLightsCount, directionalLights, spotLights,
                           viewCameraIndex, cameras, sceneGeometry, tile=None):
        """deferred.hpp:34-44. `spotLights` is a ctypes array (host span) of SpotLightPacked,
        `sceneGeometry` an abi.FillScene or None (keep the G-buffer as it is)."""
        st = sceneTexture.abi()
        n_spot = len(spotLights) if spotLights is not None else 0
        spots = C.cast(spotLights, C.POINTER(abi.SpotLightPacked)) if n_spot else None
        check(lib().szg_deferred_record_draw_commands(
            self._h, _stream_handle(cmd), drawRect, C.byref(tile) if tile is not None else None, C.byref(st),
            int(atmosphericDirectionalLightsCount), C.c_void_p(directionalLights.deviceAddress()),
            int(directionalLights.deviceSize()), spots, n_spot, int(viewCameraIndex), C.c_void_p(cameras.deviceAddress()),
            C.byref(sceneGeometry) if sceneGeometry is not None else None))

    def recordGBufferFill(self, cmd, drawRect, sceneTexture, viewCameraIndex, cameras, sceneGeometry, tile=None):
        st = sceneTexture.abi()
        check(lib().szg_deferred_record_gbuffer_fill(
            self._h, _stream_handle(cmd), drawRect, C.byref(tile) if tile is not None else None, C.byref(st),
            int(viewCameraIndex), C.c_void_p(cameras.deviceAddress()), C.byref(sceneGeometry)))

    def recordLights(self, cmd, drawRect, sceneTexture, atmosphericDirectionalLightsCount, directionalLights, spotLights,
                     viewCameraIndex, cameras, tile=None):
        st = sceneTexture.abi()
        n_spot = len(spotLights) if spotLights is not None else 0
        spots = C.cast(spotLights, C.POINTER(abi.SpotLightPacked)) if n_spot else None
        check(lib().szg_deferred_record_lights(
            self._h, _stream_handle(cmd), drawRect, C.byref(tile) if tile is not None else None, C.byref(st),
            int(atmosphericDirectionalLightsCount), C.c_void_p(directionalLights.deviceAddress()),
            int(directionalLights.deviceSize()), spots, n_spot, int(viewCameraIndex), C.c_void_p(cameras.deviceAddress())))

    def recordShadowMaps(self, cmd, directionalLights, spotLights, sceneGeometry):
        """render the pipeline-owned shadow maps for the analytic scene."""
        n_spot = len(spotLights) if spotLights is not None else 0
        spots = C.cast(spotLights, C.POINTER(abi.SpotLightPacked)) if n_spot else None
        check(lib().szg_deferred_record_shadow_maps(self._h, _stream_handle(cmd), C.c_void_p(directionalLights.deviceAddress()),
                                                    int(directionalLights.deviceSize()), spots, n_spot, C.byref(sceneGeometry)))

    # -- real scene geometry (include/szg/raster.h); `meshes` is a list of syzygy_amd.meshes.MeshInstanced
    def recordGBufferRaster(self, cmd, drawRect, sceneTexture, viewCameraIndex, cameras, meshes, tile=None, device="cuda"):
        """The G-buffer pass of recordDrawCommands (deferred.cpp:493-713) through the compute rasteriser."""
        from .meshes import mesh_array

        st = sceneTexture.abi()
        arr = mesh_array(meshes, device)
        check(lib().szg_deferred_record_gbuffer_raster(
            self._h, _stream_handle(cmd), drawRect, C.byref(tile) if tile is not None else None, C.byref(st),
            int(viewCameraIndex), C.c_void_p(cameras.deviceAddress()), arr, len(meshes)))

    def recordShadowRaster(self, cmd, directionalLights, spotLights, meshes, device="cuda"):
        """The shadow passes (shadowpass.cpp:188-270) into the pipeline-owned maps."""
        from .meshes import mesh_array

        n_spot = len(spotLights) if spotLights is not None else 0
        spots = C.cast(spotLights, C.POINTER(abi.SpotLightPacked)) if n_spot else None
        arr = mesh_array(meshes, device)
        check(lib().szg_deferred_record_shadow_raster(self._h, _stream_handle(cmd), C.c_void_p(directionalLights.deviceAddress()),
                                                      int(directionalLights.deviceSize()), spots, n_spot, arr, len(meshes)))

    def recordDrawCommandsMeshes(self, cmd, drawRect, sceneTexture, atmosphericDirectionalLightsCount, directionalLights, spotLights,
                                 viewCameraIndex, cameras, meshes, tile=None, device="cuda"):
        """deferred.hpp:34-44 with `sceneGeometry` = real meshes: shadow raster, G-buffer raster, lights."""
        from .meshes import mesh_array

        st = sceneTexture.abi()
        n_spot = len(spotLights) if spotLights is not None else 0
        spots = C.cast(spotLights, C.POINTER(abi.SpotLightPacked)) if n_spot else None
        arr = mesh_array(meshes, device)
        check(lib().szg_deferred_record_draw_commands_meshes(
            self._h, _stream_handle(cmd), drawRect, C.byref(tile) if tile is not None else None, C.byref(st),
            int(atmosphericDirectionalLightsCount), C.c_void_p(directionalLights.deviceAddress()),
            int(directionalLights.deviceSize()), spots, n_spot, int(viewCameraIndex), C.c_void_p(cameras.deviceAddress()),
            arr, len(meshes)))

    def gbuffer(self):
        return lib().szg_deferred_gbuffer(self._h).contents

    def shadowMaps(self):
        return lib().szg_deferred_shadow_maps(self._h).contents

    def setShadowMap(self, index, tensor):
        """Attach a caller-owned D32F map (2-D float32 CUDA tensor) to slot `index`, or detach with None."""
        if tensor is None:
            check(lib().szg_deferred_set_shadow_map(self._h, int(index), None))
            return
        im = _image(tensor, tensor.shape[1], tensor.shape[0], abi.SZG_FORMAT_D32_SFLOAT)
        check(lib().szg_deferred_set_shadow_map(self._h, int(index), C.byref(im)))

    def getConfiguration(self):
        cfg = abi.DeferredConfiguration()
        check(lib().szg_deferred_get_configuration(self._h, C.byref(cfg)))
        return cfg

    def setConfiguration(self, cfg):
        check(lib().szg_deferred_set_configuration(self._h, C.byref(cfg)))

    def cleanup(self):
        if self._h:
            lib().szg_deferred_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass

    # -- helpers to move G-buffer planes between host numpy arrays and the device
    def upload_gbuffer(self, planes):
        """planes: dict name -> numpy array [h, w, 4] (float16 planes / float32 position)."""
        g = self.gbuffer()
        names = {"diffuse": g.diffuse, "specular": g.specular, "normal": g.normal, "worldPosition": g.worldPosition,
                 "occlusionRoughnessMetallic": g.occlusionRoughnessMetallic}
        for name, im in names.items():
            arr = np.ascontiguousarray(planes[name])
            h, w = arr.shape[0], arr.shape[1]
            src = torch.from_numpy(arr.view(np.uint8).reshape(h, -1)).cuda()
            _memcpy2d_to(im, src, h)

    def download_gbuffer(self, width, height):
        g = self.gbuffer()
        out = {}
        for name, im, dt in (("diffuse", g.diffuse, np.float16), ("specular", g.specular, np.float16),
                             ("normal", g.normal, np.float16), ("worldPosition", g.worldPosition, np.float32),
                             ("occlusionRoughnessMetallic", g.occlusionRoughnessMetallic, np.float16)):
            raw = _memcpy2d_from(im, width * abi.TEXEL_BYTES[im.format], height)
            out[name] = raw.cpu().numpy().view(dt).reshape(height, width, 4)
        return out


class _RawDeviceArray:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 3,
                                         "strides": None}


def _alias_tensor(ptr, shape):
    return torch.as_tensor(_RawDeviceArray(ptr, shape), device="cuda")


def _hip_memcpy2d(dst_ptr, dpitch, src_ptr, spitch, width_bytes, height):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy2D.restype = C.c_int
    hip.hipMemcpy2D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    torch.cuda.synchronize()
    rc = hip.hipMemcpy2D(dst_ptr, dpitch, src_ptr, spitch, width_bytes, height, 3)  # hipMemcpyDeviceToDevice
    if rc != 0:
        raise RuntimeError(f"hipMemcpy2D failed: {rc}")


def _memcpy2d_to(im, src, rows):
    _hip_memcpy2d(im.data, im.pitch_bytes, src.data_ptr(), src.shape[1], src.shape[1], rows)


def _memcpy2d_from(im, width_bytes, rows):
    out = torch.empty((rows, width_bytes), dtype=torch.uint8, device="cuda")
    _hip_memcpy2d(out.data_ptr(), width_bytes, im.data, im.pitch_bytes, width_bytes, rows)
    return out


class SkyViewComputePipeline:
    """skyview.hpp:24-51. Use `create()`; it returns None on failure like the reference
    (skyview.cpp:713-740)."""

    def __init__(self, handle, desc):
        self._h = handle
        self.desc = desc

    @staticmethod
    def create(device_index=0, transmittance_extent=(512, 128), skyview_extent=(2048, 1024), flags=0):
        desc = abi.SkyviewDesc(int(transmittance_extent[0]), int(transmittance_extent[1]), int(skyview_extent[0]),
                               int(skyview_extent[1]), int(flags), 0)
        handle = C.c_void_p()
        status = lib().szg_skyview_create(C.byref(handle), C.byref(desc), int(device_index))
        if status != abi.SZG_OK:
            return None
        return SkyViewComputePipeline(handle, desc)

    def recordDrawCommands(self, cmd, sceneTexture, drawRect, gbuffer, shadowMaps, atmosphereIndex, atmospheres,
                           viewCameraIndex, cameras, sunLightIndex, lights, tile=None):
        """skyview.hpp:39-51: transmittance LUT -> sky-view LUT -> camera composite."""
        st = sceneTexture.abi()
        check(lib().szg_skyview_record_draw_commands(
            self._h, _stream_handle(cmd), C.byref(st), drawRect, C.byref(tile) if tile is not None else None,
            C.byref(gbuffer), C.byref(shadowMaps) if shadowMaps is not None else None, int(atmosphereIndex),
            C.c_void_p(atmospheres.deviceAddress()), int(viewCameraIndex), C.c_void_p(cameras.deviceAddress()),
            int(sunLightIndex), C.c_void_p(lights.deviceAddress())))

    def recordTransmittance(self, cmd, atmosphereIndex, atmospheres):
        check(lib().szg_skyview_record_transmittance(self._h, _stream_handle(cmd), int(atmosphereIndex),
                                                     C.c_void_p(atmospheres.deviceAddress())))

    def recordSkyViewLUT(self, cmd, atmosphereIndex, atmospheres, viewCameraIndex, cameras):
        check(lib().szg_skyview_record_skyview_lut(self._h, _stream_handle(cmd), int(atmosphereIndex),
                                                   C.c_void_p(atmospheres.deviceAddress()), int(viewCameraIndex),
                                                   C.c_void_p(cameras.deviceAddress())))

    def recordSkyViewLUTRows(self, cmd, atmosphereIndex, atmospheres, viewCameraIndex, cameras, rowBegin, rowEnd):
        """Multi-GPU extension: texel rows [rowBegin, rowEnd) only (see rowtile.allgather_skyview_lut)."""
        check(lib().szg_skyview_record_skyview_lut_rows(self._h, _stream_handle(cmd), int(atmosphereIndex),
                                                        C.c_void_p(atmospheres.deviceAddress()), int(viewCameraIndex),
                                                        C.c_void_p(cameras.deviceAddress()), int(rowBegin), int(rowEnd)))

    def lutRowSlice(self, rank, nranks):
        """[begin, end) rows of the sky-view LUT that `rank` of `nranks` computes (szg_skyview_lut_row_slice)."""
        b, e = C.c_uint32(), C.c_uint32()
        check(lib().szg_skyview_lut_row_slice(self._h, int(rank), int(nranks), C.byref(b), C.byref(e)))
        return int(b.value), int(e.value)

    def setLUTReuse(self, enable):
        """Extension (abi.h "LUT reuse across frames"): skip a LUT pass whose parameter blocks are bit-equal to those its
        texels were computed from; compared on the device, identical results. Off by default (the reference recomputes)."""
        check(lib().szg_skyview_set_lut_reuse(self._h, 1 if enable else 0))

    def invalidateLUTs(self, which=abi.SZG_LUT_TRANSMITTANCE | abi.SZG_LUT_SKYVIEW):
        """The caller wrote texels of these LUTs through a pointer / tensor it kept (abi.h szg_skyview_invalidate_luts)."""
        check(lib().szg_skyview_invalidate_luts(self._h, int(which)))

    def skyviewLUT_tensor(self):
        """The sky-view LUT memory the C library owns, aliased (zero copy) as a torch float32 tensor
        [height, width, 4] through __cuda_array_interface__."""
        im = self.skyviewLUT()
        return _alias_tensor(im.data, (im.height, im.width, 4))

    def recordComposite(self, cmd, sceneTexture, drawRect, gbuffer, shadowMaps, atmosphereIndex, atmospheres, viewCameraIndex,
                        cameras, sunLightIndex, lights, tile=None):
        st = sceneTexture.abi()
        check(lib().szg_skyview_record_composite(
            self._h, _stream_handle(cmd), C.byref(st), drawRect, C.byref(tile) if tile is not None else None,
            C.byref(gbuffer), C.byref(shadowMaps) if shadowMaps is not None else None, int(atmosphereIndex),
            C.c_void_p(atmospheres.deviceAddress()), int(viewCameraIndex), C.c_void_p(cameras.deviceAddress()),
            int(sunLightIndex), C.c_void_p(lights.deviceAddress())))

    def recordMultiScatterLUT(self, cmd, atmosphereIndex, atmospheres):
        """Extension (no reference counterpart, abi.h): multi-scattering LUT, one wavefront per texel."""
        check(lib().szg_skyview_record_multiscatter_lut(self._h, _stream_handle(cmd), int(atmosphereIndex),
                                                        C.c_void_p(atmospheres.deviceAddress())))

    def multiScatterLUT(self):
        return self._lut(lib().szg_skyview_multiscatter_lut)

    def recordAerialLUT(self, cmd, atmosphereIndex, atmospheres, viewCameraIndex, cameras, maxDistanceMm):
        """Extension (no reference counterpart, abi.h): aerial-perspective froxel LUT, exact texel values."""
        check(lib().szg_skyview_record_aerial_lut(self._h, _stream_handle(cmd), int(atmosphereIndex),
                                                  C.c_void_p(atmospheres.deviceAddress()), int(viewCameraIndex),
                                                  C.c_void_p(cameras.deviceAddress()), C.c_float(maxDistanceMm)))

    def recordCompositeFast(self, cmd, sceneTexture, drawRect, gbuffer, shadowMaps, atmosphereIndex, atmospheres,
                            viewCameraIndex, cameras, sunLightIndex, lights, tile=None):
        """Extension: APPROXIMATE composite (aerial perspective from the froxel LUT). Never part of the parity frame."""
        st = sceneTexture.abi()
        check(lib().szg_skyview_record_composite_fast(
            self._h, _stream_handle(cmd), C.byref(st), drawRect, C.byref(tile) if tile is not None else None,
            C.byref(gbuffer), C.byref(shadowMaps) if shadowMaps is not None else None, int(atmosphereIndex),
            C.c_void_p(atmospheres.deviceAddress()), int(viewCameraIndex), C.c_void_p(cameras.deviceAddress()),
            int(sunLightIndex), C.c_void_p(lights.deviceAddress())))

    def aerialLUT(self):
        lum, tr = abi.Image(), abi.Image()
        check(lib().szg_skyview_aerial_lut(self._h, C.byref(lum), C.byref(tr)))
        return lum, tr

    def _lut(self, getter):
        im = abi.Image()
        check(getter(self._h, C.byref(im)))
        return im

    def transmittanceLUT(self):
        return self._lut(lib().szg_skyview_transmittance_lut)

    def skyviewLUT(self):
        return self._lut(lib().szg_skyview_skyview_lut)

    def download_lut(self, im):
        raw = _memcpy2d_from(im, im.width * 16, im.height)
        return raw.cpu().numpy().view(np.float32).reshape(im.height, im.width, 4)

    def upload_lut(self, im, array):
        arr = np.ascontiguousarray(array, dtype=np.float32).reshape(im.height, im.width * 4)
        src = torch.from_numpy(arr.view(np.uint8).reshape(im.height, -1)).cuda()
        _memcpy2d_to(im, src, im.height)

    def destroy(self):
        if self._h:
            lib().szg_skyview_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
