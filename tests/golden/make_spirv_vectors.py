#!/usr/bin/env python3
"""Writes tests/golden/spirv_vectors.npz: outputs of the reference's own compute shaders - the COMMITTED SPIR-V of
transmittance_LUT.comp, skyview_LUT.comp, lights.comp and camera.comp, executed literally by tests/golden/spirv_interp.py on
seeded inputs - together with those inputs. tests/test_spirv_pin.py then requires the CPU oracle (its contraction rule off:
oracle/libszg_oracle_literal.so) to reproduce every one of them BIT FOR BIT. That pins the oracle's dataflow - every
operation, constant, branch and its order - against the reference's binaries instead of against a reading of the GLSL.

    python tests/golden/make_spirv_vectors.py            (needs /root/reference; ~ 1.5 min)

What the vectors do NOT pin (stated in DESIGN.md 2): the values of the implementation-defined GLSL built-ins (exp, pow, sin,
cos, asin, acos: both sides use include/szg/fpmath.h), the fixed-function texture filter (both sides: binary32 weights,
clamp-to-edge, the model below), the UNORM16 conversion of the scene colour (RTE) and where a real GPU contracts a * b + c.
Those are the freedoms Vulkan leaves to the implementation; everything the shader itself states is pinned.

Sampler / image models (the interpreter's `Image` objects; skyview.cpp:199-207, gbuffer.cpp:104-109, shadowpass.cpp:29-35):
    LUTs            LINEAR, CLAMP_TO_EDGE, no mips: u = s * W - 0.5, weights (1-a)(1-b), a(1-b), (1-a)b, ab in binary32,
                    sum ((w00 t00 + w10 t10) + w01 t01) + w11 t11
    G-buffer, depth NEAREST, CLAMP_TO_EDGE: texel floor(s * W) clamped; RGBA16F planes widen exactly
    shadow maps     NEAREST, CLAMP_TO_BORDER (0)
    scene colour    rgba16 UNORM storage image: store = RTE(clamp(x, 0, 1) * 65535), load = q / 65535
"""
import ctypes as C
import hashlib
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REFERENCE = os.environ.get("SZG_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "spirv_vectors.npz")
SHADERS = {
    "transmittance": "shaders/atmosphere/transmittance_LUT.comp.spv",
    "skyview": "shaders/atmosphere/skyview_LUT.comp.spv",
    "lights": "shaders/deferred/lights.comp.spv",
    "camera": "shaders/atmosphere/camera.comp.spv",
    "oetf_srgb": "shaders/transfer/oetf_srgb.comp.spv",
    "oetf_pure_gamma": "shaders/transfer/oetf_pure_gamma.comp.spv",
    "offscreen_vert": "shaders/deferred/offscreen.vert.spv",
    "offscreen_frag": "shaders/deferred/offscreen.frag.spv",
    "depthpass_vert": "shaders/offscreenpass/depthpass.vert.spv",
}
F32 = np.float32


def available():
    return all(os.path.exists(os.path.join(REFERENCE, p)) for p in SHADERS.values())


class Builtins:
    """The implementation-defined built-ins: the pinned algorithms both sides share (include/szg/fpmath.h)."""

    def __init__(self):
        from oracle import binding as ob

        self._ob = ob

    def _e(self, fn, x, y=None):
        xs = np.array([x], np.float32)
        ys = None if y is None else np.array([y], np.float32)
        return F32(self._ob.builtin_eval(fn, xs, ys)[0])

    def exp(self, x):
        return self._e(0, x)

    def pow(self, x, y):
        return self._e(1, x, y)

    def sin(self, x):
        return self._e(2, x)

    def cos(self, x):
        return self._e(3, x)

    def asin(self, x):
        return self._e(4, x)

    def acos(self, x):
        return self._e(5, x)

    @staticmethod
    def fmin(x, y):  # fminf: the operand that is a number
        if x != x:
            return y
        if y != y:
            return x
        return x if x < y else y

    @staticmethod
    def fmax(x, y):
        if x != x:
            return y
        if y != y:
            return x
        return x if x > y else y


def _images():
    from tests.golden import spirv_interp as si

    class Store(si.Image):
        """rgba32f storage image that is only written (the LUT kernels' output)."""

        def __init__(self, w, h):
            self.w, self.h, self.texels = w, h, {}

        def size(self):
            return (self.w, self.h)

        def store(self, x, y, t):
            self.texels[(x, y)] = [F32(c) for c in t]

    class Unorm16(si.Image):
        """rgba16 storage image over a [h, w, 4] uint16 array; writes are recorded, not applied (one invocation = one pixel)."""

        def __init__(self, array):
            self.a, self.written, self.written_f = array, {}, {}

        def size(self):
            return (self.a.shape[1], self.a.shape[0])

        def fetch(self, x, y):
            return [F32(q) / F32(65535.0) for q in self.a[y, x]]

        def store(self, x, y, t):
            self.written_f[(x, y)] = [F32(c) for c in t]
            q = []
            for c in t:
                c = Builtins.fmin(Builtins.fmax(F32(c), F32(0)), F32(1))
                q.append(int(np.rint(c * F32(65535.0))))
            self.written[(x, y)] = q

    class Nearest(si.Image):
        """NEAREST / CLAMP_TO_EDGE over a [h, w, c] (or [h, w]) array of float16 / float32."""

        def __init__(self, array):
            self.a = array

        def size(self):
            return (self.a.shape[1], self.a.shape[0])

        def sample(self, sampler, u, v):
            h, w = self.a.shape[:2]
            x = int(np.floor(F32(u) * F32(w)))
            y = int(np.floor(F32(v) * F32(h)))
            x = min(max(x, 0), w - 1)
            y = min(max(y, 0), h - 1)
            t = self.a[y, x]
            if self.a.ndim == 2:
                return [F32(t), F32(0), F32(0), F32(1)]
            return [F32(c) for c in t]

    class Border(si.Image):
        """NEAREST / CLAMP_TO_BORDER(0) over a [h, w] float32 depth map."""

        def __init__(self, array):
            self.a = array

        def size(self):
            return (self.a.shape[1], self.a.shape[0])

        def sample(self, sampler, u, v):
            h, w = self.a.shape
            fx = np.floor(F32(u) * F32(w))
            fy = np.floor(F32(v) * F32(h))
            if not (fx >= 0) or not (fy >= 0) or not (fx < w) or not (fy < h):
                return [F32(0), F32(0), F32(0), F32(0)]
            return [F32(self.a[int(fy), int(fx)]), F32(0), F32(0), F32(1)]

    class Linear(si.Image):
        """LINEAR / CLAMP_TO_EDGE over a [h, w, 4] float32 array, binary32 weights."""

        def __init__(self, array):
            self.a = array

        def size(self):
            return (self.a.shape[1], self.a.shape[0])

        def sample(self, sampler, s, t):
            h, w = self.a.shape[:2]
            one, half = F32(1), F32(0.5)
            u = F32(s) * F32(w) - half
            v = F32(t) * F32(h) - half
            fu, fv = np.floor(u), np.floor(v)
            a, b = u - fu, v - fv
            if not (np.isfinite(fu) and np.isfinite(fv)):
                raise RuntimeError("non-finite texture coordinate: not part of the vectors")
            i0, j0 = int(fu), int(fv)
            i1, j1 = i0 + 1, j0 + 1
            i0, i1 = min(max(i0, 0), w - 1), min(max(i1, 0), w - 1)
            j0, j1 = min(max(j0, 0), h - 1), min(max(j1, 0), h - 1)
            w00, w10, w01, w11 = (one - a) * (one - b), a * (one - b), (one - a) * b, a * b
            out = []
            for c in range(4):
                t00, t10, t01, t11 = (F32(self.a[j0, i0, c]), F32(self.a[j0, i1, c]), F32(self.a[j1, i0, c]), F32(self.a[j1, i1, c]))
                out.append(((w00 * t00 + w10 * t10) + w01 * t01) + w11 * t11)
            return out

    return Store, Unorm16, Nearest, Border, Linear


def _texture_class(builtins):
    from tests.golden import spirv_interp as si

    class Texture(si.Image):
        """RGBA8 material map: LINEAR, REPEAT, one level, sRGB-encoded texels decoded before filtering (include/szg/raster.h
        "textures"; material.cpp samplers): binary32 weights, sum ((w00 t00 + w10 t10) + w01 t01) + w11 t11."""

        def __init__(self, array, srgb):
            self.a, self.srgb = array, srgb

        def size(self):
            return (self.a.shape[1], self.a.shape[0])

        def _decode(self, byte):
            c = F32(byte) / F32(255.0)
            if not self.srgb:
                return c
            if c <= F32(0.04045):
                return c / F32(12.92)
            return builtins.pow((c + F32(0.055)) / F32(1.055), F32(2.4))

        @staticmethod
        def _wrap(f, n):
            fn = F32(n)
            m = f - fn * np.floor(f / fn)
            i = int(m)
            return 0 if (i >= n or i < 0) else i

        def sample(self, sampler, s, t):
            h, w = self.a.shape[:2]
            one, half = F32(1), F32(0.5)
            u = F32(s) * F32(w) - half
            v = F32(t) * F32(h) - half
            fu, fv = np.floor(u), np.floor(v)
            a, b = u - fu, v - fv
            i0, j0 = self._wrap(fu, w), self._wrap(fv, h)
            i1 = 0 if i0 + 1 == w else i0 + 1
            j1 = 0 if j0 + 1 == h else j0 + 1
            w00, w10, w01, w11 = (one - a) * (one - b), a * (one - b), (one - a) * b, a * b
            out = []
            for c in range(3):
                t00, t10, t01, t11 = (self._decode(self.a[j0, i0, c]), self._decode(self.a[j0, i1, c]),
                                      self._decode(self.a[j1, i0, c]), self._decode(self.a[j1, i1, c]))
                out.append(((w00 * t00 + w10 * t10) + w01 * t01) + w11 * t11)
            return out + [F32(1)]

    return Texture


def pack_block(m, struct_type, values):
    """Bytes of an explicitly laid out block (push constants) from {member name: int}, using the module's own offsets."""
    from tests.golden import spirv_interp as si

    t = m.types[struct_type]
    raw = bytearray(128)
    end = 0
    for i, mt in enumerate(t.members):
        name = m.member_names[(struct_type, i)]
        off = m.member_decor[(struct_type, i)][si.DEC_OFFSET][0]
        kind = m.types[mt].kind
        size = 8 if kind == "pointer" or (kind == "int" and m.types[mt].width == 64) else 4
        raw[off: off + size] = int(values[name]).to_bytes(size, "little")
        end = max(end, off + size)
    return bytes(raw[:end])


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def generate(log=print):
    from oracle import binding as ob
    from syzygy_amd import abi
    from tests import util
    from tests.golden import spirv_interp as si

    Store, Unorm16, Nearest, Border, Linear = _images()
    mods = {k: si.Module(os.path.join(REFERENCE, p)) for k, p in SHADERS.items()}
    for k, m in mods.items():
        assert m.no_contraction == 0 and (m.local_size == (16, 16, 1) or k.endswith(("_vert", "_frag"))), k
    builtins = Builtins()
    rng = np.random.default_rng(0x5A2C)
    out = {}
    t_start = time.time()

    def bindings(m):
        return {m.names[g]: (m.decor[g][si.DEC_SET][0], m.decor[g][si.DEC_BINDING][0]) for g, (pt, sc) in m.globals.items()
                if sc == si.SC_UNIFORM_CONSTANT}

    # ------------------------------------------------------------------ scenarios (parameter blocks) -------------------
    scenarios = []
    def unusual(a):  # every term of sampleExtinction non-zero (the Earth defaults have two zero coefficient vectors)
        a.absorptionRayleighPerMegameter[:] = [0.7, 1.3, 2.9]
        a.scatteringOzonePerMegameter[:] = [0.3, 0.2, 0.9]

    for elevation, cam_edit, atm_edit in ((35.0, None, None), (4.0, None, None), (-2.5, None, None), (62.0, "high", None),
                                          (18.0, "tilt", unusual), (80.0, "low", None)):
        from syzygy_amd import scene

        camera = scene.default_camera()
        if cam_edit == "high":
            camera.cameraPosition[1] = -2500.0  # +y is down in the reference's world space: 2.5 km up, every pixel is sky
        elif cam_edit == "tilt":
            camera.eulerAngles[0] += 0.35
            camera.eulerAngles[1] -= 0.6
        elif cam_edit == "low":
            camera.cameraPosition[1] *= 0.25
            camera.eulerAngles[0] -= 0.2
        scenarios.append(util.Inputs(48, 27, elevation_degrees=elevation, spots=5, camera=camera, atmosphere_edit=atm_edit))
    # The transmittance LUT keeps the reference's extent: common.glinl:13-14 compiles 512 x 128 into the coordinate maps of
    # every shader that samples it (the oracle takes them from the image, which is the same thing at this extent only). The
    # fixture carries its SHA-256, not its 1 MiB: the test recomputes it with the oracle, whose texels the first group of
    # vectors pins. The sky-view LUT is sampled through textureSize(): a small one travels with the fixture.
    TW, TH, SW, SH = 512, 128, 96, 48
    tluts = {}

    with ob.use_literal():
        # -------------------------------------------------------------- transmittance_LUT.comp --------------------------
        m = mods["transmittance"]
        b = bindings(m)
        W, H = 512, 128
        texels = [(0, 0), (511, 127), (511, 0), (0, 127), (255, 64)] + [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(43)]
        blocks, coords, results = [], [], []
        for k, inp in enumerate(scenarios[:2]):
            mem = si.Memory()
            pad = mem.alloc(b"\xff" * 128)  # atmosphere 0: never read
            addr = mem.alloc(b"\xff" * 128 + bytes(inp.atm))  # the block in use is atmosphere 1 of its buffer
            image = Store(W, H)
            it = si.Interpreter(m, mem, builtins, struct.pack("<QII", addr, 1, 0), {b["transmittance_LUT"]: image})
            for (x, y) in texels[: 48 if k == 0 else 16]:
                it.run((x, y, 0))
                blocks.append(np.frombuffer(bytes(inp.atm), np.uint8))
                coords.append((x, y))
                results.append(image.texels[(x, y)])
            it.run((W, 5, 0))  # outside the image: must return without a store
            assert (W, 5) not in image.texels
            del pad
        out.update(transmittance_atm=np.array(blocks), transmittance_xy=np.array(coords, np.int32), transmittance_extent=np.array([W, H]),
                   transmittance_texel=_bits(results))
        log(f"transmittance: {len(results)} texels, {time.time() - t_start:.0f} s")

        # -------------------------------------------------------------- skyview_LUT.comp --------------------------------
        m = mods["skyview"]
        b = bindings(m)
        W, H = 2048, 1024
        sky_inputs = {"atm": [], "cam": [], "tlut": [], "xy": [], "texel": []}
        for k, inp in enumerate(scenarios):
            tlut = tluts[k] = ob.transmittance_lut(inp.atm, TW, TH, threads=8)
            out[f"tlut_sha256_{k}"] = np.frombuffer(hashlib.sha256(tlut.tobytes()).digest(), np.uint8)
            mem = si.Memory()
            a_atm = mem.alloc(bytes(inp.atm))
            a_cam = mem.alloc(b"\xff" * C.sizeof(abi.CameraPacked) * 2 + bytes(inp.cam))  # camera 2 of its buffer
            image = Store(W, H)
            it = si.Interpreter(m, mem, builtins, struct.pack("<QQIIII", a_atm, a_cam, 0, 2, 0, 0),
                                {b["skyview_LUT"]: image, b["transmittance_LUT"]: Linear(tlut)})
            picks = [(0, 0), (W - 1, H - 1), (W // 2, H // 2 - 1), (W // 2, H // 2), (3, H // 2 + 7)]
            picks += [(int(rng.integers(W)), int(rng.integers(H))) for _ in range(19 if k == 0 else 7)]
            for (x, y) in picks:
                it.run((x, y, 0))
                sky_inputs["atm"].append(np.frombuffer(bytes(inp.atm), np.uint8))
                sky_inputs["cam"].append(np.frombuffer(bytes(inp.cam), np.uint8))
                sky_inputs["tlut"].append(k)
                sky_inputs["xy"].append((x, y))
                sky_inputs["texel"].append(image.texels[(x, y)])
        out.update(skyview_atm=np.array(sky_inputs["atm"]), skyview_cam=np.array(sky_inputs["cam"]),
                   skyview_tlut=np.array(sky_inputs["tlut"], np.int32), skyview_xy=np.array(sky_inputs["xy"], np.int32),
                   skyview_extent=np.array([W, H]), skyview_texel=_bits(sky_inputs["texel"]))
        log(f"skyview: {len(sky_inputs['texel'])} texels, {time.time() - t_start:.0f} s")

        # -------------------------------------------------------------- lights.comp + camera.comp ------------------------
        ml, mc = mods["lights"], mods["camera"]
        bl, bc = bindings(ml), bindings(mc)
        for k, inp in enumerate(scenarios):
            Wf, Hf = inp.width, inp.height
            frame = ob.HostFrame(Wf, Hf)
            ob.gbuffer_fill(frame, inp.rect, None, inp.cam, inp.synthetic.fill, threads=8)
            # shadow maps: random occluder depths in EVERY slot (sun, moon, spots), odd extents
            nslots = 2 + inp.spot_count
            maps = {s: rng.random((int(rng.integers(8, 40)), int(rng.integers(8, 40))), dtype=np.float32) for s in range(nslots)}
            maps[3] = (maps[3] > 0.5).astype(np.float32) * F32(0.9999)
            slot_arrays = [maps[s] for s in range(nslots)]
            tlut = tluts[k]
            slut = ob.skyview_lut(inp.atm, inp.cam, tlut, SW, SH, threads=8)
            out[f"slut_{k}"] = slut
            for name, plane in frame.planes().items():
                out[f"gbuffer_{name}_{k}"] = plane.copy()
            out[f"depth_{k}"] = frame.depth.copy()
            out[f"atm_{k}"] = np.frombuffer(bytes(inp.atm), np.uint8)
            out[f"cam_{k}"] = np.frombuffer(bytes(inp.cam), np.uint8)
            out[f"dirs_{k}"] = np.frombuffer(bytes(inp.dirs), np.uint8)
            out[f"spots_{k}"] = np.frombuffer(bytes(inp.spots), np.uint8)
            for s, a in maps.items():
                out[f"shadow_{k}_{s}"] = a

            mem = si.Memory()
            a_atm = mem.alloc(bytes(inp.atm))
            a_cam = mem.alloc(bytes(inp.cam))
            a_dir = mem.alloc(bytes(inp.dirs))
            a_spot = mem.alloc(bytes(inp.spots))
            planes = frame.planes()
            gb = {"gbufferDiffuse": Nearest(planes["diffuse"]), "gbufferSpecular": Nearest(planes["specular"]),
                  "gbufferNormal": Nearest(planes["normal"]), "gbufferWorldPosition": Nearest(planes["worldPosition"]),
                  "gbufferOcclusionRoughnessMetallic": Nearest(planes["occlusionRoughnessMetallic"])}
            shadows = [Border(a) for a in slot_arrays]

            # pixels: a lattice over the frame + the corners
            corners = {(0, 0), (Wf - 1, Hf - 1), (Wf - 1, 0), (0, Hf - 1)}
            if k == 0:
                pixels = sorted({(x, y) for y in range(0, Hf, 2) for x in range(k % 2, Wf, 3)} | corners)
            else:
                pixels = sorted({(x, y) for y in range(1, Hf, 4) for x in range(k % 5, Wf, 5)} | corners)

            # lights.comp: directional lights [sun, moon] with the sun skipped, the spots, every shadow slot bound
            colour0 = np.zeros((Hf, Wf, 4), np.uint16)
            image = Unorm16(colour0)
            desc = {bl[n]: v for n, v in gb.items()}
            desc.update({bl["image"]: image, bl["shadowMaps"]: shadows, bl["shadowMapSampler"]: "shadowMapSampler"})
            pcl = struct.pack("<QIIQQIIIIffff", a_cam, 0, 0, a_dir, a_spot, 2, inp.spot_count, 1, 0, 0.0, 0.0, float(Wf), float(Hf))
            it = si.Interpreter(ml, mem, builtins, pcl, desc)
            lit_f, lit_q, lit_w = [], [], []
            for (x, y) in pixels:
                it.run((x, y, 0))
                # background texels return without a store (lights.comp:126-129): they keep the clear colour (0, 0, 0, 1) of
                # the vkCmdClearColorImage in front of the dispatch (deferred.cpp:715-717)
                lit_w.append((x, y) in image.written)
                lit_f.append(image.written_f.get((x, y), [F32(0), F32(0), F32(0), F32(1)]))
                lit_q.append(image.written.get((x, y), [0, 0, 0, 65535]))
            out[f"lights_xy_{k}"] = np.array(pixels, np.int32)
            out[f"lights_stored_{k}"] = np.array(lit_w)
            out[f"lights_value_{k}"] = _bits(lit_f)
            out[f"lights_unorm_{k}"] = np.array(lit_q, np.uint16)
            log(f"lights scenario {k}: {len(pixels)} pixels, {time.time() - t_start:.0f} s")

            # the oracle's lights pass produces the prior colour camera.comp reads back (checked against the vectors above
            # by the test; here it only has to exist)
            shadow_images = (abi.Image * nslots)()
            for s in range(nslots):
                shadow_images[s] = ob.host_image(slot_arrays[s], abi.SZG_FORMAT_D32_SFLOAT)
            sm = abi.ShadowMaps(nslots, 0, C.cast(shadow_images, C.POINTER(abi.Image)))
            ob.lights(frame, inp.rect, None, sm, inp.cam, inp.dirs, 2, 1, inp.spots, inp.spot_count, threads=8)
            out[f"prior_{k}"] = frame.color.copy()

            # camera.comp
            image = Unorm16(frame.color.copy())
            desc = {bc[n]: v for n, v in gb.items()}
            desc.update({bc["image"]: image, bc["fragmentDepth"]: Nearest(frame.depth), bc["skyview_LUT"]: Linear(slut),
                         bc["transmittance_LUT"]: Linear(tlut), bc["shadowMaps"]: shadows, bc["shadowMapSampler"]: "shadowMapSampler"})
            pcc = struct.pack("<QQIIIIIIIIQII", a_atm, a_cam, 0, 0, Wf, Hf, 0, 0, Wf, Hf, a_dir, 0, 0)
            it = si.Interpreter(mc, mem, builtins, pcc, desc)
            cam_f, cam_q = [], []
            for (x, y) in pixels:
                it.run((x, y, 0))
                cam_f.append(image.written_f[(x, y)])
                cam_q.append(image.written[(x, y)])
            out[f"camera_xy_{k}"] = np.array(pixels, np.int32)
            out[f"camera_value_{k}"] = _bits(cam_f)
            out[f"camera_unorm_{k}"] = np.array(cam_q, np.uint16)
            log(f"camera scenario {k}: {len(pixels)} pixels, {time.time() - t_start:.0f} s")
    # ------------------------------------------------------------------ oetf_srgb.comp / oetf_pure_gamma.comp --------------
    # in place on the rgba16 scene colour: EVERY 16-bit code value in the red channel (green and blue walk the codes in other
    # orders, alpha must pass through), one invocation per pixel of a 256 x 256 image
    codes = np.arange(65536, dtype=np.uint32)
    colour = np.stack([codes, 65535 - codes, (codes * 7919 + 13) % 65536, (codes * 31) % 65536], axis=1).astype(np.uint16).reshape(256, 256, 4)
    out["oetf_input"] = colour
    for name in ("oetf_srgb", "oetf_pure_gamma"):
        m = mods[name]
        image = Unorm16(colour)
        it = si.Interpreter(m, si.Memory(), builtins, b"", {bindings(m)["image"]: image})
        for y in range(256):
            for x in range(256):
                it.run((x, y, 0))
        it.run((256, 0, 0))  # outside the image: no store
        assert (256, 0) not in image.written
        out[name] = np.array([image.written[(x, y)] for y in range(256) for x in range(256)], np.uint16).reshape(256, 256, 4)
        log(f"{name}: 65536 pixels, {time.time() - t_start:.0f} s")
    # ------------------------------------------------------------------ offscreen.vert / depthpass.vert / offscreen.frag -----
    Texture = _texture_class(builtins)
    nv, ni = 24, 3
    vertices = (abi.VertexPacked * nv)()
    for v in vertices:
        v.position[:] = [float(F32(x)) for x in rng.uniform(-3, 3, 3)]
        n = rng.normal(size=3)
        v.normal[:] = [float(F32(x)) for x in n / np.linalg.norm(n) * rng.uniform(0.5, 2.0)]  # not unit: the shader normalises
        v.uv_x, v.uv_y = float(F32(rng.uniform(-1, 2))), float(F32(rng.uniform(-1, 2)))
        v.color[:] = [1.0, 1.0, 1.0, 1.0]
    models = (abi.Mat4 * ni)()
    mits = (abi.Mat4 * ni)()
    for i in range(ni):
        a = np.eye(4)
        a[:3, :3] = rng.normal(size=(3, 3)) + 2.0 * np.eye(3)
        a[:3, 3] = rng.uniform(-5, 5, 3)
        models[i].m[:] = [float(F32(x)) for x in a.T.ravel()]  # column-major
        mits[i].m[:] = [float(F32(x)) for x in np.linalg.inv(a).T.T.ravel()]
    cam = scenarios[0].cam
    light = abi.Mat4()
    light.m[:] = [float(F32(x)) for x in (rng.normal(size=(4, 4)) * 0.1 + np.eye(4)).T.ravel()]
    mem = si.Memory()
    a_vert, a_model, a_mit = mem.alloc(bytes(vertices)), mem.alloc(bytes(models)), mem.alloc(bytes(mits))
    a_cam = mem.alloc(b"\xff" * C.sizeof(abi.CameraPacked) + bytes(cam))  # camera 1 of its buffer
    a_light = mem.alloc(bytes(light))
    mv, md, mf = mods["offscreen_vert"], mods["depthpass_vert"], mods["offscreen_frag"]

    def block_type(m):
        (gid,) = [g for g, (pt, sc) in m.globals.items() if sc == si.SC_PUSH_CONSTANT]
        return m.types[m.globals[gid][0]].pointee

    itv = si.Interpreter(mv, mem, builtins, pack_block(mv, block_type(mv), dict(vertexBuffer=a_vert, modelBuffer=a_model,
                         modelInverseTransposeBuffer=a_mit, cameraBuffer=a_cam, cameraIndex=1)), {})
    itd = si.Interpreter(md, mem, builtins, pack_block(md, block_type(md), dict(vertexBuffer=a_vert, modelBuffer=a_model,
                         projViewBuffer=a_light, projViewIndex=0)), {})
    vs, ds = [], []
    for inst in range(ni):
        for vi in range(nv):
            io = {si.BUILTIN_VERTEX_INDEX: vi, si.BUILTIN_INSTANCE_INDEX: inst}
            itv.run(inputs=io)
            o = itv.outputs
            vs.append(list(o["gl_PerVertex"][0][0]) + list(o["outWorldPosition"][0]) + list(o["outNormal"][0]) + list(o["outTexCoord"][0]))
            itd.run(inputs=io)
            ds.append(list(itd.outputs["gl_PerVertex"][0][0]))
    out.update(raster_vertices=np.frombuffer(bytes(vertices), np.uint8), raster_models=np.frombuffer(bytes(models), np.uint8),
               raster_mits=np.frombuffer(bytes(mits), np.uint8), raster_camera=np.frombuffer(bytes(cam), np.uint8),
               raster_light=np.frombuffer(bytes(light), np.uint8), offscreen_vert=_bits(vs), depthpass_vert=_bits(ds))
    log(f"vertex stages: {len(vs)} + {len(ds)} invocations, {time.time() - t_start:.0f} s")

    tex = {"color": rng.integers(0, 256, (5, 7, 4), dtype=np.uint8), "normal": rng.integers(0, 256, (4, 4, 4), dtype=np.uint8),
           "ORM": rng.integers(0, 256, (3, 6, 4), dtype=np.uint8)}
    bf = bindings(mf)
    itf = si.Interpreter(mf, si.Memory(), builtins, b"", {bf["color"]: Texture(tex["color"], True), bf["normal"]: Texture(tex["normal"], False),
                                                          bf["ORM"]: Texture(tex["ORM"], False)})
    frag_in, frag_out = [], []
    for _ in range(160):
        world = [F32(x) for x in rng.uniform(-20, 20, 3)]
        n = rng.normal(size=3)
        normal = [F32(x) for x in n / np.linalg.norm(n)]
        uv = [F32(x) for x in rng.uniform(-2, 3, 2)]
        dwx = [F32(x) for x in rng.normal(size=3) * 0.05]
        dwy = [F32(x) for x in rng.normal(size=3) * 0.05]
        dux = [F32(x) for x in rng.normal(size=2) * 0.01]
        duy = [F32(x) for x in rng.normal(size=2) * 0.01]
        # cotangentFrame differentiates p = -inWorldPosition (offscreen.frag:54, :65): the fixed function hands it the negated
        # differences
        itf.run(inputs={"inWorldPosition": world, "inNormal": normal, "inTexCoord": uv},
                derivatives=[[-c for c in dwx], [-c for c in dwy], dux, duy])
        assert not itf.derivatives
        o = itf.outputs
        frag_in.append(world + normal + uv + dwx + dwy + dux + duy)
        frag_out.append(list(o["outWorldPosition"][0]) + list(o["outNormal"][0]) + list(o["outDiffuseColor"][0]) +
                        list(o["outSpecularColor"][0]) + list(o["outOcclusionRoughnessMetallic"][0]))
    out.update(frag_in=_bits(frag_in), offscreen_frag=_bits(frag_out), frag_tex_color=tex["color"], frag_tex_normal=tex["normal"],
               frag_tex_orm=tex["ORM"])
    log(f"fragment stage: {len(frag_out)} invocations, {time.time() - t_start:.0f} s")
    out["scenarios"] = np.array(len(scenarios))
    out["lut_extents"] = np.array([TW, TH, SW, SH])
    return out


if __name__ == "__main__":
    if not available():
        raise SystemExit(f"{REFERENCE}: the committed SPIR-V of the reference is not here; nothing written")
    vectors = generate()
    np.savez_compressed(OUT, **vectors)
    print(f"wrote {OUT}: {os.path.getsize(OUT) / 1024:.0f} KiB")
