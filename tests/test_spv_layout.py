"""The DATA ABI pinned against the reference's own binaries.

tests/golden/spv_reflection.json is the output of the reference's vendored spirv-reflect (compiled where it lies into
oracle/_ref/reflect_spv) on the committed SPIR-V of the hot path's shaders: push-constant member offsets and sizes, the full
layouts of the structs behind the buffer references (Atmosphere, Camera, LightDirectional, LightSpot, Vertex), workgroup size,
storage-image formats and descriptor bindings. The engine validates its own C++ structs against this same reflection at
pipeline creation (deferred.cpp:30-62, pipelines.cpp:609-624).

What is asserted: include/szg/abi.h and include/szg/raster.h (measured with offsetof / sizeof by a C program compiled here,
not by reading the header) and syzygy_amd/abi.py's ctypes mirrors have exactly those offsets and sizes.

This pins LAYOUT. The arithmetic of the same binaries is pinned by tests/test_spirv_pin.py.
"""
import ctypes
import json
import os
import subprocess

import pytest

from syzygy_amd import abi
from tests.golden import make_spv_reflection as gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "spv_reflection.json")

# SpvImageFormat (SPIR-V specification, unified1): the two storage formats the path declares
SPV_FORMAT_RGBA32F = 1
SPV_FORMAT_RGBA16 = 10
# SpvReflectDescriptorType == VkDescriptorType
SAMPLER, SAMPLED_IMAGE, STORAGE_IMAGE = 0, 2, 3
COMBINED = 1

# GLSL block / struct -> the C type of include/szg that mirrors it, and GLSL member -> C member where the names differ
# (lights.comp:41-57 calls its buffer references directionalLights / spotLights, deferred.hpp:82-98 ...Buffer)
C_TYPES = {
    "Atmosphere": "szg_atmosphere_packed",
    "Camera": "szg_camera_packed",
    "LightDirectional": "szg_directional_light_packed",
    "LightSpot": "szg_spot_light_packed",
    "Vertex": "szg_vertex_packed",
}
PUSH_CONSTANTS = {
    "transmittance_LUT.comp.spv": ("szg_pc_transmittance", {}),
    "skyview_LUT.comp.spv": ("szg_pc_skyview", {}),
    "camera.comp.spv": ("szg_pc_composite", {}),
    "lights.comp.spv": ("szg_pc_lights", {"directionalLights": "directionalLightsBuffer", "spotLights": "spotLightsBuffer"}),
}


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def _structs(golden):
    """Every buffer-reference struct found below any push-constant block: name -> (padded size, array stride, members)."""
    found = {}

    def walk(member):
        if member["type_name"] in C_TYPES:
            entry = (member["padded_size"], member["type_array_stride"],
                     [(m["name"], m["offset"], m["size"]) for m in member["members"]])
            assert found.setdefault(member["type_name"], entry) == entry, member["type_name"]  # same layout in every shader
        for m in member["members"]:
            walk(m)

    for shader in golden.values():
        for block in shader["push_constants"]:
            walk(block)
    return found


def _measure_header(queries):
    """offsetof / sizeof of (type, member-or-None) pairs, measured by compiling a C program against the public headers."""
    lines = ["#include <stddef.h>", "#include <stdio.h>", '#include "szg/abi.h"', '#include "szg/raster.h"',
             "#define MEMBER_SIZE(t, m) sizeof(((t*)0)->m)", "int main(void) {"]
    for t, m in queries:
        if m is None:
            lines.append(f'  printf("%zu\\n", sizeof({t}));')
        else:
            lines.append(f'  printf("%zu %zu\\n", offsetof({t}, {m}), MEMBER_SIZE({t}, {m}));')
    lines += ["  return 0;", "}"]
    return lines


def test_committed_reflection_is_what_the_reference_binaries_say(golden):
    """Where the reference checkout and oracle/_ref/reflect_spv exist (the build container), re-reflect and compare with the
    committed fixture; elsewhere (the GPU box) the fixture stands alone."""
    live = gen.reflect()
    if live is None:
        pytest.skip("reference checkout / oracle/_ref not present here: the committed fixture is used as it is")
    assert live == golden


def test_headers_match_the_reflected_layouts(golden, tmp_path):
    structs = _structs(golden)
    assert sorted(structs) == sorted(C_TYPES)
    queries, expect = [], []
    for glsl, (padded, stride, members) in structs.items():
        queries.append((C_TYPES[glsl], None))
        expect.append(str(stride))  # the array stride of the buffer reference == sizeof of the packed C struct
        assert stride == padded and stride % 16 == 0
        for name, offset, size in members:
            queries.append((C_TYPES[glsl], name))
            expect.append(f"{offset} {size}")
    for shader, (ctype, rename) in PUSH_CONSTANTS.items():
        (block,) = golden[shader]["push_constants"]
        queries.append((ctype, None))
        expect.append(str(block["padded_size"]))
        for m in block["members"]:
            queries.append((ctype, rename.get(m["name"], m["name"])))
            expect.append(f"{m['offset']} {m['size']}")
    source = tmp_path / "layout.c"
    source.write_text("\n".join(_measure_header(queries)) + "\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(source), "-o", str(exe)], check=True)
    got = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")[:-1]
    assert len(got) == len(expect)
    wrong = [(q, e, g) for q, e, g in zip(queries, expect, got) if e != g]
    assert not wrong, wrong


def test_ctypes_mirrors_match_the_reflected_layouts(golden):
    structs = _structs(golden)
    py = {"Atmosphere": abi.AtmospherePacked, "Camera": abi.CameraPacked, "LightDirectional": abi.DirectionalLightPacked,
          "LightSpot": abi.SpotLightPacked}
    for glsl, cls in py.items():
        padded, stride, members = structs[glsl]
        assert ctypes.sizeof(cls) == stride, glsl
        for name, offset, size in members:
            field = getattr(cls, name)
            assert (field.offset, field.size) == (offset, size), (glsl, name)


def test_workgroup_formats_and_bindings(golden):
    compute = ["transmittance_LUT.comp.spv", "skyview_LUT.comp.spv", "camera.comp.spv", "lights.comp.spv", "oetf_srgb.comp.spv",
               "oetf_pure_gamma.comp.spv"]
    for name in compute:
        assert golden[name]["local_size"] == [16, 16, 1], name  # SURVEY §2b; computeDispatchCount pipelines.cpp:815-830

    def binding(shader, name):
        (b,) = [b for b in golden[shader]["bindings"] if b["name"] == name]
        return b

    # both LUTs are written as rgba32f storage images = SZG_FORMAT_RGBA32_SFLOAT, 16 B/texel (skyview.cpp:85,182)
    assert binding("transmittance_LUT.comp.spv", "transmittance_LUT")["image_format"] == SPV_FORMAT_RGBA32F
    assert binding("skyview_LUT.comp.spv", "skyview_LUT")["image_format"] == SPV_FORMAT_RGBA32F
    assert binding("transmittance_LUT.comp.spv", "transmittance_LUT")["descriptor_type"] == STORAGE_IMAGE
    # ... and read back through a combined sampler (LINEAR / CLAMP_TO_EDGE: skyview.cpp:199-207)
    assert binding("skyview_LUT.comp.spv", "transmittance_LUT")["descriptor_type"] == COMBINED
    assert binding("camera.comp.spv", "skyview_LUT")["descriptor_type"] == COMBINED
    # the scene colour is an rgba16 (UNORM) storage image in both passes = SZG_FORMAT_RGBA16_UNORM
    for shader in ("camera.comp.spv", "lights.comp.spv"):
        b = binding(shader, "image")
        assert (b["image_format"], b["descriptor_type"], b["set"], b["binding"]) == (SPV_FORMAT_RGBA16, STORAGE_IMAGE, 0, 0)
    assert abi.SZG_FORMAT_RGBA32_SFLOAT == 2 and abi.SZG_FORMAT_RGBA16_UNORM == 3
    # the five G-buffer planes: binding order inside their set == member order of szg_gbuffer (gbuffer/gbuffer.glinl:1-6)
    planes = ["gbufferDiffuse", "gbufferSpecular", "gbufferNormal", "gbufferWorldPosition", "gbufferOcclusionRoughnessMetallic"]
    fields = [f[0] for f in abi.GBuffer._fields_]
    assert fields == ["diffuse", "specular", "normal", "worldPosition", "occlusionRoughnessMetallic"]
    for shader, gset in (("camera.comp.spv", 2), ("lights.comp.spv", 1)):
        for k, plane in enumerate(planes):
            b = binding(shader, plane)
            assert (b["set"], b["binding"], b["descriptor_type"]) == (gset, k, COMBINED), (shader, plane)
    # shadow maps: an unsized array of sampled images + one sampler (shadowmap.glinl); depth of the scene texture
    for shader, sset in (("camera.comp.spv", 4), ("lights.comp.spv", 3)):
        b = binding(shader, "shadowMaps")
        assert (b["set"], b["binding"], b["descriptor_type"]) == (sset, 0, SAMPLED_IMAGE)
        assert binding(shader, "shadowMapSampler")["descriptor_type"] == SAMPLER
    b = binding("camera.comp.spv", "fragmentDepth")
    assert (b["set"], b["binding"], b["descriptor_type"]) == (0, 1, COMBINED)
    assert (binding("camera.comp.spv", "skyview_LUT")["binding"], binding("camera.comp.spv", "transmittance_LUT")["binding"]) == (0, 1)
    # OETF: in place on one storage image (the resource is RGBA16 UNORM, editor/uilayer.cpp:285-291)
    assert binding("oetf_srgb.comp.spv", "image")["descriptor_type"] == STORAGE_IMAGE


def test_the_reference_shaders_permit_contraction(golden):
    """The contraction rule of oracle/szg_oracle.cpp (dot, matrix * vector, mix, a * b + c evaluated with fused multiply-adds)
    rests on this: not one NoContraction decoration in the committed SPIR-V of the path (and no `precise` in the GLSL), while
    the shaders are full of OpFMul / OpFAdd pairs, OpDot and OpMatrixTimesVector whose evaluation Vulkan leaves to the
    implementation."""
    for name in ("transmittance_LUT.comp.spv", "skyview_LUT.comp.spv", "camera.comp.spv", "lights.comp.spv", "offscreen.vert.spv"):
        counts = golden[name]["arithmetic"]
        assert counts["no_contraction"] == 0, name
    assert golden["camera.comp.spv"]["arithmetic"]["OpFMul"] > 50 and golden["camera.comp.spv"]["arithmetic"]["OpDot"] > 5
    assert golden["lights.comp.spv"]["arithmetic"]["OpMatrixTimesVector"] >= 1


def test_raster_push_constants_name_the_buffers_raster_h_takes(golden):
    """offscreen.vert / depthpass.vert read vertices, model matrices (64 B stride) and cameras / light matrices through
    buffer references: the arrays szg_mesh_instanced and the shadow pass take (include/szg/raster.h)."""
    (vs,) = golden["offscreen.vert.spv"]["push_constants"]
    assert [m["name"] for m in vs["members"]] == ["vertexBuffer", "modelBuffer", "modelInverseTransposeBuffer", "cameraBuffer", "cameraIndex"]
    strides = {m["name"]: m["members"][0]["type_array_stride"] for m in vs["members"] if m["members"]}
    assert strides == {"vertexBuffer": 48, "modelBuffer": 64, "modelInverseTransposeBuffer": 64, "cameraBuffer": 416}
    (ds,) = golden["depthpass.vert.spv"]["push_constants"]
    assert [m["name"] for m in ds["members"]] == ["vertexBuffer", "modelBuffer", "projViewBuffer", "projViewIndex"]
    assert ctypes.sizeof(abi.Mat4) == 64
