"""Writes tests/golden/spv_reflection.json: what the reference's committed SPIR-V binaries declare about the data ABI of the
hot path, as reported by the reference's OWN vendored reflection library (thirdparty/spirv-reflect, compiled where it lies
into oracle/_ref/reflect_spv by `make -C oracle ref`; the engine validates its push-constant structs against the same
reflection: deferred.cpp:30-62, pipelines.cpp:609-624). The file is data - reflected offsets, sizes, formats, bindings -
not reference text. Run in the build container (needs /root/reference); the GPU box and the test suite only read the JSON.

    python tests/golden/make_spv_reflection.py
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REFERENCE = os.environ.get("SZG_REFERENCE", "/root/reference")
TOOL = os.path.join(ROOT, "oracle", "_ref", "reflect_spv")
# the four compute programs of the path + the programs either side of it
SHADERS = [
    "shaders/atmosphere/transmittance_LUT.comp.spv",
    "shaders/atmosphere/skyview_LUT.comp.spv",
    "shaders/atmosphere/camera.comp.spv",
    "shaders/deferred/lights.comp.spv",
    "shaders/deferred/offscreen.vert.spv",
    "shaders/deferred/offscreen.frag.spv",
    "shaders/offscreenpass/depthpass.vert.spv",
    "shaders/transfer/oetf_srgb.comp.spv",
    "shaders/transfer/oetf_pure_gamma.comp.spv",
]


def reflect():
    """The reflection of every shader above as a dict; None when the reference or the tool is absent."""
    paths = [os.path.join(REFERENCE, s) for s in SHADERS]
    if not os.path.exists(TOOL) or not all(os.path.exists(p) for p in paths):
        return None
    out = subprocess.run([TOOL] + paths, check=True, capture_output=True, text=True).stdout
    return json.loads(out)


if __name__ == "__main__":
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    data = reflect()
    if data is None:
        sys.exit("the reference checkout or oracle/_ref/reflect_spv is missing")
    target = os.path.join(ROOT, "tests", "golden", "spv_reflection.json")
    with open(target, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
        f.write("\n")
    print(f"wrote {target}: {len(data)} shaders")
