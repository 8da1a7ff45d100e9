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
    data = json.loads(out)
    # What the reflection library does not report: whether the shaders forbid fused multiply-adds anywhere. Counted from the
    # SPIR-V words themselves (SPIR-V 1.x: OpDecorate = 71, OpMemberDecorate = 72, Decoration NoContraction = 42).
    import struct

    for path in paths:
        blob = open(path, "rb").read()
        words = struct.unpack("<%dI" % (len(blob) // 4), blob)
        assert words[0] == 0x07230203
        i, counts = 5, {"no_contraction": 0, "OpFMul": 0, "OpFAdd": 0, "OpFSub": 0, "OpDot": 0, "OpMatrixTimesVector": 0}
        while i < len(words):
            op, n = words[i] & 0xFFFF, words[i] >> 16
            if (op == 71 and words[i + 2] == 42) or (op == 72 and words[i + 3] == 42):
                counts["no_contraction"] += 1
            for name, code in (("OpFMul", 133), ("OpFAdd", 129), ("OpFSub", 131), ("OpDot", 148), ("OpMatrixTimesVector", 145)):
                if op == code:
                    counts[name] += 1
            i += n
        data[os.path.basename(path)]["arithmetic"] = counts
    return data


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
