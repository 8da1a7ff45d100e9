"""The asset loaders under AddressSanitizer + UBSan (CPU build, tests/cpp/fuzz_assets.cpp): valid files and tens of
thousands of corrupted variants go through szg_gltf_load_memory / szg_decode_image_rgba; any out-of-bounds access,
overflow or leak aborts the harness."""
import os
import subprocess

import numpy as np

from tests import gltf_writer as gw
from tests.test_assets import _two_primitive_asset

HERE = os.path.dirname(os.path.abspath(__file__))


def test_asset_loaders_survive_corrupted_input_under_sanitizers(tmp_path):
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "cpp"), "fuzz_assets"], check=True)
    rng = np.random.default_rng(7)
    seeds = []

    def seed(name, data):
        (tmp_path / name).write_bytes(data)
        seeds.append(str(tmp_path / name))

    b, _ = _two_primitive_asset()
    png = gw.png_rgba8(rng.integers(0, 256, (6, 5, 4), dtype=np.uint8))
    tex = b.texture(b.image_view(png))
    tex2 = b.texture(b.image_uri(gw.data_uri_png(png)))
    b.doc["materials"] = [{"name": "a", "pbrMetallicRoughness": {"baseColorTexture": {"index": tex}, "metallicRoughnessTexture": {"index": tex2}},
                           "normalTexture": {"index": tex}, "occlusionTexture": {"index": tex}}, {"name": "b"}]
    seed("asset.glb", b.glb())
    seed("asset.gltf", b.gltf_embedded())
    seed("rgba.png", gw.png_rgba8(rng.integers(0, 256, (23, 31, 4), dtype=np.uint8), idat_split=100))
    seed("pal.png", gw.png_encode(rng.integers(0, 4, (17, 9, 1)), 3, 2, palette=rng.integers(0, 256, (4, 3)), trns=bytes([1, 2]), interlace=True))
    seed("g16.png", gw.png_encode(rng.integers(0, 65536, (8, 8, 2)), 4, 16, interlace=True, level=0))
    out = subprocess.run([os.path.join(HERE, "cpp", "fuzz_assets"), "4000"] + seeds, capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:allocator_may_return_null=1:max_allocation_size_mb=2048"))
    assert out.returncode == 0, out.stderr[-4000:]
    inputs, loaded = (int(x) for x in out.stdout.split()[::2][:2])
    assert inputs == 5 * 4001 and 5 <= loaded < inputs
