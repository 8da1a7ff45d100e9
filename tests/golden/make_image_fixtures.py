"""PNG fixtures written by an INDEPENDENT encoder (Pillow 12 = libpng/zlib) together with Pillow's own decode of each,
for tests/test_assets.py::test_images_written_and_decoded_by_pillow. Run with an interpreter that has Pillow (in the build
container: /opt/conda/bin/python tests/golden/make_image_fixtures.py); the product and the test suite never import Pillow.
PNG decodes must be identical. (JPEG is not decoded by this build: asset IO is out of the hot path's scope)"""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "images")
os.makedirs(HERE, exist_ok=True)
rng = np.random.default_rng(2024)
y, x = np.mgrid[0:40, 0:56]
rgb = np.stack([(x * 4 + y) % 256, (y * 6) % 256, ((x + 2 * y) * 3) % 256], -1).astype(np.uint8)
rgb[8:20, 10:30] = rng.integers(0, 256, (12, 20, 3))
rgb[25:, :18] = (30, 180, 220)
img = Image.fromarray(rgb, "RGB")
expected = {}


def keep(name, data):
    with open(os.path.join(HERE, name), "wb") as f:
        f.write(data)
    expected[name] = np.array(Image.open(io.BytesIO(data)).convert("RGBA"))


def png(name, image, **kw):
    b = io.BytesIO()
    image.save(b, "PNG", **kw)
    keep(name, b.getvalue())


png("rgb.png", img)
png("rgba.png", Image.fromarray(np.dstack([rgb, ((x * 5 + y * 3) % 256).astype(np.uint8)]), "RGBA"), compress_level=9)
png("grey.png", img.convert("L"))
png("grey_alpha.png", img.convert("LA"))
png("palette.png", img.convert("P", palette=Image.ADAPTIVE, colors=37))
pal = img.convert("P", palette=Image.ADAPTIVE, colors=16)
png("palette_transparent.png", pal, transparency=3, bits=4)
png("bilevel.png", img.convert("1"))
raw16 = rgb[..., 0].astype(np.uint16) * 257 + 13
png("grey16.png", Image.fromarray(raw16, "I;16"))
# Pillow clamps 16-bit samples to 255 when it converts to 8 bits; stb_image keeps the high byte, which is what is expected here
high = (raw16 >> 8).astype(np.uint8)
expected["grey16.png"] = np.dstack([high, high, high, np.full_like(high, 255)])
np.savez_compressed(os.path.join(HERE, "expected_rgba.npz"), **expected)
print({k: v.shape for k, v in expected.items()})
