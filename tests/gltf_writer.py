"""Independent encoders for the asset tests: PNG (zlib + struct) and glTF 2.0 / GLB (json + numpy). Nothing here shares
code with the C++ loader (syzygy_amd/csrc/host_assets.cpp); the tests write files with these and compare what the loader
returns with the arrays that went in."""
import base64
import json
import struct
import zlib

import numpy as np

ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def _chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def _pack_rows(samples, depth):
    """samples: [h, w * channels] unsigned ints at `depth` bits -> list of packed scanline bytes (MSB first)."""
    rows = []
    for row in samples:
        if depth == 8:
            rows.append(bytes(row.astype(np.uint8)))
        elif depth == 16:
            rows.append(row.astype(">u2").tobytes())
        else:
            bits = np.zeros(((len(row) * depth + 7) // 8) * 8, np.uint8)
            for k in range(depth):
                bits[k : len(row) * depth : depth] = (row >> (depth - 1 - k)) & 1
            rows.append(np.packbits(bits).tobytes())
    return rows


def _filter_rows(rows, bpp, rng, filters):
    out = bytearray()
    previous = bytes(len(rows[0])) if rows else b""
    for row in rows:
        f = int(rng.choice(filters))
        cur = np.frombuffer(row, np.uint8).astype(np.int32)
        up = np.frombuffer(previous, np.uint8).astype(np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if len(cur) > bpp else np.zeros(len(cur), np.int32)
        upleft = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]]) if len(cur) > bpp else np.zeros(len(cur), np.int32)
        if len(cur) <= bpp:
            left = np.zeros(len(cur), np.int32)
            upleft = np.zeros(len(cur), np.int32)
        if f == 0:
            pred = np.zeros_like(cur)
        elif f == 1:
            pred = left
        elif f == 2:
            pred = up
        elif f == 3:
            pred = (left + up) // 2
        else:
            p = left + up - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
        out.append(f)
        out += bytes(((cur - pred) & 255).astype(np.uint8))
        previous = row
    return bytes(out)


def png_encode(samples, color_type, depth, *, palette=None, trns=None, interlace=False, filters=(0, 1, 2, 3, 4), seed=0,
               level=6, strategy=zlib.Z_DEFAULT_STRATEGY, idat_split=None, extra_chunks=()):
    """samples: uint array [h, w, channels] (or [h, w]) holding raw sample values at `depth` bits."""
    rng = np.random.default_rng(seed)
    samples = np.asarray(samples)
    if samples.ndim == 2:
        samples = samples[..., None]
    h, w, ch = samples.shape
    assert ch == CHANNELS[color_type]
    bpp = max(1, ch * depth // 8)
    raw = bytearray()
    passes = ADAM7 if interlace else [(0, 0, 1, 1)]
    for x0, y0, dx, dy in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        rows = _pack_rows(sub.reshape(sub.shape[0], -1), depth)
        raw += _filter_rows(rows, bpp, rng, filters)
    comp = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
    data = comp.compress(bytes(raw)) + comp.flush()
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0))
    for kind, body in extra_chunks:
        out += _chunk(kind, body)
    if palette is not None:
        out += _chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).reshape(-1)))
    if trns is not None:
        out += _chunk(b"tRNS", bytes(trns))
    if idat_split:
        for k in range(0, len(data), idat_split):
            out += _chunk(b"IDAT", data[k : k + idat_split])
    else:
        out += _chunk(b"IDAT", data)
    return out + _chunk(b"IEND", b"")


def png_rgba8(rgba, **kw):
    return png_encode(np.asarray(rgba, np.uint8), 6, 8, **kw)


# ---------------------------------------------------------------------------------------------------------------
# glTF
# ---------------------------------------------------------------------------------------------------------------
COMPONENT = {np.dtype(np.int8): 5120, np.dtype(np.uint8): 5121, np.dtype(np.int16): 5122, np.dtype(np.uint16): 5123,
             np.dtype(np.uint32): 5125, np.dtype(np.float32): 5126}
TYPES = {1: "SCALAR", 2: "VEC2", 3: "VEC3", 4: "VEC4"}


class GltfBuilder:
    """Accumulates one binary buffer, buffer views and accessors; the caller adds meshes / materials as plain dicts."""

    def __init__(self):
        self.blob = bytearray()
        self.doc = {"asset": {"version": "2.0"}, "bufferViews": [], "accessors": [], "meshes": [], "materials": [],
                    "textures": [], "images": []}

    def view(self, data, stride=None):
        while len(self.blob) % 4:
            self.blob.append(0)
        v = {"buffer": 0, "byteOffset": len(self.blob), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        self.blob += data
        self.doc["bufferViews"].append(v)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, array, normalized=False, stride_pad=0, offset_pad=0):
        """array: [n] or [n, k] numpy array of a glTF component type. `stride_pad` bytes of filler follow every element
        (exercises byteStride), `offset_pad` bytes precede the first (exercises accessor.byteOffset)."""
        a = np.ascontiguousarray(array)
        k = 1 if a.ndim == 1 else a.shape[1]
        elem = a.dtype.itemsize * k
        if stride_pad or offset_pad:
            rows = a.reshape(len(a), -1).view(np.uint8).reshape(len(a), elem)
            padded = np.concatenate([rows, np.full((len(a), stride_pad), 0xAB, np.uint8)], axis=1)
            data = bytes([0xCD] * offset_pad) + padded.tobytes()
            view = self.view(data, stride=(elem + stride_pad) if stride_pad else None)
        else:
            view = self.view(a.tobytes())
        acc = {"bufferView": view, "componentType": COMPONENT[a.dtype], "count": len(a), "type": TYPES[k]}
        if offset_pad:
            acc["byteOffset"] = offset_pad
        if normalized:
            acc["normalized"] = True
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    def sparse_accessor(self, base, indices, values):
        """An accessor whose elements `indices` are replaced by `values` (base may be None: zeros)."""
        values = np.ascontiguousarray(values)
        k = values.shape[1]
        acc = {"componentType": COMPONENT[values.dtype], "count": len(base) if base is not None else int(max(indices)) + 1,
               "type": TYPES[k],
               "sparse": {"count": len(indices),
                          "indices": {"bufferView": self.view(np.asarray(indices, np.uint16).tobytes()), "componentType": 5123},
                          "values": {"bufferView": self.view(values.tobytes())}}}
        if base is not None:
            acc["bufferView"] = self.view(np.ascontiguousarray(base).tobytes())
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    def image_view(self, data, name=None):
        img = {"bufferView": self.view(data), "mimeType": "image/png"}
        if name:
            img["name"] = name
        self.doc["images"].append(img)
        return len(self.doc["images"]) - 1

    def image_uri(self, uri, name=None):
        img = {"uri": uri}
        if name:
            img["name"] = name
        self.doc["images"].append(img)
        return len(self.doc["images"]) - 1

    def texture(self, image):
        self.doc["textures"].append({"source": image})
        return len(self.doc["textures"]) - 1

    def document(self, buffer_uri=None):
        doc = {k: v for k, v in self.doc.items() if v}
        buf = {"byteLength": len(self.blob)}
        if buffer_uri is not None:
            buf["uri"] = buffer_uri
        doc["buffers"] = [buf]
        return doc

    def glb(self):
        js = json.dumps(self.document()).encode()
        js += b" " * (-len(js) % 4)
        blob = bytes(self.blob) + b"\0" * (-len(self.blob) % 4)
        total = 12 + 8 + len(js) + 8 + len(blob)
        return (struct.pack("<4sII", b"glTF", 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js +
                struct.pack("<II", len(blob), 0x004E4942) + blob)

    def gltf_embedded(self):
        uri = "data:application/octet-stream;base64," + base64.b64encode(bytes(self.blob)).decode()
        return json.dumps(self.document(uri)).encode()

    def gltf_external(self, bin_name):
        return json.dumps(self.document(bin_name)).encode(), bytes(self.blob)


def data_uri_png(png):
    return "data:image/png;base64," + base64.b64encode(png).decode()


def uv_sphere(stacks=12, slices=24, radius=1.0):
    """glTF-space sphere (+y up, counter-clockwise front faces seen from outside): positions, normals, uvs, indices."""
    pos, nrm, uv = [], [], []
    for i in range(stacks + 1):
        theta = np.pi * i / stacks
        for j in range(slices + 1):
            phi = 2 * np.pi * j / slices
            n = (np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi))
            nrm.append(n)
            pos.append(tuple(radius * c for c in n))
            uv.append((j / slices, i / stacks))
    idx = []
    for i in range(stacks):
        for j in range(slices):
            a, b = i * (slices + 1) + j, (i + 1) * (slices + 1) + j
            idx += [a, a + 1, b, b, a + 1, b + 1]
    return (np.array(pos, np.float32), np.array(nrm, np.float32), np.array(uv, np.float32), np.array(idx, np.uint32))
