"""Test-side JPEG tools, sharing no code with syzygy_amd/csrc/host_jpeg.cpp:

  encode()   a baseline (SOF0, Huffman) JPEG writer: any sampling factors, optimal Huffman tables (T.81 K.2), restart
             intervals, interleaved or one-scan-per-component, 8- or 16-bit quantisation tables, JFIF or bare RGB
  decode()   a slow, plain Python/numpy decoder restating stb_image's published integer arithmetic (12-bit fixed-point
             inverse DCT, triangle-filter chroma upsampling, 20-bit fixed-point YCbCr -> RGB): what the reference's
             stbi_load_from_memory(..., 4) returns for a baseline file (assets.cpp:328-335)
"""
import struct

import numpy as np
from scipy.fft import dctn

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
Q_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87,
                   80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92,
                   95, 98, 112, 100, 103, 99])
Q_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99,
                     99, 99] + [99] * 32)


def _scaled(table, quality):
    s = 5000 / quality if quality < 50 else 200 - 2 * quality
    return np.clip((table * s + 50) // 100, 1, 255).astype(np.int64)


# ---------------------------------------------------------------------------------------------------------------
# encoder
# ---------------------------------------------------------------------------------------------------------------
def _optimal_table(freq):
    """T.81 K.2: code lengths limited to 16, no all-ones code. freq: {symbol: count}. Returns (bits[1..16], symbols)."""
    f = [0] * 257
    for s, c in freq.items():
        f[s] = c
    f[256] = 1
    size = [0] * 257
    others = [-1] * 257
    while True:
        c1, best = -1, None
        for i in range(257):
            if f[i] and (best is None or f[i] <= best):
                c1, best = i, f[i]
        c2, best = -1, None
        for i in range(257):
            if f[i] and i != c1 and (best is None or f[i] <= best):
                c2, best = i, f[i]
        if c2 < 0:
            break
        f[c1] += f[c2]
        f[c2] = 0
        size[c1] += 1
        while others[c1] >= 0:
            c1 = others[c1]
            size[c1] += 1
        others[c1] = c2
        size[c2] += 1
        while others[c2] >= 0:
            c2 = others[c2]
            size[c2] += 1
    bits = [0] * 40
    for i in range(257):
        if size[i]:
            bits[size[i]] += 1
    i = 39
    while i > 16:
        while bits[i] > 0:
            j = i - 2
            while bits[j] == 0:
                j -= 1
            bits[i] -= 2
            bits[i - 1] += 1
            bits[j + 1] += 2
            bits[j] -= 1
        i -= 1
    while bits[i] == 0:
        i -= 1
    bits[i] -= 1  # the reserved code point
    symbols = [s for n in range(1, 40) for s in range(256) if size[s] == n]
    return bits[1:17], symbols


def _codes(bits, symbols):
    out, code, k = {}, 0, 0
    for n in range(1, 17):
        for _ in range(bits[n - 1]):
            out[symbols[k]] = (code, n)
            code += 1
            k += 1
        code <<= 1
    return out


def _category(v):
    return int(abs(int(v))).bit_length()


def _extra(v, n):
    return int(v) if v >= 0 else int(v) + (1 << n) - 1


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, n):
        self.acc = (self.acc << n) | (value & ((1 << n) - 1))
        self.n += n
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 255
            self.out.append(b)
            if b == 255:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def align(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def default_progressive_script(n):
    """(kind, components, Ss, Se, Ah, Al): DC with a point transform, two AC bands, then successive-approximation refinements."""
    comps = list(range(n))
    script = [("dc", comps, 0, 0, 0, 1)]
    script += [("ac", [k], 1, 5, 0, 2) for k in comps]
    script += [("ac", [k], 6, 63, 0, 2) for k in comps]
    script += [("ac", [k], 1, 63, 2, 1) for k in comps]
    script += [("dc", comps, 0, 0, 1, 0)]
    script += [("ac", [k], 1, 63, 1, 0) for k in comps]
    return script


def encode(image, sampling=((2, 2), (1, 1), (1, 1)), quality=85, restart=0, interleaved=True, jfif=True, rgb_ids=False,
           q16=False, adobe_transform=None, comment=None, progressive=None):
    """image: uint8 [h, w, 3] (RGB) or [h, w] (grey). With rgb_ids the three planes are stored as they are under the
    component ids 'R', 'G', 'B'; otherwise RGB is converted to YCbCr (JFIF). progressive: None (baseline), True (the
    default scan script) or a scan script (default_progressive_script)."""
    img = np.asarray(image)
    h, w = img.shape[:2]
    if img.ndim == 3 and img.shape[2] == 4:
        # four components stored as they are (Adobe CMYK / YCCK files); give adobe_transform 0 or 2
        planes = [img[..., k].astype(np.float64) for k in range(4)]
        sampling = list(sampling) if len(sampling) == 4 else [(1, 1)] * 4
    elif img.ndim == 2:
        planes, sampling = [img.astype(np.float64)], [(1, 1)] if sampling is None or len(sampling) != 1 else list(sampling)
    elif rgb_ids or adobe_transform == 0:
        planes = [img[..., k].astype(np.float64) for k in range(3)]
    else:
        r, g, b = (img[..., k].astype(np.float64) for k in range(3))
        planes = [0.299 * r + 0.587 * g + 0.114 * b, -0.168736 * r - 0.331264 * g + 0.5 * b + 128, 0.5 * r - 0.418688 * g - 0.081312 * b + 128]
    n = len(planes)
    sampling = list(sampling)[:n]
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    mx, my = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    tables = [_scaled(Q_LUMA, quality), _scaled(Q_CHROMA, quality)]
    if q16:
        tables[0] = tables[0] * 3 + 200  # needs 16-bit entries
    comp_tq = [0] + [1] * (n - 1)
    # quantised coefficient blocks per component, [rows of blocks][cols of blocks][64] in natural order
    blocks = []
    for k, (plane, (sh, sv)) in enumerate(zip(planes, sampling)):
        fx, fy = hmax // sh, vmax // sv
        full = np.pad(plane, ((0, my * 8 * vmax - h), (0, mx * 8 * hmax - w)), mode="edge")
        sub = full.reshape(full.shape[0] // fy, fy, full.shape[1] // fx, fx).mean(axis=(1, 3))
        bh, bw = sub.shape[0] // 8, sub.shape[1] // 8
        tiles = sub.reshape(bh, 8, bw, 8).transpose(0, 2, 1, 3) - 128.0
        coef = dctn(tiles, axes=(2, 3), norm="ortho")
        q = np.rint(coef / tables[comp_tq[k]].reshape(8, 8)).astype(np.int64)
        blocks.append(q.reshape(bh, bw, 64))

    ids = [ord(c) for c in "RGB"] if (rgb_ids and n == 3) else list(range(1, n + 1))

    def scan_units(components):
        """yield (component, block) in coding order"""
        if len(components) == 1:
            k = components[0]
            sh, sv = sampling[k]
            bw, bh = -(-(-(-w * sh // hmax)) // 8), -(-(-(-h * sv // vmax)) // 8)
            for j in range(bh):
                for i in range(bw):
                    yield [(k, blocks[k][j, i])]
        else:
            for j in range(my):
                for i in range(mx):
                    unit = []
                    for k in components:
                        sh, sv = sampling[k]
                        for y in range(sv):
                            for x in range(sh):
                                unit.append((k, blocks[k][j * sv + y, i * sh + x]))
                    yield unit

    def symbols_of(components):
        """[(kind, component, symbol, extra value, extra bits) ...] with ('rst', n) entries between restart intervals"""
        out, pred, count, rst = [], {k: 0 for k in components}, 0, 0
        for unit in scan_units(components):
            if restart and count == restart:
                out.append(("rst", rst))
                rst = (rst + 1) & 7
                pred = {k: 0 for k in components}
                count = 0
            count += 1
            for k, block in unit:
                zz = [int(block[ZIGZAG[i]]) for i in range(64)]
                diff = zz[0] - pred[k]
                pred[k] = zz[0]
                c = _category(diff)
                out.append(("dc", k, c, _extra(diff, c), c))
                run = 0
                last = max([i for i in range(1, 64) if zz[i]], default=0)
                for i in range(1, last + 1):
                    if zz[i] == 0:
                        run += 1
                        continue
                    while run > 15:
                        out.append(("ac", k, 0xF0, 0, 0))
                        run -= 16
                    c = _category(zz[i])
                    out.append(("ac", k, (run << 4) | c, _extra(zz[i], c), c))
                    run = 0
                if last < 63:
                    out.append(("ac", k, 0, 0, 0))
        return out

    if progressive:
        script = default_progressive_script(n) if progressive is True else progressive
        return _write_progressive(blocks, sampling, tables, comp_tq, ids, h, w, hmax, vmax, mx, my, restart, script, jfif and not rgb_ids,
                                  adobe_transform, scan_units)

    scans = [list(range(n))] if (interleaved or n == 1) else [[k] for k in range(n)]
    scan_symbols = [symbols_of(s) for s in scans]
    # one DC and one AC table per table class (luma / chroma), optimal for this file
    table_of = [0] + [1] * (n - 1)
    freq = {("dc", 0): {}, ("dc", 1): {}, ("ac", 0): {}, ("ac", 1): {}}
    for syms in scan_symbols:
        for s in syms:
            if s[0] != "rst":
                d = freq[(s[0], table_of[s[1]])]
                d[s[2]] = d.get(s[2], 0) + 1
    huff = {}
    for key, f in freq.items():
        if f:
            bits, symbols = _optimal_table(f)
            huff[key] = (bits, symbols, _codes(bits, symbols))

    out = bytearray(b"\xff\xd8")
    if jfif and not rgb_ids and adobe_transform is None:
        out += b"\xff\xe0" + struct.pack(">H5sHBHHBB", 16, b"JFIF\0", 0x0101, 0, 1, 1, 0, 0)
    if adobe_transform is not None:
        out += b"\xff\xee" + struct.pack(">H5sHHHB", 14, b"Adobe", 100, 0, 0, adobe_transform)
    if comment:
        out += b"\xff\xfe" + struct.pack(">H", 2 + len(comment)) + comment
    for t in sorted(set(comp_tq)):
        zz = [int(tables[t][ZIGZAG[i]]) for i in range(64)]
        if max(zz) > 255:
            out += b"\xff\xdb" + struct.pack(">HB", 2 + 1 + 128, 0x10 | t) + b"".join(struct.pack(">H", v) for v in zz)
        else:
            out += b"\xff\xdb" + struct.pack(">HB", 2 + 1 + 64, t) + bytes(zz)
    out += b"\xff\xc0" + struct.pack(">HBHHB", 8 + 3 * n, 8, h, w, n)
    for k in range(n):
        out += bytes([ids[k], (sampling[k][0] << 4) | sampling[k][1], comp_tq[k]])
    for (kind, t), (bits, symbols, _) in sorted(huff.items()):
        out += b"\xff\xc4" + struct.pack(">HB", 2 + 1 + 16 + len(symbols), (0x10 if kind == "ac" else 0) | t) + bytes(bits) + bytes(symbols)
    if restart:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart)
    for comps, syms in zip(scans, scan_symbols):
        out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(comps), len(comps))
        for k in comps:
            out += bytes([ids[k], (table_of[k] << 4) | table_of[k]])
        out += bytes([0, 63, 0])
        bits = _Bits()
        for s in syms:
            if s[0] == "rst":
                bits.align()
                bits.out += bytes([0xFF, 0xD0 + s[1]])
                continue
            code, length = huff[(s[0], table_of[s[1]])][2][s[2]]
            bits.put(code, length)
            if s[4]:
                bits.put(s[3], s[4])
        bits.align()
        out += bits.out
    return bytes(out + b"\xff\xd9")


def _point(v, al):
    """AC point transform: division by 2^al that rounds towards zero (T.81 G.1.2.2)"""
    v = int(v)
    return (abs(v) >> al) * (1 if v >= 0 else -1)


def _write_progressive(blocks, sampling, tables, comp_tq, ids, h, w, hmax, vmax, mx, my, restart, script, jfif, adobe_transform, scan_units):
    """T.81 annex G: spectral selection + successive approximation, one DHT pair (optimal, table 0) in front of every scan."""
    n = len(blocks)
    out = bytearray(b"\xff\xd8")
    if jfif and adobe_transform is None:
        out += b"\xff\xe0" + struct.pack(">H5sHBHHBB", 16, b"JFIF\0", 0x0101, 0, 1, 1, 0, 0)
    if adobe_transform is not None:
        out += b"\xff\xee" + struct.pack(">H5sHHHB", 14, b"Adobe", 100, 0, 0, adobe_transform)
    for t in sorted(set(comp_tq)):
        zz = [int(tables[t][ZIGZAG[i]]) for i in range(64)]
        if max(zz) > 255:
            out += b"\xff\xdb" + struct.pack(">HB", 2 + 1 + 128, 0x10 | t) + b"".join(struct.pack(">H", v) for v in zz)
        else:
            out += b"\xff\xdb" + struct.pack(">HB", 2 + 1 + 64, t) + bytes(zz)
    out += b"\xff\xc2" + struct.pack(">HBHHB", 8 + 3 * n, 8, h, w, n)
    for k in range(n):
        out += bytes([ids[k], (sampling[k][0] << 4) | sampling[k][1], comp_tq[k]])
    if restart:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart)

    for kind, comps, ss, se, ah, al in script:
        items, count, rst = [], 0, 0  # ("sym", v) | ("bits", v, n) | ("rst", k)
        pred = {k: 0 for k in comps}
        eobrun, pending = 0, []  # AC: end-of-band run and the correction bits that follow its symbol

        def flush_eobrun():
            nonlocal eobrun, pending
            if eobrun:
                nb = eobrun.bit_length() - 1
                items.append(("sym", nb << 4))
                if nb:
                    items.append(("bits", eobrun - (1 << nb), nb))
                eobrun = 0
            for b in pending:
                items.append(("bits", b, 1))
            pending = []

        for unit in scan_units(comps):
            if restart and count == restart:
                flush_eobrun()
                items.append(("rst", rst))
                rst, count, pred = (rst + 1) & 7, 0, {k: 0 for k in comps}
            count += 1
            for k, block in unit:
                if kind == "dc":
                    v = int(block[0]) >> al  # arithmetic shift (T.81 G.1.2.1)
                    if ah == 0:
                        diff = v - pred[k]
                        pred[k] = v
                        c = _category(diff)
                        items.append(("sym", c))
                        if c:
                            items.append(("bits", _extra(diff, c), c))
                    else:
                        items.append(("bits", v & 1, 1))
                    continue
                zz = [int(block[ZIGZAG[i]]) for i in range(64)]
                if ah == 0:
                    run = 0
                    for i in range(ss, se + 1):
                        t = _point(zz[i], al)
                        if t == 0:
                            run += 1
                            continue
                        flush_eobrun()
                        while run > 15:
                            items.append(("sym", 0xF0))
                            run -= 16
                        c = _category(t)
                        items.append(("sym", (run << 4) | c))
                        items.append(("bits", _extra(t, c), c))
                        run = 0
                    if run:
                        eobrun += 1
                        if eobrun == 0x7FFF:
                            flush_eobrun()
                    continue
                # refinement (T.81 figure G.7): newly non-zero coefficients have magnitude 1 after the point transform
                absval = [abs(zz[i]) >> al for i in range(64)]
                eob = max([i for i in range(ss, se + 1) if absval[i] == 1], default=0)
                run, corrections = 0, []
                for i in range(ss, se + 1):
                    t = absval[i]
                    if t == 0:
                        run += 1
                        continue
                    while run > 15 and i <= eob:
                        flush_eobrun()
                        items.append(("sym", 0xF0))
                        run -= 16
                        items.extend(("bits", b, 1) for b in corrections)
                        corrections = []
                    if t > 1:
                        corrections.append(t & 1)
                        continue
                    flush_eobrun()
                    items.append(("sym", (run << 4) | 1))
                    items.append(("bits", 0 if zz[i] < 0 else 1, 1))
                    items.extend(("bits", b, 1) for b in corrections)
                    corrections, run = [], 0
                if run or corrections:
                    eobrun += 1
                    pending.extend(corrections)
                    if eobrun == 0x7FFF or len(pending) > 900:
                        flush_eobrun()
        flush_eobrun()

        freq = {}
        for it in items:
            if it[0] == "sym":
                freq[it[1]] = freq.get(it[1], 0) + 1
        codes = None
        if freq:
            bits, symbols = _optimal_table(freq)
            codes = _codes(bits, symbols)
            out += b"\xff\xc4" + struct.pack(">HB", 2 + 1 + 16 + len(symbols), 0x10 if kind == "ac" else 0) + bytes(bits) + bytes(symbols)
        out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(comps), len(comps))
        for k in comps:
            out += bytes([ids[k], 0])
        out += bytes([ss, se, (ah << 4) | al])
        stream = _Bits()
        for it in items:
            if it[0] == "rst":
                stream.align()
                stream.out += bytes([0xFF, 0xD0 + it[1]])
            elif it[0] == "sym":
                stream.put(*codes[it[1]])
            else:
                stream.put(it[1], it[2])
        stream.align()
        out += stream.out
    return bytes(out + b"\xff\xd9")


# ---------------------------------------------------------------------------------------------------------------
# reference decoder (stb_image's arithmetic)
# ---------------------------------------------------------------------------------------------------------------
def _f2f(x):
    return int(x * 4096 + 0.5)


def _idct_1d(s0, s1, s2, s3, s4, s5, s6, s7):
    p2, p3 = s2, s6
    p1 = (p2 + p3) * _f2f(0.5411961)
    t2 = p1 + p3 * _f2f(-1.847759065)
    t3 = p1 + p2 * _f2f(0.765366865)
    p2, p3 = s0, s4
    t0, t1 = (p2 + p3) * 4096, (p2 - p3) * 4096
    x0, x3, x1, x2 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    t0, t1, t2, t3 = s7, s5, s3, s1
    p3, p4, p1, p2 = t0 + t2, t1 + t3, t0 + t3, t1 + t2
    p5 = (p3 + p4) * _f2f(1.175875602)
    t0, t1, t2, t3 = t0 * _f2f(0.298631336), t1 * _f2f(2.053119869), t2 * _f2f(3.072711026), t3 * _f2f(1.501321110)
    p1 = p5 + p1 * _f2f(-0.899976223)
    p2 = p5 + p2 * _f2f(-2.562915447)
    p3 = p3 * _f2f(-1.961570560)
    p4 = p4 * _f2f(-0.390180644)
    return x0, x1, x2, x3, t0 + p1 + p3, t1 + p2 + p4, t2 + p2 + p3, t3 + p1 + p4


def _idct_block(d):
    """d: 64 ints (dequantised, natural order) -> 8x8 uint8"""
    v = [0] * 64
    for i in range(8):
        col = [d[i + 8 * r] for r in range(8)]
        if not any(col[1:]):
            for r in range(8):
                v[i + 8 * r] = col[0] * 4
            continue
        x0, x1, x2, x3, t0, t1, t2, t3 = _idct_1d(*col)
        x0, x1, x2, x3 = x0 + 512, x1 + 512, x2 + 512, x3 + 512
        v[i], v[i + 56] = (x0 + t3) >> 10, (x0 - t3) >> 10
        v[i + 8], v[i + 48] = (x1 + t2) >> 10, (x1 - t2) >> 10
        v[i + 16], v[i + 40] = (x2 + t1) >> 10, (x2 - t1) >> 10
        v[i + 24], v[i + 32] = (x3 + t0) >> 10, (x3 - t0) >> 10
    out = np.zeros((8, 8), np.uint8)
    bias = 65536 + (128 << 17)
    for r in range(8):
        x0, x1, x2, x3, t0, t1, t2, t3 = _idct_1d(*v[8 * r : 8 * r + 8])
        x0, x1, x2, x3 = x0 + bias, x1 + bias, x2 + bias, x3 + bias
        row = [(x0 + t3) >> 17, (x1 + t2) >> 17, (x2 + t1) >> 17, (x3 + t0) >> 17, (x3 - t0) >> 17, (x2 - t1) >> 17, (x1 - t2) >> 17,
               (x0 - t3) >> 17]
        out[r] = [min(max(x, 0), 255) for x in row]
    return out


def _int16(x):
    x &= 0xFFFF
    return x - 0x10000 if x & 0x8000 else x


class _Reader:
    """bits of one entropy-coded segment, MSB first, 0xFF00 unstuffed; zeros after its end"""

    def __init__(self, data):
        self.data, self.pos, self.bit = data, 0, 0

    def bit1(self):
        if self.pos >= len(self.data):
            return 0
        b = (self.data[self.pos] >> (7 - self.bit)) & 1
        self.bit += 1
        if self.bit == 8:
            self.bit, self.pos = 0, self.pos + 1
        return b

    def bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit1()
        return v


def _decode_symbol(reader, table):
    code = 0
    for n in range(1, 17):
        code = (code << 1) | reader.bit1()
        if (n, code) in table:
            return table[(n, code)]
    raise ValueError("bad Huffman code")


def _extend(v, n):
    return v - (1 << n) + 1 if n and v < (1 << (n - 1)) else v


def _progressive_ac(reader, table, data, ss, se, ah, al, eobrun):
    """one block of a progressive AC scan (T.81 G.1.2.2 / G.1.2.3); returns the updated end-of-band run"""
    if ah == 0:
        if eobrun:
            return eobrun - 1
        k = ss
        while k <= se:
            rs = _decode_symbol(reader, table)
            r, sz = rs >> 4, rs & 15
            if sz == 0:
                if r < 15:
                    return (1 << r) + (reader.bits(r) if r else 0) - 1
                k += 16
                continue
            k += r
            data[ZIGZAG[k]] = _int16(_extend(reader.bits(sz), sz) * (1 << al))
            k += 1
        return 0
    bit = 1 << al

    def correct(i):
        if reader.bit1() and (data[i] & bit) == 0:
            data[i] = _int16(data[i] + bit if data[i] > 0 else data[i] - bit)

    if eobrun:
        for k in range(ss, se + 1):
            if data[ZIGZAG[k]]:
                correct(ZIGZAG[k])
        return eobrun - 1
    k = ss
    while k <= se:
        rs = _decode_symbol(reader, table)
        r, sz = rs >> 4, rs & 15
        value = 0
        if sz == 0:
            if r < 15:
                eobrun = (1 << r) - 1 + (reader.bits(r) if r else 0)
                r = 64
        else:
            assert sz == 1
            value = bit if reader.bit1() else -bit
        while k <= se:
            i = ZIGZAG[k]
            k += 1
            if data[i]:
                correct(i)
            else:
                if r == 0:
                    data[i] = value
                    break
                r -= 1
    return eobrun


def decode(data):
    """baseline or progressive JPEG bytes -> uint8 [h, w, 4] as stb_image returns it for a 4-channel request"""
    data = bytes(data)
    assert data[:2] == b"\xff\xd8"
    pos = 2
    quant, huff, comps, restart = {}, {}, [], 0
    jfif, adobe, w = False, -1, 0
    planes = {}
    progressive, coefs = False, {}
    while True:
        assert data[pos] == 0xFF
        while data[pos + 1] == 0xFF:
            pos += 1
        marker = data[pos + 1]
        pos += 2
        if marker == 0xD9:
            break
        length = struct.unpack(">H", data[pos : pos + 2])[0]
        seg = data[pos + 2 : pos + length]
        pos += length
        if marker == 0xE0 and seg[:5] == b"JFIF\0":
            jfif = True
        elif marker == 0xEE and seg[:6] == b"Adobe\0":
            adobe = seg[11]
        elif marker == 0xDB:
            while seg:
                p16, t = seg[0] >> 4, seg[0] & 15
                vals = struct.unpack(">64H", seg[1:129]) if p16 else tuple(seg[1:65])
                table = [0] * 64
                for i in range(64):
                    table[ZIGZAG[i]] = vals[i]
                quant[t] = table
                seg = seg[129:] if p16 else seg[65:]
        elif marker == 0xC4:
            while seg:
                kind, t = seg[0] >> 4, seg[0] & 15
                counts = seg[1:17]
                total = sum(counts)
                symbols = seg[17 : 17 + total]
                table, code, k = {}, 0, 0
                for n in range(1, 17):
                    for _ in range(counts[n - 1]):
                        table[(n, code)] = symbols[k]
                        code += 1
                        k += 1
                    code <<= 1
                huff[(kind, t)] = table
                seg = seg[17 + total :]
        elif marker == 0xDD:
            restart = struct.unpack(">H", seg)[0]
        elif marker in (0xC0, 0xC1, 0xC2):
            progressive = marker == 0xC2
            _, h, w, n = struct.unpack(">BHHB", seg[:6])
            comps = [{"id": seg[6 + 3 * i], "h": seg[7 + 3 * i] >> 4, "v": seg[7 + 3 * i] & 15, "tq": seg[8 + 3 * i]} for i in range(n)]
            hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
            mx, my = -(-w // (8 * hmax)), -(-h // (8 * vmax))
            for c in comps:
                c["x"], c["y"] = -(-w * c["h"] // hmax), -(-h * c["v"] // vmax)
                planes[c["id"]] = np.zeros((my * c["v"] * 8, mx * c["h"] * 8), np.uint8)
                coefs[c["id"]] = {}
        elif marker == 0xDA:
            ns = seg[0]
            ss, se, ah, al = seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns] >> 4, seg[3 + 2 * ns] & 15
            order = []
            for i in range(ns):
                c = next(c for c in comps if c["id"] == seg[1 + 2 * i])
                c["td"], c["ta"] = seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15
                order.append(c)
            # entropy-coded data up to the next marker that is not a restart marker; split at the restart markers
            segments, cur = [], bytearray()
            while True:
                b = data[pos]
                if b != 0xFF:
                    cur.append(b)
                    pos += 1
                elif data[pos + 1] == 0:
                    cur.append(0xFF)
                    pos += 2
                elif 0xD0 <= data[pos + 1] <= 0xD7:
                    segments.append(bytes(cur))
                    cur = bytearray()
                    pos += 2
                else:
                    segments.append(bytes(cur))
                    break
            if ns == 1:
                c = order[0]
                units = [[(c, j, i)] for j in range((c["y"] + 7) // 8) for i in range((c["x"] + 7) // 8)]
            else:
                units = [[(c, j * c["v"] + y, i * c["h"] + x) for c in order for y in range(c["v"]) for x in range(c["h"])]
                         for j in range(my) for i in range(mx)]
            interval = restart if restart else len(units)
            for s, first in enumerate(range(0, len(units), interval)):
                reader = _Reader(segments[s] if s < len(segments) else b"")
                pred = {c["id"]: 0 for c in order}
                eobrun = 0
                for unit in units[first : first + interval]:
                    for c, by, bx in unit:
                        if progressive:
                            cf = coefs[c["id"]].setdefault((by, bx), [0] * 64)
                            if ss == 0:
                                if ah == 0:
                                    t = _decode_symbol(reader, huff[(0, c["td"])])
                                    pred[c["id"]] += _extend(reader.bits(t), t)
                                    cf[0] = _int16(pred[c["id"]] * (1 << al))
                                elif reader.bit1():
                                    cf[0] = _int16(cf[0] + (1 << al))
                            else:
                                eobrun = _progressive_ac(reader, huff[(1, c["ta"])], cf, ss, se, ah, al, eobrun)
                            continue
                        dq = quant[c["tq"]]
                        block = [0] * 64
                        t = _decode_symbol(reader, huff[(0, c["td"])])
                        pred[c["id"]] += _extend(reader.bits(t), t)
                        block[0] = _int16(pred[c["id"]] * dq[0])
                        k = 1
                        while k < 64:
                            rs = _decode_symbol(reader, huff[(1, c["ta"])])
                            r, sz = rs >> 4, rs & 15
                            if sz == 0:
                                if rs != 0xF0:
                                    break
                                k += 16
                                continue
                            k += r
                            block[ZIGZAG[k]] = _int16(_extend(reader.bits(sz), sz) * dq[ZIGZAG[k]])
                            k += 1
                        planes[c["id"]][by * 8 : by * 8 + 8, bx * 8 : bx * 8 + 8] = _idct_block(block)
    if progressive:
        for c in comps:
            dq = quant[c["tq"]]
            for (by, bx), cf in coefs[c["id"]].items():
                planes[c["id"]][by * 8 : by * 8 + 8, bx * 8 : bx * 8 + 8] = _idct_block([_int16(cf[i] * dq[i]) for i in range(64)])
    # upsampling (stb_image's resamplers) and colour conversion
    n = len(comps)
    full = []
    for c in comps:
        hs, vs = hmax // c["h"], vmax // c["v"]
        plane = planes[c["id"]].astype(np.int64)
        wl = -(-w // hs)
        rows, ystep, ypos, l0, l1 = [], vs >> 1, 0, 0, 0
        for _ in range(h):
            bottom = ystep >= (vs >> 1)
            near, far = (plane[l1], plane[l0]) if bottom else (plane[l0], plane[l1])
            near, far = near[:wl], far[:wl]
            if hs == 1 and vs == 1:
                row = near
            elif hs == 1 and vs == 2:
                row = (3 * near + far + 2) >> 2
            elif hs == 2 and vs == 1:
                row = np.zeros(2 * wl, np.int64)
                if wl == 1:
                    row[:] = near[0]
                else:
                    row[0], row[1] = near[0], (near[0] * 3 + near[1] + 2) >> 2
                    for i in range(1, wl - 1):
                        m = 3 * near[i] + 2
                        row[2 * i], row[2 * i + 1] = (m + near[i - 1]) >> 2, (m + near[i + 1]) >> 2
                    row[2 * wl - 2], row[2 * wl - 1] = (near[wl - 2] * 3 + near[wl - 1] + 2) >> 2, near[wl - 1]
            elif hs == 2 and vs == 2:
                t = 3 * near + far
                row = np.zeros(2 * wl, np.int64)
                if wl == 1:
                    row[:] = (t[0] + 2) >> 2
                else:
                    row[0], row[2 * wl - 1] = (t[0] + 2) >> 2, (t[wl - 1] + 2) >> 2
                    for i in range(1, wl):
                        row[2 * i - 1], row[2 * i] = (3 * t[i - 1] + t[i] + 8) >> 4, (3 * t[i] + t[i - 1] + 8) >> 4
            else:
                row = np.repeat(near, hs)
            rows.append(np.asarray(row[:w]))
            ystep += 1
            if ystep >= vs:
                ystep, l0 = 0, l1
                ypos += 1
                if ypos < c["y"]:
                    l1 += 1
        full.append(np.stack(rows))
    out = np.full((h, w, 4), 255, np.uint8)

    def ycc_to_rgb(y, cb, cr):
        f = lambda x: int(np.float32(x) * np.float32(4096.0) + np.float32(0.5)) << 8  # noqa: E731
        cb, cr = cb - 128, cr - 128
        yf = (y << 20) + (1 << 19)
        masked = ((cb * -f(0.34414)) & 0xFFFFFFFF) & 0xFFFF0000
        masked = np.where(masked >= (1 << 31), masked - (1 << 32), masked)
        return [np.clip(v >> 20, 0, 255) for v in (yf + cr * f(1.40200), yf + cr * -f(0.71414) + masked, yf + cb * f(1.77200))]

    def blinn(x, k):
        t = x * k + 128
        return (t + (t >> 8)) >> 8

    if n == 4:
        k4 = full[3]
        if adobe == 0:
            rgb = [blinn(full[c], k4) for c in range(3)]
        else:
            rgb = ycc_to_rgb(full[0], full[1], full[2])
            if adobe == 2:
                rgb = [blinn(255 - v, k4) for v in rgb]
        for c in range(3):
            out[..., c] = rgb[c]
        return out
    if n == 1:
        out[..., 0] = out[..., 1] = out[..., 2] = full[0]
    elif [c["id"] for c in comps] == [ord(x) for x in "RGB"] or (adobe == 0 and not jfif):
        for k in range(3):
            out[..., k] = full[k]
    else:
        f = lambda x: int(np.float32(x) * np.float32(4096.0) + np.float32(0.5)) << 8  # noqa: E731
        y, cb, cr = full[0], full[1] - 128, full[2] - 128
        yf = (y << 20) + (1 << 19)
        r = yf + cr * f(1.40200)
        masked = ((cb * -f(0.34414)) & 0xFFFFFFFF) & 0xFFFF0000  # the 32-bit pattern of the product, low half cleared
        masked = np.where(masked >= (1 << 31), masked - (1 << 32), masked)
        g = yf + cr * -f(0.71414) + masked
        b = yf + cb * f(1.77200)
        out[..., 0] = np.clip(r >> 20, 0, 255)
        out[..., 1] = np.clip(g >> 20, 0, 255)
        out[..., 2] = np.clip(b >> 20, 0, 255)
    return out
