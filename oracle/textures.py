"""Texture path of the reference's loader, restated in numpy.  TEST INFRASTRUCTURE (see oracle/oracle.h): only
tests/ may import this; the product's implementation is caitlynrenderer_amd/csrc/host/image.cpp.

`texture_to_array_bytes` follows Caitlyn/Scene.h:321-371 (resize_image), :648-662 (same-size path) and :688-710
(floats pushed into a vector<unsigned char>): fp32 arithmetic in the reference's operation order, truncation to bytes.
The image writers below produce the test inputs (files with known pixels) for the decoders.
"""
import struct
import zlib

import numpy as np

F = np.float32


def texture_to_array_bytes(rgb, out_w=256, out_h=256):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    img_h, img_w = rgb.shape[:2]
    image = F(255) * (rgb.astype(np.float32) * (F(1.0) / F(255.0)))          # Scene.h:326-333
    if (img_w, img_h) == (out_w, out_h):
        return image.astype(np.uint8)                                        # Scene.h:653-661, then :701-703
    x_ratio = F(img_w) / F(out_w) if out_w > 1 else F(1)                     # Scene.h:337-338
    y_ratio = F(img_h) / F(out_h) if out_h > 1 else F(1)
    fx = (x_ratio * np.arange(out_w, dtype=np.float32)).astype(np.float32)[None, :]
    fy = (y_ratio * np.arange(out_h, dtype=np.float32)).astype(np.float32)[:, None]
    xl, xh = np.floor(fx).astype(np.int64), np.ceil(fx).astype(np.int64)
    yl, yh = np.floor(fy).astype(np.int64), np.ceil(fy).astype(np.int64)
    xw = (fx - xl.astype(np.float32)).astype(np.float32)[..., None]
    yw = (fy - yl.astype(np.float32)).astype(np.float32)[..., None]
    flat = image.reshape(-1, 3)
    last = flat.shape[0] - 1

    def at(yy, xx):                      # flat index as the reference forms it; past the end (UB there): last pixel
        return flat[np.minimum(yy * img_w + xx, last)]

    a, b, c, d = at(yl, xl), at(yl, xh), at(yh, xl), at(yh, xh)
    one = F(1)
    pixel = (a * (one - xw)) * (one - yw) + (b * xw) * (one - yw) + (c * yw) * (one - xw) + (d * xw) * yw   # :359-362
    return pixel.astype(np.uint8)


# ---- writers for test inputs -------------------------------------------------------------------------------------

def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _png_pack(px, depth):
    """(h, w, ch) samples -> (row bytes, stride)."""
    h, w, ch = px.shape
    if depth == 16:
        return px.astype(">u2").tobytes(), w * ch * 2
    if depth == 8:
        return px.astype(np.uint8).tobytes(), w * ch
    bits = np.unpackbits(px.astype(np.uint8).reshape(h, w, 1), axis=2)[:, :, 8 - depth:].reshape(h, -1)
    bits = np.pad(bits, ((0, 0), (0, (-bits.shape[1]) % 8)))
    return np.packbits(bits, axis=1).tobytes(), bits.shape[1] // 8


def _png_filter(rows, h, stride, bpp, filters):
    raw = bytearray()
    prev = bytearray(stride)
    for y in range(h):
        cur = bytearray(rows[y * stride:(y + 1) * stride])
        ft = (y % 5) if filters == "cycle" else int(filters)
        out = bytearray(stride)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            up = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = up
            elif ft == 3:
                pred = (a + up) >> 1
            else:
                p = a + up - c
                pa, pb, pc = abs(p - a), abs(p - up), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (up if pb <= pc else c)
            out[i] = (cur[i] - pred) & 0xFF
        raw.append(ft)
        raw += out
        prev = cur
    return raw


ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))   # x0, y0, dx, dy


def write_png(pixels, color_type=2, depth=8, palette=None, filters="cycle", interlace=False, trns=None):
    """pixels: (H, W, C) samples (C by colour type: 0 grey, 2 RGB, 3 index, 4 grey+alpha, 6 RGBA).  `interlace`: Adam7;
    `trns`: bytes of a tRNS chunk (transparency: a decoder asked for RGB drops it)."""
    px = np.asarray(pixels)
    h, w = px.shape[:2]
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    px = px.reshape(h, w, ch)
    bpp = max(1, ch * depth // 8)
    raw = bytearray()
    for x0, y0, dx, dy in (ADAM7 if interlace else ((0, 0, 1, 1),)):
        sub = px[y0::dy, x0::dx]
        if sub.size == 0:
            continue
        rows, stride = _png_pack(sub, depth)
        raw += _png_filter(rows, sub.shape[0], stride, bpp, filters)
    data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0))
    if palette is not None:
        data += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        data += _chunk(b"tRNS", bytes(trns))
    comp = zlib.compress(bytes(raw), 6)
    half = len(comp) // 2
    data += _chunk(b"IDAT", comp[:half]) + _chunk(b"IDAT", comp[half:])       # split IDAT: decoders must concatenate
    return data + _chunk(b"IEND", b"")


def write_bmp(rgb, bpp=24, top_down=False, palette=None):
    px = np.asarray(rgb, np.uint8)
    h, w = px.shape[:2]
    if bpp == 8:
        body = px.reshape(h, w)
    elif bpp == 24:
        body = px[:, :, ::-1].reshape(h, w * 3)
    else:
        body = np.concatenate([px[:, :, ::-1], np.full((h, w, 1), 255, np.uint8)], axis=2).reshape(h, w * 4)
    stride = (body.shape[1] + 3) // 4 * 4
    body = np.pad(body, ((0, 0), (0, stride - body.shape[1])))
    if not top_down:
        body = body[::-1]
    pal = b""
    if bpp == 8:
        p = np.asarray(palette, np.uint8).reshape(-1, 3)
        pal = np.concatenate([p[:, ::-1], np.zeros((p.shape[0], 1), np.uint8)], axis=1).tobytes()
    off = 14 + 40 + len(pal)
    hdr = b"BM" + struct.pack("<IHHI", off + body.size, 0, 0, off)
    info = struct.pack("<IiiHHIIiiII", 40, w, -h if top_down else h, 1, bpp, 0, body.size, 2835, 2835, len(pal) // 4, 0)
    return hdr + info + pal + np.ascontiguousarray(body).tobytes()


def write_tga(pixels, kind=2, rle=False, top_down=False, alpha=False, palette=None):
    """kind 2: RGB(A), 3: grey, 1: colour-mapped (pixels = indices)."""
    px = np.asarray(pixels, np.uint8)
    h, w = px.shape[:2]
    if kind == 2:
        body = px[:, :, ::-1]
        if alpha:
            body = np.concatenate([body, np.full((h, w, 1), 200, np.uint8)], axis=2)
    else:
        body = px.reshape(h, w, 1)
    nb = body.shape[2]
    if not top_down:
        body = body[::-1]
    flat = np.ascontiguousarray(body).reshape(-1, nb)
    if rle:
        out = bytearray()
        i = 0
        n = flat.shape[0]
        while i < n:
            run = 1
            while i + run < n and run < 128 and np.array_equal(flat[i + run], flat[i]):
                run += 1
            if run > 1:
                out.append(0x80 | (run - 1))
                out += flat[i].tobytes()
                i += run
            else:
                lit = 1
                while i + lit < n and lit < 128 and not np.array_equal(flat[i + lit], flat[i + lit - 1]):
                    lit += 1
                out.append(lit - 1)
                out += flat[i:i + lit].tobytes()
                i += lit
        data = bytes(out)
    else:
        data = flat.tobytes()
    cmap = b""
    cm = (0, 0, 0)
    if kind == 1:
        p = np.asarray(palette, np.uint8).reshape(-1, 3)
        cmap = p[:, ::-1].tobytes()
        cm = (0, p.shape[0], 24)
    ident = b"id"
    hdr = struct.pack("<BBBHHBHHHHBB", len(ident), 1 if kind == 1 else 0, kind + (8 if rle else 0), cm[0], cm[1], cm[2],
                      0, 0, w, h, 8 * nb, (0x20 if top_down else 0) | (8 if (kind == 2 and alpha) else 0))
    return hdr + ident + cmap + data


def write_pnm(pixels, maxval=255):
    px = np.asarray(pixels)
    h, w = px.shape[:2]
    grey = px.ndim == 2 or px.shape[2] == 1
    head = ("P5" if grey else "P6") + f"\n# made by the test suite\n{w} {h}\n{maxval}\n"
    body = px.astype(">u2").tobytes() if maxval > 255 else px.astype(np.uint8).tobytes()
    return head.encode() + body


# ---- a small baseline JPEG writer: files with sampling factors, colour models and marker layouts that common encoders
# ---- do not produce (test inputs for the JPEG decoder; what the reference's stb_image makes of them is the fixture)
_ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                    28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                    54, 47, 55, 62, 63])


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, length):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)                   # byte stuffing
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)     # pad with 1-bits


def _segment(marker, payload):
    return bytes([0xFF, marker]) + struct.pack(">H", len(payload) + 2) + payload


def write_jpeg(planes, sampling, quant=8, ids=None, interleaved=True, restart=0, adobe=None, jfif=True, wide_dqt=False,
               fill_bytes=False, dnl=False):
    """planes: full-resolution (H, W) uint8 arrays, one per component; sampling: [(h, v), ...].  Flat Huffman tables
    (4-bit DC categories, 8-bit AC symbols), one quantisation table (`quant`: scalar or 64 values, natural order)."""
    from scipy.fft import dctn
    n_comp = len(planes)
    height, width = planes[0].shape
    ids = list(ids) if ids is not None else list(range(1, n_comp + 1))
    h_max, v_max = max(s[0] for s in sampling), max(s[1] for s in sampling)
    mcu_x, mcu_y = -(-width // (8 * h_max)), -(-height // (8 * v_max))
    q = np.broadcast_to(np.asarray(quant, np.float64).reshape(-1), (64,)).reshape(8, 8)
    coeffs = []                                                    # per component: (blocks_y, blocks_x, 64) zigzag ints
    for plane, (h, v) in zip(planes, sampling):
        cx, cy = -(-width * h // h_max), -(-height * v // v_max)
        ys = np.minimum(np.arange(cy) * v_max // v, height - 1)     # point-sample the full-resolution plane
        xs = np.minimum(np.arange(cx) * h_max // h, width - 1)
        sub = np.asarray(plane, np.float64)[np.ix_(ys, xs)]
        sub = np.pad(sub, ((0, mcu_y * v * 8 - cy), (0, mcu_x * h * 8 - cx)), mode="edge") - 128.0
        by, bx = sub.shape[0] // 8, sub.shape[1] // 8
        blocks = sub.reshape(by, 8, bx, 8).transpose(0, 2, 1, 3)
        c = np.rint(dctn(blocks, axes=(2, 3), norm="ortho") / q).astype(np.int64).reshape(by, bx, 64)[:, :, _ZIGZAG]
        coeffs.append((c, -(-cx // 8), -(-cy // 8)))
    dc_len, ac_syms = 4, [0x00, 0xF0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]
    ac_code = {sym: k for k, sym in enumerate(ac_syms)}

    def size_of(v):
        return int(abs(int(v))).bit_length()

    def put_value(bits, v, s):
        bits.put(v if v >= 0 else v + (1 << s) - 1, s)

    def put_block(bits, zz, pred):
        diff = int(zz[0]) - pred
        s = size_of(diff)
        bits.put(s, dc_len)
        if s:
            put_value(bits, diff, s)
        run = 0
        last = np.flatnonzero(zz[1:])
        last = int(last[-1]) + 1 if last.size else 0
        for k in range(1, last + 1):
            v = int(zz[k])
            if v == 0:
                run += 1
                continue
            while run > 15:
                bits.put(ac_code[0xF0], 8)
                run -= 16
            s = size_of(v)
            bits.put(ac_code[(run << 4) | s], 8)
            put_value(bits, v, s)
            run = 0
        if last < 63:
            bits.put(ac_code[0x00], 8)
        return int(zz[0])

    def scan(components):
        head = bytes([len(components)]) + b"".join(bytes([ids[k], 0x00]) for k in components) + bytes([0, 63, 0])
        out = bytearray(_segment(0xDA, head))
        bits, pred, done, rst = _Bits(), [0] * n_comp, 0, 0
        if len(components) == 1:
            k = components[0]
            c, bw, bh = coeffs[k]
            units = [[(k, j, i)] for j in range(bh) for i in range(bw)]
        else:
            units = [[(k, j * sampling[k][1] + y, i * sampling[k][0] + x) for k in components
                      for y in range(sampling[k][1]) for x in range(sampling[k][0])] for j in range(mcu_y) for i in range(mcu_x)]
        for u, unit in enumerate(units):
            for k, by, bx in unit:
                pred[k] = put_block(bits, coeffs[k][0][by, bx], pred[k])
            done += 1
            if restart and done == restart and u + 1 < len(units):
                bits.flush()
                out += bits.out + (b"\xff\xff" if fill_bytes else b"") + bytes([0xFF, 0xD0 + rst])
                bits, pred, done, rst = _Bits(), [0] * n_comp, 0, (rst + 1) & 7
        bits.flush()
        return bytes(out + bits.out)

    data = bytearray(b"\xff\xd8")
    if jfif:
        data += _segment(0xE0, b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0")
    if adobe is not None:
        data += _segment(0xEE, b"Adobe\0\x64\0\0\0\0" + bytes([adobe]))
    data += _segment(0xFE, b"made by the test suite")
    qz = np.rint(q.reshape(-1)[_ZIGZAG]).astype(np.int64)
    data += _segment(0xDB, (b"\x10" + qz.astype(">u2").tobytes()) if wide_dqt else (b"\x00" + qz.astype(np.uint8).tobytes()))
    if fill_bytes:
        data += b"\xff\xff"                                         # fill bytes before a marker (T.81 B.1.1.2)
    data += _segment(0xC0, struct.pack(">BHHB", 8, 0 if False else height, width, n_comp)
                     + b"".join(bytes([ids[k], (sampling[k][0] << 4) | sampling[k][1], 0]) for k in range(n_comp)))
    data += _segment(0xC4, bytes([0x00]) + bytes([0, 0, 0, 12] + [0] * 12) + bytes(range(12)))
    data += _segment(0xC4, bytes([0x10]) + bytes([0] * 7 + [len(ac_syms)] + [0] * 8) + bytes(ac_syms))
    if restart:
        data += _segment(0xDD, struct.pack(">H", restart))
    if interleaved and n_comp > 1:
        data += scan(list(range(n_comp)))
        if dnl:
            data += _segment(0xDC, struct.pack(">H", height))
    else:
        for k in range(n_comp):
            data += scan([k])
    return bytes(data + b"\xff\xd9")
