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


def write_png(pixels, color_type=2, depth=8, palette=None, filters="cycle"):
    """pixels: (H, W, C) samples (C by colour type: 0 grey, 2 RGB, 3 index, 4 grey+alpha, 6 RGBA)."""
    px = np.asarray(pixels)
    h, w = px.shape[:2]
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    px = px.reshape(h, w, ch)
    if depth == 16:
        rows = px.astype(">u2").tobytes()
        stride = w * ch * 2
    elif depth == 8:
        rows = px.astype(np.uint8).tobytes()
        stride = w * ch
    else:
        bits = np.unpackbits(px.astype(np.uint8).reshape(h, w, 1), axis=2)[:, :, 8 - depth:].reshape(h, -1)
        pad = (-bits.shape[1]) % 8
        bits = np.pad(bits, ((0, 0), (0, pad)))
        rows = np.packbits(bits, axis=1).tobytes()
        stride = bits.shape[1] // 8
    bpp = max(1, ch * depth // 8)
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
    data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 0))
    if palette is not None:
        data += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
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
