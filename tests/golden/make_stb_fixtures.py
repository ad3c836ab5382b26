"""Golden vectors from the REFERENCE's own image decoder: stb_image as vendored in /root/reference/Caitlyn/stb_image.h,
compiled from where it lies into oracle/_ref/libstbref.so (oracle/ref_stb.c, `make -C oracle ref`).  Run in the BUILD
container only:

    python tests/golden/make_stb_fixtures.py

Writes tests/golden/stb_decodes.npz: for every test file its bytes (`<name>__file`) and what the reference's
`stbi_load_from_memory(..., 3)` (the call of Caitlyn/Scene.h:619) returns for it (`<name>__rgb`; shape (0, 0, 3) when stb
refuses the file).  tests/test_textures.py holds the product's decoders (csrc/host/image.cpp) to these bytes on every
machine, and to the live library where /root/reference exists.  The files are generated here (oracle/textures.py writers,
PIL for the JPEGs): DATA, not reference source.
"""
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import stbref, textures as T      # noqa: E402


def test_files():
    """name -> file bytes; deterministic."""
    rng = np.random.default_rng(2024)
    out = {}
    h, w = 23, 31
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rgb[3:9, 5:20] = rgb[3, 5]
    rgba = np.concatenate([rgb, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)
    grey = rng.integers(0, 256, (h, w, 1), dtype=np.uint8)
    for ft in ("cycle", 0, 1, 2, 3, 4):
        out[f"png_rgb8_filter_{ft}"] = T.write_png(rgb, 2, 8, filters=ft)
    out["png_rgba8"] = T.write_png(rgba, 6, 8)
    out["png_grey8"] = T.write_png(grey, 0, 8)
    out["png_greyalpha8"] = T.write_png(np.concatenate([grey, 255 - grey], axis=2), 4, 8)
    for depth in (1, 2, 4):
        out[f"png_grey{depth}"] = T.write_png(rng.integers(0, 1 << depth, (h, w, 1), dtype=np.uint8), 0, depth)
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    for depth in (1, 2, 4, 8):
        out[f"png_palette{depth}"] = T.write_png(rng.integers(0, min(16, 1 << depth), (h, w, 1), dtype=np.uint8), 3, depth, palette=pal)
    for (hh, ww) in ((23, 31), (1, 1), (3, 2), (9, 5)):
        out[f"png_adam7_rgb8_{hh}x{ww}"] = T.write_png(rgb[:hh, :ww], 2, 8, interlace=True)
    out["png_adam7_grey2"] = T.write_png(rng.integers(0, 4, (h, w, 1), dtype=np.uint8), 0, 2, interlace=True)
    out["png_adam7_palette4"] = T.write_png(rng.integers(0, 16, (h, w, 1), dtype=np.uint8), 3, 4, palette=pal, interlace=True)
    out["png_adam7_rgba16"] = T.write_png(rng.integers(0, 65536, (h, w, 4), dtype=np.uint16), 6, 16, interlace=True)
    out["png_rgb8_trns"] = T.write_png(rgb, 2, 8, trns=bytes([0, rgb[0, 0, 0], 0, rgb[0, 0, 1], 0, rgb[0, 0, 2]]))
    out["png_palette4_trns"] = T.write_png(rng.integers(0, 16, (h, w, 1), dtype=np.uint8), 3, 4, palette=pal, trns=bytes(range(0, 160, 10)))
    out["png_grey4_trns"] = T.write_png(rng.integers(0, 16, (h, w, 1), dtype=np.uint8), 0, 4, trns=bytes([0, 3]))
    out["png_rgb16"] = T.write_png(rng.integers(0, 65536, (h, w, 3), dtype=np.uint16), 2, 16)
    out["png_grey16"] = T.write_png(rng.integers(0, 65536, (h, w, 1), dtype=np.uint16), 0, 16)
    pal2 = rng.integers(0, 256, (200, 3), dtype=np.uint8)
    idx = rng.integers(0, 200, (h, w), dtype=np.uint8)
    for td in (False, True):
        out[f"bmp24_td{int(td)}"] = T.write_bmp(rgb, 24, td)
        out[f"bmp32_td{int(td)}"] = T.write_bmp(rgb, 32, td)
        out[f"bmp8_td{int(td)}"] = T.write_bmp(idx, 8, td, palette=pal2)
        for rle in (False, True):
            out[f"tga_rgb_td{int(td)}_rle{int(rle)}"] = T.write_tga(rgb, 2, rle, td)
            out[f"tga_rgba_td{int(td)}_rle{int(rle)}"] = T.write_tga(rgb, 2, rle, td, alpha=True)
            out[f"tga_grey_td{int(td)}_rle{int(rle)}"] = T.write_tga(grey[..., 0], 3, rle, td)
            out[f"tga_cmap_td{int(td)}_rle{int(rle)}"] = T.write_tga(idx, 1, rle, td, palette=pal2)
    out["pnm_p6"] = T.write_pnm(rgb)
    out["pnm_p5"] = T.write_pnm(grey[..., 0])
    out["pnm_p6_16"] = T.write_pnm(rng.integers(0, 65536, (7, 5, 3), dtype=np.uint16), 65535)
    out["pnm_p5_16"] = T.write_pnm(rng.integers(0, 65536, (7, 5), dtype=np.uint16), 65535)
    # JPEG (lossy: only the reference's own decoder defines the bytes).  Smooth content + noise, odd sizes for the MCU padding.
    try:
        from PIL import Image
    except Exception:
        Image = None
    if Image is not None:
        yy, xx = np.mgrid[0:37, 0:53]
        smooth = np.stack([128 + 100 * np.sin(xx / 7.0) * np.cos(yy / 5.0), 40 + 3 * xx + yy, 255 - 4 * yy + 0 * xx], 2)
        img = np.clip(smooth + rng.normal(0, 12, smooth.shape), 0, 255).astype(np.uint8)

        def jpg(arr, mode="RGB", **kw):
            b = io.BytesIO()
            Image.fromarray(arr if mode == "RGB" else arr[..., 0], mode).save(b, "JPEG", **kw)
            return b.getvalue()
        out["jpeg_444_q90"] = jpg(img, quality=90, subsampling=0)
        out["jpeg_422_q85"] = jpg(img, quality=85, subsampling=1)
        out["jpeg_420_q75"] = jpg(img, quality=75, subsampling=2)
        out["jpeg_420_q30_optimized"] = jpg(img, quality=30, subsampling=2, optimize=True)
        out["jpeg_grey_q80"] = jpg(img, mode="L", quality=80)
        out["jpeg_progressive_420_q80"] = jpg(img, quality=80, subsampling=2, progressive=True)
        out["jpeg_progressive_444_q95"] = jpg(img, quality=95, subsampling=0, progressive=True)
        out["jpeg_restart_420"] = jpg(img, quality=80, subsampling=2, restart_marker_blocks=2) if "restart_marker_blocks" in Image.core.__dict__ else jpg(img[:16, :16], quality=80, subsampling=2)
        out["jpeg_16x16_444"] = jpg(img[:16, :16], quality=92, subsampling=0)
        out["jpeg_8x8_420"] = jpg(img[:8, :8], quality=92, subsampling=2)
        out["jpeg_1x1"] = jpg(img[:1, :1], quality=90)
        out["jpeg_q100_noise"] = jpg(rng.integers(0, 256, (24, 40, 3), dtype=np.uint8), quality=100, subsampling=0)
        out["jpeg_q1"] = jpg(img, quality=1, subsampling=2)
        b = io.BytesIO()
        Image.fromarray(np.concatenate([img, img[..., :1]], 2), "CMYK").save(b, "JPEG", quality=85)
        out["jpeg_cmyk_adobe"] = b.getvalue()
        out["jpeg_keep_rgb"] = jpg(img, quality=85, subsampling=0, keep_rgb=True)
        out["jpeg_progressive_restart"] = jpg(img, quality=70, subsampling=2, progressive=True, restart_marker_blocks=3)
        out["jpeg_progressive_grey"] = jpg(img, mode="L", quality=70, progressive=True)
    # files from the test suite's own baseline writer: sampling factors, colour models and marker layouts encoders avoid
    yy, xx = np.mgrid[0:29, 0:43]
    planes = [np.clip(128 + 110 * np.sin(xx / (3.0 + c)) * np.cos(yy / (2.5 + c)) + rng.normal(0, 6, xx.shape), 0, 255).astype(np.uint8)
              for c in range(4)]
    for name, samp in (("411", [(4, 1), (1, 1), (1, 1)]), ("440", [(1, 2), (1, 1), (1, 1)]), ("410", [(4, 2), (1, 1), (1, 1)]),
                       ("h2v4", [(2, 4), (1, 1), (1, 1)]), ("chroma_mixed", [(2, 2), (2, 1), (1, 2)]), ("luma_sub", [(1, 1), (2, 2), (2, 2)]),
                       ("h3v3", [(3, 3), (1, 1), (1, 1)]), ("frac_interleaved", [(4, 2), (3, 1), (3, 2)])):
        out[f"jpeg_own_{name}"] = T.write_jpeg(planes[:3], samp, quant=6)
    out["jpeg_own_rgb_ids"] = T.write_jpeg(planes[:3], [(1, 1)] * 3, quant=4, ids=[ord("R"), ord("G"), ord("B")])
    out["jpeg_own_adobe0_no_jfif"] = T.write_jpeg(planes[:3], [(2, 1), (1, 1), (1, 1)], quant=4, adobe=0, jfif=False)
    out["jpeg_own_adobe0_with_jfif"] = T.write_jpeg(planes[:3], [(2, 1), (1, 1), (1, 1)], quant=4, adobe=0, jfif=True)
    out["jpeg_own_cmyk"] = T.write_jpeg(planes, [(1, 1)] * 4, quant=5, adobe=0, jfif=False)
    out["jpeg_own_ycck_420"] = T.write_jpeg(planes, [(2, 2), (1, 1), (1, 1), (2, 2)], quant=5, adobe=2, jfif=False)
    out["jpeg_own_four_no_adobe"] = T.write_jpeg(planes, [(1, 1)] * 4, quant=5)
    out["jpeg_own_non_interleaved_420"] = T.write_jpeg(planes[:3], [(2, 2), (1, 1), (1, 1)], quant=6, interleaved=False)
    out["jpeg_own_non_interleaved_restart"] = T.write_jpeg(planes[:3], [(2, 1), (1, 1), (1, 1)], quant=6, interleaved=False, restart=3)
    out["jpeg_own_restart_fill_dnl"] = T.write_jpeg(planes[:3], [(2, 2), (1, 1), (1, 1)], quant=6, restart=2, fill_bytes=True, dnl=True)
    out["jpeg_own_wide_dqt"] = T.write_jpeg(planes[:3], [(1, 1)] * 3, quant=300, wide_dqt=True)
    out["jpeg_own_grey_2x2"] = T.write_jpeg(planes[:1], [(2, 2)], quant=3)
    out["jpeg_own_frac_non_interleaved"] = T.write_jpeg(planes[:3], [(4, 3), (3, 2), (3, 3)], quant=6, interleaved=False)
    return out


def main():
    assert stbref.available(), "oracle/_ref/libstbref.so is missing: run `make -C oracle ref` (needs /root/reference)"
    files = test_files()
    out = {}
    for name, data in files.items():
        out[name + "__file"] = np.frombuffer(data, np.uint8)
        # what the reference's decoder returns for this one depends on memory it never wrote (sampling factors that do not
        # divide the largest, one scan per component): recorded as refused, which is what the product does
        rgb = None if name == "jpeg_own_frac_non_interleaved" else stbref.decode_rgb(data)
        out[name + "__rgb"] = rgb if rgb is not None else np.zeros((0, 0, 3), np.uint8)
        print(f"{name:32s} {len(data):6d} bytes -> {None if rgb is None else rgb.shape}")
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stb_decodes.npz"), **out)


if __name__ == "__main__":
    main()
