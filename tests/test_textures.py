"""Texture path of the loader (Scene.h:597-710): the image decoders (PNG, BMP, TGA, PNM, JPEG), the
reference's bilinear resize + byte truncation, and map_Kd handling.  Checkers: the known pixels the test files were
written from; oracle/textures.py (numpy restatement of Scene.h:321-371); and — for every decoder — the bytes the
reference's OWN decoder returns for the same files: tests/golden/stb_decodes.npz, made by
tests/golden/make_stb_fixtures.py from oracle/_ref/libstbref.so (the stb_image.h the reference vendors, compiled
where it lies).  No GPU needed."""
import os

import numpy as np
import pytest

from conftest import write_obj
from oracle import textures as T


@pytest.fixture(scope="module")
def host():
    import __graft_entry__ as g
    g.build()
    from caitlynrenderer_amd import host
    return host


def test_png_decoder_all_colour_types_depths_and_filters(host):
    rng = np.random.default_rng(5)
    h, w = 23, 31                                   # odd sizes: partial bytes at low bit depths
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    for ft in ("cycle", 0, 1, 2, 3, 4):
        assert np.array_equal(host.decode_image(T.write_png(rgb, 2, 8, filters=ft)), rgb), ft
    rgba = np.concatenate([rgb, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)
    assert np.array_equal(host.decode_image(T.write_png(rgba, 6, 8)), rgb)                      # alpha dropped, not multiplied
    grey = rng.integers(0, 256, (h, w, 1), dtype=np.uint8)
    assert np.array_equal(host.decode_image(T.write_png(grey, 0, 8)), grey.repeat(3, 2))
    ga = np.concatenate([grey, 255 - grey], axis=2)
    assert np.array_equal(host.decode_image(T.write_png(ga, 4, 8)), grey.repeat(3, 2))
    for depth, scale in ((1, 255), (2, 85), (4, 17)):                                            # grey scaled to 0..255
        g = rng.integers(0, 1 << depth, (h, w, 1), dtype=np.uint8)
        assert np.array_equal(host.decode_image(T.write_png(g, 0, depth)), (g * scale).astype(np.uint8).repeat(3, 2)), depth
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    for depth in (1, 2, 4, 8):                                                                   # palette indices are not scaled
        idx = rng.integers(0, min(16, 1 << depth), (h, w, 1), dtype=np.uint8)
        assert np.array_equal(host.decode_image(T.write_png(idx, 3, depth, palette=pal)), pal[idx[..., 0]]), depth
    deep = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
    assert np.array_equal(host.decode_image(T.write_png(deep, 2, 16)), (deep >> 8).astype(np.uint8))   # high byte


def test_bmp_tga_pnm_decoders(host):
    rng = np.random.default_rng(6)
    rgb = rng.integers(0, 256, (19, 27, 3), dtype=np.uint8)                 # 27*3 = 81: BMP rows need padding
    rgb[3:9, 5:20] = rgb[3, 5]                                              # runs for the RLE encoder
    grey = rng.integers(0, 256, (19, 27), dtype=np.uint8)
    pal = rng.integers(0, 256, (200, 3), dtype=np.uint8)
    idx = rng.integers(0, 200, (19, 27), dtype=np.uint8)
    for td in (False, True):
        assert np.array_equal(host.decode_image(T.write_bmp(rgb, 24, td)), rgb)
        assert np.array_equal(host.decode_image(T.write_bmp(rgb, 32, td)), rgb)
        assert np.array_equal(host.decode_image(T.write_bmp(idx, 8, td, palette=pal)), pal[idx])
        for rle in (False, True):
            assert np.array_equal(host.decode_image(T.write_tga(rgb, 2, rle, td)), rgb), (td, rle)
            assert np.array_equal(host.decode_image(T.write_tga(rgb, 2, rle, td, alpha=True)), rgb)
            assert np.array_equal(host.decode_image(T.write_tga(grey, 3, rle, td)), grey[..., None].repeat(3, 2))
            assert np.array_equal(host.decode_image(T.write_tga(idx, 1, rle, td, palette=pal)), pal[idx])
    assert np.array_equal(host.decode_image(T.write_pnm(rgb)), rgb)
    assert np.array_equal(host.decode_image(T.write_pnm(grey)), grey[..., None].repeat(3, 2))


GOLDEN_STB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stb_decodes.npz")


def _stb_cases():
    z = np.load(GOLDEN_STB)
    return z, sorted(k[:-6] for k in z.files if k.endswith("__file"))


def test_every_decoder_returns_the_bytes_of_the_references_stb_image(host):
    """The fixture holds, per file, what the reference's stbi_load_from_memory(..., 3) returned (an empty array where it
    refused).  JPEG included: lossy, so only the reference's decoder defines the bytes; the product must equal them."""
    import caitlynrenderer_amd as cr
    z, names = _stb_cases()
    kinds = set()
    for name in names:
        data, want = z[name + "__file"].tobytes(), z[name + "__rgb"]
        if want.size == 0:
            with pytest.raises(cr.CrtError):
                host.decode_image(data)
        else:
            got = host.decode_image(data)
            assert got.shape == want.shape and np.array_equal(got, want), name
        kinds.add(name[:3])
    assert kinds == {"png", "bmp", "tga", "pnm", "jpe"} and len(names) >= 90


def test_stb_fixture_is_what_the_reference_decoder_returns_today():
    """Where the reference is present (build container) the fixture is re-derived from the live library."""
    from oracle import stbref
    if not stbref.available():
        pytest.skip("oracle/_ref/libstbref.so not built (no /root/reference on this machine)")
    z, names = _stb_cases()
    for name in names:
        if name == "jpeg_own_frac_non_interleaved":      # the reference decoder's output depends on uninitialised memory
            continue
        got = stbref.decode_rgb(z[name + "__file"].tobytes())
        want = z[name + "__rgb"]
        assert (got is None and want.size == 0) or (got is not None and np.array_equal(got, want)), name


def test_png_adam7_and_transparency_chunk(host):
    rng = np.random.default_rng(11)
    for h, w in ((23, 31), (1, 1), (2, 3), (5, 1), (8, 8), (9, 17)):             # sizes where some of the 7 passes are empty
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(host.decode_image(T.write_png(rgb, 2, 8, interlace=True)), rgb), (h, w)
        g = rng.integers(0, 4, (h, w, 1), dtype=np.uint8)
        assert np.array_equal(host.decode_image(T.write_png(g, 0, 2, interlace=True)), (g * 85).astype(np.uint8).repeat(3, 2))
        deep = rng.integers(0, 65536, (h, w, 4), dtype=np.uint16)
        assert np.array_equal(host.decode_image(T.write_png(deep, 6, 16, interlace=True)), (deep[..., :3] >> 8).astype(np.uint8))
    rgb = rng.integers(0, 256, (6, 7, 3), dtype=np.uint8)
    key = bytes([0, rgb[0, 0, 0], 0, rgb[0, 0, 1], 0, rgb[0, 0, 2]])
    assert np.array_equal(host.decode_image(T.write_png(rgb, 2, 8, trns=key)), rgb)         # colour key: dropped with the alpha


def _jpeg_planes(rng, h, w, n):
    yy, xx = np.mgrid[0:h, 0:w]
    return [np.clip(128 + 110 * np.sin(xx / (2.0 + c)) * np.cos(yy / (3.0 + c)) + rng.normal(0, 5, (h, w)), 0, 255).astype(np.uint8)
            for c in range(n)]


def test_jpeg_decoder_properties(host):
    """Properties that hold whatever the decoder's rounding: a grey file decodes to the encoded plane within the
    quantisation error; the result does not depend on how the scans are laid out (interleaved, one scan per component,
    restart intervals, fill bytes): the same coefficients must give the same bytes."""
    import caitlynrenderer_amd as cr
    rng = np.random.default_rng(12)
    (y,) = _jpeg_planes(rng, 40, 52, 1)
    out = host.decode_image(T.write_jpeg([y], [(1, 1)], quant=1))
    assert out.shape == (40, 52, 3) and np.array_equal(out[..., 0], out[..., 1]) and np.array_equal(out[..., 0], out[..., 2])
    assert np.abs(out[..., 0].astype(int) - y.astype(int)).max() <= 4
    planes = _jpeg_planes(rng, 29, 43, 3)
    for samp in ([(1, 1)] * 3, [(2, 2), (1, 1), (1, 1)], [(2, 1), (1, 1), (1, 1)], [(1, 2), (1, 1), (1, 1)], [(4, 1), (1, 1), (1, 1)]):
        base = host.decode_image(T.write_jpeg(planes, samp, quant=6))
        for kw in (dict(interleaved=False), dict(restart=1), dict(restart=3, fill_bytes=True), dict(interleaved=False, restart=2), dict(dnl=True)):
            assert np.array_equal(host.decode_image(T.write_jpeg(planes, samp, quant=6, **kw)), base), (samp, kw)
    # RGB component ids: no colour conversion, the planes come back (within quantisation error)
    rgb = host.decode_image(T.write_jpeg(planes, [(1, 1)] * 3, quant=1, ids=[ord("R"), ord("G"), ord("B")]))
    assert np.abs(rgb.astype(int) - np.stack(planes, 2).astype(int)).max() <= 4
    # damaged: truncated, no end marker, 12-bit samples, arithmetic coding, a scan before any table
    good = T.write_jpeg(planes, [(2, 2), (1, 1), (1, 1)], quant=6)
    sof = good.find(b"\xff\xc0")
    twelve = bytearray(good); twelve[sof + 4] = 12
    arith = bytearray(good); arith[sof + 1] = 0xC9
    for data in (good[:len(good) // 2], good[:-2], bytes(twelve), bytes(arith), b"\xff\xd8\xff\xda\x00\x08\x01\x01\x00\x00\x3f\x00\xff\xd9",
                 b"\xff\xd8\xff\xe0" + b"\0" * 64):
        with pytest.raises(cr.CrtError):
            host.decode_image(data)


def test_entropy_data_that_ends_inside_a_coefficient_is_refused(host):
    """a marker in mid-coefficient.  The bit reader adds nothing once it has met a marker; a coefficient that needs more
    bits than are left is refused (the bit count used to go negative, and the next refill shifted by more than 31)."""
    import caitlynrenderer_amd as cr
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "stb_decodes.npz"))
    names = [k[:-6] for k in z.files if k.endswith("__file") and k.startswith("jpeg")]
    assert names
    refused = 0
    for name in names[:12]:
        good = bytes(z[name + "__file"])
        sos = good.rfind(b"\xff\xda")
        body = sos + 2 + ((good[sos + 2] << 8) | good[sos + 3])           # first byte of the last scan's entropy-coded data
        for cut in (body + 1, body + 3, body + (len(good) - body) // 2):
            data = good[:cut] + b"\xff\xd9"                                # EOI right inside the entropy-coded segment
            try:
                host.decode_image(data)
            except cr.CrtError:
                refused += 1
    assert refused > 0                                                    # no crash, no hang; the truncated scans are refused


def test_refused_and_damaged_files(host):
    import caitlynrenderer_amd as cr
    from caitlynrenderer_amd import _lib
    rgb = np.zeros((4, 4, 3), np.uint8)
    png = T.write_png(rgb)
    interlaced = bytearray(png); interlaced[28] = 2                          # IHDR interlace byte: 0 or 1 (CRC is not checked)
    for data in (bytes(interlaced), png[:40], T.write_pnm(np.zeros((4, 4, 3), np.uint16), 65535),   # 16-bit PNM: stb_image refuses
                 T.write_bmp(rgb)[:60], T.write_tga(rgb)[:20], b"P6\n4 4\n255\n" + b"\0" * 10,
                 b"not an image at all, just some text that is long enough"):
        with pytest.raises(cr.CrtError) as e:
            host.decode_image(data)
        assert e.value.code == _lib.CRT_ERR_INVALID
    # headers of a few bytes that claim gigabytes of pixels are refused before anything is allocated
    import struct
    huge_png = bytearray(png); huge_png[16:24] = struct.pack(">II", 30000, 30000)
    huge_tga = bytearray(T.write_tga(rgb)); huge_tga[12:16] = struct.pack("<HH", 30000, 30000)
    huge_rle = bytearray(huge_tga); huge_rle[2] = 10
    for data in (bytes(huge_png), bytes(huge_tga), bytes(huge_rle)):
        with pytest.raises(cr.CrtError) as e:
            host.decode_image(data)
        assert e.value.code == _lib.CRT_ERR_INVALID


@pytest.mark.parametrize("size", [(256, 256), (300, 300), (512, 384), (100, 70), (256, 100), (64, 512), (1, 1), (257, 255), (3, 1000)])
def test_resize_to_texture_array_matches_the_numpy_restatement(host, size):
    """Scene.h:321-371 in fp32 with the reference's operation order, then float -> unsigned char truncation; includes
    up-scaling, where ceil() indexes one column/row past the source (flat indexing, clamped at the very end)."""
    h, w = size
    rgb = np.random.default_rng(h * 1000 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    got, want = host.texture_to_array_bytes(rgb), T.texture_to_array_bytes(rgb)
    assert got.shape == (256, 256, 3) and np.array_equal(got, want)
    assert np.array_equal(host.texture_to_array_bytes(rgb, 64, 32), T.texture_to_array_bytes(rgb, 64, 32))


def test_same_size_path_is_the_identity(host):
    """255 * (b * (1/255.f)) truncated to a byte (Scene.h:653-661) gives b back for all 256 values."""
    ramp = np.arange(256, dtype=np.uint8).reshape(1, 256, 1).repeat(256, 0).repeat(3, 2)
    assert np.array_equal(host.texture_to_array_bytes(ramp), ramp) and np.array_equal(T.texture_to_array_bytes(ramp), ramp)


def test_loader_map_kd(host, cornell, tmp_path):
    """map_Kd in the .mtl (Scene.h:597-677): layers in order of first use, name = what follows the last backslash,
    a texture's second user keeps tex_ind = -1 (Scene.h:604 has no else branch), files resized to 256x256."""
    import caitlynrenderer_amd as cr
    mesh, _ = cornell
    rng = np.random.default_rng(8)
    wood = rng.integers(0, 256, (300, 200, 3), dtype=np.uint8)
    tiles = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    open(tmp_path / "wood.png", "wb").write(T.write_png(wood))
    open(tmp_path / "tiles.tga", "wb").write(T.write_tga(tiles, rle=True))
    uv = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0]], np.float32)
    tris = mesh.triangles.copy()
    for q in range(tris.shape[0] // 2):
        tris[2 * q, 8:12] = (0, 1, 2, 0)
        tris[2 * q + 1, 8:12] = (0, 2, 3, 0)
    m = cr.Mesh(mesh.vertices, mesh.normals, uv, tris, mesh.materials, mesh.lights, mesh.vertex_min)
    write_obj(m, str(tmp_path / "scene.obj"), map_kd={1: "tiles.tga", 2: "textures\\\\wood.png", 3: "wood.png", 4: "tiles.tga"})
    got = cr.Mesh.read_object(str(tmp_path / "scene.obj"))
    assert got.albedo_textures is not None and got.albedo_textures.shape == (2, 256, 256, 3)
    assert np.array_equal(got.albedo_textures[0], tiles)                                       # same size: passed through
    assert np.array_equal(got.albedo_textures[1], T.texture_to_array_bytes(wood))
    assert got.materials[:, 12].tolist() == [-1.0, 0.0, 1.0, -1.0, -1.0, -1.0]                 # 3 and 4 reuse a name: no layer
    assert np.array_equal(got.texcoords, np.stack([uv[:, 0], np.float32(1) - (np.float32(1) - uv[:, 1])], axis=1))
    # a JPEG texture: the layer is the reference decoder's pixels (fixture) through the reference's resize
    z = np.load(GOLDEN_STB)
    open(tmp_path / "photo.jpg", "wb").write(z["jpeg_progressive_420_q80__file"].tobytes())
    write_obj(m, str(tmp_path / "jpg.obj"), mtl_name="jpg.mtl", map_kd={2: "photo.jpg"})
    got = cr.Mesh.read_object(str(tmp_path / "jpg.obj"))
    assert np.array_equal(got.albedo_textures[0], T.texture_to_array_bytes(z["jpeg_progressive_420_q80__rgb"]))
    # a missing or undecodable texture is an error, not a crash
    write_obj(m, str(tmp_path / "bad.obj"), mtl_name="bad.mtl", map_kd={1: "nowhere.png"})
    with pytest.raises(cr.CrtError):
        cr.Mesh.read_object(str(tmp_path / "bad.obj"))
