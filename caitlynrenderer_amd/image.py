"""Image output for the resolved frame (the step right after the hot path).

`Scene.resolve()` returns the RGBA8 result of Shader/output.fs with rows bottom-up like GL
(Quad.h:16-24); image files want the top row first."""
import numpy as np


def write_ppm(path, rgba):
    """Binary PPM (P6) of an (H, W, 4|3) uint8 bottom-up image, flipped to top-down."""
    img = np.asarray(rgba, dtype=np.uint8)[::-1, :, :3]
    h, w = img.shape[:2]
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(img).tobytes())


def write_png(path, rgba):
    """PNG of an (H, W, 4|3) uint8 bottom-up image (alpha kept when present), top row first in the file."""
    from .host import encode_png
    with open(path, "wb") as f:
        f.write(encode_png(np.asarray(rgba, dtype=np.uint8), bottom_up=True))


def read_ppm(path):
    """Inverse of write_ppm: returns the bottom-up (H, W, 3) uint8 image."""
    with open(path, "rb") as f:
        data = f.read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P6"
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], np.uint8, w * h * 3).reshape(h, w, 3)[::-1]
