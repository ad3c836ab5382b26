"""ctypes binding of oracle/_ref/libstbref.so — the reference's own stb_image (Caitlyn/stb_image.h), compiled from where it
lies in /root/reference.  TEST INFRASTRUCTURE ONLY (tests/ and tests/golden/make_stb_fixtures.py); exists in the build
container only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libstbref.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(LIB_PATH)
        l.ref_stbi_load_rgb.restype = C.c_void_p
        l.ref_stbi_load_rgb.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.ref_stbi_free.restype = None
        l.ref_stbi_free.argtypes = [C.c_void_p]
        l.ref_stbi_failure.restype = C.c_char_p
        _lib = l
    return _lib


def decode_rgb(file_bytes):
    """stbi_load_from_memory(bytes, ..., 3) -> (H, W, 3) uint8 (top row first), or None if stb refuses the file."""
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    b = bytes(file_bytes)
    p = lib().ref_stbi_load_rgb(b, len(b), C.byref(w), C.byref(h), C.byref(c))
    if not p:
        return None
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(h.value, w.value, 3)).copy()
    finally:
        lib().ref_stbi_free(p)
