"""ctypes binding of oracle/_ref/librndref.so — the reference's own Caitlyn/Rnd.h compiled from where it lies in
/root/reference.  TEST INFRASTRUCTURE ONLY; exists in the build container only."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "librndref.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(LIB_PATH)
        l.ref_rnd_set_state.argtypes = [C.c_uint32]
        l.ref_rnd_state.restype = C.c_uint32
        l.ref_randf2.restype = C.c_float
        l.ref_randf.restype = C.c_float
        l.ref_pcg_hash.restype = C.c_uint32
        l.ref_pcg_hash.argtypes = [C.c_uint32]
        _lib = l
    return _lib
