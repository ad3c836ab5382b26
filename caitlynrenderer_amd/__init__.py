"""MI355X-native ray/BVH-traversal hot path of a Caitlyn-style path tracer.

The compute path lives in libcrt.so (hand-written HIP for gfx950 behind the C ABI of
include/crt.h); this package is the thin host-side mirror of the reference's Scene / Camera /
SBVH / CWBVH interface plus tile sharding helpers.  Importing it without a built libcrt.so fails.
"""
from . import _lib
from ._lib import CRT_TRACE_ANY, CRT_TRACE_BVH2, CRT_TRACE_CLOSEST, CRT_TRACE_TIE_LOWEST_ID, CrtError
from .host import CWBVH, SBVH, Camera, Mesh, Rnd, pcg_hash
from .scene import HIT_DT, RAY_DT, STATS_DT, Scene, SceneData

_lib.lib()   # fail loudly at import time if the HIP extension is missing


def has_experiments():
    """True when libcrt.so was built with the experimental kernel variants (make EXPERIMENTS=1, include/crt.h crt_set_option)."""
    return bool(_lib.lib().crt_has_experiments())


def warmup():
    """crt_warmup: HIP context + the library's code objects on the current device, so that the first scene does not pay for them."""
    _lib.check(_lib.lib().crt_warmup())


__all__ = ["has_experiments", "warmup", "Scene", "SceneData", "Camera", "Mesh", "SBVH", "CWBVH", "Rnd", "pcg_hash", "CrtError",
           "RAY_DT", "HIT_DT", "STATS_DT", "CRT_TRACE_CLOSEST", "CRT_TRACE_ANY", "CRT_TRACE_BVH2", "CRT_TRACE_TIE_LOWEST_ID"]
