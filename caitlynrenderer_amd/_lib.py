"""ctypes binding of libcrt.so (the C ABI declared in include/crt.h).

The library is the product: if it is missing this module raises — there is no Python or CPU
fallback for the traversal path.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C caitlynrenderer_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRT_LIB points the binding at another build of the same library (A/B measurements of kernel variants in one gpurun call)
LIB_PATH = os.environ.get("CRT_LIB") or os.path.join(_HERE, "libcrt.so")

CRT_ABI_VERSION = int(os.environ.get("CRT_LIB_ABI", "6"))      # CRT_LIB_ABI: with CRT_LIB, an older build of the library measured beside this one (tools/ab_run.sh)
CRT_OK, CRT_ERR_INVALID, CRT_ERR_NO_DEVICE, CRT_ERR_HIP, CRT_ERR_IO, CRT_ERR_LIMIT, CRT_ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6
CRT_TRACE_CLOSEST, CRT_TRACE_ANY, CRT_TRACE_BVH2, CRT_TRACE_TIE_LOWEST_ID = 0, 1, 2, 4
CRT_BUILD_LBVH_ON_DEVICE = 1


class CrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"crt error {code}: {msg}")
        self.code = code


class crt_camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("forward", C.c_float * 3), ("fov", C.c_float), ("focal_dist", C.c_float), ("aperture", C.c_float)]


class crt_scene_desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("vertices", C.c_void_p), ("n_vertices", C.c_size_t),
        ("normals", C.c_void_p), ("n_normals", C.c_size_t),
        ("texcoords", C.c_void_p), ("n_texcoords", C.c_size_t),
        ("triangles", C.c_void_p), ("n_triangles", C.c_size_t),
        ("tri_orig_ids", C.c_void_p),
        ("materials", C.c_void_p), ("n_materials", C.c_size_t),
        ("lights", C.c_void_p), ("n_lights", C.c_size_t),
        ("bvh", C.c_void_p), ("n_bvh", C.c_size_t),
        ("bvh8", C.c_void_p), ("n_bvh8", C.c_size_t),
        ("bvh8_tri_slots", C.c_void_p), ("n_bvh8_tris", C.c_size_t),
        ("albedo_textures", C.c_void_p), ("tex_width", C.c_uint32), ("tex_height", C.c_uint32), ("n_textures", C.c_uint32),
        ("width", C.c_uint32), ("height", C.c_uint32), ("max_depth", C.c_uint32), ("build_flags", C.c_uint32),
    ]


class crt_frame_stats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("any_rays", C.c_uint64), ("ms_total", C.c_float),
                ("ms_trace_closest", C.c_float), ("ms_trace_any", C.c_float), ("ms_shade", C.c_float),
                ("ms_raygen", C.c_float), ("n_trace_launches", C.c_uint32),
                ("nodes_closest", C.c_uint64), ("tris_closest", C.c_uint64), ("nodes_any", C.c_uint64), ("tris_any", C.c_uint64),
                ("stack_overflows", C.c_uint32),
                ("wave_steps_closest_nodes", C.c_uint64), ("wave_steps_closest_tris", C.c_uint64),
                ("wave_steps_any_nodes", C.c_uint64), ("wave_steps_any_tris", C.c_uint64), ("closest_hits", C.c_uint64),
                ("nodes_closest_uniform", C.c_uint64), ("nodes_any_uniform", C.c_uint64)]


class crt_bvh_info(C.Structure):
    _fields_ = [("n_nodes8", C.c_uint64), ("n_tris8", C.c_uint64), ("n_bvh2_nodes", C.c_uint64), ("max_depth8", C.c_uint64),
                ("built_on_device", C.c_uint32), ("bvh2_depth", C.c_uint32), ("build_wall_ms", C.c_float), ("build_upload_ms", C.c_float),
                ("build_lbvh_device_ms", C.c_float), ("build_convert_device_ms", C.c_float)]


# every symbol include/crt.h declares: name -> (restype, argtypes)
_P, _SZ, _I, _U32, _F = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_float
SYMBOLS = {
    "crt_scene_create": (_I, [C.POINTER(crt_scene_desc), C.POINTER(_P)]),
    "crt_scene_destroy": (_I, [_P]),
    "crt_set_camera": (_I, [_P, C.POINTER(crt_camera)]),
    "crt_render_frame": (_I, [_P, _F, _F]),
    "crt_render_frame_async": (_I, [_P, _F, _F]),
    "crt_render_frames": (_I, [_P, C.c_uint32, _P, _P]),
    "crt_render_frames_async": (_I, [_P, C.c_uint32, _P, _P]),
    "crt_sync": (_I, [_P]),
    "crt_set_option": (_I, [_P, C.c_char_p, _I]),
    "crt_reset": (_I, [_P]),
    "crt_read_sum": (_I, [_P, _P, _SZ]),
    "crt_resolve": (_I, [_P, _F, _P, _SZ]),
    "crt_sum_device": (_I, [_P, C.POINTER(_P)]),
    "crt_trace": (_I, [_P, _P, _SZ, _P, _I, _P]),
    "crt_trace_device": (_I, [_P, _P, _SZ, _P, _I, _P, _I]),
    "crt_resolve_device": (_I, [_P, C.c_float, C.POINTER(C.c_void_p), _I]),
    "crt_get_launch_times": (_I, [_P, _P, _SZ, C.POINTER(_SZ)]),
    "crt_debug_read_queue": (_I, [_P, _I, _U32, _P, _SZ, C.POINTER(_SZ)]),
    "crt_debug_time_graph": (_I, [_P, _U32, _P, _U32, C.POINTER(_F), C.POINTER(_F)]),
    "crt_debug_launch_form": (_I, [_P, C.POINTER(C.c_int32)]),
    "crt_debug_launch_info": (_I, [_P, C.POINTER(C.c_int32)]),
    "crt_debug_step_hist": (_I, [_P, _P]),
    "crt_set_shard": (_I, [_P, _U32, _U32, _U32]),
    "crt_set_devices": (_I, [_P, C.POINTER(C.c_int32), _U32, _U32]),
    "crt_shard_tiles": (_I, [_U32] * 7 + [_P, _SZ, C.POINTER(_SZ)]),
    "crt_get_devices": (_I, [_P, C.POINTER(_U32), C.POINTER(C.c_int32), _U32, C.POINTER(C.c_int32), C.POINTER(_F)]),
    "crt_packed_info": (_I, [_P, C.POINTER(_U32), C.POINTER(_U32), C.POINTER(_SZ)]),
    "crt_read_packed": (_I, [_P, _P, _SZ]),
    "crt_copy_packed_device": (_I, [_P, _P, _SZ, _I]),
    "crt_get_frame_stats": (_I, [_P, C.POINTER(crt_frame_stats)]),
    "crt_get_bvh_info": (_I, [_P, C.POINTER(crt_bvh_info)]),
    "crt_device_count": (_I, []),
    "crt_has_experiments": (_I, []),
    "crt_warmup": (_I, []),
    "crt_camera_look_at": (_I, [C.POINTER(_F), C.POINTER(_F), _F, C.POINTER(crt_camera)]),
    "crt_pcg_hash": (_U32, [_U32]),
    "crt_randf2": (_F, [C.POINTER(_U32)]),
    "crt_sbvh_build": (_I, [_P, _SZ, _P, _SZ, _U32, C.POINTER(_P)]),
    "crt_sbvh_num_nodes": (_SZ, [_P]),
    "crt_sbvh_num_slots": (_SZ, [_P]),
    "crt_sbvh_nodes": (_P, [_P]),
    "crt_sbvh_triangle_indices": (_P, [_P]),
    "crt_sbvh_triangles": (_P, [_P]),
    "crt_sbvh_free": (None, [_P]),
    "crt_lbvh_build": (_I, [_P, _SZ, _P, _SZ, _U32, C.POINTER(_P)]),
    "crt_lbvh_last_build_ms": (None, [C.POINTER(_F), C.POINTER(_F)]),
    "crt_cwbvh_convert": (_I, [_P, _SZ, _SZ, C.POINTER(_P)]),
    "crt_cwbvh_convert_device": (_I, [_P, _SZ, _SZ, C.POINTER(_P)]),
    "crt_cwbvh_last_convert_ms": (None, [C.POINTER(_F), C.POINTER(_F)]),
    "crt_cwbvh_num_nodes": (_SZ, [_P]),
    "crt_cwbvh_num_tris": (_SZ, [_P]),
    "crt_cwbvh_nodes": (_P, [_P]),
    "crt_cwbvh_tri_slots": (_P, [_P]),
    "crt_cwbvh_child_bvh2": (_P, [_P]),
    "crt_cwbvh_depth": (_U32, [_P]),
    "crt_cwbvh_free": (None, [_P]),
    "crt_load_obj": (_I, [C.c_char_p, C.POINTER(_F), C.POINTER(_P)]),
    "crt_mesh_counts": (_SZ, [_P] + [C.POINTER(_SZ)] * 6),
    "crt_mesh_vertices": (_P, [_P]),
    "crt_mesh_normals": (_P, [_P]),
    "crt_mesh_texcoords": (_P, [_P]),
    "crt_mesh_triangles": (_P, [_P]),
    "crt_mesh_materials": (_P, [_P]),
    "crt_mesh_lights": (_P, [_P]),
    "crt_mesh_vertex_min": (_P, [_P]),
    "crt_mesh_albedo_textures": (_P, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "crt_mesh_free": (None, [_P]),
    "crt_image_decode": (_I, [_P, _SZ, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P, _SZ]),
    "crt_image_encode_png": (_I, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _SZ, C.POINTER(C.c_size_t)]),
    "crt_texture_to_array_bytes": (_I, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "crt_last_error": (C.c_char_p, []),
    "crt_abi_version": (_U32, []),
}

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (same SONAME as
    /opt/rocm's); if both copies get loaded, the second one initialised sees no devices.  When torch
    is installed, load ITS runtime first (RTLD_GLOBAL) so libcrt.so's NEEDED libamdhip64.so.7 binds to
    it, whichever of torch / this package is imported first.  Without torch, /opt/rocm's is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        libdir = os.path.join(os.path.dirname(spec.origin), "lib")
        for name in ("libamdhip64.so",):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load libcrt.so once; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run __graft_entry__.build() or `make -C caitlynrenderer_amd/csrc`); there is no fallback path")
        _share_hip_runtime_with_torch()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)   # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if l.crt_abi_version() != CRT_ABI_VERSION:
            raise ImportError("libcrt.so ABI version mismatch; rebuild")
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise CrtError(rc, lib().crt_last_error().decode("utf-8", "replace"))
