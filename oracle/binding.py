"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
product package.  Takes the same numpy arrays the product uploads (SceneData) and runs the CPU
restatement of the reference's algorithm on them.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

CLOSEST, ANY = 0, 1
TIE_FIRST_VISITED, TIE_LOWEST_ID = 0, 1
BRUTE, BVH2, BVH8 = 0, 1, 2

RAY_DT = np.dtype([("o", "<f4", 3), ("tmax", "<f4"), ("d", "<f4", 3), ("pad", "<u4")])
HIT_DT = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("tri", "<i4")])
STATS_DT = np.dtype([("nodes", "<u2"), ("tris", "<u2")])


class orc_camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("forward", C.c_float * 3), ("fov", C.c_float), ("focal_dist", C.c_float), ("aperture", C.c_float)]


class orc_scene(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("normals", C.c_void_p), ("texcoords", C.c_void_p),
                ("triangles", C.c_void_p), ("tri_orig_ids", C.c_void_p), ("materials", C.c_void_p),
                ("lights", C.c_void_p), ("bvh2", C.c_void_p), ("bvh8", C.c_void_p), ("bvh8_tri_slots", C.c_void_p),
                ("n_triangles", C.c_int32), ("n_lights", C.c_int32), ("n_bvh2", C.c_int32), ("n_bvh8", C.c_int32),
                ("n_bvh8_tris", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("max_depth", C.c_int32),
                ("camera", orc_camera),
                ("albedo_textures", C.c_void_p), ("tex_width", C.c_int32), ("tex_height", C.c_int32), ("n_textures", C.c_int32)]


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        l.orc_sin.restype = C.c_float; l.orc_sin.argtypes = [C.c_float]
        l.orc_cos.restype = C.c_float; l.orc_cos.argtypes = [C.c_float]
        l.orc_rand.restype = C.c_float; l.orc_rand.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float]
        l.orc_pcg_hash.restype = C.c_uint32; l.orc_pcg_hash.argtypes = [C.c_uint32]
        l.orc_randf2.restype = C.c_float; l.orc_randf2.argtypes = [C.POINTER(C.c_uint32)]
        l.orc_trace.restype = None
        l.orc_trace.argtypes = [C.POINTER(orc_scene), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        l.orc_primary_rays.restype = None
        l.orc_primary_rays.argtypes = [C.POINTER(orc_scene), C.c_float, C.c_float, C.c_int, C.c_void_p]
        l.orc_render_frame.restype = None
        l.orc_render_frame.argtypes = [C.POINTER(orc_scene), C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        l.orc_render_rows.restype = None
        l.orc_render_rows.argtypes = [C.POINTER(orc_scene), C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_int]
        l.orc_resolve.restype = None
        l.orc_resolve.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]
        l.orc_hardware_threads.restype = C.c_int
        fp = C.POINTER(C.c_float)
        l.orc_disney_eval_test.restype = None
        l.orc_disney_eval_test.argtypes = [fp, fp, fp, fp, fp, fp]
        l.orc_disney_sample_test.restype = None
        l.orc_disney_sample_test.argtypes = [fp, fp, fp, fp, fp]
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Oracle:
    """CPU oracle over one SceneData (duck-typed: needs the attribute names of caitlynrenderer_amd.SceneData)."""

    def __init__(self, data, width, height, max_depth=3, camera=None):
        self._keep = []

        def arr(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt)
            self._keep.append(a)
            return a

        s = orc_scene()
        self.vertices = arr(data.vertices, np.float32)
        self.triangles = arr(data.triangles, np.int32)
        s.vertices = _p(self.vertices); s.normals = _p(arr(data.normals, np.float32))
        s.texcoords = _p(arr(data.texcoords, np.float32))
        s.triangles = _p(self.triangles); s.tri_orig_ids = _p(arr(data.tri_orig_ids, np.int32))
        s.materials = _p(arr(data.materials, np.float32))
        lights = arr(data.lights, np.float32)
        s.lights = _p(lights)
        bvh2 = arr(data.bvh, np.float32); s.bvh2 = _p(bvh2)
        bvh8 = arr(data.bvh8, np.uint8); s.bvh8 = _p(bvh8)
        slots = arr(data.bvh8_tri_slots, np.int32); s.bvh8_tri_slots = _p(slots)
        s.n_triangles = self.triangles.shape[0]
        s.n_lights = 0 if lights is None else lights.reshape(-1, 18).shape[0]
        s.n_bvh2 = 0 if bvh2 is None else bvh2.reshape(-1, 8).shape[0]
        s.n_bvh8 = 0 if bvh8 is None else bvh8.reshape(-1, 80).shape[0]
        s.n_bvh8_tris = 0 if slots is None else slots.shape[0]
        s.width, s.height, s.max_depth = int(width), int(height), int(max_depth)
        tex = getattr(data, "albedo_textures", None)
        if tex is not None:
            tex = arr(tex, np.uint8)
            assert tex.ndim == 4 and tex.shape[3] == 3, "albedo_textures must be (layers, H, W, 3) uint8"
            s.albedo_textures = _p(tex)
            s.n_textures, s.tex_height, s.tex_width = int(tex.shape[0]), int(tex.shape[1]), int(tex.shape[2])
        self.s = s
        self.set_camera(camera if camera is not None else data.camera)

    def set_camera(self, camera):
        if camera is None:
            return
        c = camera.c if hasattr(camera, "c") else camera
        for name in ("position", "right", "up", "forward"):
            for k in range(3):
                getattr(self.s.camera, name)[k] = getattr(c, name)[k]
        self.s.camera.fov = c.fov

    def trace(self, rays, accel=BVH8, mode=CLOSEST, tie=TIE_LOWEST_ID, stats=False, threads=1):
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        hits = np.empty(rays.shape[0], HIT_DT)
        st = np.zeros(rays.shape[0], STATS_DT) if stats else None
        lib().orc_trace(C.byref(self.s), accel, mode, tie, _p(rays), rays.shape[0], _p(hits), _p(st) if stats else None, threads)
        return (hits, st) if stats else hits

    def primary_rays(self, rx=0.0, ry=0.0, jitter=False):
        rays = np.empty(self.s.width * self.s.height, RAY_DT)
        lib().orc_primary_rays(C.byref(self.s), float(np.float32(rx)), float(np.float32(ry)), 1 if jitter else 0, _p(rays))
        return rays

    def render_frame(self, rx, ry, sum_buf=None, accel=BVH8, tie=TIE_LOWEST_ID, threads=1):
        """Adds one sample per pixel into sum_buf (H, W, 3) and returns (sum_buf, counters[4])."""
        if sum_buf is None:
            sum_buf = np.zeros((self.s.height, self.s.width, 3), np.float32)
        cnt = (C.c_uint64 * 4)()
        lib().orc_render_frame(C.byref(self.s), accel, tie, float(np.float32(rx)), float(np.float32(ry)), _p(sum_buf), cnt, threads)
        return sum_buf, [int(x) for x in cnt]

    def render_rows(self, rx, ry, y0, y1, sum_buf, accel=BVH8, tie=TIE_LOWEST_ID):
        cnt = (C.c_uint64 * 4)()
        lib().orc_render_rows(C.byref(self.s), accel, tie, float(np.float32(rx)), float(np.float32(ry)), _p(sum_buf), cnt, int(y0), int(y1))
        return [int(x) for x in cnt]


def resolve(sum_buf, inv_count):
    sum_buf = np.ascontiguousarray(sum_buf, dtype=np.float32)
    n = sum_buf.size // 3
    out = np.empty((n, 4), np.uint8)
    lib().orc_resolve(_p(sum_buf), n, float(np.float32(inv_count)), _p(out))
    return out.reshape(sum_buf.shape[:-1] + (4,))


def rand_sequence(px, py, rx, ry, n):
    seed = (C.c_float * 2)(px + 0.5, py + 0.5)
    return [float(lib().orc_rand(seed, float(np.float32(rx)), float(np.float32(ry)))) for _ in range(n)]


def _f(*xs):
    return (C.c_float * len(xs))(*[float(np.float32(x)) for x in xs])


def disney_eval(base, metallic, roughness, n, wo, wi):
    """(f rgb, pdf) of the oracle-defined Disney lobe for one direction pair."""
    f, pdf = _f(0, 0, 0), C.c_float()
    lib().orc_disney_eval_test(_f(*base, metallic, roughness), _f(*n), _f(*wo), _f(*wi), f, C.byref(pdf))
    return np.array(f[:], np.float32), float(pdf.value)


def disney_sample(base, metallic, roughness, n, wo, u):
    wi = _f(0, 0, 0)
    lib().orc_disney_sample_test(_f(*base, metallic, roughness), _f(*n), _f(*wo), _f(*u), wi)
    return np.array(wi[:], np.float32)


def hardware_threads():
    """Host threads this process may actually use: the affinity mask and the cgroup CPU quota, not
    the machine's core count (a one-GPU box exposes 256 logical CPUs but grants a share of them)."""
    n = int(lib().orc_hardware_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // period))
        except Exception:
            pass
    return max(1, min(n, 64))
