"""Device-side scene: the drop-in for Scene::gpu_data / update / Render (Caitlyn/Scene.h:1000-1246).

`SceneData` is the bundle of host arrays the reference uploads; `Scene` owns the device copy and
dispatches the HIP path through the C ABI.  There is no CPU fallback: constructing a `Scene`
without a GPU raises CrtError(CRT_ERR_NO_DEVICE).
"""
import ctypes as C

import numpy as np

from ._lib import (CRT_ABI_VERSION, CRT_BUILD_LBVH_ON_DEVICE, CRT_TRACE_ANY, CRT_TRACE_CLOSEST, check, crt_bvh_info, crt_frame_stats,
                   crt_scene_desc, lib)
from .host import CWBVH, SBVH, Camera, Mesh, Rnd, _ptr

RAY_DT = np.dtype([("o", "<f4", 3), ("tmax", "<f4"), ("d", "<f4", 3), ("pad", "<u4")])
HIT_DT = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("tri", "<i4")])
STATS_DT = np.dtype([("nodes", "<u2"), ("tris", "<u2")])


class SceneData:
    """Host arrays in upload order (Scene.h:1015-1062) + the CWBVH the shader was meant to get."""

    def __init__(self, mesh, sbvh, cwbvh, camera):
        self.vertices, self.normals, self.texcoords = mesh.vertices, mesh.normals, mesh.texcoords
        self.materials, self.lights = mesh.materials, mesh.lights
        self.triangles = sbvh.triangles                 # leaf order with duplicates (sbvh.h:130-139)
        self.tri_orig_ids = sbvh.triangle_indices
        self.bvh = sbvh.flat_nodes
        self.bvh8 = cwbvh.nodes if cwbvh is not None else None
        self.bvh8_tri_slots = cwbvh.tri_slots if cwbvh is not None else None
        self.camera = camera
        self.albedo_textures = getattr(mesh, "albedo_textures", None)   # (layers, H, W, 3) uint8 or None (Scene.h:1065-1078)
        self.n_source_triangles = int(mesh.triangles.shape[0])
        self.build_flags = 0

    @staticmethod
    def build(mesh, camera, sbvh_flags=0, with_cwbvh=True, builder="sbvh", convert="host"):
        """Scene::build_bvh (Scene.h:929-959) followed by the intended CWBVH::convert.
        builder="lbvh" swaps the host SBVH for the GPU linear BVH (crt_lbvh_build); convert="device" runs
        the CWBVH conversion on the GPU (crt_cwbvh_convert_device, same bytes as the host converter)."""
        sbvh = SBVH(mesh.triangles, mesh.vertices, sbvh_flags, builder=builder)
        cw = CWBVH().convert(sbvh, device=(convert == "device")) if with_cwbvh else None
        return SceneData(mesh, sbvh, cw, camera)

    @staticmethod
    def for_device_build(mesh, camera, builder="lbvh"):
        """The seven input arrays only (triangles in source order): Scene() then builds the BVH2 (builder "lbvh", or "ploc" /
        "ploc<radius>"), the CWBVH and the intersection records in HBM (crt_scene_desc.build_flags =
        CRT_BUILD_LBVH_ON_DEVICE [| CRT_BUILD_PLOC | radius << 8]) — nothing but these arrays crosses PCIe."""
        class _NoBvh:
            triangles, triangle_indices, flat_nodes = mesh.triangles, None, None
        data = SceneData(mesh, _NoBvh, None, camera)
        data.build_flags = CRT_BUILD_LBVH_ON_DEVICE
        if builder.startswith("ploc"):
            data.build_flags |= 2 | ((int(builder[4:]) if len(builder) > 4 else 0) << 8)
        elif builder.startswith("sah"):
            data.build_flags |= 4 | ((int(builder[3:]) if len(builder) > 3 else 0) << 8)
        return data

    @staticmethod
    def from_obj(path, camera):
        """Scene(file_name, …) up to gpu_data (Scene.h:447-495): load, translate, build."""
        mesh = Mesh.read_object(path, camera)
        return SceneData.build(mesh, camera)


class Scene:
    def __init__(self, data, width, height, max_depth=3):
        self._h = C.c_void_p()
        self.width, self.height, self.max_depth = int(width), int(height), int(max_depth)
        self.frame_count = 0                       # Scene.h:384
        self.rnd = Rnd()                           # Rnd.h:7
        d = crt_scene_desc()
        d.abi_version = CRT_ABI_VERSION
        keep = []

        def arr(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            keep.append(a)
            return a

        v = arr(data.vertices, np.float32); d.vertices, d.n_vertices = _ptr(v), v.shape[0]
        n = arr(data.normals, np.float32); d.normals, d.n_normals = _ptr(n), n.shape[0]
        t = arr(data.texcoords, np.float32); d.texcoords, d.n_texcoords = _ptr(t), t.shape[0]
        tr = arr(data.triangles, np.int32); d.triangles, d.n_triangles = _ptr(tr), tr.shape[0]
        if data.tri_orig_ids is not None:
            ids = arr(data.tri_orig_ids, np.int32); d.tri_orig_ids = _ptr(ids)
        m = arr(data.materials, np.float32); d.materials, d.n_materials = _ptr(m), m.shape[0]
        l = arr(data.lights, np.float32); d.lights, d.n_lights = _ptr(l), l.shape[0]
        if data.bvh is not None:
            b = arr(data.bvh, np.float32); d.bvh, d.n_bvh = _ptr(b), b.shape[0]
        if data.bvh8 is not None:
            b8 = arr(data.bvh8, np.uint8); d.bvh8, d.n_bvh8 = _ptr(b8), b8.shape[0]
            sl = arr(data.bvh8_tri_slots, np.int32); d.bvh8_tri_slots, d.n_bvh8_tris = _ptr(sl), sl.shape[0]
        tex = getattr(data, "albedo_textures", None)
        if tex is not None:
            tex = arr(tex, np.uint8)
            assert tex.ndim == 4 and tex.shape[3] == 3, "albedo_textures must be (layers, H, W, 3) uint8"
            d.albedo_textures = _ptr(tex)
            d.n_textures, d.tex_height, d.tex_width = tex.shape[0], tex.shape[1], tex.shape[2]
        d.width, d.height, d.max_depth = self.width, self.height, self.max_depth
        d.build_flags = int(getattr(data, "build_flags", 0))
        import time
        t0 = time.perf_counter()
        check(lib().crt_scene_create(C.byref(d), C.byref(self._h)))
        self.create_ms = (time.perf_counter() - t0) * 1e3        # wall time of crt_scene_create itself (the arrays were marshalled before)
        if data.camera is not None:
            self.update(data.camera)

    # -- reference-shaped API ------------------------------------------------------------
    def update(self, camera):
        """Scene::update (Scene.h:1233-1246)."""
        check(lib().crt_set_camera(self._h, C.byref(camera.c)))

    def Render(self):
        """Scene::Render (Scene.h:1158-1231) minus the present pass: draw one sample, ++frame_count."""
        r1, r2 = self.rnd.randf2(), self.rnd.randf2()          # Scene.h:1208
        self.render_frame(r1, r2)
        self.frame_count += 1
        return r1, r2

    def reset(self):
        """camera.isMoving branch (Scene.h:1160-1172)."""
        check(lib().crt_reset(self._h))
        self.frame_count = 0

    # -- explicit API ----------------------------------------------------------------------
    def render_frame(self, rx, ry, sync=True):
        fn = lib().crt_render_frame if sync else lib().crt_render_frame_async
        check(fn(self._h, float(np.float32(rx)), float(np.float32(ry))))

    def render_frames(self, rvs, sync=True):
        """crt_render_frames: rvs = sequence of (rx, ry); the same sums as render_frame per pair, fewer launches where
        the path allows (shadow rays in place: up to 8 samples per launch; on a shard or a small frame the samples of a
        launch run side by side on the waves of a workgroup, option "wave_samples")."""
        rx = np.ascontiguousarray([r[0] for r in rvs], dtype=np.float32)
        ry = np.ascontiguousarray([r[1] for r in rvs], dtype=np.float32)
        fn = lib().crt_render_frames if sync else lib().crt_render_frames_async
        check(fn(self._h, int(rx.size), _ptr(rx), _ptr(ry)))

    def sync(self):
        check(lib().crt_sync(self._h))

    def set_option(self, name, value):
        check(lib().crt_set_option(self._h, name.encode(), int(value)))

    def read_sum(self):
        out = np.empty((self.height, self.width, 3), np.float32)
        check(lib().crt_read_sum(self._h, _ptr(out), out.size))
        return out

    def resolve(self, inv_count=None):
        if inv_count is None:
            inv_count = 1.0 / max(self.frame_count, 1)
        out = np.empty((self.height, self.width, 4), np.uint8)
        check(lib().crt_resolve(self._h, float(np.float32(inv_count)), _ptr(out), out.size))
        return out

    def resolve_device(self, inv_count=None, sync=True):
        """crt_resolve_device: the tone-mapped RGBA8 frame left in device memory (what the reference's output pass leaves in the default
        framebuffer); returns the device pointer"""
        if inv_count is None:
            inv_count = 1.0 / max(self.frame_count, 1)
        p = C.c_void_p()
        check(lib().crt_resolve_device(self._h, float(np.float32(inv_count)), C.byref(p), 1 if sync else 0))
        return p.value

    def launch_times(self):
        """crt_get_launch_times: ms of every event-carrying launch since the spans were restarted (options timing / timing_accumulate)"""
        n = C.c_size_t()
        check(lib().crt_get_launch_times(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.float32)
        if n.value:
            check(lib().crt_get_launch_times(self._h, _ptr(out), n.value, C.byref(n)))
        return out

    def trace(self, rays, mode=CRT_TRACE_CLOSEST, stats=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        hits = np.empty(rays.shape[0], HIT_DT)
        st = np.zeros(rays.shape[0], STATS_DT) if stats else None
        check(lib().crt_trace(self._h, _ptr(rays), rays.shape[0], _ptr(hits), int(mode), _ptr(st) if stats else None))
        return (hits, st) if stats else hits

    def trace_device(self, d_rays, n, d_hits, mode=CRT_TRACE_CLOSEST, d_stats=None, sync=True):
        check(lib().crt_trace_device(self._h, C.c_void_p(d_rays), int(n), C.c_void_p(d_hits), int(mode),
                                     C.c_void_p(d_stats) if d_stats else None, 1 if sync else 0))

    def debug_read_queue(self, which, segment):
        """Rays entering `segment` (which=0) or its shadow rays (which=2) as left by the last frame."""
        n = C.c_size_t()
        check(lib().crt_debug_read_queue(self._h, int(which), int(segment), None, 0, C.byref(n)))
        out = np.empty(n.value, RAY_DT)
        check(lib().crt_debug_read_queue(self._h, int(which), int(segment), _ptr(out), n.value, C.byref(n)))
        return out

    def debug_launch_form(self):
        """0: the last launch ran its samples one after the other in each wave (or had one); 1: side by side on the waves
        of a workgroup (option "wave_samples")."""
        form = C.c_int32()
        check(lib().crt_debug_launch_form(self._h, C.byref(form)))
        return form.value

    def debug_launch_info(self):
        """crt_debug_launch_info of the last first-segment launch: {"form": 0 | 1 | 2 (2 = four samples of a 4x4 pixel quadrant in the
        lanes of a wave), "wide": the 6-waves-per-SIMD build ran, "one_pass": a build without the sample loop ran (k_segment<..., ONE>),
        "samples": samples per pixel of the launch, "shards": tile shards side by side (streams / devices)}"""
        info = (C.c_int32 * 4)()
        check(lib().crt_debug_launch_info(self._h, info))
        return {"form": info[0], "wide": bool(info[1] & 1), "one_pass": bool(info[1] & 2), "samples": info[2], "shards": info[3]}

    def debug_step_hist(self, stop=False):
        """crt_debug_step_hist: (closest[65], any[65]) node steps of the counting frames since the previous call by number of enabled lanes"""
        if stop:
            check(lib().crt_debug_step_hist(self._h, None))
            return None
        h = np.zeros(130, np.uint64)
        check(lib().crt_debug_step_hist(self._h, _ptr(h)))
        return h[:65].copy(), h[65:].copy()

    def set_shard(self, rank, world, tile=16):
        check(lib().crt_set_shard(self._h, int(rank), int(world), int(tile)))

    def sum_device(self):
        """crt_sum_device: device pointer (int) of the whole frame's running sum on the first device (gathered when there are several)."""
        ptr = C.c_void_p()
        check(lib().crt_sum_device(self._h, C.byref(ptr)))
        return ptr.value

    def set_devices(self, devices, tile=16):
        """crt_set_devices: this one handle renders on all of `devices` (HIP device ids, the scene's own first); read_sum / resolve
        gather the other devices' tiles to the first.  The same id more than once = virtual devices on one GPU (tests)."""
        ids = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        check(lib().crt_set_devices(self._h, ids, len(devices), int(tile)))

    def devices(self):
        """{"devices": [...], "transport": "rccl" | "copy", "last_gather_ms": ms of the last gather}"""
        n, tr, ms = C.c_uint32(), C.c_int32(), C.c_float()
        ids = (C.c_int32 * 64)()
        check(lib().crt_get_devices(self._h, C.byref(n), ids, 64, C.byref(tr), C.byref(ms)))
        return {"devices": [int(ids[k]) for k in range(n.value)], "transport": "rccl" if tr.value == 0 else "copy", "last_gather_ms": float(ms.value)}

    def packed_info(self):
        nt, tile, nf = C.c_uint32(), C.c_uint32(), C.c_size_t()
        check(lib().crt_packed_info(self._h, C.byref(nt), C.byref(tile), C.byref(nf)))
        return nt.value, tile.value, nf.value

    def read_packed(self):
        _, _, nf = self.packed_info()
        out = np.empty(nf, np.float32)
        check(lib().crt_read_packed(self._h, _ptr(out), nf))
        return out

    def copy_packed_device(self, d_dst, n_floats, sync=True):
        check(lib().crt_copy_packed_device(self._h, C.c_void_p(d_dst), int(n_floats), 1 if sync else 0))

    def frame_stats(self):
        st = crt_frame_stats()
        check(lib().crt_get_frame_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def bvh_info(self):
        st = crt_bvh_info()
        check(lib().crt_get_bvh_info(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def close(self):
        if self._h:
            lib().crt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


__all__ = ["Scene", "SceneData", "Camera", "Mesh", "SBVH", "CWBVH", "RAY_DT", "HIT_DT", "STATS_DT",
           "CRT_TRACE_CLOSEST", "CRT_TRACE_ANY", "CRT_BUILD_LBVH_ON_DEVICE"]
