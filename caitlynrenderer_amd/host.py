"""Host-side mirror of the reference's scene-preparation API, driven through the C ABI.

Same names and meaning as the reference's C++ classes so tests read like its own code would:
Camera (Caitlyn/Camera.h:4-66), Rnd (Caitlyn/Rnd.h), SBVH (Caitlyn/sbvh.h:81-153),
CWBVH (Caitlyn/cwbvh.h:51-73), Mesh.read_object (Caitlyn/Scene.h:742-926).  All of it is
[host] code inside libcrt.so; nothing here needs a GPU.

Array conventions (numpy, C-contiguous): vertices/normals (n,3) f32, texcoords (n,2) f32,
triangles (n,12) i32 = (v[4], vn[4], vt[4]), materials (n,16) f32, lights (n,18) f32,
BVH2 nodes (n,8) f32, CWBVH nodes (n,80) u8.
"""
import ctypes as C

import numpy as np

from ._lib import check, crt_camera, lib


def _copy(ptr, ctype, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (n,)).copy().view(dtype).reshape(shape)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Rnd:
    """thread_local s_RndState = 1; randf2() (Rnd.h:7, :36-40)."""

    def __init__(self, state=1):
        self.state = C.c_uint32(state)

    def randf2(self):
        return float(lib().crt_randf2(C.byref(self.state)))


def pcg_hash(x):
    return int(lib().crt_pcg_hash(C.c_uint32(x & 0xFFFFFFFF)))


class Camera:
    """Camera(pos, lookAt, fovDegrees) (Camera.h:7-19)."""

    def __init__(self, pos, look_at, fov_deg):
        self.c = crt_camera()
        p = (C.c_float * 3)(*map(float, pos))
        l = (C.c_float * 3)(*map(float, look_at))
        check(lib().crt_camera_look_at(p, l, float(fov_deg), C.byref(self.c)))

    position = property(lambda s: np.array(s.c.position[:], dtype=np.float32))
    right = property(lambda s: np.array(s.c.right[:], dtype=np.float32))
    up = property(lambda s: np.array(s.c.up[:], dtype=np.float32))
    forward = property(lambda s: np.array(s.c.forward[:], dtype=np.float32))
    fov = property(lambda s: float(s.c.fov))

    def translate(self, t):
        """Scene.h:924: camera.position += transformation_vector (fp32 add)."""
        p = (np.array(self.c.position[:], dtype=np.float32) + np.asarray(t, dtype=np.float32)).astype(np.float32)
        for k in range(3):
            self.c.position[k] = float(p[k])


class Mesh:
    """Result of Scene::Read_Object (Scene.h:742-926), vertices already translated by -vertex_min."""

    def __init__(self, vertices, normals, texcoords, triangles, materials, lights, vertex_min=None):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.texcoords = np.ascontiguousarray(texcoords, dtype=np.float32).reshape(-1, 2)
        self.triangles = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 12)
        self.materials = np.ascontiguousarray(materials, dtype=np.float32).reshape(-1, 16)
        self.lights = np.ascontiguousarray(lights, dtype=np.float32).reshape(-1, 18)
        self.vertex_min = np.zeros(3, np.float32) if vertex_min is None else np.asarray(vertex_min, dtype=np.float32)
        self.albedo_textures = None   # optional (layers, H, W, 3) uint8 array; materials[:, 12] = layer index or -1

    @staticmethod
    def read_object(path, camera=None):
        """Read_Object(file_name); translates `camera` like Scene.h:924 when given."""
        L = lib()
        h = C.c_void_p()
        cam = None
        if camera is not None:
            cam = (C.c_float * 3)(*camera.c.position[:])
        check(L.crt_load_obj(str(path).encode(), cam, C.byref(h)))
        try:
            n = [C.c_size_t() for _ in range(6)]
            L.crt_mesh_counts(h, *[C.byref(x) for x in n])
            nv, nn, nt, ntri, nm, nl = [x.value for x in n]
            m = Mesh(
                _copy(L.crt_mesh_vertices(h), C.c_float, (nv, 3), np.float32),
                _copy(L.crt_mesh_normals(h), C.c_float, (nn, 3), np.float32),
                _copy(L.crt_mesh_texcoords(h), C.c_float, (nt, 2), np.float32),
                _copy(L.crt_mesh_triangles(h), C.c_int32, (ntri, 12), np.int32),
                _copy(L.crt_mesh_materials(h), C.c_float, (nm, 16), np.float32),
                _copy(L.crt_mesh_lights(h), C.c_float, (nl, 18), np.float32),
                _copy(L.crt_mesh_vertex_min(h), C.c_float, (3,), np.float32),
            )
            tw, th, tl = C.c_int32(), C.c_int32(), C.c_int32()
            tex = L.crt_mesh_albedo_textures(h, C.byref(tw), C.byref(th), C.byref(tl))
            if tex and tl.value > 0:                      # map_Kd textures as the reference's RGB8 array (Scene.h:597-710)
                m.albedo_textures = _copy(tex, C.c_uint8, (tl.value, th.value, tw.value, 3), np.uint8)
        finally:
            L.crt_mesh_free(h)
        if camera is not None:
            for k in range(3):
                camera.c.position[k] = cam[k]
        return m


def decode_image(file_bytes):
    """crt_image_decode: (H, W, 3) uint8, top row first — what stbi_load(name, &w, &h, 0, 3) gives the reference."""
    L = lib()
    buf = np.frombuffer(bytes(file_bytes), np.uint8)
    w, h = C.c_int32(), C.c_int32()
    check(L.crt_image_decode(_ptr(buf), buf.size, C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 3), np.uint8)
    check(L.crt_image_decode(_ptr(buf), buf.size, C.byref(w), C.byref(h), _ptr(out), out.size))
    return out


def encode_png(pixels, bottom_up=False):
    """crt_image_encode_png: PNG file bytes of an (H, W, 3|4) uint8 image."""
    px = np.ascontiguousarray(pixels, dtype=np.uint8)
    assert px.ndim == 3 and px.shape[2] in (3, 4)
    L = lib()
    size = C.c_size_t()
    check(L.crt_image_encode_png(_ptr(px), px.shape[1], px.shape[0], px.shape[2], int(bottom_up), None, 0, C.byref(size)))
    out = np.empty(size.value, np.uint8)
    check(L.crt_image_encode_png(_ptr(px), px.shape[1], px.shape[0], px.shape[2], int(bottom_up), _ptr(out), out.size, C.byref(size)))
    return out[:size.value].tobytes()


def texture_to_array_bytes(rgb, out_w=256, out_h=256):
    """crt_texture_to_array_bytes: the reference's resize + byte truncation (Scene.h:321-371, :648-662, :688-710)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    assert rgb.ndim == 3 and rgb.shape[2] == 3
    out = np.empty((out_h, out_w, 3), np.uint8)
    check(lib().crt_texture_to_array_bytes(_ptr(rgb), rgb.shape[1], rgb.shape[0], out_w, out_h, _ptr(out)))
    return out


class SBVH:
    """SBVH(trs, vertices) (sbvh.h:99): flat_nodes, triangle_indices and the re-ordered triangles."""
    NO_SPATIAL_SPLITS = 1

    def __init__(self, triangles, vertices, flags=0, builder="sbvh"):
        """builder="sbvh": the reference's split-BVH on the host; "lbvh": GPU linear BVH (crt_lbvh_build); "ploc" / "ploc<radius>":
        GPU parallel locally-ordered clustering (crt_lbvh_build with CRT_GPU_BUILD_PLOC)."""
        L = lib()
        tris = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 12)
        verts = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        h = C.c_void_p()
        gpu = builder == "lbvh" or builder.startswith(("ploc", "sah"))
        if builder.startswith("sah"):                   # "sah" or "sah<small-node threshold>": crt_lbvh_build with CRT_GPU_BUILD_SAH
            flags = 4 | ((int(builder[3:]) if len(builder) > 3 else 0) << 8)
        if builder.startswith("ploc"):                  # "ploc" or "ploc<radius>": crt_lbvh_build with CRT_GPU_BUILD_PLOC | radius << 8
            radius = int(builder[4:]) if len(builder) > 4 else 0
            flags = 2 | (radius << 8)
        build = L.crt_lbvh_build if gpu else L.crt_sbvh_build
        check(build(_ptr(tris), tris.shape[0], _ptr(verts), verts.shape[0], int(flags), C.byref(h)))
        self.build_ms = None
        if gpu:
            dev, tot = C.c_float(), C.c_float()
            L.crt_lbvh_last_build_ms(C.byref(dev), C.byref(tot))
            self.build_ms = (dev.value, tot.value)
        try:
            nn, ns = L.crt_sbvh_num_nodes(h), L.crt_sbvh_num_slots(h)
            self.flat_nodes = _copy(L.crt_sbvh_nodes(h), C.c_float, (nn, 8), np.float32)
            self.triangle_indices = _copy(L.crt_sbvh_triangle_indices(h), C.c_int32, (ns,), np.int32)
            self.triangles = _copy(L.crt_sbvh_triangles(h), C.c_int32, (ns, 12), np.int32)
        finally:
            L.crt_sbvh_free(h)

    def count_leaf(self):
        return int((self.flat_nodes[:, 7] != 0).sum())

    def depth(self):
        """Deepest leaf level, root = 0 (children follow parents in BFS order)."""
        n = self.flat_nodes.shape[0]
        level = np.zeros(n, np.int32)
        inner = np.nonzero(self.flat_nodes[:, 7] == 0)[0]
        left = self.flat_nodes[inner, 3].astype(np.int64)
        for i, l in zip(inner, left):
            level[l] = level[l + 1] = level[i] + 1
        return int(level.max()) if n else 0


class CWBVH:
    """CWBVH().convert(bvh) (cwbvh.h:58): nodes (n,80) u8, tri_slots (CWBVH order -> BVH2 leaf slot)."""

    def __init__(self):
        self.nodes = np.zeros((0, 80), np.uint8)
        self.tri_slots = np.zeros((0,), np.int32)
        self.depth = 0

    def convert(self, bvh, device=False):
        """device=True runs the conversion on the GPU (crt_cwbvh_convert_device): same bytes, ~100x faster."""
        flat = bvh.flat_nodes if isinstance(bvh, SBVH) else np.ascontiguousarray(bvh, dtype=np.float32).reshape(-1, 8)
        n_slots = int(bvh.triangle_indices.shape[0]) if isinstance(bvh, SBVH) else int(
            (flat[flat[:, 7] != 0, 3] + flat[flat[:, 7] != 0, 7]).max())
        return self.convert_arrays(flat, n_slots, device)

    def convert_arrays(self, flat_nodes, n_slots, device=False):
        L = lib()
        flat = np.ascontiguousarray(flat_nodes, dtype=np.float32).reshape(-1, 8)
        h = C.c_void_p()
        self.convert_ms = None
        if device:
            check(L.crt_cwbvh_convert_device(_ptr(flat), flat.shape[0], int(n_slots), C.byref(h)))
            dev, tot = C.c_float(), C.c_float()
            L.crt_cwbvh_last_convert_ms(C.byref(dev), C.byref(tot))
            self.convert_ms = (dev.value, tot.value)
        else:
            check(L.crt_cwbvh_convert(_ptr(flat), flat.shape[0], int(n_slots), C.byref(h)))
        try:
            nn, nt = L.crt_cwbvh_num_nodes(h), L.crt_cwbvh_num_tris(h)
            self.nodes = _copy(L.crt_cwbvh_nodes(h), C.c_uint8, (nn, 80), np.uint8)
            self.tri_slots = _copy(L.crt_cwbvh_tri_slots(h), C.c_int32, (nt,), np.int32)
            self.child_bvh2 = _copy(L.crt_cwbvh_child_bvh2(h), C.c_int32, (nn, 8), np.int32)
            self.depth = int(L.crt_cwbvh_depth(h))
        finally:
            L.crt_cwbvh_free(h)
        return self
