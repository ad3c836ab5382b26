import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _build_once():
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session", autouse=True)
def built():
    _build_once()


@pytest.fixture(scope="session")
def cr(built):
    import caitlynrenderer_amd
    return caitlynrenderer_amd


@pytest.fixture(scope="session")
def ob(built):
    from oracle import binding
    return binding


@pytest.fixture(scope="session")
def survey():
    with open(os.path.join(GOLDEN, "survey_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def vectors():
    return dict(np.load(os.path.join(GOLDEN, "oracle_vectors.npz")))


class _Cam:
    """crt_camera rebuilt from the fixture (no /root/reference on the GPU box)."""

    def __init__(self, cr, d):
        from caitlynrenderer_amd._lib import crt_camera
        self.c = crt_camera()
        for k in ("position", "right", "up", "forward"):
            for i in range(3):
                getattr(self.c, k)[i] = d[k][i]
        self.c.fov = d["fov"]
        self.c.focal_dist, self.c.aperture = 0.1, 0.0
    position = property(lambda s: np.array(s.c.position[:], np.float32))
    fov = property(lambda s: float(s.c.fov))


@pytest.fixture(scope="session")
def cornell(cr):
    """(mesh, camera) of the Cornell box from the committed fixture."""
    with open(os.path.join(GOLDEN, "cornell_box.json")) as f:
        j = json.load(f)
    mesh = cr.Mesh(np.array(j["vertices"], np.float32), np.array(j["normals"], np.float32), np.zeros((0, 2), np.float32),
                   np.array(j["triangles"], np.int32), np.array(j["materials"], np.float32),
                   np.array(j["lights"], np.float32), np.array(j["vertex_min"], np.float32))
    return mesh, _Cam(cr, j["camera"])


@pytest.fixture(scope="session")
def cornell_data(cr, cornell):
    mesh, cam = cornell
    return cr.SceneData.build(mesh, cam)


@pytest.fixture(scope="session")
def tess8(cr, cornell):
    from caitlynrenderer_amd.meshgen import tessellated_cornell
    mesh, cam = cornell
    m = tessellated_cornell(mesh, 8)
    return m, cr.SceneData.build(m, cam)


@pytest.fixture(scope="session")
def tess40(cr, cornell):
    from caitlynrenderer_amd.meshgen import tessellated_cornell
    mesh, cam = cornell
    m = tessellated_cornell(mesh, 40)
    return m, cr.SceneData.build(m, cam)


def seeded_rays(mesh, n, seed, dt):
    rng = np.random.default_rng(seed)
    lo, hi = mesh.vertices.min(0), mesh.vertices.max(0)
    rays = np.zeros(n, dt)
    rays["o"] = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3))
    rays["d"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["tmax"] = np.float32(1e9)
    return rays


def numpy_brute_force(mesh, rays):
    """A third, independent implementation of the hit definition (no BVH, no C): Moeller-Trumbore over ALL triangles per ray in
    float32 numpy, same operation order as path_trace.fs:322-374 ((x*x + y*y) + z*z dots, no fused multiply-adds), nearest t,
    ties to the lowest triangle id.  Returns (tri, t, u, v); tri = -1 for a miss."""
    f = np.float32
    V = mesh.vertices.astype(np.float32)
    T = mesh.triangles[:, :3].astype(np.int64)
    v0 = V[T[:, 0]]
    e1 = (V[T[:, 1]] - v0).astype(f)
    e2 = (V[T[:, 2]] - v0).astype(f)

    def dot(a, b):
        return ((a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]).astype(f) + a[..., 2] * b[..., 2]).astype(f)

    def cross(a, b):
        return np.stack([(a[..., 1] * b[..., 2]).astype(f) - (a[..., 2] * b[..., 1]).astype(f),
                         (a[..., 2] * b[..., 0]).astype(f) - (a[..., 0] * b[..., 2]).astype(f),
                         (a[..., 0] * b[..., 1]).astype(f) - (a[..., 1] * b[..., 0]).astype(f)], -1).astype(f)
    n = rays.shape[0]
    tri = np.full(n, -1, np.int32)
    tt, uu, vv = np.zeros(n, f), np.zeros(n, f), np.zeros(n, f)
    with np.errstate(all="ignore"):
        for i in range(n):
            o, d = rays["o"][i].astype(f), rays["d"][i].astype(f)
            pv = cross(np.broadcast_to(d, e2.shape), e2)
            tv = (o - v0).astype(f)
            qv = cross(tv, e1)
            u, v, t = dot(tv, pv), dot(np.broadcast_to(d, qv.shape), qv), dot(e2, qv)
            inv = (f(1.0) / dot(e1, pv)).astype(f)
            u, v, t = (u * inv).astype(f), (v * inv).astype(f), (t * inv).astype(f)
            w = ((f(1.0) - u).astype(f) - v).astype(f)
            ok = (u >= 0) & (v >= 0) & (t >= 0) & (w >= 0) & (t < rays["tmax"][i])
            if ok.any():
                k = int(np.argmin(np.where(ok, t, np.inf)))              # first index of the minimum = lowest id among equal t
                tri[i], tt[i], uu[i], vv[i] = k, t[k], u[k], v[k]
    return tri, tt, uu, vv


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "Models"))


def write_obj(mesh, path, mtl_name="scene.mtl", map_kd=None):
    """Write a Mesh back as OBJ + MTL text (v//vn faces, or v/vt/vn when the mesh has texcoords; %.9g so every
    float32 survives the round trip); lets the C++ example and the loader tests run on the GPU box, where
    /root/reference does not exist.  map_kd: {material index: texture file name} -> `map_Kd` lines."""
    import os
    lines = [f"mtllib {mtl_name}"]
    for v in mesh.vertices + mesh.vertex_min:
        lines.append("v %.9g %.9g %.9g" % tuple(float(x) for x in v))
    for n in mesh.normals:
        lines.append("vn %.9g %.9g %.9g" % tuple(float(x) for x in n))
    with_vt = mesh.texcoords.shape[0] > 0
    for uv in mesh.texcoords:                         # the loader stores (u, 1 - v), Scene.h:801
        lines.append("vt %.9g %.9g" % (float(uv[0]), float(np.float32(1) - uv[1])))
    cur = None
    for t in mesh.triangles:
        if t[3] != cur:
            cur = int(t[3])
            lines.append(f"usemtl m{cur}")
        assert t[7] == 1, "write_obj expects vertex normals"
        if with_vt:
            lines.append("f " + " ".join(f"{int(t[k]) + 1}/{int(t[8 + k]) + 1}/{int(t[4 + k]) + 1}" for k in range(3)))
        else:
            lines.append("f " + " ".join(f"{int(t[k]) + 1}//{int(t[4 + k]) + 1}" for k in range(3)))
    open(path, "w").write("\n".join(lines) + "\n")
    m = []
    for i, mat in enumerate(mesh.materials):
        m += [f"newmtl m{i}", "Kd %.9g %.9g %.9g" % tuple(float(x) for x in mat[0:3])]
        e = mat[4:7] if mat[7] != -1 else (0.0, 0.0, 0.0)
        m.append("Ke %.9g %.9g %.9g" % tuple(float(x) for x in e))
        if map_kd and i in map_kd:
            m.append(f"map_Kd {map_kd[i]}")
    open(os.path.join(os.path.dirname(path), mtl_name), "w").write("\n".join(m) + "\n")


@pytest.fixture(scope="session")
def textured(cr, cornell):
    """Cornell box with uv coordinates on every quad and two materials switched to texture layers 0 / 1
    (16x8 and wrap-exercising uv > 1): the textured-albedo branch of path_trace.fs:471-483."""
    mesh, cam = cornell
    tris = mesh.triangles.copy()
    uv = np.array([[0.0, 0.0], [1.7, 0.0], [1.7, 2.3], [0.0, 2.3], [-0.4, 0.25], [0.9, 0.25], [0.9, 1.5], [-0.4, 1.5]], np.float32)
    for q in range(tris.shape[0] // 2):
        o = 4 * (q % 2)
        tris[2 * q, 8:12] = (o + 0, o + 1, o + 2, 0)
        tris[2 * q + 1, 8:12] = (o + 0, o + 2, o + 3, 0)
    mats = mesh.materials.copy()
    mats[3, 12] = 0.0     # Khaki (boxes, floor...) -> layer 0
    mats[2, 12] = 1.0     # HalveRed wall -> layer 1
    m = cr.Mesh(mesh.vertices, mesh.normals, uv, tris, mats, mesh.lights, mesh.vertex_min)
    rng = np.random.default_rng(42)
    m.albedo_textures = rng.integers(0, 256, size=(2, 8, 16, 3), dtype=np.uint8)
    return m, cr.SceneData.build(m, cam), cam
