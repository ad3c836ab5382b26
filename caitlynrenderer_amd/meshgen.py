"""Deterministic procedural meshes for the bench/test configs.

`tessellated_cornell(base, n)` takes the 16 Cornell quads (already translated by -vertex_min),
cuts each of the 15 non-light quads into n x n cells of two triangles (v00,v10,v11),(v00,v11,v01),
keeps the light quad as 2 triangles, displaces interior grid vertices along the quad normal by
0.02*(pcg_hash(vertex_index ^ 0x1234)/2^32 - 0.5), and gives every vertex the quad normal
(vn.w = 1).  n = 183 gives 1,004,672 triangles / 507,844 vertices; n = 8 gives 1,922 triangles.

SURVEY §8d leaves two things open that the displacement (a hash of the vertex index) and therefore the
SBVH's spatial splits depend on; both are fixed here to what reproduces the survey's own probe of the
reference builder on this mesh (appendix A / §8: n = 8 -> 1,945 leaf slots, 3,889 nodes, depth 15;
n = 183 -> 1,006,286 slots, 2,012,571 nodes, depth 25 — asserted in tests/test_host.py):
  * vertex numbering inside a quad: vertex (i, j) — i along P00->P10, j along P00->P01 — has index
    base + j*(n+1) + i (rows of constant j); quads in OBJ order, the light quad's 4 vertices in sequence;
  * grid positions by bilinear interpolation in float32.
(Round 1 numbered i-major and interpolated in float64: 1,928 / 3,855 and 1,006,291 / 2,012,581 — a different
mesh, not a different builder.)
"""
import numpy as np

from .host import Mesh


def pcg_hash_np(x):
    """Vectorised Caitlyn/Rnd.h:21-26 on uint32 arrays."""
    x = np.asarray(x, dtype=np.uint64)
    m = np.uint64(0xFFFFFFFF)
    state = (x * np.uint64(747796405) + np.uint64(2891336453)) & m
    word = (((state >> ((state >> np.uint64(28)) + np.uint64(4))) ^ state) * np.uint64(277803737)) & m
    return (((word >> np.uint64(22)) ^ word) & m).astype(np.uint32)


def tessellated_cornell(base, n):
    tris = base.triangles
    assert tris.shape[0] % 2 == 0, "base mesh must be fan-triangulated quads"
    verts64 = base.vertices.astype(np.float64)
    out_v, out_t = [], []
    n_vertices = 0
    for q in range(tris.shape[0] // 2):
        t0, t1 = tris[2 * q], tris[2 * q + 1]
        a, b, c, d = int(t0[0]), int(t0[1]), int(t0[2]), int(t1[2])   # fan (a,b,c),(a,c,d)
        mtl = int(t0[3])
        vn = t0[4:8].copy()
        emissive = base.materials[mtl, 7] != -1.0
        P00, P10, P11, P01 = verts64[a], verts64[b], verts64[c], verts64[d]
        if emissive:
            out_v.append(np.stack([P00, P10, P11, P01]).astype(np.float32))
            base_i = n_vertices
            for (i0, i1, i2) in ((0, 1, 2), (0, 2, 3)):
                out_t.append(np.array([[base_i + i0, base_i + i1, base_i + i2, mtl, vn[0], vn[1], vn[2], vn[3], -1, -1, -1, 0]], np.int32))
            n_vertices += 4
            continue
        f32 = np.float32
        s = (np.arange(n + 1, dtype=f32) / f32(n))
        S, T = np.meshgrid(s, s, indexing="ij")           # S: i (P00->P10), T: j (P00->P01)
        S, T = S[..., None], T[..., None]
        p00, p10, p11, p01 = (v.astype(f32) for v in (P00, P10, P11, P01))
        one = f32(1)
        P = (one - S) * (one - T) * p00 + S * (one - T) * p10 + S * T * p11 + (one - S) * T * p01     # float32 throughout
        idx = n_vertices + np.arange((n + 1) * (n + 1), dtype=np.int64).reshape(n + 1, n + 1).T   # idx[i, j] = base + j*(n+1) + i
        h = pcg_hash_np((idx.astype(np.uint64) ^ np.uint64(0x1234)) & np.uint64(0xFFFFFFFF)).astype(np.float64)
        disp = 0.02 * (h / 4294967296.0 - 0.5)
        interior = np.zeros((n + 1, n + 1), bool)
        interior[1:n, 1:n] = True
        N = base.normals[int(vn[0])].astype(np.float64) if vn[3] == 1 else np.cross(P10 - P00, P01 - P00)
        P = P.astype(np.float64) + (disp * interior)[..., None] * N
        grid = np.empty(((n + 1) * (n + 1), 3), np.float32)
        grid[(idx - n_vertices).reshape(-1)] = P.reshape(-1, 3).astype(np.float32)          # stored in index order
        out_v.append(grid)
        i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        v00, v10, v11, v01 = idx[i, j], idx[i + 1, j], idx[i + 1, j + 1], idx[i, j + 1]
        cells = np.empty((n, n, 2, 12), np.int32)
        cells[..., 0, 0], cells[..., 0, 1], cells[..., 0, 2] = v00, v10, v11
        cells[..., 1, 0], cells[..., 1, 1], cells[..., 1, 2] = v00, v11, v01
        cells[..., 3] = mtl
        cells[..., 4:8] = vn
        cells[..., 8:11] = -1
        cells[..., 11] = 0
        out_t.append(cells.reshape(-1, 12))
        n_vertices += (n + 1) * (n + 1)
    return Mesh(np.concatenate(out_v), base.normals, base.texcoords, np.concatenate(out_t), base.materials,
                base.lights, base.vertex_min)


MIRROR_TYPE, DISNEY_TYPE = 1.0, 17.0          # Caitlyn/Scene.h:111-132 MaterialType (albedo.w, Scene.h:576-582)


def with_disney_materials(base):
    """The Cornell box of BASELINE configs[3] ("4-bounce Disney BSDF"): the tall box becomes a mirror (material `tallBox`
    of the .mtl, `type Mirror`), the short box brushed metal and the floor a glossy dielectric (Disney_type: specular.x =
    metallic, specular.y = roughness).  Apply to the 32-triangle base mesh BEFORE tessellated_cornell — quads are
    identified by their position in Models/cornell-box.obj: tall box = quads 0-4, short box = 5-9, floor = quad 14.
    The material model itself has no reference code (oracle-defined)."""
    assert base.triangles.shape[0] == 32 and base.materials.shape[0] == 6, "expects the base Cornell mesh"
    tris = base.triangles.copy()
    mats = base.materials.copy()
    mats[5, 0:4] = (0.95, 0.95, 0.95, MIRROR_TYPE)                       # tallBox: mirror
    mats[4, 0:4] = (0.955, 0.638, 0.538, DISNEY_TYPE)                    # shortBox: copper-like metal
    mats[4, 8:12] = (1.0, 0.35, 0.0, 0.0)                                # metallic, roughness, -, specular.w = 0 (NEE on)
    floor = np.array([[0.75, 0.75, 0.75, DISNEY_TYPE, 0, 0, 0, -1, 0.0, 0.25, 0, 0, -1, -1, -1, -1]], np.float32)
    mats = np.concatenate([mats, floor])                                 # material 6: glossy grey dielectric
    tris[0:10, 3] = 5
    tris[10:20, 3] = 4
    tris[28:30, 3] = 6
    m = Mesh(base.vertices, base.normals, base.texcoords, tris, mats, base.lights, base.vertex_min)
    if getattr(base, "albedo_textures", None) is not None:
        m.albedo_textures = base.albedo_textures
    return m
