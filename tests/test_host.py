"""Host-side API (camera, RNG, loader, builders) against the SURVEY §8c known answers and the
structural invariants the reference's algorithms imply."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, REFERENCE, ROOT, have_reference


def test_host_rng_known_answers(cr, ob, survey):
    r = cr.Rnd()
    got = [(r.randf2(), r.randf2()) for _ in range(4)]
    np.testing.assert_allclose(got, survey["host_rng"]["random_vectors"], rtol=0, atol=5e-8)
    r = cr.Rnd()
    r.randf2()
    assert hex(r.state.value) == survey["host_rng"]["states_frame1"][0]
    r.randf2()
    assert hex(r.state.value) == survey["host_rng"]["states_frame1"][1]
    # product and oracle agree on the hash
    for x in (0, 1, 12345, 0xFFFFFFFF):
        assert cr.pcg_hash(x) == ob.lib().orc_pcg_hash(x)


def test_host_rng_equals_the_references_own_rnd_h(cr, ob):
    """tests/golden/ref_rnd.json was produced by the reference's Caitlyn/Rnd.h itself, compiled where it lies
    (oracle/_ref/librndref.so, tests/golden/make_ref_rnd_fixture.py): the product's crt_randf2 / crt_pcg_hash and the oracle's
    restatement must give the same bits and states; where the library exists (build container) the fixture is re-derived live."""
    import ctypes as C
    import struct
    ref = json.load(open(os.path.join(GOLDEN, "ref_rnd.json")))
    assert len(ref["sequences"]["1"]) == 256
    for start, seq in ref["sequences"].items():
        r = cr.Rnd(int(start))
        o_state = C.c_uint32(int(start))
        for want_bits, want_state in seq:
            got = r.randf2()
            assert struct.unpack("<I", struct.pack("<f", got))[0] == want_bits and r.state.value == want_state, (start, want_state)
            og = ob.lib().orc_randf2(C.byref(o_state))
            assert struct.unpack("<I", struct.pack("<f", og))[0] == want_bits and o_state.value == want_state
    for x, h in ref["pcg_hash"].items():
        assert cr.pcg_hash(int(x)) == h and ob.lib().orc_pcg_hash(int(x)) == h
    from oracle import rndref
    if rndref.available():
        L = rndref.lib()
        for start, seq in ref["sequences"].items():
            L.ref_rnd_set_state(int(start))
            for want_bits, want_state in seq:
                v = L.ref_randf2()
                assert struct.unpack("<I", struct.pack("<f", v))[0] == want_bits and L.ref_rnd_state() == want_state
        rng = np.random.default_rng(4)
        for x in rng.integers(0, 2 ** 32, 2000, dtype=np.uint64):
            assert cr.pcg_hash(int(x)) == L.ref_pcg_hash(int(x))


def test_camera_matches_reference_construction(cr, survey):
    cam = cr.Camera((-2.755610, 2.745992, 7.58545), (-2.755610, 2.745992, 6.58545), 40.0)   # Scene.h:468
    ka = survey["cornell_load"]
    np.testing.assert_allclose(cam.forward, ka["forward"], atol=1e-6)
    np.testing.assert_allclose(cam.right, ka["right"], atol=1e-6)
    np.testing.assert_allclose(cam.up, ka["up"], atol=1e-6)
    assert abs(cam.fov - ka["fov"]) < 1e-7
    # a tilted camera stays orthonormal
    c2 = cr.Camera((0, 0, 0), (1, 0.5, -2), 55.0)
    M = np.stack([c2.right, c2.up, c2.forward])
    np.testing.assert_allclose(M @ M.T, np.eye(3), atol=1e-6)


@pytest.mark.skipif(not have_reference(), reason="reference data files only exist in the build container")
def test_loader_on_reference_cornell(cr, survey, cornell):
    cam = cr.Camera((-2.755610, 2.745992, 7.58545), (-2.755610, 2.745992, 6.58545), 40.0)
    m = cr.Mesh.read_object(os.path.join(REFERENCE, "Models", "cornell-box.obj"), cam)
    ka = survey["cornell_load"]
    assert (m.vertices.shape[0], m.normals.shape[0], m.triangles.shape[0], m.materials.shape[0], m.lights.shape[0]) == \
        (ka["counts"]["vertices"], ka["counts"]["normals"], ka["counts"]["triangles"], ka["counts"]["materials"], ka["counts"]["lights"])
    np.testing.assert_allclose(m.vertex_min, ka["vertex_min"], atol=1e-6)
    np.testing.assert_allclose(cam.position, ka["camera_position"], atol=1e-6)
    # the committed fixture is exactly what the loader produces from the reference's data file
    fx, _ = cornell
    assert np.array_equal(m.vertices, fx.vertices) and np.array_equal(m.triangles, fx.triangles)
    assert np.array_equal(m.materials, fx.materials) and np.array_equal(m.lights, fx.lights)


def test_loader_semantics_on_handwritten_obj(cr, tmp_path):
    (tmp_path / "m.mtl").write_text(
        "newmtl A\nKa 1 1 1\nKd 0.5 0.25 0.125\nKs 1 1 1\nKe 0 0 0\n"
        "newmtl Lamp\nKd 1 1 1\n//Ke 9 9 9\nKe 2 3 4\n"
        "newmtl Mir\ntype Mirror\nKd 0.1 0.2 0.3\nKe 0 0 0\n")
    (tmp_path / "m.obj").write_text(
        "# comment\nmtllib m.mtl\nv 1 1 1\nv 3 1 1\nv 3 3 1\nv 1 3 1\nv 2 2 5\n"
        "vt 0 0\nvt 1 0.25\nvn 0 0 1\n"
        "usemtl A\nf 1//1 2//1 3//1 4//1\n"        # quad -> fan of 2, v//vn
        "usemtl Lamp\nf -5/1 -4/2 -1/1\n"          # negative indices, v/vt, no normals -> integer normal
        "usemtl Mir\nf 1/1/1 2/2/1 5/1/1\n"        # v/vt/vn
        "f 1 2 3\n"                                # bare corners: no branch in the reference -> nothing
        "l 1 2\ns off\n")
    m = cr.Mesh.read_object(str(tmp_path / "m.obj"))
    assert m.triangles.shape[0] == 4
    np.testing.assert_allclose(m.vertex_min, [1, 1, 1])
    np.testing.assert_allclose(m.vertices[0], [0, 0, 0])          # translated by -vertex_min (Scene.h:915-925)
    assert m.triangles[0].tolist() == [0, 1, 2, 0, 0, 0, 0, 1, -1, -1, -1, 0]
    assert m.triangles[1].tolist() == [0, 2, 3, 0, 0, 0, 0, 1, -1, -1, -1, 0]
    # cross((2,0,0),(1,1,4)) = (0,-8,2) truncated to ints, w = 0 (Scene.h:849-852)
    assert m.triangles[2].tolist() == [0, 1, 4, 1, 0, -8, 2, 0, 0, 1, 0, 0]
    assert m.triangles[3].tolist() == [0, 1, 4, 2, 0, 0, 0, 1, 0, 1, 0, 0]
    np.testing.assert_allclose(m.texcoords, [[0, 1], [1, 0.75]])  # vt stored as (u, 1-v), Scene.h:801
    np.testing.assert_allclose(m.materials[0][:8], [0.5, 0.25, 0.125, 0, 0, 0, 0, -1])
    np.testing.assert_allclose(m.materials[1][4:8], [2, 3, 4, 0])  # first emissive material -> light index 0
    assert m.materials[2][3] == 1.0                                # type Mirror -> albedo.w = Mirror_type
    assert m.lights.shape[0] == 1
    L = m.lights[0]
    np.testing.assert_allclose(L[0:3], [0, 0, 0]); np.testing.assert_allclose(L[3:6], [2, 0, 0]); np.testing.assert_allclose(L[6:9], [1, 1, 4])
    np.testing.assert_allclose(L[15:18], [np.sqrt(68.0), 1.0, 0.0], rtol=1e-6)   # |u x v|, pdf = area / sum


def test_loader_short_m_lines_are_not_read_past_their_end(cr, tmp_path):
    """Scene.h:890-899 takes the first line that starts with 'm' as `mtllib` and skips six characters; a shorter line
    has nothing behind them (no read past the terminator)."""
    (tmp_path / "m.mtl").write_text("newmtl A\nKd 0.5 0.25 0.125\nKe 0 0 0\n")
    (tmp_path / "s.obj").write_text("m\nmtl\nmtllib\nmtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nusemtl A\nf 1//1 2//1 3//1\n")
    m = cr.Mesh.read_object(str(tmp_path / "s.obj"))
    assert m.triangles.shape[0] == 1 and m.materials.shape[0] == 1
    np.testing.assert_allclose(m.materials[0][:3], [0.5, 0.25, 0.125])


def test_sbvh_cornell_known_answers(cr, cornell, survey):
    mesh, _ = cornell
    b = cr.SBVH(mesh.triangles, mesh.vertices)
    ka = survey["cornell_sbvh"]
    assert b.flat_nodes.shape[0] == ka["n_nodes"] and b.count_leaf() == ka["n_leaves"]
    assert b.triangle_indices.tolist() == ka["triangle_indices"]
    for name, idx in (("node0", 0), ("node2", 2)):
        n = b.flat_nodes[idx]
        np.testing.assert_allclose(n[0:3], ka[name]["min"], atol=1e-6)
        np.testing.assert_allclose(n[4:7], ka[name]["max"], atol=1e-6)
        assert n[3] == ka[name]["min_w"] and n[7] == ka[name]["max_w"]
    assert np.array_equal(b.triangles, mesh.triangles[b.triangle_indices])
    # regression pin of the full arrays (same values on every machine)
    with open(os.path.join(GOLDEN, "cornell_bvh.json")) as f:
        g = json.load(f)
    assert b.flat_nodes.view(np.uint32).ravel().tolist() == g["flat_nodes_bits"]
    cw = cr.CWBVH().convert(b)
    assert cw.nodes.ravel().tolist() == g["bvh8_bytes"] and cw.tri_slots.tolist() == g["bvh8_tri_slots"]


def _check_bvh2(flat, tris, verts, n_source):
    n = flat.shape[0]
    leaf = flat[:, 7] != 0
    assert (flat[leaf, 7] == 1).all()                        # convert_to_bvh1: single-triangle leaves
    inner = np.nonzero(~leaf)[0]
    left = flat[inner, 3].astype(np.int64)
    assert (left > inner).all() and (left + 1 < n).all()
    # BFS numbering: the k-th interior node's children are 2k+1, 2k+2 (sbvh.h:595-606)
    assert np.array_equal(left, 2 * np.arange(len(inner)) + 1)
    slots = flat[leaf, 3].astype(np.int64)
    assert sorted(slots.tolist()) == list(range(tris.shape[0]))
    # children lie inside their parent except the full-triangle boxes of split 2-leaves
    tri_lo = verts[tris[:, :3]].min(1)
    tri_hi = verts[tris[:, :3]].max(1)
    for i in np.nonzero(leaf)[0]:
        s = int(flat[i, 3])
        assert (flat[i, 0:3] <= tri_hi[s] + 1e-6).all() and (flat[i, 4:7] >= tri_lo[s] - 1e-6).all()
    return len(inner)


def test_sbvh_structure_and_duplicates(cr, tess8):
    mesh, data = tess8
    _check_bvh2(data.bvh, data.triangles, mesh.vertices, mesh.triangles.shape[0])
    ids = data.tri_orig_ids
    assert set(ids.tolist()) == set(range(mesh.triangles.shape[0]))   # every source triangle referenced
    assert len(ids) >= mesh.triangles.shape[0]
    nosplit = cr.SBVH(mesh.triangles, mesh.vertices, cr.SBVH.NO_SPATIAL_SPLITS)
    assert len(nosplit.triangle_indices) == mesh.triangles.shape[0]   # pure SAH sweep: a permutation
    assert sorted(nosplit.triangle_indices.tolist()) == list(range(mesh.triangles.shape[0]))


def test_sbvh_spatial_splits_duplicate_long_triangles(cr):
    # two long crossing slivers force spatial splits: the classic SBVH case
    v = np.array([[0, 0, 0], [10, 0.1, 0], [10, 0, 0.1], [0, 5, 5], [0.1, -5, -5], [0, -5, -5.1],
                  [5, 5, 0], [5.1, -5, 0.1], [5, -5, 0]], np.float32)
    t = np.zeros((3, 12), np.int32)
    t[:, :3] = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
    b = cr.SBVH(t, v)
    assert b.flat_nodes.shape[0] == 2 * len(b.triangle_indices) - 1
    assert set(b.triangle_indices.tolist()) == {0, 1, 2}


def _decode_children(node):
    """(slot, is_inner, lo, hi, meta) of the occupied slots of one 80-byte node8."""
    p = node[0:12].view(np.float32)
    e = node[12:15].astype(np.uint32)
    scale = (e << 23).view(np.float32)
    meta = node[24:32]
    q = node[32:80].reshape(3, 2, 8)   # axis, lo/hi, slot
    out = []
    for s in range(8):
        if meta[s] == 0:
            continue
        lo = p.astype(np.float64) + q[:, 0, s].astype(np.float64) * scale.astype(np.float64)
        hi = p.astype(np.float64) + q[:, 1, s].astype(np.float64) * scale.astype(np.float64)
        inner = bool((int(meta[s]) & (int(meta[s]) << 1)) & 0x10)
        out.append((s, inner, lo, hi, int(meta[s])))
    return p, out


def _validate_cwbvh(cw, flat):
    """Every child box decodes to a superset of the BVH2 node it stands for (clipped to the node8's own
    quantisation frame); imask and meta agree; inner children are consecutive from child_base; leaf
    triangles are exactly the BVH2 subtree's slots; every slot is referenced once."""
    cw_nodes, tri_slots, child_bvh2 = cw.nodes, cw.tri_slots, cw.child_bvh2
    seen_tri = np.zeros(len(tri_slots), bool)

    def subtree_slots(n):
        if flat[n, 7] != 0:
            return list(range(int(flat[n, 3]), int(flat[n, 3]) + int(flat[n, 7])))
        l = int(flat[n, 3])
        return subtree_slots(l) + subtree_slots(l + 1)

    stack = [(0, 1)]
    depth = 0
    n_visited = 0
    while stack:
        ni, lvl = stack.pop()
        depth = max(depth, lvl)
        n_visited += 1
        node = cw_nodes[ni]
        p, kids = _decode_children(node)
        scale = ((node[12:15].astype(np.uint32) << 23).view(np.float32)).astype(np.float64)
        frame_lo = p.astype(np.float64)
        frame_hi = frame_lo + 255.0 * scale
        imask = int(node[15])
        child_base = int(node[16:20].view(np.uint32)[0])
        tri_base = int(node[20:24].view(np.uint32)[0])
        rank = 0
        assert {s for s, *_ in kids} == {s for s in range(8) if child_bvh2[ni, s] >= 0}
        for (s, inner, lo, hi, meta) in kids:
            b2 = int(child_bvh2[ni, s])
            want_lo = np.maximum(flat[b2, 0:3].astype(np.float64), frame_lo)
            want_hi = np.minimum(flat[b2, 4:7].astype(np.float64), frame_hi)
            ok = want_lo <= want_hi
            assert (lo[ok] <= want_lo[ok]).all() and (hi[ok] >= want_hi[ok]).all()
            assert inner == bool((imask >> s) & 1)
            if inner:
                assert meta == ((24 + s) | 0x20)
                stack.append((child_base + rank, lvl + 1))
                rank += 1
            else:
                cnt = {1: 1, 3: 2, 7: 3}[meta >> 5]
                off = meta & 31
                got = [int(tri_slots[tri_base + off + j]) for j in range(cnt)]
                assert got == subtree_slots(b2)
                for j in range(cnt):
                    assert not seen_tri[tri_base + off + j]
                    seen_tri[tri_base + off + j] = True
    assert seen_tri.all()
    assert n_visited == cw_nodes.shape[0]
    assert sorted(tri_slots.tolist()) == list(range(len(tri_slots)))
    assert depth == cw.depth
    return depth


def test_cwbvh_structure(cr, cornell, cornell_data, tess8):
    mesh, _ = cornell
    sb = cr.SBVH(mesh.triangles, mesh.vertices)
    cw = cr.CWBVH().convert(sb)
    assert _validate_cwbvh(cw, sb.flat_nodes) == 2 and cw.nodes.shape == (3, 80)
    assert np.array_equal(cw.nodes, cornell_data.bvh8)
    m8, d8 = tess8
    cw8 = cr.CWBVH().convert_arrays(d8.bvh, d8.triangles.shape[0])
    assert np.array_equal(cw8.nodes, d8.bvh8) and np.array_equal(cw8.tri_slots, d8.bvh8_tri_slots)
    depth = _validate_cwbvh(cw8, d8.bvh)
    assert depth <= 16
    # compression: well under one node per triangle (SURVEY a8 estimates 0.2-0.35)
    assert d8.bvh8.shape[0] < 0.4 * d8.triangles.shape[0]


@pytest.mark.parametrize("n_tris", [1, 2, 3, 4, 5, 9])
def test_cwbvh_tiny_scenes(cr, ob, n_tris):
    rng = np.random.default_rng(n_tris)
    v = rng.random((3 * n_tris, 3)).astype(np.float32) * 4
    t = np.zeros((n_tris, 12), np.int32)
    t[:, :3] = np.arange(3 * n_tris).reshape(-1, 3)
    b = cr.SBVH(t, v)
    cw = cr.CWBVH().convert(b)
    _validate_cwbvh(cw, b.flat_nodes)
    assert cw.nodes.shape[0] >= 1


def test_cwbvh_flat_and_degenerate_boxes(cr):
    # axis-aligned walls: extent 0 on one axis -> the exponent must not come from log2(0) (cwbvh.h:305-311)
    v = np.array([[0, 0, 0], [4, 0, 0], [4, 0, 4], [0, 0, 4], [0, 3, 0], [4, 3, 0], [4, 3, 4], [0, 3, 4]], np.float32)
    t = np.zeros((4, 12), np.int32)
    t[:, :3] = [[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7]]
    b = cr.SBVH(t, v)
    cw = cr.CWBVH().convert(b)
    e = cw.nodes[:, 12:15]
    assert (e >= 1).all() and (e <= 254).all()
    _validate_cwbvh(cw, b.flat_nodes)


def test_cwbvh_rejects_fat_leaves(cr):
    flat = np.zeros((1, 8), np.float32)
    flat[0] = [0, 0, 0, 0, 1, 1, 1, 5]      # a BVH2 leaf with 5 triangles cannot be encoded (max 3, cwbvh.h:101)
    with pytest.raises(cr.CrtError):
        cr.CWBVH().convert_arrays(flat, 5)


def test_meshgen_counts(cr, cornell):
    from caitlynrenderer_amd.meshgen import pcg_hash_np, tessellated_cornell
    mesh, _ = cornell
    m = tessellated_cornell(mesh, 8)
    assert m.triangles.shape[0] == 15 * 8 * 8 * 2 + 2 == 1922
    assert m.vertices.shape[0] == 15 * 81 + 4
    assert 15 * 183 * 183 * 2 + 2 == 1004672 and 15 * 184 * 184 + 4 == 507844   # config 3
    assert (m.triangles[:, 7] == 1).all()                                        # vertex normals present
    xs = np.array([0, 1, 77, 0xFFFFFFFF], np.uint32)
    assert [int(x) for x in pcg_hash_np(xs)] == [cr.pcg_hash(int(x)) for x in xs]
    # deterministic
    m2 = tessellated_cornell(mesh, 8)
    assert np.array_equal(m.vertices.view(np.uint32), m2.vertices.view(np.uint32))


def _bvh2_depth(flat):
    depth = np.zeros(flat.shape[0], np.int32)
    for i in np.nonzero(flat[:, 7] == 0)[0]:
        l = int(flat[i, 3])
        depth[l] = depth[l + 1] = depth[i] + 1
    return int(depth.max())


@pytest.mark.parametrize("n", [8, 183])
def test_sbvh_reproduces_the_survey_probes_with_spatial_splits(cr, cornell, survey, n):
    """The two reference-derived SBVH results that exercise spatial-split duplicates (n = 8; §8: n = 183 —
    the survey ran the reference's own sbvh.h on the §8d mesh): leaf slots, BVH2 nodes and depth are reproduced exactly.
    Round 1 missed them by a handful of nodes because meshgen numbered grid vertices i-major and interpolated in float64
    (the displacement hashes the vertex index); the builder was not the difference."""
    from caitlynrenderer_amd.meshgen import tessellated_cornell
    want = survey["sbvh_on_tessellated_cornell"][f"n{n}"]
    mesh = tessellated_cornell(cornell[0], n)
    assert mesh.triangles.shape[0] == want["triangles"]
    sb = cr.SBVH(mesh.triangles, mesh.vertices)
    assert sb.triangle_indices.shape[0] == want["leaf_slots"]
    assert sb.flat_nodes.shape[0] == want["bvh2_nodes"] == 2 * want["leaf_slots"] - 1
    assert _bvh2_depth(sb.flat_nodes) == want["depth"]
    if "duplicates" in want:
        assert want["leaf_slots"] - want["triangles"] == want["duplicates"]
        assert mesh.vertices.shape[0] == want["vertices"]


def test_sbvh_is_identical_for_any_thread_count(cr, tess40):
    """Subtrees are built concurrently from private copies and stitched in the sequential order, so
    the tree must not depend on CRT_BUILD_THREADS (csrc/host/sbvh.cpp build_rec)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import sys, hashlib; sys.path.insert(0, %r); import __graft_entry__ as g; import caitlynrenderer_amd as cr;"
            "from caitlynrenderer_amd.meshgen import tessellated_cornell; m = tessellated_cornell(g._cornell()[0], 40);"
            "sb = cr.SBVH(m.triangles, m.vertices); print(hashlib.sha1(sb.flat_nodes.tobytes() + sb.triangle_indices.tobytes()).hexdigest())") % ROOT
    digests = set()
    for th in ("1", "3", "8"):
        env = dict(os.environ, CRT_BUILD_THREADS=th)
        digests.add(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip())
    import hashlib
    mesh, data = tess40
    digests.add(hashlib.sha1(data.bvh.tobytes() + data.tri_orig_ids.tobytes()).hexdigest())
    assert len(digests) == 1


def test_obj_round_trip_and_ppm(cr, cornell, tmp_path):
    """write_obj -> Read_Object reproduces the scene (up to the fp32 re-translation of vertices), and the
    PPM writer flips the bottom-up frame."""
    from conftest import write_obj
    from caitlynrenderer_amd.image import read_ppm, write_ppm
    mesh, _ = cornell
    write_obj(mesh, str(tmp_path / "c.obj"))
    m = cr.Mesh.read_object(str(tmp_path / "c.obj"))
    assert np.array_equal(m.triangles[:, :8], mesh.triangles[:, :8])
    np.testing.assert_allclose(m.vertices, mesh.vertices, atol=2e-6)
    assert np.array_equal(m.lights.shape, mesh.lights.shape) and m.materials.shape == mesh.materials.shape
    np.testing.assert_allclose(m.materials[:, :8], mesh.materials[:, :8])
    img = (np.arange(5 * 7 * 4) % 251).astype(np.uint8).reshape(5, 7, 4)
    write_ppm(str(tmp_path / "a.ppm"), img)
    assert np.array_equal(read_ppm(str(tmp_path / "a.ppm")), img[:, :, :3])
    # PNG writer of the boundary (crt_image_encode_png): the file decodes to the picture, top row first
    from caitlynrenderer_amd import host
    from caitlynrenderer_amd.image import write_png
    write_png(str(tmp_path / "a.png"), img)
    assert np.array_equal(host.decode_image(open(tmp_path / "a.png", "rb").read()), img[::-1, :, :3])
    rng = np.random.default_rng(3)
    for shape in ((1, 1, 3), (9, 4, 4), (120, 77, 3)):
        px = rng.integers(0, 256, shape, dtype=np.uint8)
        px[: shape[0] // 2] = px[0, 0]                                    # flat rows and noisy rows: different filters win
        data = host.encode_png(px)
        assert data[:8] == b"\x89PNG\r\n\x1a\n" and np.array_equal(host.decode_image(data), px[..., :3])
        import zlib, struct
        at = 8
        while at < len(data):                                             # every chunk carries a valid CRC
            n, tag = struct.unpack(">I4s", data[at:at + 8])
            assert struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0] == zlib.crc32(data[at + 4:at + 8 + n])
            at += 12 + n


def test_cpp_example_builds_and_reports_usage(cr):
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "examples", "render_obj")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_flatnode_link_word_round_trips_in_both_of_its_forms(tmp_path):
    """host/flatnode_link.hpp: a FlatNode link below 2^24 is the reference's float (FlatNode.h:34-40, exact), one of 2^24 or more its bit pattern,
    and a reader tells them apart by the value (floats written from integers are >= 1.0; slot 0 is 0.0f either way).  Compiled for the host here;
    the device builders and walks include the same header."""
    import subprocess
    src = tmp_path / "link.cpp"
    src.write_text('#include <cstdio>\n#include <cstring>\n#include "host/flatnode_link.hpp"\n'
                   'int main() {\n'
                   '  const unsigned probe[] = {0u, 1u, 2u, 77u, (1u << 23) - 1u, (1u << 23), (1u << 24) - 1u, (1u << 24), (1u << 24) + 1u, 33554431u, 123456789u, crt::kMaxLinkBits - 1u};\n'
                   '  for (unsigned i : probe) {\n'
                   '    const float w = crt::link_enc(i);\n'
                   '    if ((unsigned)crt::link_of(w) != i) { std::printf("round trip of %u failed\\n", i); return 1; }\n'
                   '    if (i < (1u << 24) && w != (float)i) { std::printf("%u is not the float form\\n", i); return 1; }\n'
                   '    if (i >= (1u << 24) && !(w < 1.0f)) { std::printf("%u reads as a float\\n", i); return 1; }\n'
                   '  }\n'
                   '  for (unsigned i = 1; i < (1u << 24); i += 9973u) if (crt::link_of((float)i) != (int)i) return 2;   // every float-valued link a caller can hand in\n'
                   '  std::puts("ok");\n  return 0;\n}\n')
    exe = tmp_path / "link"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "caitlynrenderer_amd", "csrc"), "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout
