"""The oracle (CPU restatement, oracle/oracle.c) against the SURVEY §8c known answers, against its own
committed vectors, and against brute force."""
import numpy as np
import pytest

from conftest import seeded_rays


def test_bvh2_primary_census_known_answers(ob, cornell_data, survey):
    """Reference-order BVH2 walk (path_trace.fs:511-667) on 1920x1080 pixel-centre rays."""
    ka = survey["bvh2_primary_census_1920x1080_no_jitter"]
    o = ob.Oracle(cornell_data, 1920, 1080, 3)
    rays = o.primary_rays(jitter=False)
    assert rays.shape[0] == ka["rays"]
    hits, st = o.trace(rays, ob.BVH2, ob.CLOSEST, ob.TIE_FIRST_VISITED, stats=True, threads=4)
    assert int((hits["tri"] >= 0).sum()) == ka["hits"]
    c = hits[ka["centre_pixel"]["py"] * 1920 + ka["centre_pixel"]["px"]]
    assert c["tri"] == ka["centre_pixel"]["triangle"] and abs(c["t"] - ka["centre_pixel"]["t"]) < 1e-6
    assert abs(st["nodes"].mean() - ka["nodes_per_ray"]) < 0.005 and abs(st["tris"].mean() - ka["tris_per_ray"]) < 0.005
    # the lowest-id tie rule only differs from the reference's first-visited rule on exact ties: none here
    h2 = o.trace(rays, ob.BVH2, ob.CLOSEST, ob.TIE_LOWEST_ID, threads=4)
    assert np.array_equal(h2["tri"], hits["tri"]) and np.array_equal(h2["t"].view(np.uint32), hits["t"].view(np.uint32))
    # CWBVH walk finds the same hits bit for bit
    h8 = o.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, threads=4)
    assert np.array_equal(h8["tri"], hits["tri"]) and np.array_equal(h8["t"].view(np.uint32), hits["t"].view(np.uint32))


@pytest.mark.parametrize("scene", ["cornell", "tess8"])
def test_committed_vectors(ob, cr, cornell, cornell_data, tess8, vectors, scene):
    data = cornell_data if scene == "cornell" else tess8[1]
    o = ob.Oracle(data, 96, 54, 3, cornell[1])
    rays = vectors[f"{scene}_rays"].view(ob.RAY_DT).ravel()
    h, st = o.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, stats=True)
    assert np.array_equal(h.view(np.uint32).reshape(-1, 4), vectors[f"{scene}_hits"])
    assert np.array_equal(st.view(np.uint16).reshape(-1, 2), vectors[f"{scene}_stats"])
    ra = rays.copy()
    ra["tmax"] = np.float32(2.5)
    assert np.array_equal((o.trace(ra, ob.BVH8, ob.ANY)["tri"] >= 0).astype(np.uint8), vectors[f"{scene}_any"])
    rnd = cr.Rnd()
    s = np.zeros((54, 96, 3), np.float32)
    for _ in range(2):
        o.render_frame(rnd.randf2(), rnd.randf2(), s)
    assert np.array_equal(s.view(np.uint32), vectors[f"{scene}_sum2_bits"])


def test_pinned_sin_cos_and_rand(ob, vectors):
    xs = vectors["sin_x"].view(np.float32)
    got_s = np.array([ob.lib().orc_sin(float(x)) for x in xs], np.float32)
    got_c = np.array([ob.lib().orc_cos(float(x)) for x in xs], np.float32)
    assert np.array_equal(got_s.view(np.uint32), vectors["sin_y"]) and np.array_equal(got_c.view(np.uint32), vectors["cos_y"])
    # correctly rounded for all practical purposes: equal to the float rounding of a float64 libm result
    assert (got_s == np.sin(xs.astype(np.float64)).astype(np.float32)).mean() > 0.999
    assert np.abs(got_s.astype(np.float64) - np.sin(xs.astype(np.float64))).max() < 6e-8
    assert np.abs(got_c.astype(np.float64) - np.cos(xs.astype(np.float64))).max() < 6e-8
    seq = np.array(ob.rand_sequence(960, 540, 0.6591631, 0.910802, 16), np.float32)
    assert np.array_equal(seq.view(np.uint32), vectors["rand_seq"])
    assert ((seq >= 0) & (seq < 1)).all()
    # the definition itself (path_trace.fs:38-42), recomputed in numpy
    sx = sy = None
    sx, sy = np.float32(960.5), np.float32(540.5)
    rv = np.float32(0.6591631) * np.float32(0.910802)
    for k in range(4):
        sx, sy = np.float32(sx - rv), np.float32(sy - rv)
        d = np.float32(np.float32(sx * np.float32(12.9898)) + np.float32(sy * np.float32(78.233)))
        v = np.float32(np.float32(np.sin(np.float64(d))) * np.float32(43758.5453))
        assert seq[k] == np.float32(v - np.floor(v))


@pytest.mark.parametrize("scene", ["cornell", "tess8"])
def test_brute_force_bvh2_cwbvh_agree(ob, cornell, cornell_data, tess8, scene):
    mesh, data = (cornell[0], cornell_data) if scene == "cornell" else tess8
    o = ob.Oracle(data, 64, 64, 3, cornell[1])
    rays = seeded_rays(mesh, 6000, 11, ob.RAY_DT)
    hb = o.trace(rays, ob.BRUTE, ob.CLOSEST, ob.TIE_LOWEST_ID, threads=4)
    h2 = o.trace(rays, ob.BVH2, ob.CLOSEST, ob.TIE_LOWEST_ID)
    h8 = o.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID)
    for h in (h2, h8):
        # identical (id, t) except where slab rounding meets an exact edge hit: allow none on these seeds
        assert np.array_equal(h["tri"], hb["tri"]) and np.array_equal(h["t"].view(np.uint32), hb["t"].view(np.uint32))
    ra = rays.copy()
    ra["tmax"] = np.float32(1.5)
    ab = o.trace(ra, ob.BRUTE, ob.ANY, threads=4)["tri"] >= 0
    assert np.array_equal(o.trace(ra, ob.BVH2, ob.ANY)["tri"] >= 0, ab)
    assert np.array_equal(o.trace(ra, ob.BVH8, ob.ANY)["tri"] >= 0, ab)
    # any-hit is consistent with closest-hit: occluded <=> nearest hit closer than tmax
    assert np.array_equal(ab, (hb["tri"] >= 0) & (hb["t"] < 1.5))


@pytest.mark.parametrize("scene,n", [("tess8", 1500), ("tess40", 250)])
def test_numpy_brute_force_agrees_with_all_three_walks(ob, cornell, tess8, tess40, scene, n):
    """Breaks the symmetry between oracle.c and the HIP kernel with a third implementation that shares no code with either:
    vectorised numpy over all triangles (tests/conftest.py numpy_brute_force).  Bit-equal (id, t, u, v)."""
    from conftest import numpy_brute_force
    mesh, data = tess8 if scene == "tess8" else tess40
    o = ob.Oracle(data, 64, 64, 3, cornell[1])
    rays = seeded_rays(mesh, n, 23, ob.RAY_DT)
    rays["tmax"][::7] = np.float32(2.0)                                    # some rays with a finite reach
    tri, t, u, v = numpy_brute_force(mesh, rays)
    assert (tri >= 0).sum() > n // 2 and (tri < 0).sum() > 0
    for accel in (ob.BRUTE, ob.BVH2, ob.BVH8):
        h = o.trace(rays, accel, ob.CLOSEST, ob.TIE_LOWEST_ID, threads=4)
        assert np.array_equal(h["tri"], tri), accel
        hit = tri >= 0
        for a, b in ((h["t"], t), (h["u"], u), (h["v"], v)):
            assert np.array_equal(a[hit].view(np.uint32), b[hit].view(np.uint32)), accel


def test_duplicate_references_report_original_ids(ob, tess8, cornell):
    mesh, data = tess8
    o = ob.Oracle(data, 64, 64, 3, cornell[1])
    rays = seeded_rays(mesh, 2000, 5, ob.RAY_DT)
    h = o.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID)
    hit = h["tri"] >= 0
    assert hit.any() and h["tri"][hit].max() < mesh.triangles.shape[0]


def test_axis_parallel_and_degenerate_rays(ob, cornell, cornell_data):
    """d components of exactly 0 give 0*inf = NaN slabs; fmin/fmax semantics must keep all three walks equal."""
    mesh = cornell[0]
    o = ob.Oracle(cornell_data, 64, 64, 3, cornell[1])
    rays = np.zeros(12, ob.RAY_DT)
    dirs = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    for i, d in enumerate(dirs):
        rays[i]["o"], rays[i]["d"] = (2.78, 2.75, 2.8), d
        rays[6 + i]["o"], rays[6 + i]["d"] = (0.0, 2.75, 2.8), d       # origin ON a wall plane
    rays["tmax"] = np.float32(1e9)
    hb = o.trace(rays, ob.BRUTE, ob.CLOSEST, ob.TIE_LOWEST_ID)
    for accel in (ob.BVH2, ob.BVH8):
        h = o.trace(rays, accel, ob.CLOSEST, ob.TIE_LOWEST_ID)
        a, b = h[:6], hb[:6]
        assert np.array_equal(a["tri"], b["tri"]) and np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))
        # origin exactly on a box face with a zero direction component: the slab test sees 0*inf = NaN
        # and may cull what brute force finds (an inherent property of slab tests, same in the
        # reference's hit_bbox); whatever a walk does report must still be the true nearest hit
        a, b = h[6:], hb[6:]
        rep = a["tri"] >= 0
        assert np.array_equal(a["tri"][rep], b["tri"][rep]) and np.array_equal(a["t"][rep].view(np.uint32), b["t"][rep].view(np.uint32))
    assert (hb["tri"][[0, 1, 2, 3, 5]] >= 0).all() and hb["tri"][4] == -1   # the box is open towards +z


def test_zero_direction_components_do_not_degenerate(ob, cornell, tess8):
    """fract(sin()*43758.5453) returns exactly 0 about once in 400 calls, so cosine samples equal to an
    axis-aligned normal (two zero components) are routine.  The CWBVH walk clamps such components to
    +-2^-80 for the slab test: hits equal brute force and the visit count stays that of a normal ray."""
    mesh, data = tess8
    o = ob.Oracle(data, 64, 64, 3, cornell[1])
    rng = np.random.default_rng(1)
    rays = np.zeros(600, ob.RAY_DT)
    rays["o"] = (0.3 + 4.9 * rng.random((600, 3))).astype(np.float32)
    dirs = np.array([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (0, 0.6, 0.8), (0.6, 0, -0.8), (-0.0, 1, 0)], np.float32)
    rays["d"] = dirs[np.arange(600) % len(dirs)]
    rays["tmax"] = np.float32(1e9)
    hb = o.trace(rays, ob.BRUTE, ob.CLOSEST, ob.TIE_LOWEST_ID, threads=4)
    h8, st = o.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, stats=True)
    assert np.array_equal(h8["tri"], hb["tri"]) and np.array_equal(h8["t"].view(np.uint32), hb["t"].view(np.uint32))
    generic = rays.copy()
    generic["d"] = (generic["d"] + np.float32(0.01)) / np.linalg.norm(generic["d"] + np.float32(0.01), axis=1, keepdims=True)
    _, sg = o.trace(generic, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, stats=True)
    assert st["nodes"].mean() < 1.5 * sg["nodes"].mean() + 2 and st["nodes"].max() < 200
    # the shader RNG really does return exact zeros at that rate
    seqs = np.array([ob.rand_sequence(px, 7, 0.6591631, 0.910802, 4) for px in range(0, 1920, 2)], np.float32)
    assert 0.0005 < (seqs == 0).mean() < 0.01 and (seqs < 1).all()
    # non-finite origins and directions terminate immediately instead of walking the whole tree
    bad = np.zeros(3, ob.RAY_DT)
    bad["o"] = [(np.nan, 1, 1), (1, np.inf, 1), (1, 1, 1)]
    bad["d"] = [(0, 0, 1), (0, 0, 1), (np.nan, np.nan, np.nan)]
    bad["tmax"] = 1e9
    hbad, sbad = o.trace(bad, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, stats=True)
    assert (hbad["tri"] == -1).all() and sbad["nodes"][:2].max() == 0 and sbad["nodes"][2] < 50


def test_integrator_statistics_and_energy(ob, cr, cornell, cornell_data):
    """Frame sums: oracle paths through BVH2 and CWBVH give the same radiance; rays are counted."""
    o = ob.Oracle(cornell_data, 128, 72, 3, cornell[1])
    s8, c8 = o.render_frame(0.6591631, 0.910802, accel=ob.BVH8, threads=4)
    s2, c2 = o.render_frame(0.6591631, 0.910802, accel=ob.BVH2, threads=4)
    assert np.array_equal(s8.view(np.uint32), s2.view(np.uint32))
    assert c8[0] == c2[0] and c8[1] == c2[1] and 128 * 72 <= c8[0] <= 3 * 128 * 72
    assert np.isfinite(s8).all() and (s8 >= 0).all() and 0 < s8.max() < 50
    # depth 1 emits at most one shadow ray per primary hit
    o1 = ob.Oracle(cornell_data, 128, 72, 1, cornell[1])
    _, c1 = o1.render_frame(0.6591631, 0.910802, threads=4)
    assert c1[0] == 128 * 72 and c1[1] <= c1[0]


def test_resolve_matches_output_shader_formula(ob):
    rng = np.random.default_rng(3)
    s = (rng.random((40, 30, 3)) * 6).astype(np.float32)
    got = ob.resolve(s, 0.25)
    c = s.astype(np.float64) * 0.25
    lum = 0.3 * c[..., 0] + 0.6 * c[..., 1] + 0.1 * c[..., 2]
    want = np.clip((c / (1 + lum / 2)[..., None]) ** (1 / 2.2), 0, 1) * 255 + 0.5
    assert np.abs(got[..., :3].astype(np.int32) - want.astype(np.int32)).max() <= 1
    assert (got[..., 3] == 255).all()


def test_config1_sah_bvh_primary_rays_cpu(ob, cr, cornell):
    """BASELINE config 1: Cornell box, SAH BVH (exact-sweep object splits only, sbvh.h:338-378 with spatial
    splits disabled), primary rays only, CPU scalar traversal.  Same hits as the SBVH tree and as brute force."""
    mesh, cam = cornell
    sah = cr.SceneData.build(mesh, cam, sbvh_flags=cr.SBVH.NO_SPATIAL_SPLITS)
    full = cr.SceneData.build(mesh, cam)
    assert sah.bvh.shape[0] == 2 * mesh.triangles.shape[0] - 1 and sorted(sah.tri_orig_ids.tolist()) == list(range(32))
    o_sah, o_full = ob.Oracle(sah, 1920, 1080, 1, cam), ob.Oracle(full, 1920, 1080, 1, cam)
    rays = o_sah.primary_rays(0.6591631, 0.910802, jitter=True)          # frame-1 randomVector
    h_sah = o_sah.trace(rays, ob.BVH2, ob.CLOSEST, ob.TIE_FIRST_VISITED, threads=4)
    h_full = o_full.trace(rays, ob.BVH2, ob.CLOSEST, ob.TIE_FIRST_VISITED, threads=4)
    assert np.array_equal(h_sah["tri"], h_full["tri"]) and np.array_equal(h_sah["t"].view(np.uint32), h_full["t"].view(np.uint32))
    sample = slice(None, None, 53)
    hb = o_sah.trace(rays[sample], ob.BRUTE, ob.CLOSEST, ob.TIE_LOWEST_ID, threads=4)
    assert np.array_equal(hb["tri"], h_sah["tri"][sample])
    assert 0.5 < (h_sah["tri"] >= 0).mean() < 0.6                      # 56 % of a 16:9 frame sees the box


def test_textured_albedo_definition(ob, cr, textured):
    """The texture branch (path_trace.fs:471-483): bilinear GL_LINEAR / GL_REPEAT fetch of the RGB8 array and
    pow(., 2.2), checked against an independent numpy evaluation on a frame where only texture data changes."""
    mesh, data, cam = textured
    assert data.albedo_textures is not None and data.texcoords.shape == (8, 2)
    o = ob.Oracle(data, 96, 54, 1, cam)
    s1, _ = o.render_frame(0.6591631, 0.910802)
    # constant textures equal to a material colour c must reproduce the untextured render with albedo c^2.2
    flat = type("D", (), {})()
    flat.__dict__.update(data.__dict__)
    flat.albedo_textures = np.full((2, 8, 16, 3), 128, np.uint8)
    s_flat, _ = ob.Oracle(flat, 96, 54, 1, cam).render_frame(0.6591631, 0.910802)
    plain = type("D", (), {})()
    plain.__dict__.update(data.__dict__)
    plain.albedo_textures = None
    plain.materials = data.materials.copy()
    c = np.float32(np.float64(np.float32(128) / np.float32(255)) ** np.float64(np.float32(2.2)))
    for m in (2, 3):
        plain.materials[m, 0:3] = c
        plain.materials[m, 12] = -1
    s_plain, _ = ob.Oracle(plain, 96, 54, 1, cam).render_frame(0.6591631, 0.910802)
    np.testing.assert_allclose(s_flat, s_plain, rtol=2e-6, atol=1e-7)
    assert np.abs(s1 - s_flat).max() > 0.01          # the random texture really modulates the image
    assert np.isfinite(s1).all()
