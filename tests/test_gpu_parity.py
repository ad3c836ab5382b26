"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Hit ids, t, u, v and the
per-ray visit counters are bit-exact; radiance is within abs 1e-5 + rel 1e-4 per channel (the oracle
and the kernels share every floating-point rule, so in practice the sums are bit-identical too)."""
import numpy as np
import pytest

from conftest import seeded_rays

pytestmark = pytest.mark.gpu

RX1, RY1 = 0.6591631174087524, 0.9108020067214966      # frame-1 randomVector
RX2, RY2 = 0.13842908, 0.1292837                        # frame-2 randomVector

# kernel variants that lost every measurement are compiled only with `make EXPERIMENTS=1` (include/crt.h, crt_set_option); the
# default library refuses their options, and the cases below that use them run only against an experiments build
class _Experiments:
    def __bool__(self):                                 # asked inside the tests, i.e. after conftest's `built` fixture
        import caitlynrenderer_amd
        return caitlynrenderer_amd.has_experiments()


EXPERIMENTS = _Experiments()
EXPERIMENTAL_OPTIONS = {"oversubscribe", "waves_per_workgroup", "trace_occupancy"}


def _assert_hits_equal(got, want):
    assert np.array_equal(got["tri"], want["tri"])
    for k in ("t", "u", "v"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k


@pytest.fixture(scope="module")
def scenes(cr, ob, cornell, cornell_data, tess8, tess40):
    out = {}
    for name, data in (("cornell", cornell_data), ("tess8", tess8[1]), ("tess40", tess40[1])):
        out[name] = (cr.Scene(data, 256, 144, 3), ob.Oracle(data, 256, 144, 3, cornell[1]), data)
    yield out
    for s, _, _ in out.values():
        s.close()


def test_native_library_is_the_one_running(cr):
    import torch
    from caitlynrenderer_amd import _lib
    assert torch.cuda.is_available() and _lib.lib().crt_device_count() >= 1
    assert "caitlynrenderer_amd/libcrt.so" in open("/proc/self/maps").read()


@pytest.mark.parametrize("name", ["cornell", "tess8", "tess40"])
def test_closest_hit_bit_exact(cr, ob, cornell, tess8, tess40, scenes, name):
    scene, orc, data = scenes[name]
    mesh = {"cornell": cornell[0], "tess8": tess8[0], "tess40": tess40[0]}[name]
    rays = np.concatenate([seeded_rays(mesh, 50000, 3, cr.RAY_DT), orc.primary_rays(RX1, RY1, jitter=True).astype(cr.RAY_DT)])
    got, gst = scene.trace(rays, cr.CRT_TRACE_CLOSEST, stats=True)
    want, wst = orc.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, stats=True, threads=8)
    _assert_hits_equal(got, want)
    assert np.array_equal(gst["nodes"], wst["nodes"]) and np.array_equal(gst["tris"], wst["tris"])
    assert (got["tri"] >= 0).mean() > 0.3


@pytest.mark.parametrize("name", ["cornell", "tess8", "tess40"])
def test_any_hit_bit_exact(cr, ob, cornell, tess8, tess40, scenes, name):
    scene, orc, data = scenes[name]
    mesh = {"cornell": cornell[0], "tess8": tess8[0], "tess40": tess40[0]}[name]
    rays = seeded_rays(mesh, 60000, 4, cr.RAY_DT)
    rays["tmax"] = np.random.default_rng(9).random(len(rays)).astype(np.float32) * 6
    got, gst = scene.trace(rays, cr.CRT_TRACE_ANY, stats=True)
    want, wst = orc.trace(rays, ob.BVH8, ob.ANY, stats=True, threads=8)
    assert np.array_equal(got["tri"] >= 0, want["tri"] >= 0)
    assert np.array_equal(gst["nodes"], wst["nodes"]) and np.array_equal(gst["tris"], wst["tris"])
    # property: occluded <=> closest hit nearer than tmax
    c = scene.trace(np.array(rays, copy=True), cr.CRT_TRACE_CLOSEST)
    far = rays.copy(); far["tmax"] = np.float32(1e9)
    c = scene.trace(far, cr.CRT_TRACE_CLOSEST)
    assert np.array_equal(got["tri"] >= 0, (c["tri"] >= 0) & (c["t"] < rays["tmax"]))


@pytest.mark.parametrize("name,n", [("tess8", 3000), ("tess40", 400)])
def test_hip_walk_equals_a_numpy_brute_force_that_shares_no_code_with_the_oracle(cr, tess8, tess40, scenes, name, n):
    """The HIP CWBVH walk against a third implementation (vectorised numpy over all triangles, tests/conftest.py): (id, t, u, v)
    bit for bit, and occlusion consistent with it.  The oracle is not involved."""
    from conftest import numpy_brute_force
    scene, _, _ = scenes[name]
    mesh = {"tess8": tess8[0], "tess40": tess40[0]}[name]
    rays = seeded_rays(mesh, n, 29, cr.RAY_DT)
    rays["tmax"][::5] = np.float32(2.5)
    tri, t, u, v = numpy_brute_force(mesh, rays)
    got = scene.trace(rays, cr.CRT_TRACE_CLOSEST)
    hit = tri >= 0
    assert hit.sum() > n // 2 and np.array_equal(got["tri"], tri)
    for a, b in ((got["t"], t), (got["u"], u), (got["v"], v)):
        assert np.array_equal(a[hit].view(np.uint32), b[hit].view(np.uint32))
    occ = scene.trace(rays, cr.CRT_TRACE_ANY)["tri"] >= 0
    assert np.array_equal(occ, hit)


def test_tile_processing_order_changes_nothing_but_the_time(cr, ob, cornell, tess8):
    """The order in which the tiles of the frame are handed to the GPU (centre-out, then by measured cost, dealt to the eight
    XCD groups in snake order) is a scheduling decision: sums, ray counts and visit counters are the oracle's for every
    frame before, during and after the measurement, at every tile size and for a shard; a camera change measures again."""
    mesh, data = tess8
    cam0 = cornell[1]
    cam1 = cr.Camera((1.0, 4.5, 9.0), (2.5, 2.0, 2.0), 55.0)
    for W, H, tile, shard in ((333, 217, 64, None), (333, 217, 16, None), (640, 360, 64, (2, 3)), (200, 150, 40, None)):
        results = {}
        for adaptive in (1, 0):
            scene = cr.Scene(data, W, H, 2)
            scene.set_option("adaptive_tiles", adaptive)
            scene.set_shard(shard[0] if shard else 0, shard[1] if shard else 1, tile)
            scene.set_option("count_visits", 1)
            rnd = cr.Rnd()
            sums, stats = [], []
            for cam in (cam0, cam1):
                scene.update(cam)
                for _ in range(4):                                  # frame 1 measures, a later one adopts the order
                    scene.render_frame(rnd.randf2(), rnd.randf2())
                    st = scene.frame_stats()
                    stats.append((st["closest_rays"], st["any_rays"], st["nodes_closest"] + st["nodes_any"], st["tris_closest"] + st["tris_any"]))
                    scene.sync()
                sums.append(scene.read_sum().copy())
            results[adaptive] = (sums, stats)
            scene.close()
        for a, b in zip(results[1][0], results[0][0]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (W, H, tile, shard)
        assert results[1][1] == results[0][1]
        if shard is None:                                           # and both are the oracle's
            rnd = cr.Rnd()
            ref = np.zeros((H, W, 3), np.float32)
            k = 0
            for cam in (cam0, cam1):
                orc = ob.Oracle(data, W, H, 2, cam)
                for _ in range(4):
                    _, cnt = orc.render_frame(rnd.randf2(), rnd.randf2(), ref, threads=8)
                    assert results[1][1][k] == (cnt[0], cnt[1], cnt[2], cnt[3]), (W, H, tile, k)
                    k += 1
            assert np.array_equal(results[1][0][1].view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("name", ["cornell", "tess8", "tess8_mat", "textured"])
def test_render_frames_equals_the_same_frames_one_by_one(cr, ob, cornell, tess8, textured, name):
    """crt_render_frames: n frames in ceil(n / 8) launches on a one-segment path walked in place; on paths of several segments up to
    8 frames share each segment's launch (per-sample path state, the samples' radiance added in frame order by a last kernel); frame
    by frame otherwise.  The samples of a launch's first segment sit on the waves of a workgroup (wave_samples: what a frame this
    small picks by itself, added in sample order through LDS) or follow one another in one wave (wave_samples = 0).  The sum buffer
    is the same bit for bit every way, and equal to the oracle's.  Sharded frames, the BVH2 walks, the new materials, textures, the
    shadow queue, the bounce pools and 2 - 4 segments are all in the loop."""
    from caitlynrenderer_amd.meshgen import tessellated_cornell, with_disney_materials
    mesh_c, cam = cornell
    if name == "cornell":
        data = cr.SceneData.build(mesh_c, cam)
    elif name == "tess8":
        data = tess8[1]
    elif name == "tess8_mat":
        data = cr.SceneData.build(tessellated_cornell(with_disney_materials(mesh_c), 8), cam)
    else:
        data = textured[1]
    W, H = 233, 131
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(19)]                  # 19 = 8 + 8 + 3
    for depth, opts, shard in ((1, {}, None), (1, {"lanes_per_ray": 1}, None), (1, {"accel": 1}, None), (1, {"inplace_shadow": 0}, None),
                               (2, {}, None), (1, {"waves_per_workgroup": 4}, (1, 3)), (4, {}, None), (3, {"inplace_shadow": 2}, (0, 2)),
                               (3, {"bounce_refill": 1}, None), (2, {"accel": 2}, None), (4, {"inplace_shadow": 2, "shadow_pool": 256, "shadow_refill_min": 8}, None),
                               (3, {"inplace_shadow": 0}, None), (3, {"bounce_refill": 1, "inplace_shadow": 2, "refill_pool": 128}, None), (1, {"inplace_shadow": 2}, None),
                               (4, {"bounce_refill": 1, "inplace_shadow": 2, "persistent": 1, "refill_min": 16, "shadow_refill_min": 16}, None), (3, {"inplace_shadow": 2, "persistent": 1, "shadow_refill_min": 8}, (1, 2)),
                               (1, {"wave_samples": 0}, None), (3, {"wave_samples": 0}, (1, 2)), (1, {"wave_samples": 1, "accel": 1}, (2, 3)),
                               (4, {"wave_samples": 1, "inplace_shadow": 2}, None), (1, {"wide_first": 1}, None), (2, {"wide_first": 0, "wave_samples": 0}, (0, 2)),
                               (1, {"wave_samples": 3}, None), (1, {"wave_samples": 3, "wide_first": 1}, (1, 2)), (3, {"wave_samples": 3}, None),
                               (3, {"ray_bins": 1}, None), (4, {"ray_bins": 1}, (1, 2)), (2, {"ray_bins": 3, "wave_samples": 0}, None)):
        if name in ("tess8_mat",) and opts.get("accel"):
            continue                                                          # the BVH2 frame mode is the Lambert-only shader
        if not EXPERIMENTS and EXPERIMENTAL_OPTIONS & set(opts):
            continue                                                          # variants of the CRT_EXPERIMENTS build only
        a, b = cr.Scene(data, W, H, depth), cr.Scene(data, W, H, depth)
        for s in (a, b):
            s.update(cam)
            for k, v in opts.items():
                s.set_option(k, v)
            if shard:
                s.set_shard(shard[0], shard[1], 64)
        for r in rvs:
            a.render_frame(*r)
        b.render_frames(rvs)
        sa, sb = a.read_sum(), b.read_sum()
        assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32)), (name, depth, opts)
        # the stats of a batched launch are its totals: 3 samples in the last launch of 19 = 8 + 8 + 3 where batching applies
        st_a, st_b = a.frame_stats(), b.frame_stats()
        # launches share samples unless the FIRST segment defers its shadow rays (the batched builds walk them in place) or the frame
        # goes through the BVH2 walks (rendered one by one)
        batched = opts.get("inplace_shadow", 1) != 0 and opts.get("accel", 0) == 0
        if depth == 1:
            assert st_b["closest_rays"] == (3 if batched else 1) * st_a["closest_rays"]
        elif batched:
            # several segments: 4 samples per launch (19 = 4 x 4 + 3), each with its own path state; radiance added in frame order afterwards
            assert 2.5 * st_a["closest_rays"] < st_b["closest_rays"] < 3.5 * st_a["closest_rays"]
        else:
            assert st_b["closest_rays"] == st_a["closest_rays"]
        assert st_b["stack_overflows"] == 0
        if not shard and name != "textured":
            orc = ob.Oracle(data, W, H, depth, cam)
            ref = np.zeros((H, W, 3), np.float32)
            o_accel, o_tie = (ob.BVH2, ob.TIE_FIRST_VISITED) if opts.get("accel") == 1 else (ob.BVH8, ob.TIE_LOWEST_ID)
            for r in rvs:
                orc.render_frame(r[0], r[1], ref, accel=o_accel, tie=o_tie, threads=8)
            assert np.array_equal(sb.view(np.uint32), ref.view(np.uint32)), (name, depth, opts)
        a.close(); b.close()
    # a counting frame is never batched, and n = 0 is a no-op
    c = cr.Scene(data, W, H, 1)
    c.update(cam)
    c.render_frames([])
    c.set_option("count_visits", 1)
    c.render_frames(rvs[:3])
    assert c.frame_stats()["closest_rays"] == W * H and c.frame_stats()["nodes_closest"] > 0      # one frame's worth: the last of the three
    c.close()


def test_launch_form_follows_the_size_of_the_launch(cr, cornell, cornell_data, tess8):
    """wave_samples = 2 (default).  A launch of 4 or 8 samples on a tree of 64+ nodes puts four samples of a 4x4 pixel quadrant in the
    lanes of one wave (form 2).  Otherwise (6 samples here; any launch on the 32-triangle box) it runs its samples side by side on the
    waves of a workgroup (form 1) when it has too few 64-pixel batches to fill the GPU's wave slots with their samples one after the
    other — a shard, a small frame — and keeps them in one wave (form 0) when it is bound by throughput (the whole 1080p frame: 32,400
    batches on 5,120 slots); that choice is made from the measured tile costs once there are any.  0, 1 and 3 force a form; a single
    frame has nothing to choose."""
    _, data = tess8
    _, cam = cornell
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(8)]
    def forms(W, H, shard=None, option=None, n=8, d=None):
        s = cr.Scene(d if d is not None else data, W, H, 1)
        s.update(cam)
        if option is not None:
            s.set_option("wave_samples", option)
        if shard:
            s.set_shard(shard[0], shard[1], 16)
        out = []
        for _ in range(4):                      # the first launch measures the tiles; a later one has adopted the costs
            s.render_frames(rvs[:n])
            out.append(s.debug_launch_form())
        s.render_frame(*rvs[0])
        out.append(s.debug_launch_form())
        s.close()
        return out
    assert forms(1920, 1080) == [2, 2, 2, 2, 0]
    assert forms(1920, 1080, (3, 8)) == [2, 2, 2, 2, 0]
    # 6 samples go as 4 + 2: the last launch of the call is the pair, in the form the launch's size asks for
    assert forms(1920, 1080, n=6) == [0, 0, 0, 0, 0]
    assert forms(1920, 1080, (3, 8), n=6) == [1, 1, 1, 1, 0]
    assert forms(320, 200, n=3) == [1, 1, 1, 1, 0]
    assert forms(1920, 1080, d=cornell_data) == [0, 0, 0, 0, 0]              # a 3-node tree: more, shorter waves cost more than they save
    assert forms(1920, 1080, (3, 8), d=cornell_data) == [1, 1, 1, 1, 0]
    assert forms(1920, 1080, (3, 8), option=0) == [0, 0, 0, 0, 0]
    assert forms(1920, 1080, option=1) == [1, 1, 1, 1, 0]
    assert forms(1920, 1080, option=3, n=6) == [0, 0, 0, 0, 0]               # 3 = lanes or nothing: the trailing pair runs one after the other


def test_edge_cases(cr, ob, scenes, cornell):
    scene, orc, _ = scenes["cornell"]
    assert len(scene.trace(np.zeros(0, cr.RAY_DT))) == 0                      # empty input
    rays = np.zeros(70, cr.RAY_DT)                                            # ragged: not a multiple of 64
    dirs = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    for i in range(70):
        rays[i]["o"] = (2.78, 2.75, 2.8) if i % 2 == 0 else (0.0, 2.75, 2.8)  # inside / exactly on a wall plane
        rays[i]["d"] = dirs[i % 6]                                            # zero components: 0*inf = NaN slabs
        rays[i]["tmax"] = [1e9, 0.0, -1.0, 1e-30, 3.0][i % 5]                 # includes tmax <= 0
    got = scene.trace(rays, cr.CRT_TRACE_CLOSEST)
    _assert_hits_equal(got, orc.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID))
    got = scene.trace(rays, cr.CRT_TRACE_ANY)
    assert np.array_equal(got["tri"] >= 0, orc.trace(rays, ob.BVH8, ob.ANY)["tri"] >= 0)


def test_zero_components_and_non_finite_rays(cr, ob, scenes):
    scene, orc, _ = scenes["tess40"]
    rng = np.random.default_rng(2)
    rays = np.zeros(4096, cr.RAY_DT)
    rays["o"] = (0.3 + 4.9 * rng.random((4096, 3))).astype(np.float32)
    dirs = np.array([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (0, 0.6, 0.8), (0.6, 0, -0.8),
                     (-0.0, 1, 0), (0, 0, 0)], np.float32)
    rays["d"] = dirs[np.arange(4096) % len(dirs)]
    rays["tmax"] = np.float32(1e9)
    rays["o"][5] = (np.nan, 1, 1); rays["o"][6] = (1, -np.inf, 1); rays["d"][7] = (np.nan, 1, 0)
    got, gst = scene.trace(rays, cr.CRT_TRACE_CLOSEST, stats=True)
    want, wst = orc.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, stats=True, threads=8)
    _assert_hits_equal(got, want)
    assert np.array_equal(gst["nodes"], wst["nodes"]) and np.array_equal(gst["tris"], wst["tris"])
    assert gst["nodes"].max() < 400                      # nothing degenerates into a full-tree walk
    got = scene.trace(rays, cr.CRT_TRACE_ANY)
    assert np.array_equal(got["tri"] >= 0, orc.trace(rays, ob.BVH8, ob.ANY, threads=8)["tri"] >= 0)


def test_trace_pool_sizes_and_refill_thresholds_agree(cr, ob, scenes):
    """crt_trace walks its rays in pools of 64 (default), 128 or 256 per wave, refilled once `refill_min` lanes are idle: hits and
    per-ray visit counters are those of the oracle whatever the pool, the threshold or the (ragged) ray count."""
    scene, orc, _ = scenes["tess40"]
    rng = np.random.default_rng(11)
    for n in (1, 63, 64, 65, 129, 1000, 4096 + 257, 3 * 4096 + 70):
        rays = np.zeros(n, cr.RAY_DT)
        rays["o"] = (0.3 + 4.9 * rng.random((n, 3))).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
        rays["tmax"] = np.float32(1e9)
        want, wst = orc.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, stats=True, threads=8)
        occ = orc.trace(rays, ob.BVH8, ob.ANY, threads=8)["tri"] >= 0
        for pool, refill in ((64, 8), (128, 8), (128, 64), (256, 1), (256, 40)):
            scene.set_option("trace_pool", pool)
            scene.set_option("refill_min", refill)
            got, gst = scene.trace(rays, cr.CRT_TRACE_CLOSEST, stats=True)
            _assert_hits_equal(got, want)
            assert np.array_equal(gst["nodes"], wst["nodes"]) and np.array_equal(gst["tris"], wst["tris"]), (n, pool, refill)
            assert np.array_equal(scene.trace(rays, cr.CRT_TRACE_ANY)["tri"] >= 0, occ), (n, pool, refill)
    scene.set_option("trace_pool", 64)
    scene.set_option("refill_min", 8)
    with pytest.raises(cr.CrtError):
        scene.set_option("trace_pool", 100)


def test_exact_ties_resolve_to_lowest_original_id(cr, ob):
    """Two coincident triangles: every hit is an exact tie; the lower original id must win."""
    v = np.array([[0, 0, 0], [4, 0, 0], [0, 4, 0], [0, 0, 0], [4, 0, 0], [0, 4, 0], [0, 0, 1], [4, 0, 1], [0, 4, 1]], np.float32)
    t = np.zeros((3, 12), np.int32)
    t[:, :3] = [[6, 7, 8], [3, 4, 5], [0, 1, 2]]        # ids 1 and 2 coincide
    mats = np.zeros((1, 16), np.float32); mats[0, 4:8] = -1; mats[0, 12:16] = -1
    mesh = cr.Mesh(v, np.zeros((0, 3)), np.zeros((0, 2)), t, mats, np.zeros((0, 18)))
    cam = cr.Camera((1, 1, -5), (1, 1, 0), 40)
    data = cr.SceneData.build(mesh, cam)
    scene = cr.Scene(data, 32, 32, 1)
    rays = np.zeros(64, cr.RAY_DT)
    rays["o"] = [(0.5 + 0.03 * i, 0.5, -3) for i in range(64)]
    rays["d"] = (0, 0, 1)
    rays["tmax"] = 1e9
    got = scene.trace(rays)
    assert (got["tri"] == 1).all() and (got["t"] == 3.0).all()
    _assert_hits_equal(got, ob.Oracle(data, 32, 32, 1, cam).trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID))
    scene.close()


@pytest.mark.parametrize("name,depth", [("cornell", 1), ("cornell", 3), ("tess8", 3), ("tess40", 4)])
def test_radiance_matches_oracle(cr, ob, cornell, scenes, name, depth):
    _, _, data = scenes[name]
    W, H = 200, 120                                   # not a multiple of the 64-pixel tile: ragged tiles
    scene = cr.Scene(data, W, H, depth)
    orc = ob.Oracle(data, W, H, depth, cornell[1])
    rnd = cr.Rnd()
    ref = np.zeros((H, W, 3), np.float32)
    n_closest = n_any = 0
    for frame in range(4):                            # frames 1..4 with the host RNG's randomVectors
        rx, ry = rnd.randf2(), rnd.randf2()
        scene.render_frame(rx, ry)
        _, cnt = orc.render_frame(rx, ry, ref, threads=8)
        st = scene.frame_stats()
        assert st["closest_rays"] == cnt[0] and st["any_rays"] == cnt[1]
        out = scene.read_sum()
        err = np.abs(out - ref)
        assert (err <= 1e-5 + 1e-4 * np.abs(ref)).all(), (frame, float(err.max()))      # the stated tolerance
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (frame, float(err.max()))   # and in fact bit-exact
    assert ref.max() > 0.5
    # resolve: Shader/output.fs byte for byte (the gamma is pinned: a byte is the number of thresholds its tone-mapped value has reached)
    img = scene.resolve(0.25)
    want = ob.resolve(ref, 0.25)
    assert np.array_equal(img, want) and len(np.unique(img[..., :3])) > 100
    # and the pinned gamma is the formula's: against powf evaluated here, at most one 8-bit step apart and equal on nearly every byte
    c = ref.astype(np.float32) * np.float32(0.25)
    lum = np.float32(0.3) * c[..., 0] + np.float32(0.6) * c[..., 1] + np.float32(0.1) * c[..., 2]
    x = c * (np.float32(1.0) / (np.float32(1.0) + lum / np.float32(2.0)))[..., None]
    formula = (np.clip(np.power(x, np.float32(1.0 / 2.2), dtype=np.float32), 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    diff = np.abs(formula.astype(np.int32) - img[..., :3].astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3
    # reset clears the sum (Scene.h:1160-1172)
    scene.reset()
    assert not scene.read_sum().any()
    scene.close()


def test_full_resolution_cornell_frame(cr, ob, cornell, cornell_data):
    """Config 2: 1920x1080, 1 spp, primary + shadow; hit-id image and radiance against the oracle."""
    W, H = 1920, 1080
    scene = cr.Scene(cornell_data, W, H, 1)
    orc = ob.Oracle(cornell_data, W, H, 1, cornell[1])
    rays = orc.primary_rays(RX1, RY1, jitter=True)
    got = scene.trace(rays.astype(cr.RAY_DT))
    want = orc.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, threads=8)
    _assert_hits_equal(got, want)
    scene.render_frame(RX1, RY1)
    ref, cnt = orc.render_frame(RX1, RY1, threads=8)
    out = scene.read_sum()
    err = np.abs(out - ref)
    assert (err <= 1e-5 + 1e-4 * np.abs(ref)).all(), float(err.max())
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), float(err.max())
    st = scene.frame_stats()
    assert st["closest_rays"] == W * H == cnt[0] and st["any_rays"] == cnt[1]
    # size-independent properties: determinism and additivity of the running sum
    scene.render_frame(RX1, RY1)
    out2 = scene.read_sum()
    assert np.array_equal(out2.view(np.uint32), (out + out).view(np.uint32))
    scene.close()


def test_full_resolution_mesh_frame_with_visit_counters(cr, ob, cornell, tess40):
    """1920x1080 on the 48k-triangle mesh, 3 segments: radiance bit-exact, ray counts and the summed
    node/triangle visit counters (the roofline's algorithmic-bytes inputs) equal to the oracle's."""
    W, H = 1920, 1080
    scene = cr.Scene(tess40[1], W, H, 3)
    orc = ob.Oracle(tess40[1], W, H, 3, cornell[1])
    scene.set_option("count_visits", 1)
    scene.render_frame(RX1, RY1)
    ref, cnt = orc.render_frame(RX1, RY1, threads=16)
    st = scene.frame_stats()
    assert (st["closest_rays"], st["any_rays"]) == (cnt[0], cnt[1])
    assert st["nodes_closest"] + st["nodes_any"] == cnt[2] and st["tris_closest"] + st["tris_any"] == cnt[3]
    out = scene.read_sum()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), float(np.abs(out - ref).max())
    scene.close()


@pytest.mark.parametrize("inplace", [1, 0])
def test_async_frames_and_timing_options_do_not_change_results(cr, scenes, inplace):
    """Frames queued back to back (counter banks alternate, the kernels clear the next frame's bank) with the
    event spans reduced / accumulated give the same sum and ray counts as synchronous frames with full timing.
    inplace = 1: one launch per path segment; inplace = 0: the segments' shadow rays deferred to one more launch per frame."""
    _, _, data = scenes["tess8"]
    W, H, depth, frames = 320, 200, 3, 7
    per_frame = depth + (0 if inplace else 1)      # launches that carry events (the fold kernel is not a traversal launch)
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(frames)]
    a = cr.Scene(data, W, H, depth)
    a.set_option("inplace_shadow", inplace)
    a.render_frame(*rvs[0])
    st0 = a.frame_stats()                      # launches carry no events unless asked to: nothing timed, nothing counted
    assert st0["n_trace_launches"] == 0 and st0["ms_trace_closest"] == 0 and st0["ms_total"] == 0
    a.reset()
    a.set_option("timing", 2)
    for rx, ry in rvs:
        a.render_frame(rx, ry)
    want, want_st = a.read_sum(), a.frame_stats()
    assert want_st["n_trace_launches"] == per_frame and want_st["ms_trace_closest"] > 0 and (want_st["ms_trace_any"] > 0) == (not inplace)
    a.close()
    for timing in (0, 1, 2):
        b = cr.Scene(data, W, H, depth)
        b.set_option("inplace_shadow", inplace)
        b.set_option("timing", timing)
        b.set_option("timing_accumulate", frames * depth * 2)
        for rx, ry in rvs:
            b.render_frame(rx, ry, sync=False)
        b.sync()
        st = b.frame_stats()
        assert np.array_equal(b.read_sum().view(np.uint32), want.view(np.uint32)), timing
        assert (st["closest_rays"], st["any_rays"]) == (want_st["closest_rays"], want_st["any_rays"])
        assert st["n_trace_launches"] == (0, frames * depth, frames * per_frame)[timing]
        assert (st["ms_trace_closest"] > 0) == (timing > 0) and (st["ms_trace_any"] > 0) == (timing > 1 and not inplace)
        b.set_option("timing_accumulate", 0)
        b.set_option("timing", 2)
        b.render_frame(*rvs[0])
        assert b.frame_stats()["n_trace_launches"] == per_frame
        b.close()


@pytest.mark.parametrize("name", ["cornell", "tess8"])
def test_scheduling_and_loop_variants_give_identical_frames(cr, scenes, name):
    """Grid shape (one wave / four waves per workgroup, one batch per workgroup / persistent static schedule), the
    vote ratio of the traversal loop and the bounce-ray pools only reorder work: sums and ray counts stay identical."""
    _, _, data = scenes[name]
    W, H, depth = 328, 200, 3
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(3)]

    def run(options):
        s = cr.Scene(data, W, H, depth)
        for k, v in options.items():
            s.set_option(k, v)
        for rx, ry in rvs:
            s.render_frame(rx, ry)
        out, st = s.read_sum(), s.frame_stats()
        assert st["stack_overflows"] == 0
        s.close()
        return out, (st["closest_rays"], st["any_rays"])

    want, want_counts = run({})
    assert want.max() > 0.1
    if EXPERIMENTS:
        # waves_per_workgroup is a per-scene setting: a 4-wave scene next to a 1-wave scene leaves the latter alone
        other = cr.Scene(data, 64, 64, 1)
        other.set_option("waves_per_workgroup", 4)
        got, counts = run({})
        other.close()
        assert counts == want_counts and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    else:
        # the default build carries none of the variants that lost their measurements: asking for one is an error, their defaults are accepted
        other = cr.Scene(data, 64, 64, 1)
        for k, v in (("waves_per_workgroup", 4), ("oversubscribe", 1)):
            with pytest.raises(cr.CrtError, match="CRT_EXPERIMENTS"):
                other.set_option(k, v)
        for k, v in (("waves_per_workgroup", 1), ("oversubscribe", 0), ("bounce_refill", 0)):
            other.set_option(k, v)
        other.close()
    for options in ({"waves_per_workgroup": 4}, {"waves_per_workgroup": 2}, {"lanes_per_ray": 1}, {"lanes_per_ray": 8, "tri_min": 1},
                    {"waves_per_workgroup": 2, "oversubscribe": 2}, {"waves_per_workgroup": 2, "tri_min": 0},
                    {"accel": 1, "waves_per_workgroup": 4, "_ref": {"accel": 1}}, {"oversubscribe": 1}, {"oversubscribe": 3, "waves_per_workgroup": 4},
                    {"trace_occupancy": 2, "oversubscribe": 1}, {"tri_min": 0}, {"tri_min": 1}, {"tri_min": 5}, {"ray_bins": 1}, {"ray_bins": 2}, {"ray_bins": 4}, {"ray_bins": 5},
                    {"ray_bins": 3, "lanes_per_ray": 1}, {"ray_bins": 1, "inplace_shadow": 0}, {"ray_bins": 4, "inplace_shadow": 2},
                    {"bounce_refill": 1}, {"bounce_refill": 1, "refill_min": 1}, {"bounce_refill": 1, "refill_min": 40, "refill_pool": 128},
                    {"bounce_refill": 1, "refill_min": 65, "refill_pool": 64}, {"bounce_refill": 1, "lanes_per_ray": 1, "refill_pool": 512}, {"inplace_shadow": 0},
                    {"inplace_shadow": 2}, {"inplace_shadow": 2, "shadow_pool": 256, "shadow_refill_min": 8}, {"inplace_shadow": 2, "shadow_pool": 128, "shadow_refill_min": 64, "lanes_per_ray": 1},
                    {"inplace_shadow": 0, "shadow_pool": 512, "shadow_refill_min": 1}, {"inplace_shadow": 2, "bounce_refill": 1},
                    {"inplace_shadow": 2, "persistent": 1, "shadow_refill_min": 16}, {"shadow_waves": 8}, {"shadow_waves": 1, "shadow_pool": 64}, {"sort_shadow": 1}, {"sort_shadow": 1, "persistent": 0}, {"inplace_shadow": 0, "sort_shadow": 1, "shadow_pool": 64}, {"bounce_refill": 1, "persistent": 1, "refill_min": 24}, {"bounce_refill": 1, "persistent": 1, "inplace_shadow": 0, "refill_min": 1, "shadow_refill_min": 64},
                    {"bounce_refill": 1, "persistent": 1, "inplace_shadow": 2, "lanes_per_ray": 1, "shadow_refill_min": 8},
                    {"inplace_shadow": 0, "bounce_refill": 1}, {"inplace_shadow": 0, "tri_min": 0}, {"inplace_shadow": 0, "oversubscribe": 2}):
        if not EXPERIMENTS and EXPERIMENTAL_OPTIONS & set(options):
            continue
        ref_opts = options.pop("_ref", None)             # a variant compared with another baseline (the BVH2 frame mode)
        w, wc = (want, want_counts) if ref_opts is None else run(ref_opts)
        got, counts = run(options)
        assert counts == wc, options
        assert np.array_equal(got.view(np.uint32), w.view(np.uint32)), options


@pytest.mark.parametrize("name,depth", [("tess8", 3), ("tess40", 4)])
def test_deferred_shadow_rays_keep_sums_and_counters(cr, ob, scenes, disney_scenes, name, depth):
    """`inplace_shadow` 2 / 0: the NEE shadow rays of the bounce segments / of every segment wait in the frame's NEE queue and are walked by ONE
    any-hit launch behind the last segment (lock-step batches of 64, or pools with lane refill), an occluded ray clears its contribution
    slot, and a last kernel adds every path's slots in segment order.  Same rays, same walks, the same additions in the same order: radiance,
    ray counts and visit totals equal the in-place form's and the oracle's; what changes is how full the waves of the any-hit walks are."""
    _, _, data = scenes[name]
    W, H = 320, 184
    for d in (data, disney_scenes[0][name]):
        orc = ob.Oracle(d, W, H, depth, disney_scenes[1])
        rnd = cr.Rnd()
        rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(2)]
        ref = np.zeros((H, W, 3), np.float32)
        cnt = None
        for rx, ry in rvs:
            _, cnt = orc.render_frame(rx, ry, ref, threads=8)
        steps = {}
        for key, opts in (("default", {}), ("inplace", {"inplace_shadow": 1}), ("bounce", {"inplace_shadow": 2, "persistent": 0, "shadow_pool": 64, "shadow_refill_min": 65, "sort_shadow": 0}), ("sorted", {"sort_shadow": 1}), ("all", {"inplace_shadow": 0, "persistent": 0, "sort_shadow": 0}),
                          ("bounce_pool", {"inplace_shadow": 2, "shadow_pool": 256, "shadow_refill_min": 8, "persistent": 0}), ("bounce_one_lane", {"inplace_shadow": 2, "lanes_per_ray": 1}),
                          ("refill", {"bounce_refill": 1, "persistent": 0, "inplace_shadow": 1}), ("wavefront", {"bounce_refill": 1, "inplace_shadow": 2, "refill_pool": 128, "shadow_pool": 128, "shadow_refill_min": 16, "persistent": 0}),
                          ("persistent", {"bounce_refill": 1, "inplace_shadow": 2, "persistent": 1, "refill_min": 16, "shadow_refill_min": 16})):
            s = cr.Scene(d, W, H, depth)
            for k, v in opts.items():
                s.set_option(k, v)
            s.set_option("count_visits", 1)
            for rx, ry in rvs:
                s.render_frame(rx, ry)
            st = s.frame_stats()
            assert (st["closest_rays"], st["any_rays"]) == (cnt[0], cnt[1]) and st["stack_overflows"] == 0, key
            assert st["nodes_closest"] + st["nodes_any"] == cnt[2] and st["tris_closest"] + st["tris_any"] == cnt[3], key
            assert np.array_equal(s.read_sum().view(np.uint32), ref.view(np.uint32)), key
            steps[key] = (st["wave_steps_any_nodes"], st["wave_steps_closest_nodes"], st["nodes_closest"], st["tris_closest"], st["nodes_any"], st["tris_any"])
            s.close()
        assert len({v[2:] for v in steps.values()}) == 1                                  # the same visits, block by block
        assert steps["bounce"][0] < steps["inplace"][0] and steps["all"][0] < steps["inplace"][0]      # fuller waves: fewer wave-level any-hit node steps
        assert steps["sorted"][0] < steps["default"][0]                                   # rays that start together walk together: fewer still once sorted by origin cell
        assert steps["refill"][1] < steps["inplace"][1]                                   # and fewer closest-hit ones through the refilled pools
        if name == "tess40":                                                              # (a persistent grid spreads a small frame's rays over more waves than it has batches)
            assert steps["persistent"][1] < steps["inplace"][1] and steps["persistent"][0] < steps["inplace"][0]


@pytest.mark.parametrize("T", [16, 24])
def test_tile_shards_compose_to_the_full_frame(cr, ob, cornell, cornell_data, T):
    """world=3 shards rendered on one GPU: the union of the ranks' pixels is bit-identical to world=1
    (power-of-two tiles index pixels with shifts, other multiples of 8 with integer divisions)."""
    from caitlynrenderer_amd import tiles
    W, H = 200, 120
    full = cr.Scene(cornell_data, W, H, 3)
    full.set_shard(0, 1, T)
    full.render_frame(RX1, RY1)
    want = full.read_sum()
    acc = np.zeros_like(want)
    frame = np.zeros_like(want)
    for r in range(3):
        s = cr.Scene(cornell_data, W, H, 3)
        s.set_shard(r, 3, T)
        s.render_frame(RX1, RY1)
        part = s.read_sum()
        assert not (acc != 0)[part != 0].any()        # shards are disjoint
        acc += part
        nt, tile, nf = s.packed_info()
        assert nt == len(tiles.local_tiles(W, H, T, r, 3)) and tile == T
        tiles.untile_into(frame, s.read_packed(), tiles.local_tiles(W, H, T, r, 3), T)
        s.close()
    assert np.array_equal(acc.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(frame.view(np.uint32), want.view(np.uint32))
    full.close()


@pytest.mark.parametrize("name,depth,k", [("tess8", 3, 2), ("tess40", 4, 2), ("tess8", 2, 3), ("cornell", 1, 2)])
def test_streams_option_splits_the_frame_over_streams_of_one_gpu(cr, ob, cornell, scenes, name, depth, k):
    """Option "streams": k tile shards of the frame on k streams of the one GPU (a multi-segment frame is a chain of dependent launches;
    another shard's launches fill their tails), the scene buffers shared.  Same sums as the oracle and the same statistics, frame by
    frame and batched; the packed buffer of the caller's shard is assembled from the streams' parts; a shard set with crt_set_shard can
    be split the same way (its own tiles dealt to the streams); crt_set_shard and streams = 1 undo the split."""
    _, _, data = scenes[name]
    _, cam = cornell
    W, H = 250, 140
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(6)]
    orc = ob.Oracle(data, W, H, depth, cam)
    ref = np.zeros((H, W, 3), np.float32)
    cnt_last = None
    for r in rvs:
        _, cnt_last = orc.render_frame(r[0], r[1], ref, threads=8)
    sc = cr.Scene(data, W, H, depth)
    sc.set_option("streams", k)
    assert sc.devices()["devices"] == [0] * k
    sc.render_frame(*rvs[0])
    sc.render_frames(rvs[1:5])
    sc.set_option("count_visits", 1)
    sc.render_frame(*rvs[5])
    st = sc.frame_stats()
    assert (st["closest_rays"], st["any_rays"]) == (cnt_last[0], cnt_last[1]) and st["stack_overflows"] == 0
    assert st["nodes_closest"] + st["nodes_any"] == cnt_last[2] and st["tris_closest"] + st["tris_any"] == cnt_last[3]
    out = sc.read_sum()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), float(np.abs(out - ref).max())
    plain = cr.Scene(data, W, H, depth)
    for r in rvs:
        plain.render_frame(*r)
    assert sc.packed_info() == plain.packed_info()
    assert np.array_equal(sc.read_packed().view(np.uint32), plain.read_packed().view(np.uint32))      # tile for tile, in the shard's own order
    plain.close()
    with pytest.raises(cr.CrtError):
        sc.set_option("streams", 5)
    sc.set_option("streams", 0)                     # the library's own pick: 3 for a few-node scene, 2 for a multi-segment path, else 1
    assert len(sc.devices()["devices"]) == (3 if name == "cornell" else 2 if depth > 1 else 1)
    # back to one stream: a plain scene again (the sum restarts, as with every re-sharding)
    sc.set_option("streams", 1)
    assert sc.devices()["devices"] == [0]
    sc.render_frame(*rvs[0])
    one = np.zeros((H, W, 3), np.float32)
    orc.render_frame(rvs[0][0], rvs[0][1], one, threads=8)
    assert np.array_equal(sc.read_sum().view(np.uint32), one.view(np.uint32))
    # a caller that shards the frame itself takes the split away, and may split its own shard again
    sc.set_option("streams", 2)
    sc.set_shard(1, 3, 16)
    assert sc.devices()["devices"] == [0]
    ref_shard = cr.Scene(data, W, H, depth)
    ref_shard.set_shard(1, 3, 16)
    sc.set_option("streams", k)
    assert sc.devices()["devices"] == [0] * k
    for s_ in (sc, ref_shard):
        s_.render_frame(*rvs[0])
        s_.render_frames(rvs[1:4])
    assert sc.packed_info() == ref_shard.packed_info()
    assert np.array_equal(sc.read_packed().view(np.uint32), ref_shard.read_packed().view(np.uint32)) and ref_shard.read_packed().max() > 0
    assert np.array_equal(sc.read_sum().view(np.uint32), ref_shard.read_sum().view(np.uint32))      # the shard's tiles in the frame, nothing else
    import torch
    n_floats = sc.packed_info()[2]
    d_buf = torch.zeros(n_floats, dtype=torch.float32, device="cuda")
    sc.copy_packed_device(d_buf.data_ptr(), n_floats)
    assert np.array_equal(d_buf.cpu().numpy().view(np.uint32), ref_shard.read_packed().view(np.uint32))
    sc.set_option("streams", 1)                     # the caller's shard again, on one stream
    sc.render_frame(*rvs[0])
    ref_shard.reset(); ref_shard.render_frame(*rvs[0])
    assert sc.packed_info() == ref_shard.packed_info()
    assert np.array_equal(sc.read_packed().view(np.uint32), ref_shard.read_packed().view(np.uint32))
    sc.close(); ref_shard.close()


def test_more_streams_or_devices_than_tiles(cr, ob, cornell, cornell_data):
    """A frame of one or two tiles on up to four streams (or virtual devices): some shards hold no tile at all and still take part in
    every call; sums, packed buffer and resolve are those of the plain scene."""
    _, cam = cornell
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(5)]
    for (W, H) in ((8, 8), (17, 9), (1, 1), (40, 8)):
        for depth in (1, 3):
            orc = ob.Oracle(cornell_data, W, H, depth, cam)
            ref = np.zeros((H, W, 3), np.float32)
            for r in rvs:
                orc.render_frame(r[0], r[1], ref, threads=2)
            plain = cr.Scene(cornell_data, W, H, depth)
            plain.render_frame(*rvs[0]); plain.render_frames(rvs[1:5])
            for k in (2, 3, 4):
                sc = cr.Scene(cornell_data, W, H, depth)
                if k == 4:
                    sc.set_devices([0] * k, 16)
                else:
                    sc.set_option("streams", k)
                sc.render_frame(*rvs[0]); sc.render_frames(rvs[1:5])
                assert np.array_equal(sc.read_sum().view(np.uint32), ref.view(np.uint32)), (W, H, depth, k)
                assert np.array_equal(sc.resolve(0.2), plain.resolve(0.2))
                if k < 4:
                    assert np.array_equal(sc.read_packed().view(np.uint32), plain.read_packed().view(np.uint32))
                sc.close()
            plain.close()


@pytest.mark.parametrize("name,depth,n_dev", [("cornell", 1, 2), ("tess8", 3, 4), ("tess8", 2, 3)])
def test_one_handle_several_devices_gather_inside_the_c_abi(cr, ob, cornell, scenes, name, depth, n_dev):
    """crt_set_devices: ONE scene handle and one frame loop, as the reference has them (main.cpp:262-300), rendering on several
    devices; crt_read_sum / crt_resolve gather the other devices' tiles to the first and return the whole frame.  On
    this one-GPU box the devices are virtual — the same GPU listed n times: separate replicas, streams, shards and gather buffers,
    copies instead of RCCL — which exercises everything but the transport.  Frames one by one and batched, camera change, reset,
    options and statistics: bit-identical to the oracle and to the single-device scene."""
    from caitlynrenderer_amd import _lib
    _, _, data = scenes[name]
    _, cam = cornell
    W, H = 250, 140                                   # ragged tiles
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(7)]
    orc = ob.Oracle(data, W, H, depth, cam)
    ref = np.zeros((H, W, 3), np.float32)
    cnt_last = None
    for r in rvs:
        _, cnt_last = orc.render_frame(r[0], r[1], ref, threads=8)
    multi = cr.Scene(data, W, H, depth)
    multi.set_devices([0] * n_dev, 16)
    info = multi.devices()
    assert info["devices"] == [0] * n_dev and info["transport"] == "copy"
    multi.set_option("count_visits", 1)
    for r in rvs[:3]:
        multi.render_frame(*r)
    multi.set_option("count_visits", 0)
    multi.render_frames(rvs[3:6])
    multi.set_option("count_visits", 1)
    multi.render_frame(*rvs[6])
    st = multi.frame_stats()
    assert (st["closest_rays"], st["any_rays"]) == (cnt_last[0], cnt_last[1]) and st["stack_overflows"] == 0      # summed over the devices
    assert st["nodes_closest"] + st["nodes_any"] == cnt_last[2] and st["tris_closest"] + st["tris_any"] == cnt_last[3]
    out = multi.read_sum()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), float(np.abs(out - ref).max())
    assert multi.devices()["last_gather_ms"] > 0
    single = cr.Scene(data, W, H, depth)
    for r in rvs:
        single.render_frame(*r)
    assert np.array_equal(multi.resolve(1.0 / 7).view(np.uint32), single.resolve(1.0 / 7).view(np.uint32))
    # a shard of its own is refused; reset and a new camera reach every device; back to one device
    with pytest.raises(cr.CrtError) as e:
        multi.set_shard(0, 2, 16)
    assert e.value.code == _lib.CRT_ERR_INVALID
    cam2 = cr.Camera((2.0, 2.5, 9.0), (2.8, 2.7, 0.0), 50.0)
    for s in (multi, single):
        s.reset()
        s.update(cam2)
        s.render_frame(*rvs[0])
    assert np.array_equal(multi.read_sum().view(np.uint32), single.read_sum().view(np.uint32)) and single.read_sum().max() > 0
    multi.set_devices([0], 16)
    assert multi.devices()["devices"] == [0]
    multi.reset(); single.reset()
    for s in (multi, single):
        s.render_frame(*rvs[1])
    assert np.array_equal(multi.read_sum().view(np.uint32), single.read_sum().view(np.uint32))
    with pytest.raises(cr.CrtError):
        multi.set_devices([0, 99], 16)
    with pytest.raises(cr.CrtError):
        multi.set_devices([], 16)
    multi.close(); single.close()


def test_two_real_devices_rccl_and_copy_transports_agree(cr, ob, cornell, scenes):
    """Only on a machine with >= 2 GPUs (skipped on the one-GPU test box): ONE handle on two distinct devices — replication over the
    fabric, tile sharding, and the gather inside crt_read_sum through RCCL send / recv and through hipMemcpyPeerAsync — each
    bit-identical to the single-device sum; the calling thread's current HIP device is the scene's own after every entry point."""
    import torch
    from caitlynrenderer_amd import _lib
    if _lib.lib().crt_device_count() < 2:
        pytest.skip("needs two GPUs")
    _, _, data = scenes["tess8"]
    W, H, depth = 250, 140, 3
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(5)]
    single = cr.Scene(data, W, H, depth)
    single.render_frame(*rvs[0]); single.render_frames(rvs[1:])
    want = single.read_sum()
    single.close()
    for transport in (0, 1):
        multi = cr.Scene(data, W, H, depth)
        multi.set_devices([0, 1], 16)
        if transport == 0 and multi.devices()["transport"] != "rccl":
            multi.close()
            continue                                                  # librccl.so not loadable here: the copies are all there is
        multi.set_option("gather_transport", transport)
        multi.render_frame(*rvs[0]); assert torch.cuda.current_device() == 0
        multi.render_frames(rvs[1:]); assert torch.cuda.current_device() == 0
        got = multi.read_sum(); assert torch.cuda.current_device() == 0
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), transport
        multi.reset(); assert torch.cuda.current_device() == 0
        multi.close()


def test_device_resident_trace_and_torch_interop(cr, ob, cornell, scenes):
    """crt_trace_device on torch-owned HBM buffers (the bench path): same bits as the host-buffer entry."""
    import torch
    scene, orc, _ = scenes["tess40"]
    rays = orc.primary_rays(RX1, RY1, jitter=True).astype(cr.RAY_DT)
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1, 32)).cuda()
    d_hits = torch.empty((len(rays), 16), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    scene.set_option("timing", 2)
    scene.trace_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), cr.CRT_TRACE_CLOSEST)
    got = d_hits.cpu().numpy().view(cr.HIT_DT).ravel()
    assert scene.frame_stats()["ms_trace_closest"] > 0
    scene.set_option("timing", 0)
    _assert_hits_equal(got, scene.trace(rays))


def _shard_worker(rank, world, port, W, H, T, out_dir):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    import caitlynrenderer_amd as cr
    from caitlynrenderer_amd import tiles
    import __graft_entry__ as g
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)                       # both ranks share the one GPU of the test box
    mesh, cam = g._cornell()
    scene = cr.Scene(cr.SceneData.build(mesh, cam), W, H, 3)
    scene.set_shard(rank, world, T)
    if rank == 1:
        scene.set_option("streams", 2)             # this rank renders its shard as two tile sets on two streams: the same packed buffer
    rnd = cr.Rnd()
    for _ in range(2):
        scene.render_frame(rnd.randf2(), rnd.randf2())
    packed = torch.from_numpy(scene.read_packed())
    frame = tiles.gather_frame(packed, W, H, T, rank, world)
    np.save(os.path.join(out_dir, f"frame{rank}.npy"), frame)
    scene.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_render_shards_and_gather(cr, ob, cornell, cornell_data, tmp_path):
    """One process per rank (the bench.py --gpus N structure, gloo instead of RCCL because the test box has
    one GPU): each rank renders its Morton-dealt tiles with the HIP path, the packed tiles are all-gathered
    and un-tiled; the result is bit-identical to the oracle's full frame."""
    import socket
    import torch.multiprocessing as mp
    W, H, T = 200, 120, 16
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_shard_worker, args=(2, port, W, H, T, str(tmp_path)), nprocs=2, join=True)
    orc = ob.Oracle(cornell_data, W, H, 3, cornell[1])
    rnd = cr.Rnd()
    ref = np.zeros((H, W, 3), np.float32)
    for _ in range(2):
        orc.render_frame(rnd.randf2(), rnd.randf2(), ref, threads=8)
    for r in range(2):
        got = np.load(tmp_path / f"frame{r}.npy")
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_cpp_host_scene_matches_python_path(cr, ob, cornell, tmp_path):
    """examples/render_obj.cpp (crt::Scene: Read_Object -> build_bvh -> gpu_data -> Render x N, the
    reference's call sequence) gives the same running sum as the Python mirror and the oracle, and a PPM."""
    import os
    import subprocess
    from conftest import ROOT, write_obj
    from caitlynrenderer_amd.image import read_ppm
    mesh, _ = cornell
    obj = str(tmp_path / "cornell.obj")
    write_obj(mesh, obj)
    W, H, frames, depth = 160, 96, 3, 3
    out = subprocess.run([os.path.join(ROOT, "examples", "render_obj"), obj, str(tmp_path / "o.ppm"), str(W), str(H), str(frames),
                          str(depth), str(tmp_path / "sum.f32")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    got = np.fromfile(tmp_path / "sum.f32", np.float32).reshape(H, W, 3)
    # the same frames as two tile shards on two streams (crt::Scene::set_option): the same sum
    out2 = subprocess.run([os.path.join(ROOT, "examples", "render_obj"), obj, str(tmp_path / "o2.ppm"), str(W), str(H), str(frames),
                           str(depth), str(tmp_path / "sum2.f32")], capture_output=True, text=True, env=dict(os.environ, RENDER_OBJ_STREAMS="2"))
    assert out2.returncode == 0, out2.stderr
    assert np.array_equal(np.fromfile(tmp_path / "sum2.f32", np.float32).view(np.uint32), got.reshape(-1).view(np.uint32))
    cam = cr.Camera((-2.755610, 2.745992, 7.58545), (-2.755610, 2.745992, 6.58545), 40.0)   # Scene.h:468
    data = cr.SceneData.from_obj(obj, cam)
    orc = ob.Oracle(data, W, H, depth, cam)
    rnd = cr.Rnd()
    ref = np.zeros((H, W, 3), np.float32)
    for _ in range(frames):
        orc.render_frame(rnd.randf2(), rnd.randf2(), ref, threads=8)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # crt::Scene::RenderFrames (all frames in one crt_render_frames call; depth 1 so that they share launches): same sums, and a PNG
    out = subprocess.run([os.path.join(ROOT, "examples", "render_obj"), obj, str(tmp_path / "o.png"), str(W), str(H), "11", "1",
                          str(tmp_path / "sum1.f32")], capture_output=True, text=True, env=dict(os.environ, RENDER_OBJ_BATCHED="1"))
    assert out.returncode == 0, out.stderr
    orc1 = ob.Oracle(data, W, H, 1, cam)
    rnd1, ref1 = cr.Rnd(), np.zeros((H, W, 3), np.float32)
    for _ in range(11):
        orc1.render_frame(rnd1.randf2(), rnd1.randf2(), ref1, threads=8)
    assert np.array_equal(np.fromfile(tmp_path / "sum1.f32", np.float32).reshape(H, W, 3).view(np.uint32), ref1.view(np.uint32))
    from caitlynrenderer_amd import host
    png = host.decode_image(open(tmp_path / "o.png", "rb").read())
    assert png.shape == (H, W, 3) and png.std() > 5
    # the same with three path segments: the frames share each segment's launch, the sums are those of `frames` single frames above
    out = subprocess.run([os.path.join(ROOT, "examples", "render_obj"), obj, str(tmp_path / "o3.ppm"), str(W), str(H), str(frames),
                          str(depth), str(tmp_path / "sum3.f32")], capture_output=True, text=True, env=dict(os.environ, RENDER_OBJ_BATCHED="1"))
    assert out.returncode == 0, out.stderr
    assert np.array_equal(np.fromfile(tmp_path / "sum3.f32", np.float32).reshape(H, W, 3).view(np.uint32), ref.view(np.uint32))
    scene = cr.Scene(data, W, H, depth)
    for _ in range(frames):
        scene.Render()
    assert np.array_equal(scene.read_sum().view(np.uint32), ref.view(np.uint32)) and scene.frame_count == frames
    img = read_ppm(str(tmp_path / "o.ppm"))
    assert np.abs(img.astype(np.int32) - scene.resolve()[:, :, :3].astype(np.int32)).max() == 0
    assert img.max() > 100
    scene.close()


@pytest.fixture(scope="module")
def mesh1m(cr, cornell):
    """BASELINE configs[2..4] geometry: the 1,004,672-triangle tessellated Cornell box, built once."""
    from caitlynrenderer_amd.meshgen import tessellated_cornell
    mesh, cam = cornell
    big = tessellated_cornell(mesh, 183)
    assert big.triangles.shape[0] == 1004672 and big.vertices.shape[0] == 507844
    return big, cr.SceneData.build(big, cam), cam


def test_million_triangle_mesh_full_frame(cr, ob, mesh1m):
    """BASELINE config 3 geometry (1,004,672 triangles, 4 frames = 4 spp) at 1920x1080, 2 segments: radiance
    bit-identical to the oracle after every frame, ray counts and visit counters equal; plus the
    size-independent property that any-hit agrees with closest-hit on every shadow ray of the last frame."""
    big, data, cam = mesh1m
    dev = cr.CWBVH().convert_arrays(data.bvh, data.triangles.shape[0], device=True)    # the device converter at full size
    assert np.array_equal(dev.nodes, data.bvh8) and np.array_equal(dev.tri_slots, data.bvh8_tri_slots)
    W, H = 1920, 1080
    scene = cr.Scene(data, W, H, 2)
    info = scene.bvh_info()
    assert info["n_tris8"] == data.triangles.shape[0] >= 1004672 and info["max_depth8"] <= 16
    orc = ob.Oracle(data, W, H, 2, cam)
    scene.set_option("count_visits", 1)
    rnd = cr.Rnd()
    ref = np.zeros((H, W, 3), np.float32)
    for frame in range(4):
        rx, ry = rnd.randf2(), rnd.randf2()
        scene.render_frame(rx, ry)
        _, cnt = orc.render_frame(rx, ry, ref, threads=16)
        st = scene.frame_stats()
        assert (st["closest_rays"], st["any_rays"]) == (cnt[0], cnt[1]) and st["stack_overflows"] == 0
        assert st["nodes_closest"] + st["nodes_any"] == cnt[2] and st["tris_closest"] + st["tris_any"] == cnt[3]
        # uniform node steps (first-segment walks: the node through the scalar cache when every enabled lane asks for the same one):
        # a part of the same visits — the totals above already equal the oracle's — and a large one on primary rays (every ray
        # starts at the root, and the waves whose rays do not share the octant are the few that straddle the image centre)
        assert 0.9 * W * H < st["nodes_closest_uniform"] < st["nodes_closest"] and 0 < st["nodes_any_uniform"] < st["nodes_any"]
        out = scene.read_sum()
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (frame, float(np.abs(out - ref).max()))
    # the same frame with every segment's shadow rays deferred (inplace_shadow = 0): same sum, and the NEE queue feeds the
    # any-hit / closest-hit consistency check below
    scene.set_option("inplace_shadow", 0)
    scene.reset()
    scene.render_frame(rx, ry)
    last = np.zeros((H, W, 3), np.float32)
    orc.render_frame(rx, ry, last, threads=16)
    assert np.array_equal(scene.read_sum().view(np.uint32), last.view(np.uint32))
    shadow = scene.debug_read_queue(2, 1)
    assert len(shadow) > 100000
    occ = scene.trace(shadow, cr.CRT_TRACE_ANY)["tri"] >= 0
    far = shadow.copy(); far["tmax"] = np.float32(1e9)
    c = scene.trace(far, cr.CRT_TRACE_CLOSEST)
    assert np.array_equal(occ, (c["tri"] >= 0) & (c["t"] < shadow["tmax"]))
    scene.close()


def test_config4_million_triangles_four_segments(cr, ob, mesh1m):
    """BASELINE configs[3] at full size: the 1,004,672-triangle CWBVH, 4 path segments (incoherent bounce rays), 1920x1080.
    Two frames; after each the accumulated radiance is bit-identical to the oracle's, the closest / any-hit ray counts and the
    node / triangle visit totals are equal, and no stack push was dropped.  (The reference has no BSDF but Lambert —
    path_trace.fs:274-310 — so this is its integrator run one segment longer than the shader's hard-coded 3, :867.)  Then three
    frames through one crt_render_frames call, as the bench's step renders them."""
    _, data, cam = mesh1m
    W, H, depth = 1920, 1080, 4
    scene = cr.Scene(data, W, H, depth)
    orc = ob.Oracle(data, W, H, depth, cam)
    scene.set_option("count_visits", 1)
    rnd = cr.Rnd()
    ref = np.zeros((H, W, 3), np.float32)
    rvs = []
    for frame in range(2):
        rx, ry = rnd.randf2(), rnd.randf2()
        rvs.append((rx, ry))
        scene.render_frame(rx, ry)
        _, cnt = orc.render_frame(rx, ry, ref, threads=16)
        st = scene.frame_stats()
        assert (st["closest_rays"], st["any_rays"]) == (cnt[0], cnt[1]) and cnt[0] > 4_500_000 and st["stack_overflows"] == 0
        assert st["nodes_closest"] + st["nodes_any"] == cnt[2] and st["tris_closest"] + st["tris_any"] == cnt[3]
        # wave-level step counters of the counting kernels: every block execution offers 64 lane slots
        for v, w in (("nodes_closest", "wave_steps_closest_nodes"), ("tris_closest", "wave_steps_closest_tris"),
                     ("nodes_any", "wave_steps_any_nodes"), ("tris_any", "wave_steps_any_tris")):
            assert 0 < st[w] and st[w] <= st[v] <= 64 * st[w], (v, st[v], st[w])
        out = scene.read_sum()
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (frame, float(np.abs(out - ref).max()))
    scene.close()
    # the bench's form of this workload: the frames of a step share each segment's launch (crt_render_frames, per-sample path
    # state, radiance added in frame order at the end) — three frames at once, first segment in either launch form
    rvs.append((rnd.randf2(), rnd.randf2()))
    orc.render_frame(rvs[2][0], rvs[2][1], ref, threads=16)
    for ws in (0, 1):
        batched = cr.Scene(data, W, H, depth)
        batched.set_option("wave_samples", ws)
        batched.render_frames(rvs)
        assert np.array_equal(batched.read_sum().view(np.uint32), ref.view(np.uint32)), ws
        assert batched.frame_stats()["stack_overflows"] == 0
        batched.close()
    # configs[3]'s "sorting" half: the bounce rays regrouped by (direction octant, origin cell) between the segments (option
    # ray_bins; from the second frame on the bins have their places).  Same sums, same ray counts and visit totals — only the
    # number of wave-level traversal steps drops.
    steps = {}
    for bins in (0, 1, 4, 5):                       # 4 / 5: the append finds the rays of a bin by ranking through LDS
        s = cr.Scene(data, W, H, depth)
        s.set_option("ray_bins", bins)
        for r in rvs[:2]:
            s.render_frame(*r)
        s.set_option("count_visits", 1)
        s.render_frame(*rvs[2])
        st = s.frame_stats()
        assert np.array_equal(s.read_sum().view(np.uint32), ref.view(np.uint32)), bins
        steps[bins] = (st["closest_rays"], st["any_rays"], st["nodes_closest"], st["tris_closest"], st["nodes_any"], st["tris_any"], st["wave_steps_closest_nodes"])
        s.close()
    assert steps[0][:6] == steps[1][:6] == steps[4][:6] == steps[5][:6] and steps[1][6] < 0.95 * steps[0][6] and steps[4][6] < 0.95 * steps[0][6]


def test_config5_4k_frame_of_the_million_triangle_mesh(cr, ob, mesh1m):
    """BASELINE configs[4]'s workload: the 3840x2160 frame (2,040 tiles of 64x64) of the 1,004,672-triangle mesh.  The whole
    frame on one rank and ranks 0, 3 and 7 of 8 are each bit-identical to the oracle on their pixels (ray counts included for the
    full frame); the eight shards are disjoint and cover the frame, so composing them is the single-GPU frame.  Rank 5 of 8 then
    renders five frames in one crt_render_frames launch over 16x16 tiles, as the bench's ranks do, in both launch forms."""
    from caitlynrenderer_amd import tiles
    _, data, cam = mesh1m
    W, H = 3840, 2160
    orc = ob.Oracle(data, W, H, 1, cam)
    ref, cnt = orc.render_frame(RX1, RY1, threads=16)
    full = cr.Scene(data, W, H, 1)
    full.render_frame(RX1, RY1)
    st = full.frame_stats()
    assert (st["closest_rays"], st["any_rays"]) == (cnt[0], cnt[1]) and cnt[0] == W * H and st["stack_overflows"] == 0
    assert np.array_equal(full.read_sum().view(np.uint32), ref.view(np.uint32))
    full.close()
    covered = np.zeros((H, W), np.int32)
    for r in range(8):
        for tx, ty in tiles.local_tiles(W, H, 64, r, 8):
            covered[ty * 64:(ty + 1) * 64, tx * 64:(tx + 1) * 64] += 1
    assert (covered == 1).all()
    for r in (0, 3, 7):
        shard = cr.Scene(data, W, H, 1)
        shard.set_shard(r, 8, 64)
        shard.render_frame(RX1, RY1)
        part = shard.read_sum()
        mine = np.zeros((H, W), bool)
        for tx, ty in tiles.local_tiles(W, H, 64, r, 8):
            mine[ty * 64:(ty + 1) * 64, tx * 64:(tx + 1) * 64] = True
        assert np.array_equal(part[mine].view(np.uint32), ref[mine].view(np.uint32)) and not part[~mine].any(), r
        assert shard.packed_info()[0] == len(tiles.local_tiles(W, H, 64, r, 8)) == 255
        shard.close()
    # what a rank of the N = 8 bench does: its 16x16 tiles, a step's frames through crt_render_frames — 5 frames in one launch, the
    # samples one after the other in each wave and side by side on the waves of a workgroup (3 waves, two passes)
    rnd = cr.Rnd()
    rvs = [(RX1, RY1)] + [(rnd.randf2(), rnd.randf2()) for _ in range(4)]
    for rv in rvs[1:]:
        orc.render_frame(rv[0], rv[1], ref, threads=16)
    mine = np.zeros((H, W), bool)
    for tx, ty in tiles.local_tiles(W, H, 16, 5, 8):
        mine[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
    for ws in (0, 1):
        shard = cr.Scene(data, W, H, 1)
        shard.set_option("wave_samples", ws)
        shard.set_shard(5, 8, 16)
        shard.render_frames(rvs)
        part = shard.read_sum()
        assert np.array_equal(part[mine].view(np.uint32), ref[mine].view(np.uint32)) and not part[~mine].any(), ws
        assert shard.frame_stats()["closest_rays"] == 5 * int(mine.sum()) and shard.frame_stats()["stack_overflows"] == 0
        shard.close()


def test_config4_with_mirror_and_disney_materials_at_full_size(cr, ob, mesh1m, cornell):
    """BASELINE configs[3] as worded — "4-bounce Disney BSDF" — on the mesh the bench's `incoherent_disney` block runs: the
    1,004,672-triangle tree with the mirror tall box and the GGX / Disney-diffuse short box and floor, 4 path segments, 1920x1080.
    One frame against the oracle bit for bit (sum, ray counts, visit totals), then a step as the bench renders it (4 frames per
    crt_render_frames call) against the same frames one by one."""
    import copy
    from caitlynrenderer_amd.meshgen import tessellated_cornell, with_disney_materials
    _, data0, cam = mesh1m
    mesh = tessellated_cornell(with_disney_materials(cornell[0]), 183)   # same geometry and tree: only the material table and the material ids change
    data = copy.copy(data0)
    data.materials = mesh.materials
    data.triangles = data0.triangles.copy()
    data.triangles[:, 3] = mesh.triangles[data0.tri_orig_ids, 3]
    W, H, depth = 1920, 1080, 4
    scene = cr.Scene(data, W, H, depth)
    scene.set_option("count_visits", 1)
    orc = ob.Oracle(data, W, H, depth, cam)
    ref, cnt = orc.render_frame(RX1, RY1, threads=16)
    scene.render_frame(RX1, RY1)
    st = scene.frame_stats()
    assert (st["closest_rays"], st["any_rays"]) == (cnt[0], cnt[1]) and cnt[0] > 4_000_000 and st["stack_overflows"] == 0
    assert st["nodes_closest"] + st["nodes_any"] == cnt[2] and st["tris_closest"] + st["tris_any"] == cnt[3]
    one = scene.read_sum()
    assert np.array_equal(one.view(np.uint32), ref.view(np.uint32)), float(np.abs(one - ref).max())
    assert 0 < st["closest_hits"] < st["closest_rays"]
    scene.set_option("count_visits", 0)
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(4)]
    for r in rvs:
        scene.render_frame(*r)
    want = scene.read_sum()
    scene.close()
    batched = cr.Scene(data, W, H, depth)
    batched.render_frame(RX1, RY1)
    batched.render_frames(rvs)
    got = batched.read_sum()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    batched.close()
    # ... and the batched step against the ORACLE on the same five frames (a band of rows through the boxes; the whole frame five times over
    # would take the CPU minutes): what the step adds per frame — not only its first frame — keeps the oracle's bits
    y0 = (H // 2 - 16) // 8 * 8
    rows = np.zeros((H, W, 3), np.float32)
    for rx, ry in [(RX1, RY1)] + rvs:
        orc.render_rows(rx, ry, y0, y0 + 32, rows)
    assert rows[y0:y0 + 32].max() > 0
    assert np.array_equal(got[y0:y0 + 32].view(np.uint32), rows[y0:y0 + 32].view(np.uint32))


def test_wide_first_segment_build_on_the_whole_million_triangle_frame(cr, ob, mesh1m):
    """The 6-waves-per-SIMD (80 VGPR) build of the first-segment kernel is what the bench's headline launch runs (a throughput-bound
    launch picks it by itself): the unsharded 1920x1080 frame of the 1 M mesh through crt_render_frames, forced to either build,
    against the oracle bit for bit."""
    _, data, cam = mesh1m
    W, H = 1920, 1080
    orc = ob.Oracle(data, W, H, 1, cam)
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(4)]
    ref = np.zeros((H, W, 3), np.float32)
    for r in rvs:
        orc.render_frame(r[0], r[1], ref, threads=16)
    for wide in (1, 0, 2):
        s = cr.Scene(data, W, H, 1)
        s.set_option("wide_first", wide)
        s.set_option("wave_samples", 0)
        s.render_frames(rvs)
        assert np.array_equal(s.read_sum().view(np.uint32), ref.view(np.uint32)), wide
        assert s.frame_stats()["stack_overflows"] == 0
        s.close()


def _bench_step(scene, rvs):
    """bench.py run_block's step(): crt_render_frames (async) of a step's frames, default options."""
    scene.render_frames(rvs, sync=False)


def test_scene_beyond_two_to_the_23_triangles_builds_on_the_device(cr, cornell):
    """16,785,122 triangles (n = 748): past the 2^23 triangles at which a FlatNode's float links stop being exact (FlatNode.h:34-40, uploaded
    as RGBA32F: Scene.h:1057-1062) and past 2^24 triangle slots.  A scene crt_scene_create builds on the device keeps links of 2^24 or more
    as bit patterns (host/flatnode_link.hpp), so it is accepted; the oracle takes FlatNode arrays and stays at the reference's limit, so the
    properties here are its stand-ins: two DIFFERENT trees over the same triangles (binned SAH, Morton LBVH) return the same (id, t, u, v)
    bit for bit on 100,000 rays; the BVH2 walk over the device's own FlatNode array (links above 2^24 in their bit form) returns them too;
    any-hit agrees with closest-hit; a numpy brute force over all 16.8 M triangles agrees on 8 rays; frames render with no stack overflow
    and equal sums on both trees; the host-array builders still refuse the size."""
    from caitlynrenderer_amd.meshgen import tessellated_cornell
    from conftest import numpy_brute_force
    base, cam = cornell
    mesh = tessellated_cornell(base, 748)
    n = mesh.triangles.shape[0]
    assert n == 16785122 and n > 2 ** 24
    with pytest.raises(cr.CrtError):
        cr.SBVH(mesh.triangles, mesh.vertices, builder="lbvh")                      # crt_lbvh_build hands FlatNodes to the host: floats only
    W, H = 960, 540
    rng = np.random.default_rng(11)
    rays = seeded_rays(mesh, 100000, 41, cr.RAY_DT)
    results = {}
    for builder in ("sah", "lbvh"):
        scene = cr.Scene(cr.SceneData.for_device_build(mesh, cam, builder=builder), W, H, 2)
        info = scene.bvh_info()
        assert info["n_tris8"] == n and info["built_on_device"] == 1 and info["n_bvh2_nodes"] == 2 * n - 1 > 2 ** 25 and info["max_depth8"] <= 16
        got = scene.trace(rays, cr.CRT_TRACE_CLOSEST)
        hit = got["tri"] >= 0
        assert hit.sum() > 80000 and got["tri"].max() > 2 ** 23                       # original ids beyond the old cap are hit
        cut = rays.copy()
        cut["tmax"] = (rng.random(len(rays)) * 8).astype(np.float32)
        occ = scene.trace(cut, cr.CRT_TRACE_ANY)["tri"] >= 0
        assert np.array_equal(occ, hit & (got["t"] < cut["tmax"])), builder
        # the shipped shader's BVH2 walk on the device's FlatNode array, lowest-id ties: the same hits
        sub = rays[:20000]
        b2 = scene.trace(sub, cr.CRT_TRACE_BVH2 | cr.CRT_TRACE_TIE_LOWEST_ID)
        _assert_hits_equal(b2, got[:20000])
        scene.render_frame(RX1, RY1)
        scene.render_frame(RX2, RY2)
        img = scene.read_sum()
        assert scene.frame_stats()["stack_overflows"] == 0 and np.isfinite(img).all() and img.max() > 0.5
        results[builder] = (got, img)
        scene.close()
    _assert_hits_equal(results["sah"][0], results["lbvh"][0])
    assert np.array_equal(results["sah"][1].view(np.uint32), results["lbvh"][1].view(np.uint32))      # the frame does not depend on the tree
    tri, t, u, v = numpy_brute_force(mesh, rays[:8])
    got = results["sah"][0][:8]
    assert np.array_equal(got["tri"], tri) and (tri >= 0).sum() >= 6
    for a, b in ((got["t"], t), (got["u"], u), (got["v"], v)):
        assert np.array_equal(a[tri >= 0].view(np.uint32), b[tri >= 0].view(np.uint32))


def test_the_launches_bench_times_equal_the_oracle_at_full_size(cr, ob, mesh1m):
    """What bench.py's timed region runs on BASELINE configs[2], held to the oracle at full size: a step is ONE
    crt_render_frames call of 4 frames with every option at its default — which must come out as launch form 2 (four samples of a 4x4
    pixel quadrant in the lanes of a wave) on the 6-waves-per-SIMD build of the first-segment kernel — on (a) the host-built tree of the
    1,004,672-triangle mesh (the headline block) and (b) the tree the GPU SAH builder makes of the same mesh inside crt_scene_create (the
    `gpu_tree` block; the oracle walks the same builder's tree).  Two steps each — the second runs on the cost-sorted tile order the first
    one measured — the whole 1920x1080 frame against the oracle bit for bit."""
    big, data, cam = mesh1m
    W, H = 1920, 1080
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(8)]
    sah = cr.SceneData.build(big, cam, builder="sah", convert="device")
    for label, make, tree in (("host-built", lambda: cr.Scene(data, W, H, 1), data),
                              ("gpu-built", lambda: cr.Scene(cr.SceneData.for_device_build(big, cam, builder="sah"), W, H, 1), sah)):
        orc = ob.Oracle(tree, W, H, 1, cam)
        ref = np.zeros((H, W, 3), np.float32)
        for r in rvs:
            orc.render_frame(r[0], r[1], ref, threads=16)
        s = make()
        if label == "gpu-built":
            assert s.bvh_info()["n_nodes8"] == sah.bvh8.shape[0] and s.bvh_info()["built_on_device"] == 1
        for k in range(2):
            _bench_step(s, rvs[4 * k:4 * k + 4])
            s.sync()
            info = s.debug_launch_info()
            assert info == {"form": 2, "wide": True, "one_pass": True, "samples": 4, "shards": 1}, (label, k, info)
        out = s.read_sum()
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (label, float(np.abs(out - ref).max()))
        assert s.frame_stats()["stack_overflows"] == 0
        s.close()


def test_the_incoherent_blocks_of_the_bench_equal_the_oracle_at_full_size(cr, ob, mesh1m):
    """bench.py's `incoherent` block (BASELINE configs[3], the reference's Lambert integrator): 4 path segments, a step = one
    crt_render_frames call of 4 frames, option "streams" 0 — the library's pick, two tile shards of the frame on two streams — and the
    two-segment block `d2` (the other reading of "primary + 1 bounce") the same way.  Whole 1920x1080 frames against the oracle."""
    _, data, cam = mesh1m
    W, H = 1920, 1080
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(4)]
    for depth in (4, 2):
        orc = ob.Oracle(data, W, H, depth, cam)
        ref = np.zeros((H, W, 3), np.float32)
        for r in rvs:
            orc.render_frame(r[0], r[1], ref, threads=16)
        s = cr.Scene(data, W, H, depth)
        s.set_option("streams", 0)
        assert len(s.devices()["devices"]) == 2
        _bench_step(s, rvs)
        s.sync()
        info = s.debug_launch_info()
        assert info["form"] == 2 and info["samples"] == 4 and info["shards"] == 2, info
        out = s.read_sum()
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (depth, float(np.abs(out - ref).max()))
        assert s.frame_stats()["stack_overflows"] == 0
        s.close()


def test_scene_larger_than_the_infinity_cache(cr, ob, cornell):
    """The bench's `hbm_resident` workload: the Cornell scene tessellated to 8,112,002 triangles (n = 520) — 390 MB of intersection
    records + 73 MB of nodes, more than the 256 MiB Infinity Cache, so its fetches are HBM fetches — built by the GPU SAH builder
    (16.2 M BVH2 nodes: the largest tree whose float links are still exact).  Parity at that size: 100,000 sampled rays (primary
    rays of frame 1 and random rays inside the box) bit-exact against the oracle on ids, t, u, v; any-hit agrees with closest-hit on
    every one of them; the whole two-segment 1080p frame equals the oracle's (sum, ray counts, visit totals); and the scene crt_scene_create builds by itself from the
    source-order arrays (what the bench block renders) gives the same frame as the one uploaded from the host arrays."""
    from caitlynrenderer_amd.meshgen import tessellated_cornell
    base, cam = cornell
    mesh = tessellated_cornell(base, 520)
    assert mesh.triangles.shape[0] == 8112002
    data = cr.SceneData.build(mesh, cam, builder="sah", convert="device")
    W, H = 1920, 1080
    scene = cr.Scene(data, W, H, 2)
    info = scene.bvh_info()
    assert info["n_tris8"] == 8112002 and 80 * info["n_nodes8"] + 48 * info["n_tris8"] > 256 * 2 ** 20 and info["max_depth8"] <= 16
    orc = ob.Oracle(data, W, H, 2, cam)
    rng = np.random.default_rng(5)
    prim = orc.primary_rays(RX1, RY1, jitter=True)
    rays = np.zeros(100000, cr.RAY_DT)
    pick = rng.choice(len(prim), 50000, replace=False)
    for k in ("o", "d", "tmax"):
        rays[k][:50000] = prim[k][pick]
    rays["o"][50000:] = (0.2 + 5.1 * rng.random((50000, 3))).astype(np.float32)
    dirs = rng.normal(size=(50000, 3)).astype(np.float32)
    rays["d"][50000:] = dirs / np.linalg.norm(dirs, axis=1, keepdims=True).astype(np.float32)
    rays["tmax"][50000:] = np.float32(1e9)
    got, gst = scene.trace(rays, cr.CRT_TRACE_CLOSEST, stats=True)
    want, wst = orc.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, stats=True, threads=16)
    _assert_hits_equal(got, want)
    assert np.array_equal(gst["nodes"], wst["nodes"]) and np.array_equal(gst["tris"], wst["tris"])
    assert (got["tri"] >= 0).sum() > 70000
    # any-hit == closest-hit below tmax, on rays cut short at a random distance
    cut = rays.copy()
    cut["tmax"] = (rng.random(100000) * 8).astype(np.float32)
    occ = scene.trace(cut, cr.CRT_TRACE_ANY)["tri"] >= 0
    assert np.array_equal(occ, (got["tri"] >= 0) & (got["t"] < cut["tmax"]))
    # the WHOLE two-segment frame against the oracle (sum, ray counts, visit totals); then against the device-built scene of the same builder
    scene.set_option("count_visits", 1)
    scene.render_frame(RX1, RY1)
    out = scene.read_sum()
    st = scene.frame_stats()
    ref, cnt = orc.render_frame(RX1, RY1, threads=16)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)) and ref.max() > 0, float(np.abs(out - ref).max())
    assert (st["closest_rays"], st["any_rays"]) == (cnt[0], cnt[1]) and cnt[0] > 2_500_000
    assert st["nodes_closest"] + st["nodes_any"] == cnt[2] and st["tris_closest"] + st["tris_any"] == cnt[3]
    assert st["stack_overflows"] == 0
    scene.set_option("count_visits", 0)
    scene.close()
    dev = cr.Scene(cr.SceneData.for_device_build(mesh, cam, builder="sah"), W, H, 2)
    assert dev.bvh_info()["n_nodes8"] == info["n_nodes8"]
    dev.render_frame(RX1, RY1)
    assert np.array_equal(dev.read_sum().view(np.uint32), out.view(np.uint32))
    dev.close()
    # the launches bench.py's hbm_resident blocks time: the device-built scene, a step = 4 frames through one crt_render_frames call with
    # default options, at one segment (lanes form on the 6-wave build: the whole frame against the oracle) and at four (two streams: 80 rows)
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(4)]
    for depth, shards in ((1, 1), (4, 2)):
        orc_d = ob.Oracle(data, W, H, depth, cam)
        rows = np.zeros((H, W, 3), np.float32)
        y_lo, y_hi = (0, H) if depth == 1 else (504, 584)      # one segment: the whole frame, four frames over; four segments: 80 rows through the boxes
        for r in rvs:
            if depth == 1:
                orc_d.render_frame(r[0], r[1], rows, threads=16)
            else:
                orc_d.render_rows(r[0], r[1], y_lo, y_hi, rows)
        dev = cr.Scene(cr.SceneData.for_device_build(mesh, cam, builder="sah"), W, H, depth)
        dev.set_option("streams", 0)
        _bench_step(dev, rvs)
        dev.sync()
        li = dev.debug_launch_info()
        assert li["form"] == 2 and li["samples"] == 4 and li["shards"] == shards and (li["wide"] or depth > 1), li
        got = dev.read_sum()
        assert np.array_equal(got[y_lo:y_hi].view(np.uint32), rows[y_lo:y_hi].view(np.uint32)) and rows[y_lo:y_hi].max() > 0, depth
        assert dev.frame_stats()["stack_overflows"] == 0
        dev.close()


@pytest.mark.gpu
def test_float_planes_are_built_when_a_frame_first_walks_the_cwbvh(cr, scenes):
    """The float copy of the nodes' planes (uniform node steps) is built by the first CWBVH frame, not by crt_scene_create: a scene that
    rendered through its BVH2 first — alone, and with tile shards on two streams made BEFORE any CWBVH frame (the shard on the same GPU borrows
    the planes its primary then builds) — gives the frames of a scene that walked the CWBVH from the start."""
    data = scenes["tess40"][2]
    W, H, depth = 320, 200, 3
    ref = cr.Scene(data, W, H, depth)
    ref.render_frame(RX1, RY1)
    ref.render_frame(RX2, RY2)
    want = ref.read_sum()
    ref.close()
    for streams in (1, 2):
        s = cr.Scene(data, W, H, depth)
        s.set_option("accel", 1)
        if streams > 1:
            s.set_option("streams", streams)
        s.render_frame(RX1, RY1)                # BVH2 walk: no planes yet
        s.reset()
        s.set_option("accel", 0)
        s.render_frame(RX1, RY1)
        s.render_frame(RX2, RY2)
        assert np.array_equal(s.read_sum().view(np.uint32), want.view(np.uint32)), streams
        s.close()


@pytest.mark.gpu
def test_node_step_histograms_add_up_to_the_wave_step_counters(cr, scenes):
    """crt_debug_step_hist (measurement aid): node steps of the counting kernels by enabled lanes (mode 0) or by distinct (node, octant) keys
    among the enabled lanes (option step_hist_mode 1).  Either way a frame's histogram holds exactly the frame's wave-level node steps, the
    lane-weighted sum of mode 0 is the visit count, and a step never has more distinct nodes than enabled lanes."""
    data = scenes["tess40"][2]
    W, H = 256, 192
    for depth in (1, 3):
        per_mode = []
        for mode in (0, 1):
            scene = cr.Scene(data, W, H, depth)
            scene.set_option("count_visits", 1)
            scene.set_option("lanes_per_ray", 1)          # the histogram covers the one-lane-per-ray loops (a group phase's steps are not in it)
            scene.set_option("step_hist_mode", mode)
            scene.debug_step_hist()                 # start
            scene.render_frame(RX1, RY1)
            st = scene.frame_stats()
            closest, anyh = scene.debug_step_hist()
            scene.debug_step_hist(stop=True)
            scene.close()
            assert int(closest.sum()) == st["wave_steps_closest_nodes"] > 0 and int(anyh.sum()) == st["wave_steps_any_nodes"] > 0, (depth, mode)
            assert closest[0] == 0 and anyh[0] == 0
            k = np.arange(65, dtype=np.uint64)
            if mode == 0:
                assert int((closest * k).sum()) == st["nodes_closest"] and int((anyh * k).sum()) == st["nodes_any"]
            else:
                assert int((closest * k).sum()) <= st["nodes_closest"] and int((anyh * k).sum()) <= st["nodes_any"]
                assert closest[1] > 0                  # the root step of a primary wave is a single node
            per_mode.append((closest, anyh))
        # one segment: the same waves take the same steps in both runs (bounce rays come in the order their producers finished), and the distinct-node
        # count of a step is at most its lane count, so the cumulative distributions are ordered
        if depth == 1:
            for a, b in zip(per_mode[0], per_mode[1]):
                assert a.sum() == b.sum() and np.all(np.cumsum(b) >= np.cumsum(a))


@pytest.fixture(scope="module")
def disney_scenes(cr, cornell):
    """Cornell box with a mirror tall box, a brushed-metal short box and a glossy floor (meshgen.with_disney_materials),
    plain and tessellated; the material model is oracle-defined (no reference code), so parity here is HIP == oracle."""
    from caitlynrenderer_amd.meshgen import tessellated_cornell, with_disney_materials
    mesh, cam = cornell
    base = with_disney_materials(mesh)
    out = {"cornell": cr.SceneData.build(base, cam)}
    for n in (8, 40):
        out[f"tess{n}"] = cr.SceneData.build(tessellated_cornell(base, n), cam)
    return out, cam


@pytest.mark.parametrize("name,depth,inplace", [("cornell", 1, 1), ("cornell", 4, 1), ("cornell", 4, 0), ("tess8", 3, 1), ("tess8", 5, 0),
                                                ("tess40", 4, 1), ("tess40", 4, 0)])
def test_mirror_and_disney_materials_match_the_oracle(cr, ob, disney_scenes, name, depth, inplace):
    """f3: perfect mirror + GGX / Disney-diffuse lobe.  Three frames: accumulated radiance bit-identical to
    the oracle, ray counts and visit totals equal — in place and through the shadow queue (where a Disney path that samples
    below the horizon ends mid-path with its shadow ray still queued)."""
    datas, cam = disney_scenes
    data = datas[name]
    W, H = 320, 180
    scene = cr.Scene(data, W, H, depth)
    scene.set_option("inplace_shadow", inplace)
    scene.set_option("count_visits", 1)
    orc = ob.Oracle(data, W, H, depth, cam)
    rnd = cr.Rnd()
    ref = np.zeros((H, W, 3), np.float32)
    for frame in range(3):
        rx, ry = rnd.randf2(), rnd.randf2()
        scene.render_frame(rx, ry)
        _, cnt = orc.render_frame(rx, ry, ref, threads=8)
        st = scene.frame_stats()
        assert (st["closest_rays"], st["any_rays"]) == (cnt[0], cnt[1]) and st["stack_overflows"] == 0
        assert st["nodes_closest"] + st["nodes_any"] == cnt[2] and st["tris_closest"] + st["tris_any"] == cnt[3]
        out = scene.read_sum()
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (frame, float(np.abs(out - ref).max()))
    assert np.isfinite(ref).all() and ref.max() > 0.5
    scene.close()


def test_special_materials_restrictions_and_options(cr, ob, disney_scenes):
    """The BVH2 frame mode is the shipped (Lambert-only) shader: refused for a scene with Mirror / Disney materials; the
    bounce pools fall back to the lock-step segment; scheduling options leave the image untouched."""
    from caitlynrenderer_amd import _lib
    datas, cam = disney_scenes
    data = datas["tess8"]
    W, H, depth = 200, 120, 4
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(2)]

    def run(options):
        s = cr.Scene(data, W, H, depth)
        for k, v in options.items():
            s.set_option(k, v)
        for rx, ry in rvs:
            s.render_frame(rx, ry)
        out = s.read_sum()
        s.close()
        return out

    want = run({})
    for options in ({"bounce_refill": 1}, {"waves_per_workgroup": 4}, {"waves_per_workgroup": 2}, {"inplace_shadow": 2},
                    {"oversubscribe": 2}, {"tri_min": 0}, {"inplace_shadow": 0, "tri_min": 3}):
        if not EXPERIMENTS and EXPERIMENTAL_OPTIONS & set(options):
            continue
        assert np.array_equal(run(options).view(np.uint32), want.view(np.uint32)), options
    s = cr.Scene(data, W, H, depth)
    with pytest.raises(cr.CrtError) as e:
        s.set_option("accel", 1)
    assert e.value.code == _lib.CRT_ERR_INVALID and "Lambert" in str(e.value)
    s.close()


def test_textured_disney_material(cr, ob, textured):
    """The albedo texture feeds the Disney lobe's base colour exactly as it feeds Lambert's albedo (path_trace.fs:471-483)."""
    mesh, _, cam = textured
    mats = mesh.materials.copy()
    mats[3, 3] = 17.0; mats[3, 8:10] = (0.2, 0.45)          # the textured Khaki material becomes a Disney one
    mats[2, 3] = 1.0                                          # the textured red wall a (tinted) mirror
    m = cr.Mesh(mesh.vertices, mesh.normals, mesh.texcoords, mesh.triangles, mats, mesh.lights, mesh.vertex_min)
    m.albedo_textures = mesh.albedo_textures
    data = cr.SceneData.build(m, cam)
    W, H, depth = 256, 144, 3
    scene = cr.Scene(data, W, H, depth)
    orc = ob.Oracle(data, W, H, depth, cam)
    rnd = cr.Rnd()
    ref = np.zeros((H, W, 3), np.float32)
    for _ in range(2):
        rx, ry = rnd.randf2(), rnd.randf2()
        scene.render_frame(rx, ry)
        orc.render_frame(rx, ry, ref, threads=8)
    assert np.array_equal(scene.read_sum().view(np.uint32), ref.view(np.uint32))
    scene.close()


@pytest.mark.parametrize("name", ["cornell", "tess40"])
def test_bvh2_reference_order_walk_on_device(cr, ob, cornell, tess40, scenes, name):
    """CRT_TRACE_BVH2: the shipped shader's own walk (path_trace.fs:511-819) on the FlatNode array, bit-exact
    against the oracle's restatement of it: hit ids, t/u/v and the per-ray node/triangle counters, under the
    shader's first-visited rule and under the lowest-id rule; and it agrees with the CWBVH walk."""
    scene, orc, data = scenes[name]
    mesh = cornell[0] if name == "cornell" else tess40[0]
    rays = np.concatenate([seeded_rays(mesh, 40000, 13, cr.RAY_DT), orc.primary_rays(RX1, RY1, jitter=True).astype(cr.RAY_DT)])
    got, gst = scene.trace(rays, cr.CRT_TRACE_CLOSEST | cr.CRT_TRACE_BVH2, stats=True)
    want, wst = orc.trace(rays, ob.BVH2, ob.CLOSEST, ob.TIE_FIRST_VISITED, stats=True, threads=8)
    _assert_hits_equal(got, want)
    assert np.array_equal(gst["nodes"], wst["nodes"]) and np.array_equal(gst["tris"], wst["tris"])
    got2 = scene.trace(rays, cr.CRT_TRACE_CLOSEST | cr.CRT_TRACE_BVH2 | cr.CRT_TRACE_TIE_LOWEST_ID)
    _assert_hits_equal(got2, orc.trace(rays, ob.BVH2, ob.CLOSEST, ob.TIE_LOWEST_ID, threads=8))
    _assert_hits_equal(got2, scene.trace(rays, cr.CRT_TRACE_CLOSEST))          # BVH2 == CWBVH on (id, t, u, v)
    ra = rays.copy()
    ra["tmax"] = np.random.default_rng(5).random(len(ra)).astype(np.float32) * 6
    ga, gsa = scene.trace(ra, cr.CRT_TRACE_ANY | cr.CRT_TRACE_BVH2, stats=True)
    wa, wsa = orc.trace(ra, ob.BVH2, ob.ANY, stats=True, threads=8)
    assert np.array_equal(ga["tri"] >= 0, wa["tri"] >= 0)
    assert np.array_equal(gsa["nodes"], wsa["nodes"]) and np.array_equal(gsa["tris"], wsa["tris"])


def test_4k_frame_config5_geometry_on_one_gpu(cr, ob, cornell, cornell_data):
    """BASELINE config 5's 3840x2160 framebuffer (2,040 tiles) rendered by ONE rank and by rank 3 of 8:
    bit-exact against the oracle; the shard holds exactly its Morton-dealt tiles."""
    from caitlynrenderer_amd import tiles
    W, H = 3840, 2160
    orc = ob.Oracle(cornell_data, W, H, 1, cornell[1])
    ref, cnt = orc.render_frame(RX1, RY1, threads=16)
    full = cr.Scene(cornell_data, W, H, 1)
    full.render_frame(RX1, RY1)
    assert np.array_equal(full.read_sum().view(np.uint32), ref.view(np.uint32))
    st = full.frame_stats()
    assert (st["closest_rays"], st["any_rays"]) == (cnt[0], cnt[1]) and cnt[0] == W * H
    full.close()
    shard = cr.Scene(cornell_data, W, H, 1)
    shard.set_shard(3, 8, 64)
    shard.render_frame(RX1, RY1)
    part = shard.read_sum()
    mine = np.zeros((H, W), bool)
    for tx, ty in tiles.local_tiles(W, H, 64, 3, 8):
        mine[ty * 64:(ty + 1) * 64, tx * 64:(tx + 1) * 64] = True
    assert np.array_equal(part[mine].view(np.uint32), ref[mine].view(np.uint32)) and not part[~mine].any()
    assert shard.packed_info()[0] == len(tiles.local_tiles(W, H, 64, 3, 8)) == 255
    shard.close()


def test_textured_albedo_matches_oracle(cr, ob, textured):
    """a7's texture branch on the device: same bilinear/repeat definition and pow(., 2.2) as the oracle.
    Stated tolerance abs 1e-5 + rel 1e-4 (the device and libm double pow may differ in the last place before
    the rounding to float); in practice bit-identical."""
    mesh, data, cam = textured
    W, H = 320, 180
    scene = cr.Scene(data, W, H, 3)
    orc = ob.Oracle(data, W, H, 3, cam)
    rnd = cr.Rnd()
    ref = np.zeros((H, W, 3), np.float32)
    for _ in range(3):
        rx, ry = rnd.randf2(), rnd.randf2()
        scene.render_frame(rx, ry)
        orc.render_frame(rx, ry, ref, threads=8)
    out = scene.read_sum()
    err = np.abs(out - ref)
    assert (err <= 1e-5 + 1e-4 * np.abs(ref)).all(), float(err.max())
    assert (out.view(np.uint32) != ref.view(np.uint32)).mean() < 1e-3
    untex = cr.Scene(cr.SceneData.build(cr.Mesh(mesh.vertices, mesh.normals, mesh.texcoords, mesh.triangles,
                                                np.where(np.arange(16) == 12, -1, mesh.materials).astype(np.float32), mesh.lights), cam), W, H, 3)
    untex.render_frame(0.6591631, 0.910802)
    scene.reset(); scene.render_frame(0.6591631, 0.910802)
    assert np.abs(untex.read_sum() - scene.read_sum()).max() > 0.01     # the texture is really used
    untex.close(); scene.close()


def test_scene_input_variants(cr, ob, cornell, cornell_data, scenes):
    """crt_scene_desc accepts the BVH2 alone (converted to CWBVH inside, what a reference caller has today),
    a caller-built bvh8 alone, or both; all three trace identically.  BVH2-mode tracing needs the BVH2."""
    import copy
    from caitlynrenderer_amd import _lib
    _, orc, _ = scenes["cornell"]
    rays = seeded_rays(cornell[0], 20000, 21, cr.RAY_DT)
    want = orc.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, threads=8)
    only2 = copy.copy(cornell_data); only2.bvh8 = None; only2.bvh8_tri_slots = None
    only8 = copy.copy(cornell_data); only8.bvh = None
    for data in (only2, only8, cornell_data):
        s = cr.Scene(data, 64, 64, 1)
        _assert_hits_equal(s.trace(rays), want)
        if data.bvh is None:
            with pytest.raises(cr.CrtError) as e:
                s.trace(rays, cr.CRT_TRACE_BVH2)
            assert e.value.code == _lib.CRT_ERR_INVALID
        s.close()
    # a corrupted BVH2 leaf is refused even when a valid bvh8 comes with it: the BVH2 walk (accel 1/2, CRT_TRACE_BVH2)
    # loops over the leaf's slot range
    leaf = int(np.nonzero(cornell_data.bvh[:, 7] != 0)[0][3])
    for col, val in ((7, 200.0), (3, float(cornell_data.triangles.shape[0])), (3, -4.0), (7, np.nan), (3, np.nan), (7, 1e30)):
        bad2 = copy.copy(cornell_data)
        bad2.bvh = cornell_data.bvh.copy()
        bad2.bvh[leaf, col] = val
        with pytest.raises(cr.CrtError) as e:
            cr.Scene(bad2, 64, 64, 1)
        assert e.value.code == _lib.CRT_ERR_INVALID and "BVH2" in str(e.value), (col, val)
    # a corrupted caller-supplied CWBVH is refused instead of being walked
    bad = copy.copy(cornell_data)
    bad.bvh8 = cornell_data.bvh8.copy()
    bad.bvh8[0, 16:20] = np.frombuffer(np.uint32(10 ** 6).tobytes(), np.uint8)   # child_base_index out of range
    with pytest.raises(cr.CrtError) as e:
        cr.Scene(bad, 64, 64, 1)
    assert e.value.code == _lib.CRT_ERR_INVALID and "CWBVH rejected" in str(e.value)


def _check_lbvh(flat, tris, verts):
    n = flat.shape[0]
    leaf = flat[:, 7] != 0
    assert (flat[leaf, 7] == 1).all() and n == 2 * tris.shape[0] - 1
    inner = np.nonzero(~leaf)[0]
    left = flat[inner, 3].astype(np.int64)
    assert np.array_equal(left, 2 * np.arange(len(inner)) + 1)              # BFS, children adjacent
    assert sorted(flat[leaf, 3].astype(np.int64).tolist()) == list(range(tris.shape[0]))
    tri_lo, tri_hi = verts[tris[:, :3]].min(1), verts[tris[:, :3]].max(1)
    slots = flat[leaf, 3].astype(np.int64)
    assert np.array_equal(flat[leaf, 0:3], tri_lo[slots]) and np.array_equal(flat[leaf, 4:7], tri_hi[slots])   # exact leaf boxes
    # every interior box is exactly the union of its children (refit), checked bottom-up in one sweep
    lo, hi = flat[:, 0:3].copy(), flat[:, 4:7].copy()
    assert np.array_equal(lo[inner], np.minimum(lo[left], lo[left + 1])) and np.array_equal(hi[inner], np.maximum(hi[left], hi[left + 1]))


@pytest.mark.parametrize("builder", ["lbvh", "ploc", "ploc4", "ploc64", "sah"])
@pytest.mark.parametrize("name", ["cornell", "tess8", "tess40"])
def test_gpu_lbvh_builder(cr, ob, cornell, tess8, tess40, scenes, name, builder):
    """crt_lbvh_build: Morton sort, then Karras' radix tree + device refit (lbvh) or parallel locally-ordered
    clustering (ploc, search radius 16 / 4 / 64) or the top-down binned-SAH build (sah).  The tree is valid, the scene built on it (-> host CWBVH conversion)
    returns exactly the hits of the reference-builder scene."""
    mesh = {"cornell": cornell[0], "tess8": tess8[0], "tess40": tess40[0]}[name]
    sb = cr.SBVH(mesh.triangles, mesh.vertices, builder=builder)
    assert sb.build_ms is not None and sb.build_ms[0] > 0
    _check_lbvh(sb.flat_nodes, sb.triangles, mesh.vertices)
    assert sorted(sb.triangle_indices.tolist()) == list(range(mesh.triangles.shape[0]))       # no duplicates: a permutation
    assert np.array_equal(sb.triangles, mesh.triangles[sb.triangle_indices])
    data = cr.SceneData.build(mesh, cornell[1], builder=builder)
    scene_l = cr.Scene(data, 64, 64, 1)
    scene_s, orc, _ = scenes[name]
    rays = np.concatenate([seeded_rays(mesh, 40000, 31, cr.RAY_DT), orc.primary_rays(RX1, RY1, jitter=True).astype(cr.RAY_DT)])
    got = scene_l.trace(rays)
    _assert_hits_equal(got, scene_s.trace(rays))                            # same closest hits through either tree
    _assert_hits_equal(got, ob.Oracle(data, 64, 64, 1, cornell[1]).trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, threads=8))
    _assert_hits_equal(scene_l.trace(rays, cr.CRT_TRACE_BVH2 | cr.CRT_TRACE_TIE_LOWEST_ID), got)
    # deterministic: same tree on a second build
    sb2 = cr.SBVH(mesh.triangles, mesh.vertices, builder=builder)
    assert np.array_equal(sb2.flat_nodes.view(np.uint32), sb.flat_nodes.view(np.uint32)) and np.array_equal(sb2.triangle_indices, sb.triangle_indices)
    scene_l.close()
    if builder in ("ploc", "sah") and name != "cornell":
        # the point of PLOC: a better tree than the spatial-median LBVH — fewer node visits on the same rays
        lb = cr.Scene(cr.SceneData.build(mesh, cornell[1], builder="lbvh"), 64, 64, 1)
        pl = cr.Scene(data, 64, 64, 1)
        _, st_l = lb.trace(rays, stats=True)
        _, st_p = pl.trace(rays, stats=True)
        assert st_p["nodes"].astype(np.int64).sum() < st_l["nodes"].astype(np.int64).sum()
        lb.close(); pl.close()


@pytest.mark.parametrize("builder", ["lbvh", "ploc", "sah"])
def test_device_built_scenes_of_one_two_and_many_coincident_triangles(cr, builder):
    """Degenerate inputs of the build-on-device path: a single triangle (a leaf root), two, and 300 copies of one triangle
    (identical boxes and centroids: every SAH bin but one is empty, PLOC sees only ties)."""
    from caitlynrenderer_amd._lib import crt_camera
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]], np.float32)
    nrm = np.array([[0, 0, 1]], np.float32)
    mats = np.zeros((1, 16), np.float32); mats[0, 0:3] = 0.5; mats[0, 4:8] = (0, 0, 0, -1); mats[0, 12:16] = -1
    cam = cr.Camera((0.3, 0.3, 3.0), (0.3, 0.3, 0.0), 40.0)
    for tri_rows in ([[0, 1, 2]], [[0, 1, 2], [2, 3, 4]], [[0, 1, 2]] * 300):
        tris = np.array([[a, b, c, 0, 0, 0, 0, 1, -1, -1, -1, 0] for a, b, c in tri_rows], np.int32)
        mesh = cr.Mesh(v, nrm, np.zeros((0, 2), np.float32), tris, mats, np.zeros((0, 18), np.float32))
        s = cr.Scene(cr.SceneData.for_device_build(mesh, cam, builder=builder), 32, 32, 1)
        info = s.bvh_info()
        assert info["n_tris8"] == len(tri_rows) and info["n_bvh2_nodes"] == 2 * len(tri_rows) - 1
        rays = np.zeros(2, cr.RAY_DT)
        rays["o"] = [(0.2, 0.2, 2.0), (5.0, 5.0, 2.0)]; rays["d"] = (0, 0, -1); rays["tmax"] = 1e9
        h = s.trace(rays)
        assert h["tri"][0] == 0 and h["t"][0] == 2.0 and h["tri"][1] == -1          # lowest id among coincident triangles
        s.close()


def test_gpu_lbvh_edge_cases(cr):
    for n in (1, 2, 3):
        v = (np.arange(9 * n, dtype=np.float32).reshape(-1, 3) * np.float32(0.37)) % 5
        t = np.zeros((n, 12), np.int32)
        t[:, :3] = np.arange(3 * n).reshape(-1, 3)
        sb = cr.SBVH(t, v, builder="lbvh")
        _check_lbvh(sb.flat_nodes, sb.triangles, v)
    # identical centroids (equal Morton codes): index tie-break keeps the keys unique
    v = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), (50, 1))
    t = np.zeros((50, 12), np.int32)
    t[:, :3] = np.arange(150).reshape(-1, 3)
    sb = cr.SBVH(t, v, builder="lbvh")
    _check_lbvh(sb.flat_nodes, sb.triangles, v)
    cw = cr.CWBVH().convert(sb)
    assert cw.depth <= 16


@pytest.mark.parametrize("builder", ["lbvh", "ploc", "ploc4", "sah"])
def test_gpu_builders_refuse_non_finite_vertices_and_always_end(cr, builder):
    """one inf coordinate made every union area with that triangle inf; its PLOC cluster never found a neighbour and the
    single-workgroup tail looped for ever.  Every builder now gives such triangles a point box (so all later kernels see finite
    data and end), flags the vertex, and the call returns CRT_ERR_INVALID."""
    from caitlynrenderer_amd import _lib
    rng = np.random.default_rng(11)
    n = 700                                                    # below 1024 clusters: PLOC goes straight to the tail kernel
    v = rng.random((3 * n, 3)).astype(np.float32) * 4
    t = np.zeros((n, 12), np.int32)
    t[:, :3] = np.arange(3 * n).reshape(-1, 3)
    for bad in (np.inf, np.nan, -3.0e19):
        vb = v.copy()
        vb[37, 1] = bad
        with pytest.raises(cr.CrtError) as e:
            cr.SBVH(t, vb, builder=builder)
        assert e.value.code == _lib.CRT_ERR_INVALID and "vertex coordinate" in str(e.value)
    # 5000 coincident boxes and a grid of equal boxes: areas tie everywhere; the builders still end with a valid tree
    for verts in (np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), (5000, 1)),):
        tt = np.zeros((5000, 12), np.int32)
        tt[:, :3] = np.arange(15000).reshape(-1, 3)
        sb = cr.SBVH(tt, verts, builder=builder)
        assert sb.flat_nodes.shape[0] == 2 * 5000 - 1


def test_a_failed_growth_of_the_batch_buffers_leaves_the_scene_usable(cr, ob, cornell, tess8):
    """crt_render_frames on a multi-segment path grows path state and queues (~250 B per pixel and sample).  A failed
    allocation there used to leave null queues behind a scene that still claimed to be ready.  Now the new buffers are allocated
    first and swapped in only when all of them exist: the call fails with CRT_ERR_NOMEM, and the scene renders on, frame by frame
    and — at the next attempt — batched, with the oracle's bits."""
    from caitlynrenderer_amd import _lib
    _, data = tess8
    _, cam = cornell
    W, H, depth = 200, 120, 3
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(6)]
    s = cr.Scene(data, W, H, depth)
    s.render_frame(*rvs[0])
    s.set_option("debug_fail_batch_alloc", 1)
    with pytest.raises(cr.CrtError) as e:
        s.render_frames(rvs[1:5])
    assert e.value.code == _lib.CRT_ERR_NOMEM
    s.render_frame(*rvs[1])                                     # still works one by one ...
    s.render_frames(rvs[2:6])                                   # ... and batched, once the allocation succeeds
    orc = ob.Oracle(data, W, H, depth, cam)
    ref = np.zeros((H, W, 3), np.float32)
    for r in rvs:
        orc.render_frame(r[0], r[1], ref, threads=8)
    assert np.array_equal(s.read_sum().view(np.uint32), ref.view(np.uint32))
    s.close()
    # with tile shards on several streams (the default of crt::Scene for such paths) every shard's buffers are grown BEFORE any
    # shard renders — a growth that fails on the second shard leaves the first shard's sums untouched, not half a batch ahead
    for failing in (1, 2):                                          # the scene's own shard, its first peer
        s = cr.Scene(data, W, H, depth)
        s.set_option("streams", 2)
        s.render_frame(*rvs[0])
        before = s.read_sum()
        s.set_option("debug_fail_batch_alloc", failing)
        with pytest.raises(cr.CrtError) as e:
            s.render_frames(rvs[1:5])
        assert e.value.code == _lib.CRT_ERR_NOMEM
        assert np.array_equal(s.read_sum().view(np.uint32), before.view(np.uint32)), failing
        s.render_frame(*rvs[1])
        s.render_frames(rvs[2:6])
        assert np.array_equal(s.read_sum().view(np.uint32), ref.view(np.uint32)), failing
        s.close()


def _same_cwbvh(a, b):
    assert a.depth == b.depth and a.nodes.shape == b.nodes.shape
    assert np.array_equal(a.nodes, b.nodes), int((a.nodes != b.nodes).any(axis=1).sum())
    assert np.array_equal(a.tri_slots, b.tri_slots) and np.array_equal(a.child_bvh2, b.child_bvh2)


@pytest.mark.parametrize("name", ["cornell", "tess8", "tess40"])
@pytest.mark.parametrize("builder", ["sbvh", "lbvh"])
def test_device_cwbvh_conversion_is_byte_identical_to_the_host_converter(cr, cornell, tess8, tess40, name, builder):
    """crt_cwbvh_convert_device: cost tables bottom-up with an arrival counter, node8 tree level by
    level, depth-first numbering from subtree sizes — same 80-byte nodes, triangle order and child map as the host."""
    mesh = {"cornell": cornell[0], "tess8": tess8[0], "tess40": tess40[0]}[name]
    sb = cr.SBVH(mesh.triangles, mesh.vertices, builder=builder)
    host = cr.CWBVH().convert(sb)
    dev = cr.CWBVH().convert(sb, device=True)
    assert dev.convert_ms is not None and dev.convert_ms[0] > 0
    _same_cwbvh(dev, host)
    _same_cwbvh(cr.CWBVH().convert(sb, device=True), host)                  # and deterministic (atomics only order work)


def test_device_cwbvh_conversion_edge_cases_and_errors(cr, cornell):
    from caitlynrenderer_amd import _lib
    # a single leaf, a root with two leaves, a 3-triangle leaf
    one = np.array([[0, 0, 0, 0, 1, 1, 1, 1]], np.float32)
    two = np.array([[0, 0, 0, 1, 2, 1, 1, 0], [0, 0, 0, 0, 1, 1, 1, 1], [1, 0, 0, 1, 2, 1, 1, 1]], np.float32)
    fat = np.array([[0, 0, 0, 1, 2, 1, 1, 0], [0, 0, 0, 0, 1, 1, 1, 3], [1, 0, 0, 3, 2, 1, 1, 2]], np.float32)
    for flat, n_slots in ((one, 1), (two, 2), (fat, 5)):
        _same_cwbvh(cr.CWBVH().convert_arrays(flat, n_slots, device=True), cr.CWBVH().convert_arrays(flat, n_slots))
    # refused inputs: same classes of error as the host converter
    sb = cr.SBVH(cornell[0].triangles, cornell[0].vertices)
    n_slots = sb.triangle_indices.shape[0]
    big_leaf = sb.flat_nodes.copy(); big_leaf[np.nonzero(big_leaf[:, 7] != 0)[0][0], 7] = 4
    bad_link = sb.flat_nodes.copy(); bad_link[1, 3] = 0
    out_of_range = sb.flat_nodes.copy(); out_of_range[np.nonzero(out_of_range[:, 7] != 0)[0][0], 3] = n_slots + 5
    twice = sb.flat_nodes.copy(); leaves = np.nonzero(twice[:, 7] != 0)[0]; twice[leaves[0], 3] = twice[leaves[1], 3]
    # links that point outside the array, are negative or NaN: an out-of-bounds read if a pass followed them
    inner = np.nonzero(sb.flat_nodes[:, 7] == 0)[0]
    far_link = sb.flat_nodes.copy(); far_link[inner[2], 3] = sb.flat_nodes.shape[0] + 1000
    neg_link = sb.flat_nodes.copy(); neg_link[inner[1], 3] = -5
    nan_link = sb.flat_nodes.copy(); nan_link[inner[3], 3] = np.nan
    huge_link = sb.flat_nodes.copy(); huge_link[inner[0], 3] = 3e38
    for flat in (big_leaf, bad_link, out_of_range, twice, far_link, neg_link, nan_link, huge_link):
        for device in (False, True):
            with pytest.raises(cr.CrtError) as e:
                cr.CWBVH().convert_arrays(flat, n_slots, device=device)
            assert e.value.code == _lib.CRT_ERR_INVALID, (device, str(e.value))


def test_textured_scene_from_obj_mtl_and_image_files(cr, ob, textured, tmp_path):
    """SURVEY 8f-4 end to end: OBJ + MTL with map_Kd textures (PNG, RLE TGA and progressive JPEG files) -> loader (decode, the
    reference's 256x256 resize) -> the Python path, the C++ crt::Scene path and the oracle agree on the running sum."""
    import os
    import subprocess
    from conftest import ROOT, write_obj
    from oracle import textures as T
    mesh, _, _ = textured
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:300, 0:200]
    checker = (((yy // 25 + xx // 25) % 2) * 180 + 40).astype(np.uint8)[..., None].repeat(3, 2) ^ rng.integers(0, 32, (300, 200, 3), dtype=np.uint8)
    noise = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    open(tmp_path / "checker.png", "wb").write(T.write_png(checker))
    open(tmp_path / "noise.tga", "wb").write(T.write_tga(noise, rle=True, top_down=True))
    golden = np.load(os.path.join(ROOT, "tests", "golden", "stb_decodes.npz"))     # a JPEG and what the reference's stb_image makes of it
    open(tmp_path / "photo.jpg", "wb").write(golden["jpeg_progressive_420_q80__file"].tobytes())
    obj = str(tmp_path / "textured.obj")
    write_obj(mesh, obj, map_kd={2: "noise.tga", 3: "checker.png", 4: "photo.jpg"})
    cam = cr.Camera((-2.755610, 2.745992, 7.58545), (-2.755610, 2.745992, 6.58545), 40.0)   # Scene.h:468
    data = cr.SceneData.from_obj(obj, cam)
    assert data.albedo_textures.shape == (3, 256, 256, 3) and np.array_equal(data.albedo_textures[0], noise)
    assert np.array_equal(data.albedo_textures[1], T.texture_to_array_bytes(checker))
    assert np.array_equal(data.albedo_textures[2], T.texture_to_array_bytes(golden["jpeg_progressive_420_q80__rgb"]))
    W, H, frames, depth = 200, 120, 3, 3
    orc = ob.Oracle(data, W, H, depth, cam)
    scene = cr.Scene(data, W, H, depth)
    rnd = cr.Rnd()
    ref = np.zeros((H, W, 3), np.float32)
    for _ in range(frames):
        rx, ry = rnd.randf2(), rnd.randf2()
        scene.render_frame(rx, ry)
        orc.render_frame(rx, ry, ref, threads=8)
    out = scene.read_sum()
    err = np.abs(out - ref)
    assert (err <= 1e-5 + 1e-4 * np.abs(ref)).all(), float(err.max())
    run = subprocess.run([os.path.join(ROOT, "examples", "render_obj"), obj, str(tmp_path / "o.ppm"), str(W), str(H), str(frames),
                          str(depth), str(tmp_path / "sum.f32")], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    got = np.fromfile(tmp_path / "sum.f32", np.float32).reshape(H, W, 3)
    assert np.array_equal(got.view(np.uint32), out.view(np.uint32))          # C++ host path == Python host path, bit for bit
    scene.close()


def test_hip_bvh2_walk_reproduces_the_survey_census(cr, ob, cornell_data, survey):
    """The reference-side numbers of SURVEY 8c (probe of the reference's own BVH2 on its own scene: 1,163,374 of
    2,073,600 pixel-centre rays hit, centre pixel hits triangle 8 at t = 10.515989, 6.12 nodes + 1.28 triangle tests
    per ray) measured on the HIP path itself — BVH2 walk in the shader's order, and the CWBVH walk for the hits."""
    ka = survey["bvh2_primary_census_1920x1080_no_jitter"]
    scene = cr.Scene(cornell_data, 1920, 1080, 1)
    rays = ob.Oracle(cornell_data, 1920, 1080, 1).primary_rays(jitter=False).astype(cr.RAY_DT)
    assert rays.shape[0] == ka["rays"]
    hits, st = scene.trace(rays, cr.CRT_TRACE_CLOSEST | cr.CRT_TRACE_BVH2, stats=True)
    assert int((hits["tri"] >= 0).sum()) == ka["hits"]
    c = hits[ka["centre_pixel"]["py"] * 1920 + ka["centre_pixel"]["px"]]
    assert c["tri"] == ka["centre_pixel"]["triangle"] and abs(c["t"] - ka["centre_pixel"]["t"]) < 1e-6
    assert abs(st["nodes"].mean() - ka["nodes_per_ray"]) < 0.005 and abs(st["tris"].mean() - ka["tris_per_ray"]) < 0.005
    h8 = scene.trace(rays, cr.CRT_TRACE_CLOSEST)
    assert int((h8["tri"] >= 0).sum()) == ka["hits"] and np.array_equal(h8["tri"], hits["tri"])
    assert np.array_equal(h8["t"].view(np.uint32), hits["t"].view(np.uint32))
    scene.close()


@pytest.mark.parametrize("builder", ["lbvh", "ploc", "sah"])
@pytest.mark.parametrize("name", ["cornell", "tess8", "tess40"])
def test_scene_built_entirely_on_the_device(cr, ob, cornell, tess8, tess40, name, builder):
    """crt_scene_desc.build_flags = CRT_BUILD_LBVH_ON_DEVICE: only the input arrays are uploaded; LBVH, CWBVH conversion,
    leaf-order triangles and intersection records are produced in HBM.  Frames, hits, ray counts and visit counters are
    identical to a scene created from crt_lbvh_build's host arrays (the same tree going the long way round), hits agree with
    the oracle on that tree, and the BVH2 it keeps serves the BVH2 walks."""
    from caitlynrenderer_amd import _lib
    mesh = {"cornell": cornell[0], "tess8": tess8[0], "tess40": tess40[0]}[name]
    cam = cornell[1]
    W, H, depth = 256, 144, 3
    dev = cr.Scene(cr.SceneData.for_device_build(mesh, cam, builder=builder), W, H, depth)
    info = dev.bvh_info()
    assert info["built_on_device"] == 1 and info["n_tris8"] == mesh.triangles.shape[0] and info["n_bvh2_nodes"] == 2 * mesh.triangles.shape[0] - 1
    assert info["build_wall_ms"] > 0 and info["build_lbvh_device_ms"] > 0 and info["build_convert_device_ms"] > 0
    data = cr.SceneData.build(mesh, cam, builder=builder, convert="device")
    ref = cr.Scene(data, W, H, depth)
    assert ref.bvh_info()["n_nodes8"] == info["n_nodes8"] and ref.bvh_info()["max_depth8"] == info["max_depth8"]
    rnd = cr.Rnd()
    for s in (dev, ref):
        s.set_option("count_visits", 1)
    for _ in range(2):
        rx, ry = rnd.randf2(), rnd.randf2()
        dev.render_frame(rx, ry); ref.render_frame(rx, ry)
        a, b = dev.frame_stats(), ref.frame_stats()
        for k in ("closest_rays", "any_rays", "nodes_closest", "tris_closest", "nodes_any", "tris_any"):
            assert a[k] == b[k], k
        assert a["stack_overflows"] == 0
    assert np.array_equal(dev.read_sum().view(np.uint32), ref.read_sum().view(np.uint32))
    rays = seeded_rays(mesh, 20000, 5, cr.RAY_DT)
    orc = ob.Oracle(data, W, H, depth, cam)
    want = orc.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, threads=8)
    _assert_hits_equal(dev.trace(rays), want)
    got2 = dev.trace(rays, cr.CRT_TRACE_BVH2 | cr.CRT_TRACE_TIE_LOWEST_ID)      # the BVH2 stayed on the device
    assert np.array_equal(got2["tri"], want["tri"]) and np.array_equal(got2["t"].view(np.uint32), want["t"].view(np.uint32))
    dev.close(); ref.close()
    # refused inputs: the same classes of error as the host checks, found by a kernel over the uploaded array
    bad = cr.SceneData.for_device_build(mesh, cam)
    bad.triangles = mesh.triangles.copy(); bad.triangles[3, 1] = mesh.vertices.shape[0] + 7
    with pytest.raises(cr.CrtError) as e:
        cr.Scene(bad, 64, 64, 1)
    assert e.value.code == _lib.CRT_ERR_INVALID and "vertex index" in str(e.value)
    bad.triangles = mesh.triangles.copy(); bad.triangles[5, 3] = 99
    with pytest.raises(cr.CrtError) as e:
        cr.Scene(bad, 64, 64, 1)
    assert e.value.code == _lib.CRT_ERR_INVALID and "material index" in str(e.value)


def test_scene_from_bvh2_only_converts_on_the_device(cr, tess40, scenes):
    """crt_scene_create with a BVH2 and no CWBVH converts internally — on the GPU for trees of 4096+ nodes; since the
    device converter is byte-identical to the host one, the scene behaves exactly like the one given both."""
    import copy
    mesh, data = tess40
    only2 = copy.copy(data)
    only2.bvh8 = None
    only2.bvh8_tri_slots = None
    assert data.bvh.shape[0] >= 4096
    a, b = cr.Scene(only2, 96, 64, 2), scenes["tess40"][0]
    assert a.bvh_info() == b.bvh_info()
    rays = seeded_rays(mesh, 50000, 77, cr.RAY_DT)
    got, gst = a.trace(rays, stats=True)
    want, wst = b.trace(rays, stats=True)
    _assert_hits_equal(got, want)
    assert np.array_equal(gst["nodes"], wst["nodes"]) and np.array_equal(gst["tris"], wst["tris"])   # same tree, same walk
    a.close()


@pytest.mark.parametrize("name,depth", [("cornell", 3), ("tess8", 3), ("tess40", 2)])
def test_frames_through_the_reference_bvh2_walk(cr, ob, cornell, scenes, name, depth):
    """crt_set_option("accel", 1): whole frames rendered with the walk the reference ships (path_trace.fs:511-819 on the
    FlatNode array, raw 1/d, first-visited tie rule) — the live shader as a frame renderer.  Radiance, ray counts and
    visit counters bit-identical to the oracle's BVH2 integrator; accel 2 = same walk with the lowest-id tie rule."""
    _, _, data = scenes[name]
    W, H = 240, 136
    for accel, tie in ((1, ob.TIE_FIRST_VISITED), (2, ob.TIE_LOWEST_ID)):
        scene = cr.Scene(data, W, H, depth)
        scene.set_option("accel", accel)
        scene.set_option("count_visits", 1)
        orc = ob.Oracle(data, W, H, depth, cornell[1])
        rnd = cr.Rnd()
        ref = np.zeros((H, W, 3), np.float32)
        for frame in range(3):
            rx, ry = rnd.randf2(), rnd.randf2()
            scene.render_frame(rx, ry)
            _, cnt = orc.render_frame(rx, ry, ref, accel=ob.BVH2, tie=tie, threads=8)
            st = scene.frame_stats()
            assert (st["closest_rays"], st["any_rays"]) == (cnt[0], cnt[1]), (accel, frame)
            assert st["nodes_closest"] + st["nodes_any"] == cnt[2] and st["tris_closest"] + st["tris_any"] == cnt[3], (accel, frame)
            out = scene.read_sum()
            assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (accel, frame, float(np.abs(out - ref).max()))
        if accel == 2:
            # same tie rule as the CWBVH path: the two walks agree except where a zero direction component makes the
            # shader's raw 1/d slab test drop a box
            cw = cr.Scene(data, W, H, depth)
            rnd = cr.Rnd()
            for frame in range(3):
                cw.render_frame(rnd.randf2(), rnd.randf2())
            same = (cw.read_sum().view(np.uint32) == out.view(np.uint32)).all(axis=2).mean()
            assert same > 0.995, same
            cw.close()
        scene.close()
    # a scene without a BVH2 cannot switch
    import copy
    only8 = copy.copy(data); only8.bvh = None
    s8 = cr.Scene(only8, 32, 32, 1)
    with pytest.raises(cr.CrtError):
        s8.set_option("accel", 1)
    s8.close()


def test_graph_replay_measurement_aid(cr, scenes):
    """crt_debug_time_graph: the frame loop is capturable as a hipGraph; replaying it accumulates exactly like queued frames."""
    import ctypes as C
    from caitlynrenderer_amd._lib import lib, check
    _, _, data = scenes["tess8"]
    W, H, depth, n, reps = 160, 96, 2, 4, 3
    rnd = cr.Rnd()
    rxy = np.array([rnd.randf2() for _ in range(2 * n)], np.float32)
    s = cr.Scene(data, W, H, depth)
    a, b = C.c_float(), C.c_float()
    check(lib().crt_debug_time_graph(s._h, n, rxy.ctypes.data_as(C.c_void_p), reps, C.byref(a), C.byref(b)))
    assert a.value > 0 and b.value > 0
    got = s.read_sum()
    # what ran: 2 set-up frames with the first vector, reps x n queued frames, (1 warm-up + reps) graph replays of n frames
    ref = cr.Scene(data, W, H, depth)
    for _ in range(2):
        ref.render_frame(float(rxy[0]), float(rxy[1]))
    for _ in range(reps + 1 + reps):
        for f in range(n):
            ref.render_frame(float(rxy[2 * f]), float(rxy[2 * f + 1]))
    assert np.array_equal(got.view(np.uint32), ref.read_sum().view(np.uint32))
    with pytest.raises(cr.CrtError):
        check(lib().crt_debug_time_graph(s._h, 3, rxy.ctypes.data_as(C.c_void_p), 1, C.byref(a), C.byref(b)))   # odd frame count
    s.close(); ref.close()


def test_warmup_loads_the_code_objects_and_can_be_repeated(cr, cornell_data):
    """crt_warmup (the synchronous form of what the first crt_scene_create starts in the background): returns CRT_OK, also when called again and
    after scenes exist; a scene created right behind it renders the same frame as one created without it."""
    cr.warmup()
    a = cr.Scene(cornell_data, 96, 64, 2)
    a.render_frame(RX1, RY1)
    want = a.read_sum()
    cr.warmup()                                       # second call: everything is loaded, the background thread long joined
    b = cr.Scene(cornell_data, 96, 64, 2)
    b.render_frame(RX1, RY1)
    assert np.array_equal(b.read_sum().view(np.uint32), want.view(np.uint32)) and want.max() > 0
    a.close(); b.close()


def test_bench_line_contract(tmp_path):
    """`python bench.py` (N = 1): exactly one JSON line, short enough for the driver's record (< 8 KB), whose top-level fields are
    configs[2] — the 1,004,672-triangle workload BASELINE.json's targets are quoted on — with the roofline object against the roof
    that binds (vector-instruction issue; every figure recomputable from the line), the CPU baseline, compact extras for the other
    configs, the reference's own claims (README.md:21-22) measured on this GPU and the drop-in frame loop."""
    import importlib.util
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-hbm-resident"], capture_output=True, text=True,
                         cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, run.stdout
    assert len(lines[0]) < 8000, len(lines[0])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mray/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "1920x1080" in d["metric"]
    cfg = d["config"]
    assert "1004672 tris" in cfg["workload"] and cfg["resolution"] == "1920x1080" and cfg["spp_per_step"] == 4 and cfg["path_segments"] == 1
    assert "model" not in cfg and cfg["stack_overflows"] == 0
    sp = d["step_ms_spread"]
    assert sp["n"] == 6 and 0 < sp["min"] <= sp["median"] <= sp["max"] < 2 * d["ms_per_step"]

    import shutil
    have_rocprof = shutil.which("rocprofv3") is not None or os.path.exists("/opt/rocm/bin/rocprofv3")
    r = d["roofline"]
    assert r["bound"] == "valu_issue" and r["unit"] == "Gwave-instr/s" and r["peak"] == 1228.8 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.05 < r["frac_executed"] < r["frac"] <= 1.0 and r["launch_ms"] > 0 and r["launches_timed"] == 6 and r["samples_per_launch"] == 4
    assert r["algorithmic_gbps"] > 1000 and r["algorithmic_over_peak"] > 1.0 and (r["traffic"] is None or r["traffic"] > 0) and "HBM roof does not bind" in r["note"]
    # every derived figure is recomputable from the line: its counters (ONE STEP, counted in the timed launch form) x profiles/isa_counts.json /
    # launch time, SURVEY 8d's bytes with the 24 B per pixel-sample
    spec = importlib.util.spec_from_file_location("crt_roofline_t", os.path.join(ROOT, "tools", "roofline.py"))
    rl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rl)
    isa = json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))
    c = r["counters"]
    assert c["primary_rays"] == 4 * 1920 * 1080 == c["closest_rays"] and 0 < c["closest_hits"] < c["closest_rays"]
    assert 0.5 * c["nodes_closest"] < c["nodes_closest_uniform"] < c["nodes_closest"] and 0 < c["nodes_any_uniform"] < c["nodes_any"]      # 4 x 4-pixel waves agree on most nodes
    for k, v in rl.recompute_line_block(r, isa).items():
        assert abs(v - r[k]) <= 2e-3 * max(1.0, abs(v)), (k, v, r[k])
    assert r["algorithmic_bytes_per_launch"] == 80 * (c["nodes_closest"] + c["nodes_any"]) + 52 * (c["tris_closest"] + c["tris_any"]) + 24 * c["primary_rays"]
    if have_rocprof:
        # the counter passes are run by this very invocation (child processes under rocprofv3 --pmc); on a box slow enough for a pass to
        # run into its time limit bench.py says so on stderr and falls back to the committed passes — the checks below hold either way
        assert r["traffic_source"] in ("live", "committed"), r["traffic_source"]
        if r["traffic_source"] != "live":
            assert "live pmc" in run.stderr, run.stderr[-1500:]
        assert r["traffic"] > 0 and 0.2 < r["lane_util"] <= 1.0 and 0.1 < r["issue_busy"] <= 1.0, r
        assert r["traffic"] < r["algorithmic_bytes_per_launch"]       # the scene is cache-resident: no wasted re-reads
        # useful work cannot exceed executed work: frac_executed <= issue_busy x lane_util, on the headline and on every extra that carries counters
        assert abs(r["counter_frac"] - r["issue_busy"] * r["lane_util"]) < 2e-3 and r["frac_executed"] <= r["counter_frac"] and 0 < r["non_traversal_share"] < 1
        assert abs(r["hbm_frac"] - r["traffic_gbps"] / 8000.0) < 1e-3
        for k, e in d["extras"].items():
            if e.get("counter_frac"):
                assert e["frac_executed"] <= e["counter_frac"] + 1e-3, (k, e["frac_executed"], e["counter_frac"])
    assert cfg["launch"] == {"form": 2, "wide": True, "one_pass": True, "samples": 4, "shards": 1} and d["sum_rows_match_oracle"] is True
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mray/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["visit_counters_match_gpu"] is True
    assert d["value"] > 1000 and abs(d["value"] - cfg["rays_per_step"] / d["ms_per_step"] / 1e3) / d["value"] < 0.01
    ex = d["extras"]
    for k in ("cornell", "gpu_tree", "d2", "incoherent", "incoherent_disney", "scale_base"):
        assert ex[k]["value"] > 500 and ex[k]["launch_ms"] > 0 and 0 < ex[k]["frac_executed"] <= ex[k]["frac"] <= 1.0 and ex[k]["sum_rows_match_oracle"] is True, k
    assert " d2 " in ex["d2"]["workload"] and ex["incoherent"]["value"] < ex["d2"]["value"] < d["value"]
    # the builders' code object is loaded by then (crt_scene_create's warm-up thread): the device time of the build says so; the wall time of the
    # call includes the host's copy of the input arrays, which varies from box to box (9.6 ms in profiles/r04_bench_default.json, 25 ms seen)
    assert ex["gpu_tree"]["device_build"]["bvh2_device_ms"] < 8 and ex["gpu_tree"]["device_build"]["scene_create_wall_ms"] < 60
    assert ex["cornell"]["value"] > d["value"] and ex["cornell"]["samples_per_launch"] == 1
    assert ex["gpu_tree"]["value"] > 0.9 * d["value"] and ex["gpu_tree"]["device_build"]["builder"] == "sah" and ex["gpu_tree"]["device_build"]["bvh2_device_ms"] > 0
    assert " d4 " in ex["incoherent"]["workload"] and "disney" in ex["incoherent_disney"]["workload"] and "3840x2160" in ex["scale_base"]["workload"]
    # the reference's live walk (BVH2, path_trace.fs:511-819) on the same frames, and its README's two claims as this GPU measures them
    for k in ("bvh2_cornell", "bvh2_mesh1m", "bvh2_mesh1m_sah"):
        assert ex[k]["value"] > 500 and " bvh2" in ex[k]["workload"] and ex[k]["sum_rows_match_oracle"] is True, k
    cl = d["reference_claims"]
    assert cl["cwbvh_over_bvh2_mesh1m"] > 1.5 and cl["cwbvh_over_bvh2_cornell"] > 1.0 and 0.9 < cl["sbvh_over_sah_bvh2_walk"] < 1.5 and "README.md:21-22" in cl["readme"]
    # the drop-in frame loop: one sample + one resolve per displayed frame, image left in HBM or copied to the host
    for k in ("frame_loop_cornell", "frame_loop_mesh1m"):
        fl = ex[k]
        assert 0 < fl["segment_launch_ms"] < fl["ms_per_frame_image_in_hbm"] <= fl["ms_per_frame_image_in_host_memory"] and fl["rgba_rows_match_oracle"] is True, (k, fl)


def test_bench_self_launch_under_rccl_on_one_gpu(tmp_path):
    """The N > 1 code path (process group over RCCL, sharding, gather inside the timed region, the rank-0-alone base) at the only
    world size a 1-GPU box allows: launched under torch.distributed.run with one rank, configs[4]'s workload requested explicitly."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29613", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "mesh40", "--resolution", "3840x2160",
                          "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["resolution"] == "3840x2160" and d["config"]["spp_per_step"] == 4
    assert "RCCL gather" in d["config"]["gather"] and d["value"] > 1000 and "3840x2160" in d["metric"]
    assert d["gather_ms"] >= 0 and len(d["rank_device_ms_per_step"]) == 1 and d["rank_device_ms_per_step"][0] > 0


def test_bench_one_process_several_devices(tmp_path):
    """`bench.py --one-process`: the N devices behind ONE scene handle (crt_set_devices), gather inside the C ABI at the end of the
    timed region — here with 4 virtual devices on the one GPU (copies, not RCCL), configs[4]'s frame on a smaller mesh."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--one-process", "--virtual-devices", "4", "--workload", "mesh40",
                          "--resolution", "3840x2160", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-live-pmc"],
                         capture_output=True, text=True, cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-2000:]
    d = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 4 and d["config"]["devices"] == [0, 0, 0, 0] and d["config"]["parallelism"] == "tiles/4"
    assert "inside the C ABI" in d["config"]["gather"] and d["gather_ms"] > 0 and d["value"] > 500
    assert d["roofline"]["counters"]["primary_rays"] == 4 * 3840 * 2160         # the counters describe one STEP: 4 samples of every pixel


def test_short_reciprocal_and_square_root_are_exact_on_every_float(tmp_path):
    """rt_math.hpp computes 1 / x as v_rcp_f32 + one Newton step and sqrt(x) as v_rsq_f32 + one Newton step for 2^-100 <= |x| <= 2^100 and says
    the bits are those of the IEEE division / sqrtf the floating-point contract asks for.  A float has 2^32 values: the two
    checkers under tools/ubench run every one of them through both forms on the GPU (candidate A of each file is the shipped sequence)."""
    import os
    import re
    import subprocess
    from conftest import ROOT
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    for name in ("rcp_exhaustive", "sqrt_exhaustive"):
        exe = str(tmp_path / name)
        subprocess.run([hipcc, "-O3", "-ffp-contract=off", "--offload-arch=gfx950", "-o", exe, os.path.join(ROOT, "tools", "ubench", name + ".hip")],
                       check=True, capture_output=True, timeout=300)
        out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
        m = re.search(r"candidate A differs on (\d+)", out)
        assert m and int(m.group(1)) == 0, out
