"""Randomised parity soak (run on the GPU box): random cameras, resolutions, depths, builders and scenes; the HIP path
must match the oracle bit for bit (radiance sums, ray counts, visit counters) on every case.
usage: python tools/soak.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
g.build()
import caitlynrenderer_amd as cr
from caitlynrenderer_amd.meshgen import tessellated_cornell, with_disney_materials
from oracle import binding as ob

EXPERIMENTS = cr.has_experiments()

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
base, cam0 = g._cornell()
dis = with_disney_materials(base)      # mirror tall box, GGX / Disney-diffuse short box and floor (oracle-defined materials)
meshes = {"cornell": base, "tess8": tessellated_cornell(base, 8), "tess24": tessellated_cornell(base, 24),
          "cornell_mat": dis, "tess8_mat": tessellated_cornell(dis, 8), "tess24_mat": tessellated_cornell(dis, 24)}
datas = {}
bad = 0
t_start = time.time()
for case in range(n_cases):
    name = list(meshes)[rng.integers(0, len(meshes))]
    builder = str(rng.choice(["sbvh", "sbvh", "sbvh", "lbvh", "ploc", "ploc5", "sah", "sah"]))
    convert = "device" if rng.random() < 0.5 else "host"
    key = (name, builder, convert)
    if key not in datas:
        datas[key] = cr.SceneData.build(meshes[name], cam0, builder=builder, convert=convert)
    data = datas[key]
    # one case in five: the scene itself is built on the device from the source-order arrays (same builder, same tree)
    on_device = builder != "sbvh" and rng.random() < 0.4
    W, H = int(rng.integers(40, 420)), int(rng.integers(30, 260))
    depth = int(rng.integers(1, 5))
    # camera: inside or outside the box (box spans roughly 0..5.6), any direction, fov 15..100 degrees
    pos = rng.uniform(-3.0, 9.0, 3).astype(np.float32)
    tgt = rng.uniform(0.0, 5.6, 3).astype(np.float32)
    if rng.random() < 0.15:
        tgt = pos + np.array([0, 0, -1], np.float32) * np.float32(rng.uniform(0.5, 3))      # axis-aligned view: zero components
    cam = cr.Camera(tuple(float(x) for x in pos), tuple(float(x) for x in tgt), float(rng.uniform(15, 100)))
    scene = cr.Scene(cr.SceneData.for_device_build(meshes[name], cam0, builder=builder) if on_device else data, W, H, depth)
    scene.update(cam)
    jitter = 1                                   # the oracle's frame loop always jitters, like the shader
    scene.set_option("count_visits", 1)
    # pipeline variants: all must give the oracle's bits
    accel = 0 if name.endswith("_mat") else int(rng.choice([0, 0, 0, 1, 2]))      # the BVH2 frame mode is the Lambert-only shader
    opts = {"streams": int(rng.choice([1, 1, 1, 2, 2, 3]))}      # the frame on one stream, or its tile shards side by side on two or three
    opts["accel"] = accel
    if accel == 0:
        # shadow rays in place / deferred to one launch per frame (every segment's, or the bounce segments'), closest hits of the bounce
        # segments fused or through refilled pools
        opts.update(inplace_shadow=int(rng.choice([1, 1, 1, 2, 2, 0])), tri_min=int(rng.choice([0, 1, 2, 2, 3])), lanes_per_ray=int(rng.choice([1, 8, 8])),
                    bounce_refill=int(rng.random() < 0.3), refill_pool=int(rng.choice([64, 128, 256, 512])), refill_min=int(rng.choice([1, 8, 8, 32, 65])),
                    shadow_pool=int(rng.choice([64, 64, 128, 256])), shadow_refill_min=int(rng.choice([65, 65, 8, 32])), persistent=int(rng.random() < 0.4), sort_shadow=int(rng.random() < 0.6))
        if EXPERIMENTS:          # variants of a `make EXPERIMENTS=1` library only
            opts.update(oversubscribe=int(rng.choice([0, 0, 0, 1, 2])), waves_per_workgroup=int(rng.choice([1, 1, 2, 4])))
    opts["ray_bins"] = int(rng.choice([0, 0, 1, 1, 2, 3, 4, 4, 5]))     # bounce rays in emission order or binned by (octant, origin cell)
    opts["adaptive_tiles"] = int(rng.random() < 0.7)          # cost-sorted or centre-out tile order: the same pixels either way
    opts["wave_samples"] = int(rng.choice([0, 1, 2, 2, 3]))   # the samples of a launch in one wave, on the waves of a workgroup, or four in the lanes of a wave
    opts["wide_first"] = int(rng.choice([0, 1, 2]))           # the 5- or the 6-waves-per-SIMD build of the first-segment kernel
    for k, v in opts.items():
        scene.set_option(k, v)
    o_accel, o_tie = (ob.BVH8, ob.TIE_LOWEST_ID) if accel == 0 else (ob.BVH2, ob.TIE_FIRST_VISITED if accel == 1 else ob.TIE_LOWEST_ID)
    orc = ob.Oracle(data, W, H, depth, cam)
    rnd = cr.Rnd()
    for _ in range(int(rng.integers(0, 50))):
        rnd.randf2()
    ref = np.zeros((H, W, 3), np.float32)
    ok = True
    if rng.random() < 0.25:
        # crt_render_frames: the frames of the case in one call (one launch where the path allows: one segment walked in place, no
        # counting kernels) — the sum and the ray counts of the call must be those of the frames one by one
        scene.set_option("count_visits", 0)
        n_fr = int(rng.integers(1, 12))
        rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(n_fr)]
        scene.render_frames(rvs)
        tot = np.zeros(2, np.int64)
        last = np.zeros(2, np.int64)
        per_launch = 8 if (accel == 0 and opts.get("inplace_shadow", 1) != 0) else 1       # crt_device.cpp batch_limit
        # the library's own split of a call into launches (crt_render_frames_async): up to per_launch frames each, and where four samples
        # can sit in the lanes of a wave (wave_samples >= 2, a tree of 64+ nodes, CWBVH) a launch of 5..7 frames goes as 4 + the rest
        fours = opts.get("wave_samples", 2) >= 2 and scene.bvh_info()["n_nodes8"] >= 64 and accel == 0
        left, in_last = n_fr, 0
        while left:
            k = min(per_launch, left)
            if fours and per_launch > 1 and k > 4 and k % 4:
                k -= k % 4
            in_last, left = k, left - k
        for k, (rx, ry) in enumerate(rvs):
            _, cnt = orc.render_frame(rx, ry, ref, accel=o_accel, tie=o_tie, threads=16)
            if k >= n_fr - in_last:
                last += (cnt[0], cnt[1])
        st = scene.frame_stats()
        out = scene.read_sum()
        same = np.array_equal(out.view(np.uint32), ref.view(np.uint32))
        counts = (st["closest_rays"], st["any_rays"]) == (last[0], last[1])
        if not (same and counts):
            bad += 1
            print(f"MISMATCH case {case} (render_frames x{n_fr}): {key} on_device={on_device} {opts} {W}x{H} depth {depth} sum_equal {same} counts {counts} "
                  f"{(st['closest_rays'], st['any_rays'])} vs {tuple(last)}", flush=True)
        scene.close()
        if case % 10 == 9:
            print(f"case {case + 1}/{n_cases}: {bad} mismatching, {time.time() - t_start:.0f}s", flush=True)
        continue
    for frame in range(3):
        rx, ry = rnd.randf2(), rnd.randf2()
        scene.render_frame(rx, ry)
        _, cnt = orc.render_frame(rx, ry, ref, accel=o_accel, tie=o_tie, threads=16)
        st = scene.frame_stats()
        out = scene.read_sum()
        same = np.array_equal(out.view(np.uint32), ref.view(np.uint32))
        counts = (st["closest_rays"], st["any_rays"]) == (cnt[0], cnt[1]) and st["nodes_closest"] + st["nodes_any"] == cnt[2] and st["tris_closest"] + st["tris_any"] == cnt[3]
        if not (same and counts):
            ok = False
            print(f"MISMATCH case {case} frame {frame}: {key} on_device={on_device} {opts} {W}x{H} depth {depth} jitter {jitter} pos {pos} tgt {tgt} "
                  f"sum_equal {same} max|d| {np.abs(out - ref).max():.3g} n_diff {(out != ref).sum()} counts {counts}", flush=True)
            break
    bad += 0 if ok else 1
    scene.close()
    if case % 10 == 9:
        print(f"case {case + 1}/{n_cases}: {bad} mismatching, {time.time() - t_start:.0f}s", flush=True)
print(f"soak done: {n_cases} cases, {bad} mismatching")
sys.exit(1 if bad else 0)
