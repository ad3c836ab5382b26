#!/usr/bin/env python3
"""bench.py — Mray/s of the HIP ray/BVH-traversal hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cornell|mesh1m|meshN] [--depth D]

One step = one pass of the hot path over one batch: one sample per pixel of this rank's tiles
(ray generation -> CWBVH closest-hit -> shade/NEE -> CWBVH any-hit -> accumulate), everything
resident in HBM.  Default workload = BASELINE.json configs[1]: Cornell box, CWBVH, 1 spp,
primary + shadow, 1920x1080 on one GPU.  N > 1 (launched by torch.distributed.run, one rank per
GPU, RCCL) shards framebuffer tiles over the ranks with no data-path collective (weak scaling: the
frame grows with N so every rank keeps ~1920x1080 pixels) and gathers the per-tile radiance to rank 0
once, inside the timed region, at read-back.

Prints ONE JSON line on rank 0 with the driver's keys plus "roofline" (dominant kernel = the
closest-hit traversal: algorithmic bytes of SURVEY §8d / hipEvent launch time on the kernel's own
stream) and "cpu_baseline" (the CPU oracle on the same workload, bounded sample, rank 0, N = 1).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, MI355X_MICROARCH.md "Chip-level parameters"
NODE_BYTES, TRI_BYTES, FB_BYTES = 80, 52, 24   # SURVEY.md §8d algorithmic bytes per node fetch / triangle test / pixel-sample


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_workload(name, builder="sbvh", convert="host"):
    import numpy as np
    import __graft_entry__ as g
    import caitlynrenderer_amd as cr
    from caitlynrenderer_amd.meshgen import tessellated_cornell
    mesh, cam = g._cornell()
    label = "cornell-box 32 tris (Models/cornell-box.obj), CWBVH"
    if name != "cornell":
        n = 183 if name == "mesh1m" else int(name[4:])
        mesh = tessellated_cornell(mesh, n)
        label = f"procedural tessellated Cornell n={n}: {mesh.triangles.shape[0]} tris, CWBVH"
    t0 = time.time()
    data = cr.SceneData.build(mesh, cam, builder=builder, convert=convert)
    if builder == "lbvh":
        label += " over a GPU-built LBVH"
    if convert == "device":
        label += ", CWBVH converted on the GPU"
    return data, cam, label, time.time() - t0


def frame_size(n_gpus):
    if n_gpus == 1:
        return 1920, 1080
    w = int(round(1920 * math.sqrt(n_gpus) / 16.0)) * 16
    return w, int(round(w * 9 / 16))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cornell")
    ap.add_argument("--depth", type=int, default=1, help="path segments per sample (1 = primary + shadow)")
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--builder", default="sbvh", choices=["sbvh", "lbvh"],
                    help="sbvh = the reference's split-BVH on the host (default); lbvh = GPU linear BVH (crt_lbvh_build)")
    ap.add_argument("--convert", default="host", choices=["host", "device"],
                    help="BVH2 -> CWBVH conversion on the host (default) or on the GPU (crt_cwbvh_convert_device, same bytes)")
    ap.add_argument("--accel", default="cwbvh", choices=["cwbvh", "bvh2"],
                    help="cwbvh = the 8-wide compressed BVH (default, the metric's configuration); bvh2 = frames through the "
                         "reference's live BVH2 walk (path_trace.fs:511-819), for comparison")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: anything a library prints there (RCCL's version banner at
    # communicator creation, for one) is sent to stderr instead
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the traversal path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run (also at N = 1: same code path)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g
    g.build()
    import caitlynrenderer_amd as cr
    from caitlynrenderer_amd import tiles

    data, cam, label, build_s = build_workload(args.workload, args.builder, args.convert)
    W, H = frame_size(world)
    scene = cr.Scene(data, W, H, args.depth)
    scene.set_shard(rank, world, args.tile)
    if args.accel == "bvh2":
        scene.set_option("accel", 1)
        label = label.replace("CWBVH", "BVH2 walked as the shipped shader does")
    info = scene.bvh_info()
    if rank == 0:
        log(f"[bench] {label}; {W}x{H}, depth {args.depth}, {world} rank(s); BVH build {build_s:.1f}s; "
            f"{info['n_nodes8']} node8, {info['n_tris8']} tris, depth {info['max_depth8']}")

    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(args.warmup + args.steps + 1)]

    # ---- untimed: algorithmic bytes of one step (visit counters from the counting kernels) ----
    scene.set_option("count_visits", 1)
    scene.render_frame(*rvs[0])
    cs = scene.frame_stats()
    scene.set_option("count_visits", 0)
    scene.reset()

    _, tile, n_floats = scene.packed_info()
    gather_buf = torch.zeros(tiles.max_local_tiles(W, H, tile, world) * tile * tile * 3, dtype=torch.float32, device="cuda")
    recv = torch.empty(world * gather_buf.numel(), dtype=torch.float32, device="cuda") if (use_dist and rank == 0) else None

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        scene.sync()

    def read_back():
        """RCCL gather over xGMI of the per-tile radiance to rank 0 (at read-back only)."""
        scene.copy_packed_device(gather_buf.data_ptr(), n_floats)
        tiles.gather_packed_to_root(gather_buf, recv, world)

    for i in range(args.warmup):
        scene.render_frame(*rvs[1 + i], sync=False)
    scene.sync()
    if use_dist:    # warm the collective too
        read_back()
    # HIP events on every closest-hit launch of the timed region, on the scene's own stream (attached to the dispatch:
    # they take the kernel's own start/stop timestamps); the any-hit launches are timed in the untimed tail below
    scene.set_option("timing", 1)
    scene.set_option("timing_accumulate", args.steps * max(1, args.depth))
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        scene.render_frame(*rvs[1 + args.warmup + i], sync=False)
    scene.sync()
    if use_dist:
        read_back()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-step ray counts and per-kernel device time of the LAST timed step (all steps do identical work
    # up to the per-frame random vector); then a short event-timed tail for a stable launch average
    st = scene.frame_stats()
    launch_ms_timed = st["ms_trace_closest"] / max(1, st["n_trace_launches"])   # mean over the timed region's launches
    n_timed_launches = st["n_trace_launches"]
    scene.set_option("timing_accumulate", 0)
    scene.set_option("timing", 2)
    any_ms, total_ms = [], []
    for i in range(min(10, args.steps)):
        scene.render_frame(*rvs[1 + args.warmup + i])
        s = scene.frame_stats()
        any_ms.append(s["ms_trace_any"] / max(1, args.depth))
        total_ms.append(s["ms_total"])
    rays_step = st["closest_rays"] + st["any_rays"]
    if use_dist:
        t = torch.tensor([float(rays_step)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rays_all = float(t.item())
    else:
        rays_all = float(rays_step)
    value = rays_all * args.steps / dt / 1e6

    if rank == 0:
        launches = max(1, args.depth)
        node_bytes = 96 if args.accel == "bvh2" else NODE_BYTES      # own 2 texels + 4 child texels per BVH2 visit
        alg_closest = (node_bytes * cs["nodes_closest"] + TRI_BYTES * cs["tris_closest"]) / launches
        # tiny trees: k_segment also walks the NEE shadow rays (no k_shadow launch), so their visits are this launch's bytes too
        fused_shadow = st["any_rays"] > 0 and float(np.median(any_ms)) == 0.0
        if fused_shadow:
            alg_closest += (node_bytes * cs["nodes_any"] + TRI_BYTES * cs["tris_any"]) / launches
        t_closest = launch_ms_timed * 1e-3
        achieved = alg_closest / t_closest / 1e9 if t_closest > 0 else 0.0
        if fused_shadow:
            kernel_label = "k_segment (raygen + CWBVH closest hit + shading + in-place NEE any-hit walk)"
        elif args.depth > 1:
            kernel_label = ("closest-hit launches, mean per path segment: k_segment (raygen + closest hit + shading + queue emission) "
                            "for segment 0, k_closest_queue (lane-refill pools) + k_segment<PRETRACED> (shading) for bounce segments")
        else:
            kernel_label = "k_segment (raygen + CWBVH closest hit + shading + queue emission)"
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written from a separate rocprofv3 --pmc pass
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{args.workload}_d{args.depth}", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mray/s (primary+1 bounce) at 1920x1080",
            "value": round(value, 2), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "resolution": f"{W}x{H}", "spp_per_step": 1, "path_segments": args.depth,
                       "rays_per_step": int(rays_all), "closest_rays_rank0": int(st["closest_rays"]),
                       "any_rays_rank0": int(st["any_rays"]), "tile": tile, "parallelism": f"tiles/{world}",
                       "gather": "one RCCL gather of the packed tiles to rank 0 per timed region" if use_dist else "none"},
            "roofline": {"bound": "hbm", "kernel": kernel_label, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(alg_closest),
                         "bytes_per_ray": round(alg_closest / max(1, (cs["closest_rays"] + (cs["any_rays"] if fused_shadow else 0)) / launches), 2),
                         "nodes_per_ray": round(cs["nodes_closest"] / max(1, cs["closest_rays"]), 3),
                         "tris_per_ray": round(cs["tris_closest"] / max(1, cs["closest_rays"]), 3),
                         "any_hit_nodes_per_ray": round(cs["nodes_any"] / max(1, cs["any_rays"]), 3),
                         "any_hit_tris_per_ray": round(cs["tris_any"] / max(1, cs["any_rays"]), 3),
                         "launch_ms": round(t_closest * 1e3, 4), "launches_timed": int(n_timed_launches),
                         "any_hit_launch_ms": round(float(np.median(any_ms)), 4),
                         "frame_device_ms": round(float(np.median(total_ms)), 4),
                         "note": ("working set fits the 256 MiB Infinity Cache: measured HBM traffic << algorithmic bytes; the kernel is VALU-issue bound"
                                  + ("; this tree of %d nodes lives in L1, so the algorithmic-bytes rate can exceed the HBM peak" % info["n_nodes8"]
                                     if info["n_nodes8"] < 64 else ""))},
        }
        if not args.no_cpu_baseline and world == 1 and args.accel == "cwbvh":
            out["cpu_baseline"] = cpu_baseline(data, cam, W, H, args.depth, rvs[0], cs)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    scene.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(data, cam, W, H, depth, rv, cs):
    """The CPU oracle (a scalar port of the same algorithm on the same CWBVH) on a bounded sample of
    the same workload: whole frames for small scenes, a band of pixel rows when a frame would take
    too long; all host cores.  Also cross-checks the GPU visit counters on that sample."""
    import numpy as np
    from oracle import binding as ob
    orc = ob.Oracle(data, W, H, depth, cam)
    threads = ob.hardware_threads()
    y0, y1 = 0, H
    t0 = time.perf_counter()
    probe = np.zeros((H, W, 3), np.float32)
    cnt = orc.render_rows(rv[0], rv[1], H // 2 - 4, H // 2 + 4, probe)
    rate = (cnt[0] + cnt[1]) / (time.perf_counter() - t0)              # rays/s, one thread
    est_full = (cs["closest_rays"] + cs["any_rays"]) / (rate * threads)
    if est_full > 12.0:                                                  # keep the sample near 10 s
        rows = max(8, int(H * 10.0 / est_full) // 8 * 8)
        y0 = (H - rows) // 2 // 8 * 8
        y1 = y0 + rows
    times, rays = [], 0
    for rep in range(3 if est_full < 4 else 1):
        buf = np.zeros((H, W, 3), np.float32)
        t0 = time.perf_counter()
        if (y0, y1) == (0, H):
            _, c = orc.render_frame(rv[0], rv[1], buf, threads=threads)
        else:
            import concurrent.futures as cf
            bands = [(a, min(a + 8, y1)) for a in range(y0, y1, 8)]
            with cf.ThreadPoolExecutor(threads) as ex:
                cs_ = list(ex.map(lambda b: orc.render_rows(rv[0], rv[1], b[0], b[1], buf), bands))
            c = [sum(x[k] for x in cs_) for k in range(4)]
        times.append(time.perf_counter() - t0)
        rays = c[0] + c[1]
    check = None
    if (y0, y1) == (0, H):
        check = bool(c[0] == cs["closest_rays"] and c[1] == cs["any_rays"] and
                     c[2] == cs["nodes_closest"] + cs["nodes_any"] and c[3] == cs["tris_closest"] + cs["tris_any"])
    return {"value": round(rays / float(np.median(times)) / 1e6, 3), "unit": "Mray/s", "cores": threads, "kind": "port",
            "single_thread_value": round(rate / 1e6, 3),          # one thread, 8 pixel rows through the image centre
            "sample": f"rows {y0}..{y1} of {H} ({rays} rays, same frame and CWBVH as the GPU step)",
            "visit_counters_match_gpu": check}


if __name__ == "__main__":
    main()
