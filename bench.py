#!/usr/bin/env python3
"""bench.py — Mray/s of the HIP ray/BVH-traversal hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--workload auto|cornell|mesh1m|meshN] [--depth D] [--resolution WxH] [--spp S] [--scaling strong|weak]

One step = one pass of the hot path over one batch of the workload: `spp_per_step` samples per pixel of this rank's
tiles (ray generation -> CWBVH closest hit -> shading / NEE -> CWBVH any hit -> accumulate), everything resident in HBM.

Workloads (BASELINE.json `configs`, made concrete in SURVEY.md §8d):
  N = 1, --workload auto (the default):
    value / roofline / cpu_baseline  = configs[1]: Cornell box, CWBVH, 1 spp primary + shadow, 1920x1080;
    "north_star"                     = configs[2]: the 1,004,672-triangle mesh, 4 spp per step, 1920x1080, primary + shadow —
                                       the workload BASELINE.json's targets are quoted on, with its own roofline and cpu_baseline;
    "north_star_gpu_tree"            = the same over a tree built on the GPU (binned SAH, everything assembled in HBM);
    "incoherent" / "incoherent_disney" = configs[3]: same mesh and 4 spp per step, 4 path segments (incoherent bounce rays), with the reference's
                                       Lambert integrator and with the oracle-defined mirror + GGX/Disney-diffuse materials;
    "scale_base"                     = configs[4] at N = 1: the 3840x2160 frame of that mesh on one GPU (what the N > 1 lines
                                       divide by);
    "cornell_8_frames_per_launch"    = for information: configs[1]'s scene with 8 frames per crt_render_frames call (one launch).
  N > 1, --workload auto: configs[4]: ONE fixed 3840x2160 frame of the 1 M-triangle mesh, 4 spp per step, its 16x16 tiles dealt
    to the N ranks (strong scaling, no data-path collective), one RCCL gather of the per-tile radiance to rank 0 inside the timed
    region; "n1_same_workload" is the same frame rendered by rank 0 alone in the same job.  `--scaling weak` keeps the round-1
    behaviour (the frame grows with N, ~1920x1080 pixels per rank).
  An explicit --workload measures just that one (used by the profiling scripts).

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts its N ranks itself, as fresh child processes
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`) BEFORE anything touches the GPU,
forwards rank 0's JSON line and exits with the children's status; launched under torch.distributed.run by someone else it
simply is one of the ranks.

Prints ONE JSON line on rank 0 with the driver's keys plus "roofline" (dominant kernel = the fused segment kernel:
algorithmic bytes of SURVEY §8d / HIP-event launch time on the kernel's own stream; `traffic` (L2<->fabric bytes per launch) and
`valu_issue` (issue slots busy x lanes enabled — the real ceiling of this kernel) come from rocprofv3 --pmc passes: at N = 1 in
auto mode this invocation runs them itself, as child processes before its own first GPU call (`traffic_source: "live"`, ~1.5 min;
--no-live-pmc or a missing rocprofv3 falls back to the committed passes of profiles/pmc_traffic.json, `"committed"`)) and
"cpu_baseline" (the CPU oracle on the same workload, bounded sample, rank 0, N = 1).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, MI355X_MICROARCH.md "Chip-level parameters"
NODE_BYTES, TRI_BYTES, FB_BYTES = 80, 52, 24   # SURVEY.md §8d algorithmic bytes per node fetch / triangle test / pixel-sample
METRIC = "Mray/s (primary+1 bounce) at 1920x1080"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="auto", help="auto (see the module docstring), cornell, mesh1m or meshN (tessellation n)")
    ap.add_argument("--depth", type=int, default=1, help="path segments per sample (1 = primary + shadow)")
    ap.add_argument("--resolution", default=None, help="WxH; default 1920x1080 (3840x2160 for N > 1 strong scaling)")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per step (default: 1 for cornell, 4 for the mesh workloads)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = one fixed frame split over the ranks (configs[4], default); weak = the frame grows with N")
    ap.add_argument("--tile", type=int, default=16, help="tile edge in pixels (multiple of 8): unit of the ranks' shards and of the cost-sorted launch order")
    ap.add_argument("--builder", default="sbvh",
                    help="sbvh = the reference's split-BVH on the host (default); lbvh = GPU linear BVH (crt_lbvh_build); "
                         "ploc / ploc<radius> = GPU parallel locally-ordered clustering")
    ap.add_argument("--convert", default="host", choices=["host", "device"],
                    help="BVH2 -> CWBVH conversion on the host (default) or on the GPU (crt_cwbvh_convert_device, same bytes)")
    ap.add_argument("--accel", default="cwbvh", choices=["cwbvh", "bvh2"],
                    help="cwbvh = the 8-wide compressed BVH (default, the metric's configuration); bvh2 = frames through the "
                         "reference's live BVH2 walk (path_trace.fs:511-819), for comparison")
    ap.add_argument("--materials", default="lambert", choices=["lambert", "disney"],
                    help="mesh workloads: lambert = the reference's only BSDF (default); disney = the boxes get the GGX / Disney-diffuse "
                         "material of configs[3] (oracle-defined, no reference code)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=INT", help="crt_set_option passthrough (tuning experiments)")
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="untimed rendering before the warm-up steps, in ms of wall time (default 100): clocks and caches reach the steady "
                         "state the metric is about; 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="auto workload at N = 1: do not run the rocprofv3 --pmc passes (roofline.traffic / valu_issue then come from the "
                         "committed profiles/pmc_traffic.json and say so)")
    ap.add_argument("--no-extra", action="store_true", help="auto workload: only the headline block")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU and no rendering: launcher, process group (gloo), shard bookkeeping, gather and the JSON line only "
                         "(what the CPU tests exercise); the line says dry_run and reports no throughput")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher

def self_launch(args):
    """Plain `python bench.py --gpus N`: start the N ranks as fresh children before this process has made a single GPU call
    (a process that initialised the GPU must never be replaced or forked), forward rank 0's line, return the exit status."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}")
    run = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    if run.returncode == 0 and len(lines) != 1:
        log(f"[bench] expected one JSON line from rank 0, got {len(lines)}")
        return 1
    for l in lines:
        print(l, flush=True)
    return run.returncode


# ------------------------------------------------------------------------------------------------ workloads

_SCENE_CACHE = {}


def build_workload(name, builder="sbvh", convert="host", materials="lambert"):
    key = (name, builder, convert, materials)
    if key in _SCENE_CACHE:
        return _SCENE_CACHE[key]
    import copy
    import __graft_entry__ as g
    import caitlynrenderer_amd as cr
    from caitlynrenderer_amd.meshgen import tessellated_cornell
    mesh, cam = g._cornell()
    if materials == "disney":
        from caitlynrenderer_amd.meshgen import with_disney_materials
        mesh = with_disney_materials(mesh)
        lam = _SCENE_CACHE.get((name, builder, convert, "lambert"))
        if lam is not None:
            # same geometry, same tree: only the material table and the triangles' material ids change
            data0, cam0, label0, build_s = lam
            if name != "cornell":
                mesh = tessellated_cornell(mesh, 183 if name == "mesh1m" else int(name[4:]))
            data = copy.copy(data0)
            data.materials = mesh.materials
            data.triangles = data0.triangles.copy()
            data.triangles[:, 3] = mesh.triangles[data0.tri_orig_ids, 3]
            label = label0 + ", mirror tall box + GGX/Disney-diffuse short box and floor (oracle-defined materials)"
            _SCENE_CACHE[key] = (data, cam0, label, build_s)
            return _SCENE_CACHE[key]
    label = "cornell-box 32 tris (Models/cornell-box.obj), CWBVH"
    if name != "cornell":
        n = 183 if name == "mesh1m" else int(name[4:])
        mesh = tessellated_cornell(mesh, n)
        label = f"procedural tessellated Cornell n={n}: {mesh.triangles.shape[0]} tris, CWBVH"
    if materials == "disney":
        label += ", mirror tall box + GGX/Disney-diffuse short box and floor (oracle-defined materials)"
    t0 = time.time()
    data = cr.SceneData.build(mesh, cam, builder=builder, convert=convert)
    if builder == "lbvh":
        label += " over a GPU-built LBVH"
    elif builder.startswith("ploc"):
        label += f" over a GPU-built PLOC tree ({builder})"
    if convert == "device":
        label += ", CWBVH converted on the GPU"
    _SCENE_CACHE[key] = (data, cam, label, time.time() - t0)
    return _SCENE_CACHE[key]


def weak_frame_size(n_gpus):
    if n_gpus == 1:
        return 1920, 1080
    w = int(round(1920 * math.sqrt(n_gpus) / 16.0)) * 16
    return w, int(round(w * 9 / 16))


class Ctx:
    """Rank / world / collective plumbing shared by every block of one run."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.use_dist = "RANK" in os.environ       # launched by torch.distributed.run (also at N = 1: same code path)
        self.device = "cpu" if args.dry_run else "cuda"

    def init(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if not self.args.dry_run:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs a GPU: the traversal path has no CPU fallback")
            torch.cuda.set_device(self.local_rank)
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.args.dry_run:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local_rank))

    def barrier(self, scene=None):
        if self.use_dist:
            self.dist.barrier()
        if not self.args.dry_run:
            self.torch.cuda.synchronize()
        if scene is not None:
            scene.sync()

    def max_over_ranks(self, x):
        if not self.use_dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if not self.use_dist:
            return x
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.use_dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


_LIVE_PMC = {}          # "<workload>_d<depth>" -> entry measured by live_pmc() in this run


def pmc_entry(workload, depth):
    """Counter figures of this workload's dominant kernel: measured in this run when live_pmc() ran (source "live"), otherwise what
    the committed rocprofv3 --pmc passes say (profiles/pmc_traffic.json, source "committed")."""
    key = f"{workload}_d{depth}"
    if key in _LIVE_PMC:
        return dict(_LIVE_PMC[key], source="live")
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        e = json.load(open(tpath)).get(key, {})
        return dict(e, source="committed") if e else {}
    except Exception:
        return {}


def live_pmc(workloads, budget_s=240.0):
    """The hardware-counter passes of THIS run: for each (workload, depth) four `rocprofv3 --pmc <group> -- python3 bench.py
    --workload ... --no-live-pmc` children (separate passes per counter group, as MI355X_MICROARCH.md prescribes), started
    before this process has touched the GPU.  Counters perturb timing, so the children's own throughput is discarded; what is
    kept is bytes and instruction counts per launch.  Any failure leaves the committed figures in place."""
    import importlib.util
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rocprof is None:
        log("[bench] live pmc: rocprofv3 not found, using the committed passes")
        return
    import torch  # noqa: F401  (no GPU call: only pages the library in, so that the children's own imports fit their time limit on a fresh box)
    spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(ROOT, "tools", "pmc_traffic.py"))
    pt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pt)
    t_start = time.time()
    top = tempfile.mkdtemp(prefix="crt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for name, depth, spp in workloads:
            key, dirs, ok = f"{name}_d{depth}", {}, True
            for kind, counters in pt.PASSES.items():
                if time.time() - t_start > budget_s:
                    ok = False
                    log(f"[bench] live pmc: time budget used up before {key}/{kind}")
                    break
                d = os.path.join(top, f"pmc_{kind}_{key}")
                cmd = [rocprof, "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--gpus", "1",
                       "--workload", name, "--depth", str(depth), "--spp", str(spp), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-live-pmc"]
                try:
                    run = subprocess.run(cmd, cwd=top, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=90)
                except subprocess.TimeoutExpired:
                    run = None
                if run is None or run.returncode != 0:
                    ok = False
                    log(f"[bench] live pmc: pass {kind} of {key} failed ({'timeout' if run is None else 'rc %d' % run.returncode}), "
                        "no further passes: " + ("" if run is None else run.stderr[-300:]))
                    return                                           # counters do not work here: do not spend minutes finding out twice
                dirs[kind] = d
            if ok:
                e = pt.entry_from_dirs(dirs, key)
                if e and "l2_fabric_bytes_per_launch" in e:
                    e["samples_per_launch"] = spp                              # what one launch of the pass rendered (crt_render_frames)
                    _LIVE_PMC[key] = e
                    log(f"[bench] live pmc {key}: {e['l2_fabric_bytes_per_launch']} B/launch L2<->fabric, valu_issue {e.get('valu_issue')}")
    finally:
        shutil.rmtree(top, ignore_errors=True)


def run_block(ctx, name, W, H, depth, spp, sharded, cpu_base, scaling, materials=None, device_built=None):
    """Measure one workload: returns the dict of the bench line for it (rank 0) or None (other ranks).
    device_built = "lbvh" | "ploc<r>" | "sah": the scene is built by crt_scene_create itself from the source-order arrays
    (BVH2, CWBVH and records produced in HBM) instead of from host-built trees."""
    import numpy as np
    import caitlynrenderer_amd as cr
    from caitlynrenderer_amd import tiles
    args, torch = ctx.args, ctx.torch
    rank, world = (ctx.rank, ctx.world) if sharded else (0, 1)
    takes_part = sharded or ctx.rank == 0
    use_dist = ctx.use_dist and sharded
    K, Wu = args.steps, args.warmup

    out = None
    if takes_part:
        data, cam, label, build_s = build_workload(name, args.builder, args.convert, materials or args.materials)
        build_info = None
        if device_built:
            from caitlynrenderer_amd.meshgen import tessellated_cornell
            import __graft_entry__ as g
            mesh, _ = g._cornell()
            if name != "cornell":
                mesh = tessellated_cornell(mesh, 183 if name == "mesh1m" else int(name[4:]))
            t_b = time.perf_counter()
            scene = cr.Scene(cr.SceneData.for_device_build(mesh, cam, builder=device_built), W, H, depth)
            bi = scene.bvh_info()
            build_info = {"builder": device_built, "scene_create_wall_ms": round((time.perf_counter() - t_b) * 1e3, 2),
                          "upload_ms": round(bi["build_upload_ms"], 2), "bvh2_device_ms": round(bi["build_lbvh_device_ms"], 2),
                          "cwbvh_device_ms": round(bi["build_convert_device_ms"], 2)}
            label = label.split(",")[0] + f", CWBVH over a BVH2 built on the GPU ({device_built}), everything assembled in HBM by crt_scene_create"
            build_s = build_info["scene_create_wall_ms"] / 1e3
        else:
            scene = cr.Scene(data, W, H, depth)
        scene.set_shard(rank, world, args.tile)
        for kv in args.option:
            k, v = kv.split("=")
            scene.set_option(k, int(v))
        if args.accel == "bvh2":
            scene.set_option("accel", 1)
            label = label.replace("CWBVH", "BVH2 walked as the shipped shader does")
        info = scene.bvh_info()
        if ctx.rank == 0:
            log(f"[bench] {label}; {W}x{H}, depth {depth}, {spp} spp/step, {world} rank(s); BVH build {build_s:.1f}s; "
                f"{info['n_nodes8']} node8, {info['n_tris8']} tris, depth {info['max_depth8']}")
        rnd = cr.Rnd()
        rvs = [(rnd.randf2(), rnd.randf2()) for _ in range((Wu + K) * spp + 1)]

        # ---- untimed: algorithmic bytes of one frame (visit counters from the counting kernels) ----
        scene.set_option("count_visits", 1)
        scene.render_frame(*rvs[0])
        cs = scene.frame_stats()
        scene.set_option("count_visits", 0)
        scene.reset()

        _, tile, n_floats = scene.packed_info()
        gather_buf = recv = None
        if use_dist:
            gather_buf = torch.zeros(tiles.max_local_tiles(W, H, tile, world) * tile * tile * 3, dtype=torch.float32, device="cuda")
            recv = torch.empty(world * gather_buf.numel(), dtype=torch.float32, device="cuda") if rank == 0 else None

        def read_back():
            """RCCL gather over xGMI of the per-tile radiance to rank 0 (at read-back only)."""
            scene.copy_packed_device(gather_buf.data_ptr(), n_floats)
            tiles.gather_packed_to_root(gather_buf, recv, world)

        def step(k):
            """one step = spp samples per pixel; crt_render_frames shares launches among them where the path allows
            (one segment, shadow rays in place: up to 8 samples per launch), with the same sums bit for bit"""
            part = rvs[1 + k * spp:1 + (k + 1) * spp]
            if spp > 1:
                scene.render_frames(part, sync=False)
            else:
                scene.render_frame(*part[0], sync=False)

        # settle: the GPU comes out of the idle period in which this workload was built with its clocks down, and frame times keep
        # falling for the first ~150 frames (Cornell: 0.090 -> 0.075 ms); the metric is the steady state of a progressive renderer, so
        # untimed frames are rendered for --settle-ms of wall time before the W warm-up steps (which then also run at speed)
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            for k in range(max(1, Wu)):
                step(k % max(1, Wu))
            scene.sync()
        for k in range(Wu):
            step(k)
        scene.sync()
        if use_dist:    # warm the collective too
            read_back()
        # HIP events on every segment launch of the timed region, on the scene's own stream (attached to the dispatch:
        # they take the kernel's own start/stop timestamps)
        scene.set_option("timing", 1)
        scene.set_option("timing_accumulate", K * spp * max(1, depth))
    if sharded:
        ctx.barrier(scene)
    elif takes_part:
        torch.cuda.synchronize(); scene.sync()
    if takes_part:
        t0 = time.perf_counter()
        for k in range(K):
            step(Wu + k)
        scene.sync()
        if use_dist:
            read_back()
    if sharded:
        ctx.barrier(scene)
    if takes_part:
        if not sharded:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if sharded:
            dt = ctx.max_over_ranks(dt)

        # ray counts of the LAST timed frame (every frame does identical work up to the per-frame random vector) and
        # the mean launch time over the timed region; then a short event-timed tail for the per-kernel split
        st = scene.frame_stats()
        launch_ms_timed = st["ms_trace_closest"] / max(1, st["n_trace_launches"])
        n_timed_launches = st["n_trace_launches"]
        # samples per pixel one launch rendered: 1, or the step's spp where crt_render_frames batched them
        samples_per_launch = max(1, round(K * spp * max(1, depth) / max(1, n_timed_launches)))
        scene.set_option("timing_accumulate", 0)
        scene.set_option("timing", 2)
        any_ms, total_ms = [], []
        for i in range(min(10, K * spp)):
            scene.render_frame(*rvs[1 + Wu * spp + i])
            s = scene.frame_stats()
            any_ms.append(s["ms_trace_any"] / max(1, depth))
            total_ms.append(s["ms_total"])
        rays_frame = (st["closest_rays"] + st["any_rays"]) / samples_per_launch      # the stats describe the last launch
        rays_all = ctx.sum_over_ranks(rays_frame) if sharded else float(rays_frame)
        value = rays_all * K * spp / dt / 1e6

    if takes_part and ctx.rank == 0:
        launches = max(1, depth)
        node_bytes = 96 if args.accel == "bvh2" else NODE_BYTES      # own 2 texels + 4 child texels per BVH2 visit
        alg = (node_bytes * cs["nodes_closest"] + TRI_BYTES * cs["tris_closest"]) / launches * samples_per_launch
        # the segment kernel also walks the NEE shadow rays in place (no k_shadow launch): their visits are this launch's bytes too
        fused_shadow = st["any_rays"] > 0 and float(np.median(any_ms)) == 0.0
        if fused_shadow:
            alg += (node_bytes * cs["nodes_any"] + TRI_BYTES * cs["tris_any"]) / launches * samples_per_launch
        t_launch = launch_ms_timed * 1e-3
        achieved = alg / t_launch / 1e9 if t_launch > 0 else 0.0
        if fused_shadow:
            kernel_label = "k_segment (raygen / queue fetch + CWBVH closest hit + shading + in-place NEE any-hit walk), mean per path segment"
        else:
            kernel_label = "k_segment (raygen / queue fetch + CWBVH closest hit + shading + queue emission), mean per path segment"
        # the counter passes run the host-built tree, Lambert, 1920x1080: other blocks carry no counter figures of their own
        pmc = pmc_entry(name, depth) if ((W, H) == (1920, 1080) and not device_built and materials in (None, "lambert")) else {}
        traffic = pmc.get("l2_fabric_bytes_per_launch", pmc.get("hbm_bytes_per_launch"))
        if traffic is not None:      # a pass that rendered fewer samples per launch than this block's launches: scaled to the same unit
            traffic = int(traffic * samples_per_launch / max(1, pmc.get("samples_per_launch", 1)))
        roofline = {
            "bound": "hbm", "kernel": kernel_label, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "traffic_source": pmc.get("source"),
            "traffic_is": "L2<->fabric bytes per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024 from rocprofv3 --pmc passes of this workload ("
                          + ("run by this bench.py invocation as child processes before its own measurements" if pmc.get("source") == "live"
                             else "the committed ones, profiles/pmc_traffic.json")
                          + "); the scene sits in the 256 MiB Infinity Cache, so true HBM bytes are lower still",
            "limiter": "valu_issue",
            "valu_issue": pmc.get("valu_issue"),
            "algorithmic_bytes_per_launch": int(alg),
            "bytes_per_ray": round(alg / samples_per_launch / max(1, (cs["closest_rays"] + (cs["any_rays"] if fused_shadow else 0)) / launches), 2),
            "nodes_per_ray": round(cs["nodes_closest"] / max(1, cs["closest_rays"]), 3),
            "tris_per_ray": round(cs["tris_closest"] / max(1, cs["closest_rays"]), 3),
            "any_hit_nodes_per_ray": round(cs["nodes_any"] / max(1, cs["any_rays"]), 3),
            "any_hit_tris_per_ray": round(cs["tris_any"] / max(1, cs["any_rays"]), 3),
            "launch_ms": round(t_launch * 1e3, 4), "launches_timed": int(n_timed_launches), "samples_per_launch": int(samples_per_launch),
            "any_hit_launch_ms": round(float(np.median(any_ms)), 4),
            "frame_device_ms": round(float(np.median(total_ms)), 4),
            "note": ("frac = SURVEY §8d algorithmic bytes / launch time / HBM peak; the kernel itself is bound by VALU issue (valu_issue.frac = "
                     "issue slots busy x lanes enabled, from the PMC pass), not by HBM"
                     + ("; this tree of %d nodes lives in L1, so the algorithmic-bytes rate says nothing about the memory system" % info["n_nodes8"]
                        if info["n_nodes8"] < 64 else "")),
        }
        out = {
            "value": round(value, 2), "unit": "Mray/s", "ms_per_step": round(dt / K * 1e3, 4), "scaling": scaling,
            "config": {"workload": label, "resolution": f"{W}x{H}", "spp_per_step": spp, "path_segments": depth,
                       "rays_per_step": int(rays_all) * spp, "closest_rays_rank0": int(st["closest_rays"] // samples_per_launch),
                       "any_rays_rank0": int(st["any_rays"] // samples_per_launch), "tile": tile, "parallelism": f"tiles/{world}", "settle_ms": args.settle_ms,
                       "stack_overflows": int(st["stack_overflows"]),
                       "gather": "one RCCL gather of the packed tiles to rank 0 per timed region" if use_dist else "none"},
            "roofline": roofline,
        }
        if build_info:
            out["config"]["device_build"] = build_info
        if cpu_base and not args.no_cpu_baseline and args.accel == "cwbvh":
            out["cpu_baseline"] = cpu_baseline(data, cam, W, H, depth, rvs[0], cs)
    if takes_part:
        scene.close()
    return out


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


def dry_block(ctx, W, H, spp, scaling):
    """--dry-run: everything around the rendering — shard bookkeeping, the gather of packed tile buffers of the real
    size over the process group, barrier + max-over-ranks timing — with no GPU and no rendering."""
    from caitlynrenderer_amd import tiles
    torch, args = ctx.torch, ctx.args
    tile = args.tile
    cap = tiles.max_local_tiles(W, H, tile, ctx.world) * tile * tile * 3
    mine = tiles.local_tiles(W, H, tile, ctx.rank, ctx.world)
    send = torch.full((cap,), float(ctx.rank + 1), dtype=torch.float32)
    recv = torch.empty(ctx.world * cap, dtype=torch.float32) if ctx.rank == 0 else None
    ctx.barrier()
    t0 = time.perf_counter()
    if ctx.use_dist:
        tiles.gather_packed_to_root(send, recv, ctx.world)
    ctx.barrier()
    dt = ctx.max_over_ranks(time.perf_counter() - t0)
    n_tiles = ctx.sum_over_ranks(len(mine))
    if ctx.rank != 0:
        return None
    if ctx.use_dist:
        got = recv.view(ctx.world, cap)
        assert all(float(got[r][0]) == r + 1 and float(got[r][-1]) == r + 1 for r in range(ctx.world)), "gather delivered the wrong slices"
    assert int(n_tiles) == len(tiles.tile_order(W, H, tile)), "the shards do not cover the frame exactly once"
    return {"value": 0.0, "unit": "Mray/s", "ms_per_step": round(dt * 1e3, 4), "scaling": scaling, "dry_run": True,
            "config": {"workload": "dry run: no rendering", "resolution": f"{W}x{H}", "spp_per_step": spp, "tile": tile,
                       "tiles": int(n_tiles), "parallelism": f"tiles/{ctx.world}", "gather_floats_per_rank": cap},
            "roofline": None}


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))

    # stdout carries exactly ONE JSON line: anything a library prints there (RCCL's version banner at
    # communicator creation, for one) is sent to stderr instead
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if (args.gpus == 1 and "RANK" not in os.environ and args.workload == "auto" and args.accel == "cwbvh"
            and not (args.dry_run or args.no_live_pmc or args.option)):
        # hardware counters of this very run, from child processes, before this process makes its first GPU call
        live_pmc([("cornell", args.depth, args.spp or 1)] + ([] if args.no_extra else [("mesh1m", 1, 4), ("mesh1m", 4, 4)]))

    ctx = Ctx(args)
    if ctx.world != args.gpus:
        args.gpus = ctx.world
    ctx.init()
    if not args.dry_run:
        import __graft_entry__ as g
        g.build()

    N = ctx.world
    auto = args.workload == "auto"
    if N == 1:
        name = "cornell" if auto else args.workload
        scaling = "weak"              # one GPU: per-GPU work is what it is
        W, H = 1920, 1080
    elif args.scaling == "strong":
        name = "mesh1m" if auto else args.workload
        scaling = "strong"
        W, H = 3840, 2160
    else:
        name = "cornell" if auto else args.workload
        scaling = "weak"
        W, H = weak_frame_size(N)
    if args.resolution:
        W, H = (int(x) for x in args.resolution.lower().split("x"))
    spp = args.spp or (1 if name == "cornell" else 4)

    if args.dry_run:
        head = dry_block(ctx, W, H, spp, scaling)
        extra = {}
    else:
        head = run_block(ctx, name, W, H, args.depth, spp, True, N == 1, scaling)
        extra = {}
        if auto and not args.no_extra and args.accel == "cwbvh":
            if N == 1:
                extra["north_star"] = run_block(ctx, "mesh1m", 1920, 1080, 1, 4, True, True, "weak")
                extra["north_star_gpu_tree"] = run_block(ctx, "mesh1m", 1920, 1080, 1, 4, True, False, "weak", device_built="sah")
                extra["incoherent"] = run_block(ctx, "mesh1m", 1920, 1080, 4, 4, True, False, "weak")
                extra["incoherent_disney"] = run_block(ctx, "mesh1m", 1920, 1080, 4, 4, True, False, "weak", materials="disney")
                extra["scale_base"] = run_block(ctx, "mesh1m", 3840, 2160, 1, 4, True, False, "strong")
                extra["cornell_8_frames_per_launch"] = run_block(ctx, "cornell", 1920, 1080, 1, 8, True, False, "weak")
            elif args.scaling == "strong":
                ctx.barrier()
                extra["n1_same_workload"] = run_block(ctx, name, W, H, args.depth, spp, False, False, "strong")
                ctx.barrier()

    if ctx.rank == 0:
        out = {"metric": METRIC, "value": head["value"], "unit": head["unit"], "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": head["scaling"],
               "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": head["config"], "roofline": head["roofline"]}
        if "cpu_baseline" in head:
            out["cpu_baseline"] = head["cpu_baseline"]
        if head.get("dry_run"):
            out["dry_run"] = True
        descr = {"north_star": "BASELINE.json configs[2] — the workload its targets (>= 1 Gray/s, >= 50 % HBM roofline) are quoted on",
                 "north_star_gpu_tree": "configs[2] again, over a tree built on the GPU: crt_scene_create with CRT_BUILD_LBVH_ON_DEVICE | CRT_BUILD_SAH "
                                        "(binned-SAH BVH2, CWBVH conversion and records all in HBM; config.device_build has the times) instead of the host SBVH",
                 "incoherent": "BASELINE.json configs[3] ray mix with the reference's own (Lambert-only) integrator — 4 path segments on the same mesh, "
                               "4 spp per step as in configs[2] (the step's four frames share each segment's launch: crt_render_frames)",
                 "incoherent_disney": "BASELINE.json configs[3] as worded: 4 path segments with a mirror tall box and GGX / Disney-diffuse short "
                                      "box and floor (the material model has no reference code: oracle-defined, HIP == oracle bit for bit)",
                 "scale_base": "BASELINE.json configs[4] at N = 1: what the N > 1 lines of `bench.py --gpus N` divide by",
                 "cornell_8_frames_per_launch": "NOT the headline: configs[1]'s scene with 8 frames handed to crt_render_frames per step, i.e. one launch per 8 "
                                                "samples — what the top-level line (one launch per frame, as the reference's frame loop issues them) leaves "
                                                "in launch gaps and kernel tails",
                 "n1_same_workload": "the same frame rendered by rank 0 alone in this job (strong-scaling base)"}
        for k, v in extra.items():
            if v is not None:
                v["what"] = descr[k]
                out[k] = v
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ctx.close()


if __name__ == "__main__":
    main()
