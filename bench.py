#!/usr/bin/env python3
"""bench.py — Mray/s of the HIP ray/BVH-traversal hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--workload auto|cornell|mesh1m|meshN] [--depth D] [--resolution WxH] [--spp S] [--scaling strong|weak]

One step = one pass of the hot path over one batch of the workload: `spp_per_step` samples per pixel of this rank's
tiles (ray generation -> CWBVH closest hit -> shading / NEE -> CWBVH any hit -> accumulate), everything resident in HBM.

Workloads (BASELINE.json `configs`, made concrete in SURVEY.md §8d):
  N = 1, --workload auto (the default):
    value / roofline / cpu_baseline  = configs[2]: the 1,004,672-triangle mesh, CWBVH, 4 spp per step, 1920x1080, primary + shadow —
                                       the workload BASELINE.json's targets (>= 1 Gray/s, >= 50 % roofline) are quoted on;
    "extras" (compact: value, ms_per_step, launch_ms, frac, algorithmic_gbps, traffic):
      "cornell"            configs[1]: Cornell box, CWBVH, 1 spp primary + shadow, 1920x1080 (one launch per frame);
      "gpu_tree"           configs[2] over a tree built on the GPU (binned SAH, everything assembled in HBM by crt_scene_create);
      "d2"                 configs[2]'s mesh with 2 path segments: the other reading of the metric's "primary + 1 bounce" (the headline is
                           primary + shadow, SURVEY 8d's reading);
      "incoherent"         configs[3]: same mesh, 4 path segments (incoherent bounce rays), the reference's Lambert integrator;
      "incoherent_disney"  configs[3] with the oracle-defined mirror + GGX / Disney-diffuse materials;
      "scale_base"         configs[4] at N = 1: the 3840x2160 frame of that mesh on one GPU (what the N > 1 lines divide by);
      "hbm_resident"       the same scene tessellated to ~8 M triangles (n = 520): records + nodes exceed the 256 MiB Infinity
                           Cache, so its traffic figure is HBM traffic (1 segment and 4 segments).
  N > 1, --workload auto: configs[4]: ONE fixed 3840x2160 frame of the 1 M-triangle mesh, 4 spp per step, its 16x16 tiles dealt
    to the N ranks (strong scaling, no data-path collective), one RCCL gather of the per-tile radiance to rank 0 inside the timed
    region; the line carries gather_ms, per-rank device time, the same frame rendered by rank 0 alone and the efficiency against it.
    `--scaling weak` keeps the round-1 behaviour (the frame grows with N, ~1920x1080 pixels per rank).
  An explicit --workload measures just that one (used by the profiling scripts).

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts its N ranks itself, as fresh child processes
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`) BEFORE anything touches the GPU,
forwards rank 0's JSON line and exits with the children's status; launched under torch.distributed.run by someone else it
simply is one of the ranks.

Prints ONE JSON line (< 8 KB) on rank 0 with the driver's keys plus
  "roofline": the dominant kernel (the fused segment kernel) against the roof that binds it — VECTOR-INSTRUCTION ISSUE, not HBM (every
      BASELINE scene is cache-resident: SURVEY 8d's algorithmic bytes / time exceeds the HBM peak and is reported as `algorithmic_gbps`,
      never as a fraction).  achieved = TRAVERSAL wave-instructions of a launch — (node visits x I_node + triangle tests x I_tri) / 64,
      straight-line blocks counted in the ISA — / its mean HIP-event duration (blocks that render tile shards side by side on several
      streams — `config.streams` > 1 — take the step's wall time / its launches per shard instead: the time a segment of the WHOLE
      frame takes); peak = 1024 SIMDs x 2.4 GHz / 2 cycles.  `issue_busy`, `lane_util` and their product `counter_frac` come from
      rocprofv3 --pmc passes this invocation runs itself as child processes before its own first GPU call (`traffic_source: "live"`;
      --no-live-pmc or a missing rocprofv3: the committed passes of profiles/pmc_traffic.json, "committed"); frac <= counter_frac by
      construction, `non_traversal_share` = 1 - frac / counter_frac is the ray's shell (ray generation, shading, NEE, bounce
      sampling) plus the loop's bookkeeping.  `traffic` = L2<->fabric bytes per launch, (2 FETCH_SIZE + WRITE_SIZE) x 1024 from the
      same passes; `hbm_frac` = traffic / time / 8 TB/s (HBM traffic proper only for the scene that exceeds the Infinity Cache).
      Definitions and the script that recomputes every figure: tools/roofline.py, profiles/isa_counts.json.
  "cpu_baseline": the CPU oracle on the same workload, bounded sample, rank 0, N = 1.
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

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, MI355X_MICROARCH.md "Chip-level parameters" (algorithmic_gbps is quoted against it in DESIGN.md only)
VALU_PEAK_GINSTR = 1024 * 2.4 / 2.0   # G wave64-instructions/s: 1024 SIMD-32s, one wave64 VALU instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)
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
    ap.add_argument("--builder", default="auto",
                    help="auto (default) = the reference's split-BVH on the host at N = 1, the binned-SAH tree crt_scene_create builds on each rank's GPU "
                         "at N > 1 (9.5 ms instead of every rank building the SBVH on the shared host cores; 17.19 against 17.21 Gray/s); "
                         "sbvh = the host SBVH everywhere; lbvh = GPU linear BVH (crt_lbvh_build); ploc / ploc<radius> = GPU parallel locally-ordered clustering")
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
    ap.add_argument("--no-oracle-check", action="store_true",
                    help="skip `sum_rows_match_oracle` (after the timed region every block renders one more step in the timed form and compares "
                         "16 rows of the sum with the CPU oracle)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="auto workload at N = 1: do not run the rocprofv3 --pmc passes (roofline.traffic / valu_issue then come from the "
                         "committed profiles/pmc_traffic.json and say so)")
    ap.add_argument("--device-built", default=None, metavar="BUILDER",
                    help="explicit workload: crt_scene_create builds the tree itself on the GPU from the source-order arrays (lbvh, ploc<r>, sah)")
    ap.add_argument("--no-extra", action="store_true", help="auto workload: only the headline block")
    ap.add_argument("--no-hbm-resident", action="store_true", help="auto workload: skip the 8 M-triangle (> Infinity Cache) blocks")
    ap.add_argument("--one-process", action="store_true",
                    help="N GPUs behind ONE scene handle in this one process (crt_set_devices: replication, tile sharding and the RCCL gather "
                         "inside the C ABI) instead of one process per GPU under torch.distributed; the timed region ends with crt_sum_device")
    ap.add_argument("--virtual-devices", type=int, default=0, metavar="K",
                    help="with --one-process: K logical devices on GPU 0 (shards, streams and gather buffers as on K GPUs; copies instead of RCCL) — "
                         "rehearses the path on a one-GPU machine; the number it prints is NOT a scaling figure")
    ap.add_argument("--streams", type=int, default=0, metavar="K",
                    help="tile shards of a rank's frame rendered side by side on K streams of its GPU (crt_set_option streams); default 0 = the "
                         "library's pick: 3 for scenes of a few nodes, 2 for multi-segment paths, else 1")
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
    env.setdefault("CRT_BUILD_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))      # every rank builds the SBVH: share the cores (host/sbvh.cpp usable_threads)
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


def build_workload(name, builder="sbvh", convert="host", materials="lambert", sbvh_flags=0):
    key = (name, builder, convert, materials) if not sbvh_flags else (name, builder, convert, materials, sbvh_flags)
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
    data = cr.SceneData.build(mesh, cam, builder=builder, convert=convert, sbvh_flags=sbvh_flags) if sbvh_flags else cr.SceneData.build(mesh, cam, builder=builder, convert=convert)
    if sbvh_flags:
        label += " over the SAH BVH (exact-sweep object splits only, sbvh.h:338-378)"
    if builder == "lbvh":
        label += " over a GPU-built LBVH"
    elif builder.startswith("ploc"):
        label += f" over a GPU-built PLOC tree ({builder})"
    if convert == "device":
        label += ", CWBVH converted on the GPU"
    _SCENE_CACHE[key] = (data, cam, label, time.time() - t0)
    return _SCENE_CACHE[key]


_MESH_CACHE = {}


def source_mesh(name):
    """(mesh in source order, camera) of a workload name: cornell, mesh1m (n = 183) or mesh<n>"""
    if name not in _MESH_CACHE:
        import __graft_entry__ as g
        from caitlynrenderer_amd.meshgen import tessellated_cornell
        mesh, cam = g._cornell()
        if name != "cornell":
            mesh = tessellated_cornell(mesh, 183 if name == "mesh1m" else int(name[4:]))
        _MESH_CACHE[name] = (mesh, cam)
    return _MESH_CACHE[name]


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
        # --one-process: the N devices sit behind one scene handle in this process (the gather is crt_sum_device's business)
        self.one_proc_ids = None
        if args.one_process and not args.dry_run:
            self.one_proc_ids = [0] * args.virtual_devices if args.virtual_devices > 0 else list(range(max(1, args.gpus)))
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

    def gather_floats(self, x):
        """[x of rank 0, x of rank 1, ...] on every rank"""
        if not self.use_dist:
            return [float(x)]
        t = self.torch.zeros(self.world, dtype=self.torch.float64, device=self.device)
        t[self.rank] = float(x)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

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


def live_pmc(workloads, budget_s=330.0):
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
        for name, depth, spp, *more in workloads:
            key, dirs, ok = f"{name}_d{depth}", {}, True
            for kind, counters in pt.PASSES.items():
                if time.time() - t_start > budget_s:
                    ok = False
                    log(f"[bench] live pmc: time budget used up before {key}/{kind}")
                    break
                d = os.path.join(top, f"pmc_{kind}_{key}")
                cmd = [rocprof, "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--gpus", "1",
                       "--workload", name, "--depth", str(depth), "--spp", str(spp), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-live-pmc",
                       "--settle-ms", "0", "--streams", "1"] + [str(x) for x in more]     # one stream: a dispatch is a whole segment of the frame
                try:
                    run = subprocess.run(cmd, cwd=top, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=150)
                except subprocess.TimeoutExpired:
                    run = None
                if run is None or run.returncode != 0:
                    ok = False
                    log(f"[bench] live pmc: pass {kind} of {key} failed ({'timeout' if run is None else 'rc %d' % run.returncode}), "
                        "no further passes: " + ("" if run is None else run.stderr[-300:]))
                    return                                           # counters do not work here: do not spend minutes finding out twice
                dirs[kind] = d
            if ok:
                e = pt.entry_from_dirs(dirs, key, tail=min(10, 5 * spp) * depth)     # without the children's own single-sample tail frames
                if e and "l2_fabric_bytes_per_launch" in e:
                    e["samples_per_launch"] = spp                              # what one launch of the pass rendered (crt_render_frames)
                    _LIVE_PMC[key] = e
                    log(f"[bench] live pmc {key}: {e['l2_fabric_bytes_per_launch']} B/launch L2<->fabric, valu_issue {e.get('valu_issue')}")
    finally:
        shutil.rmtree(top, ignore_errors=True)


def run_block(ctx, name, W, H, depth, spp, sharded, cpu_base, scaling, materials=None, device_built=None, accel=None, sbvh_flags=0):
    """Measure one workload: returns the dict of the bench line for it (rank 0) or None (other ranks).
    device_built = "lbvh" | "ploc<r>" | "sah": the scene is built by crt_scene_create itself from the source-order arrays
    (BVH2, CWBVH and records produced in HBM) instead of from host-built trees."""
    import numpy as np
    import caitlynrenderer_amd as cr
    from caitlynrenderer_amd import tiles
    args, torch = ctx.args, ctx.torch
    accel = accel or args.accel
    rank, world = (ctx.rank, ctx.world) if sharded else (0, 1)
    takes_part = sharded or ctx.rank == 0
    use_dist = ctx.use_dist and sharded
    K, Wu = args.steps, args.warmup

    out = None
    if takes_part:
        build_info = None
        if device_built:
            # crt_scene_create gets the source-order arrays only and builds BVH2, CWBVH and records in HBM: no host tree at all
            mesh, cam = source_mesh(name)
            data = None
            t_b = time.perf_counter()
            scene = cr.Scene(cr.SceneData.for_device_build(mesh, cam, builder=device_built), W, H, depth)
            bi = scene.bvh_info()
            build_info = {"builder": device_built, "scene_create_wall_ms": round(scene.create_ms, 2),      # wall time of the crt_scene_create call
                          "upload_ms": round(bi["build_upload_ms"], 2), "bvh2_device_ms": round(bi["build_lbvh_device_ms"], 2),
                          "cwbvh_device_ms": round(bi["build_convert_device_ms"], 2)}
            label = (f"procedural tessellated Cornell n={183 if name == 'mesh1m' else name[4:]}: {mesh.triangles.shape[0]} tris" if name != "cornell" else "cornell-box 32 tris") \
                + f", CWBVH over a BVH2 built on the GPU ({device_built}), everything assembled in HBM by crt_scene_create"
            build_s = build_info["scene_create_wall_ms"] / 1e3
        else:
            data, cam, label, build_s = build_workload(name, "sbvh" if args.builder == "auto" else args.builder, args.convert, materials or args.materials, sbvh_flags)
            scene = cr.Scene(data, W, H, depth)
        one_proc = ctx.one_proc_ids if sharded else None     # N devices behind this one handle (crt_set_devices) instead of N ranks
        if one_proc:
            scene.set_devices(one_proc, args.tile)
            world = len(one_proc)
        else:
            scene.set_shard(rank, world, args.tile)
        # a multi-segment frame is a chain of dependent launches: two tile shards of this rank's part of the frame on two streams of the GPU
        # fill each other's launch tails (option "streams", bit-identical; 1 M triangles, 4 segments: +6 % on the whole frame, +12 % on an
        # eighth of the 4K frame).  One launch per step gains nothing on a whole frame and 0..2 % on a shard.
        # ... and a frame of a few-node scene (the 32-triangle Cornell box: 58 us per launch) is bound by the gaps between launches: three
        # tile shards on three streams keep the GPU busy through them (+13 %; four streams are bound by the host's launch rate)
        streams = 1
        if not one_proc and accel == "cwbvh":
            scene.set_option("streams", args.streams)            # 0 (the default here): the library's own pick, as described above
            streams = len(scene.devices()["devices"])
        for kv in args.option:
            k, v = kv.split("=")
            scene.set_option(k, int(v))
        if accel == "bvh2":
            scene.set_option("accel", 1)
            label = label.replace("CWBVH", "BVH2 walked as the shipped shader does")
        info = scene.bvh_info()
        if ctx.rank == 0:
            log(f"[bench] {label}; {W}x{H}, depth {depth}, {spp} spp/step, {world} rank(s); BVH build {build_s:.1f}s; "
                f"{info['n_nodes8']} node8, {info['n_tris8']} tris, depth {info['max_depth8']}")
        rnd = cr.Rnd()
        rvs = [(rnd.randf2(), rnd.randf2()) for _ in range((Wu + K) * spp + 1)]

        # ---- untimed: the visit counters of ONE STEP (all its samples and segments), from counting kernels in the form the timed launches
        # have (count_visits 2: four samples in the lanes of a wave where the timed step runs so; the uniform node steps depend on which
        # rays share a wave).  A step of k launches is counted launch by launch.
        scene.set_option("count_visits", 2)
        cs = None
        first = rvs[1 + Wu * spp:1 + (Wu + 1) * spp]            # the frames of the first timed step
        parts = [first[i:i + 4] for i in range(0, spp, 4)] if spp % 4 == 0 else [[r] for r in first]
        for part in parts:
            if len(part) == 4:
                scene.render_frames(part)        # one counting launch per segment in the lanes form, or four single frames where that form does not apply
            else:
                scene.render_frame(*part[0])
            st_c = scene.frame_stats()
            keys = ("closest_rays", "any_rays", "closest_hits", "nodes_closest", "tris_closest", "nodes_any", "tris_any", "nodes_closest_uniform", "nodes_any_uniform")
            cs = {k: int(st_c.get(k, 0)) + (cs[k] if cs else 0) for k in keys}
        scene.set_option("count_visits", 0)
        scene.reset()

        tile, n_floats = args.tile, 0
        if use_dist:
            _, tile, n_floats = scene.packed_info()
        gather_buf = recv = None
        if use_dist:
            gather_buf = torch.zeros(tiles.max_local_tiles(W, H, tile, world) * tile * tile * 3, dtype=torch.float32, device="cuda")
            recv = torch.empty(world * gather_buf.numel(), dtype=torch.float32, device="cuda") if rank == 0 else None

        def read_back():
            """RCCL gather over xGMI of the per-tile radiance to rank 0 (at read-back only)."""
            if one_proc:
                scene.sum_device()           # gather + un-tile inside the C ABI, the frame stays in device 0's memory
                return
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
        if use_dist or one_proc:    # warm the collective too
            read_back()
        # launches one step makes (a probe step with events on its launches): the samples of a step share launches where the path allows
        scene.set_option("timing", 1)
        scene.set_option("timing_accumulate", spp * max(1, depth))
        step(0)
        scene.sync()
        launches_per_step = max(1, int(scene.frame_stats()["n_trace_launches"]))
        scene.set_option("timing_accumulate", 0)
        if streams == 1:
            # HIP events on every segment launch of the timed region, on the scene's own stream (attached to the dispatch:
            # they take the kernel's own start/stop timestamps)
            scene.set_option("timing_accumulate", K * (spp * max(1, depth) + 2))
        else:
            # several streams: the launch time of the roofline is the step's wall time / launches (below), and an event-carrying dispatch
            # would only keep its neighbours on the other streams from overlapping it
            scene.set_option("timing", 0)
    if sharded:
        ctx.barrier(scene)
    elif takes_part:
        torch.cuda.synchronize(); scene.sync()
    gather_ms = 0.0
    if takes_part:
        t0 = time.perf_counter()
        for k in range(K):
            step(Wu + k)
        scene.sync()
        render_s = time.perf_counter() - t0            # this rank's rendering, without the gather
        if use_dist or one_proc:
            read_back()
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - t0 - render_s) * 1e3
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
        launch_info = scene.debug_launch_info()          # how the timed steps' first-segment launches ran (form, build, samples, shards)
        st = scene.frame_stats()
        n_timed_launches = st["n_trace_launches"] if streams == 1 else launches_per_step * K
        n_segment_launches = K * launches_per_step if streams == 1 else n_timed_launches      # without the deferred any-hit launches (event kind 2)
        launch_ms_timed = (st["ms_trace_closest"] + st["ms_trace_any"]) / max(1, n_segment_launches)
        # spread of the timed steps: device time of each step = the sum of its launches' own durations (one stream), else of wall-clock
        # steps rendered one by one after the clock has stopped (several streams carry no events: they would keep the shards from overlapping)
        if streams == 1 and depth == 1:
            lt = scene.launch_times()
            per = max(1, len(lt) // K)
            step_ms = [float(lt[i * per:(i + 1) * per].sum()) for i in range(K)] if len(lt) >= K else []
            spread_src = "device time of each timed step (its launches' events)"
        else:
            step_ms = []
            for k in range(min(K, 20)):
                t1 = time.perf_counter(); step(Wu + k); scene.sync(); step_ms.append((time.perf_counter() - t1) * 1e3)
            spread_src = f"wall time of {len(step_ms)} steps rendered one by one after the timed region"
        spread = ({"min": round(min(step_ms), 4), "median": round(float(np.median(step_ms)), 4), "max": round(max(step_ms), 4), "n": len(step_ms), "what": spread_src}
                  if step_ms else None)
        # samples per pixel one launch rendered: 1, or the step's spp where crt_render_frames batched them
        samples_per_launch = max(1, round(K * spp * max(1, depth) / max(1, n_segment_launches)))
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
        # primary rays of one frame of this rank = its in-frame pixels
        n_primary = W * H if one_proc else sum(min(tile, W - tx * tile) * min(tile, H - ty * tile) for tx, ty in tiles.local_tiles(W, H, tile, rank, world))
        rank_ms = ctx.gather_floats(render_s / K * 1e3) if use_dist else [render_s / K * 1e3]

    if takes_part and ctx.rank == 0:
        launches = max(1, depth)                                      # segment launches per step and sample batch
        node_bytes = 96 if accel == "bvh2" else NODE_BYTES      # own 2 texels + 4 child texels per BVH2 visit
        rl = roofline_module()
        # the counters describe this rank's share of ONE step (spp samples, every segment; deferred shadow rays are that step's work too)
        counters = dict(cs, primary_rays=int(n_primary) * spp)
        launches_per_step = launches * max(1, spp // samples_per_launch)      # segment launches one step is made of
        alg_bytes = rl.algorithmic_bytes(counters, node_bytes) / launches_per_step
        t_launch = launch_ms_timed * 1e-3
        if streams > 1 or depth > 1:
            # the shards' launches run side by side: a segment of the WHOLE frame takes the step's wall time / its launches per shard (a
            # single launch's own duration would count the time it shares the GPU with the other shards' launches more than once); and a
            # multi-segment step is more than its segment launches (the deferred shadow rays' launch, the fold): the step's wall time / segments
            t_launch = dt / max(1.0, K * launches_per_step)
        # counter passes exist per (workload, depth) at 1920x1080 on the host-built tree with the reference's materials
        pmc = pmc_entry(name, depth) if ((W, H) == (1920, 1080) and accel == "cwbvh" and not sbvh_flags and (not device_built or name == HBM_RESIDENT)
                                         and materials in (None, "lambert")) else {}
        traffic = pmc.get("l2_fabric_bytes_per_launch")
        if traffic is not None:      # a pass that rendered fewer samples per launch than this block's launches: scaled to the same unit
            traffic = int(traffic * samples_per_launch / max(1, pmc.get("samples_per_launch", 1)))
        roofline = valu_roofline(counters, t_launch, launches_per_step, samples_per_launch, accel, pmc)
        if one_proc and len(set(one_proc)) < len(one_proc):
            # virtual devices share one GPU: a launch's duration there says nothing about the kernel (the number this mode prints is not a
            # scaling figure either)
            roofline["achieved"] = roofline["frac"] = None
        traffic_gbps = round(traffic / t_launch / 1e9, 1) if traffic and t_launch > 0 else None
        alg_gbps = round(alg_bytes / t_launch / 1e9, 1) if t_launch > 0 else None
        roofline.update({
            "traffic": traffic, "traffic_source": pmc.get("source"),
            # bytes crossing L2 <-> fabric per second; for a scene larger than the 256 MiB Infinity Cache (the hbm_resident blocks) this is
            # HBM bandwidth, for the cache-resident BASELINE scenes mostly MALL hits
            "traffic_gbps": traffic_gbps,
            "hbm_frac": round(traffic_gbps / HBM_PEAK_GBS, 4) if traffic_gbps else None,      # L2<->fabric bytes / s over the HBM peak: HBM traffic proper only when the scene exceeds the Infinity Cache
            "algorithmic_gbps": alg_gbps,
            "algorithmic_over_peak": round(alg_gbps / HBM_PEAK_GBS, 4) if alg_gbps else None,  # SURVEY 8d's bytes / s over the HBM peak: above 1 on every cache-resident scene
            "algorithmic_bytes_per_launch": int(alg_bytes),
            "l2_hit_rate": pmc.get("l2_hit_rate"),
            "launch_ms": round(t_launch * 1e3, 4), "launches_timed": int(n_segment_launches), "samples_per_launch": int(samples_per_launch),
            "path_segments": depth, "frame_device_ms": round(float(np.median(total_ms)), 4),
            "nodes_per_ray": round(cs["nodes_closest"] / max(1, cs["closest_rays"]), 3), "tris_per_ray": round(cs["tris_closest"] / max(1, cs["closest_rays"]), 3),
            "any_hit_nodes_per_ray": round(cs["nodes_any"] / max(1, cs["any_rays"]), 3), "any_hit_tris_per_ray": round(cs["tris_any"] / max(1, cs["any_rays"]), 3),
            "counters": counters,
        })
        if accel == "bvh2":
            roofline["accel"] = "bvh2"
        out = {
            "value": round(value, 2), "unit": "Mray/s", "ms_per_step": round(dt / K * 1e3, 4), "step_ms_spread": spread, "scaling": scaling,
            "config": {"workload": label, "resolution": f"{W}x{H}", "spp_per_step": spp, "path_segments": depth,
                       "rays_per_step": int(rays_all) * spp, "tile": tile, "parallelism": f"tiles/{world}",
                       "n_nodes8": int(info["n_nodes8"]), "n_tris8": int(info["n_tris8"]), "stack_overflows": int(st["stack_overflows"]), "streams": int(streams),
                       "gather": ("one RCCL gather of the packed tiles to rank 0 per timed region" if use_dist else
                                  f"inside the C ABI (crt_sum_device, one process, {scene.devices()['transport']}), once per timed region" if one_proc else "none")},
            "roofline": roofline,
        }
        if use_dist:
            out["gather_ms"] = round(gather_ms, 4)
            out["rank_device_ms_per_step"] = [round(x, 4) for x in rank_ms]
            # what the collective library itself saw: the driver's record then shows RCCL with N ranks
            out["collective"] = {"backend": ctx.dist.get_backend(), "ranks_seen": int(ctx.dist.get_world_size()), "transport": "torch.distributed gather of the packed tiles to rank 0 (RCCL grouped send / recv under the nccl backend)",
                                 "build_threads_per_rank": int(os.environ.get("CRT_BUILD_THREADS", "0") or 0)}
        if one_proc:
            out["gather_ms"] = round(gather_ms, 4)
            dv = scene.devices()
            out["config"]["devices"] = dv["devices"]
            out["collective"] = {"backend": "libcrt (crt_set_devices)", "ranks_seen": len(dv["devices"]), "transport": dv["transport"]}
        if build_info:
            out["config"]["device_build"] = build_info
        out["config"]["launch"] = launch_info
        if not args.no_oracle_check and (world == 1 or one_proc):
            # after the clock has stopped: ONE more step, exactly as the timed region ran it, on a cleared sum, and 16 rows of the result
            # against the CPU oracle on the same frames (the checker, never the thing measured)
            out["sum_rows_match_oracle"] = rows_match_oracle(scene, step, Wu, spp, rvs, data, name, device_built, materials or args.materials, cam, W, H, depth, accel)
        if cpu_base and not args.no_cpu_baseline and accel == "cwbvh":
            out["cpu_baseline"] = cpu_baseline(data, cam, W, H, depth, rvs[1 + Wu * spp:1 + (Wu + 1) * spp], cs)
    if takes_part:
        scene.close()
    return out


def rows_match_oracle(scene, step, Wu, spp, rvs, data, name, device_built, materials, cam, W, H, depth, accel="cwbvh"):
    """True / False: the sum buffer after one step of the timed form equals the oracle's on 16 pixel rows through the image centre, bit
    for bit.  A scene crt_scene_create built on the GPU is checked against the oracle walking the same builder's tree (downloaded through
    the host-array entry points).  None when the check could not run."""
    import numpy as np
    import caitlynrenderer_amd as cr
    try:
        from oracle import binding as ob
        if data is None:
            key = (name, device_built, "device", "lambert")
            if key not in _SCENE_CACHE:
                mesh, _ = source_mesh(name)
                t0 = time.time()
                _SCENE_CACHE[key] = (cr.SceneData.build(mesh, cam, builder=device_built, convert="device"), cam, "", time.time() - t0)
            data = _SCENE_CACHE[key][0]
        scene.set_option("timing", 0)
        scene.reset()
        step(Wu)                                      # the first timed step's frames
        scene.sync()
        got = scene.read_sum()
        y0 = (H // 2 - 8) // 8 * 8
        orc = ob.Oracle(data, W, H, depth, cam)
        rows = np.zeros((H, W, 3), np.float32)
        o_accel, o_tie = (ob.BVH2, ob.TIE_FIRST_VISITED) if accel == "bvh2" else (ob.BVH8, ob.TIE_LOWEST_ID)
        for rx, ry in rvs[1 + Wu * spp:1 + (Wu + 1) * spp]:
            orc.render_rows(rx, ry, y0, y0 + 16, rows, accel=o_accel, tie=o_tie)
        return bool(np.array_equal(got[y0:y0 + 16].view(np.uint32), rows[y0:y0 + 16].view(np.uint32)) and rows[y0:y0 + 16].max() > 0)
    except Exception as e:                             # the check must never take the measurement down with it
        log(f"[bench] oracle row check did not run: {e!r}")
        return None


_ISA = None
_RL = None


def roofline_module():
    global _RL
    if _RL is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("crt_roofline", os.path.join(ROOT, "tools", "roofline.py"))
        _RL = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_RL)
    return _RL


ROOFLINE_NOTE = ("the scene is cache-resident, so the HBM roof does not bind (algorithmic_over_peak > 1: cache-served bytes; hbm_frac: measured "
                 "L2<->fabric bytes / 8 TB/s); frac = algorithmic traversal wave-instr (every node visit at the general 8-wide test) / time / peak; "
                 "frac_executed prices uniform node steps at their executed cost and is <= issue_busy x lane_util")


def valu_roofline(c, t_launch, launches, samples_per_launch, accel="cwbvh", pmc=None):
    """The roof that binds the segment kernel: vector-instruction issue (tools/roofline.py has the definition and recomputes it).
    c = the counters of one step, launches = segment launches that step is made of, t_launch = mean duration of one of them;
    peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction; with counters (pmc): issue_busy, lane_util, counter_frac = their
    product (>= frac_executed), non_traversal_share."""
    global _ISA
    if _ISA is None:
        try:
            _ISA = json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))
        except Exception:
            _ISA = {}
    r = {"bound": "valu_issue", "kernel": "k_segment (raygen / queue fetch + CWBVH closest hit + shading + NEE any-hit walk), mean over the step's segment launches",
         "achieved": None, "peak": round(VALU_PEAK_GINSTR, 1), "unit": "Gwave-instr/s", "frac": None, "note": ROOFLINE_NOTE}
    if _ISA.get("I_node") and accel == "cwbvh" and t_launch > 0:
        got = roofline_module().roofline_block(c, _ISA, t_launch * 1e3, launches, pmc, samples_per_launch)
        r.update({k: got[k] for k in ("achieved", "frac", "frac_executed", "traversal_wave_instr_per_launch", "executed_traversal_wave_instr_per_launch",
                                      "issue_busy", "lane_util", "counter_frac", "non_traversal_share") if k in got})
    return r


def cpu_baseline(data, cam, W, H, depth, rvs, cs):
    """The CPU oracle (a scalar port of the same algorithm on the same CWBVH) on a bounded sample of the same workload: the frames of one
    step, whole for small scenes, a band of pixel rows when that would take too long; all host cores.  Also cross-checks the GPU visit
    counters of that step (cs) when the sample is the whole step."""
    import numpy as np
    from oracle import binding as ob
    orc = ob.Oracle(data, W, H, depth, cam)
    threads = ob.hardware_threads()
    y0, y1 = 0, H
    t0 = time.perf_counter()
    probe = np.zeros((H, W, 3), np.float32)
    cnt = orc.render_rows(rvs[0][0], rvs[0][1], H // 2 - 4, H // 2 + 4, probe)
    rate = (cnt[0] + cnt[1]) / (time.perf_counter() - t0)              # rays/s, one thread
    est_full = (cs["closest_rays"] + cs["any_rays"]) / (rate * threads)
    if est_full > 12.0:                                                  # keep the sample near 10 s
        rows = max(8, int(H * 10.0 / est_full) // 8 * 8)
        y0 = (H - rows) // 2 // 8 * 8
        y1 = y0 + rows
    times, rays, c = [], 0, [0, 0, 0, 0]
    for rep in range(3 if est_full < 4 else 1):
        buf = np.zeros((H, W, 3), np.float32)
        c = [0, 0, 0, 0]
        t0 = time.perf_counter()
        for rx, ry in rvs:
            if (y0, y1) == (0, H):
                _, c1 = orc.render_frame(rx, ry, buf, threads=threads)
            else:
                import concurrent.futures as cf
                bands = [(a, min(a + 8, y1)) for a in range(y0, y1, 8)]
                with cf.ThreadPoolExecutor(threads) as ex:
                    cs_ = list(ex.map(lambda b: orc.render_rows(rx, ry, b[0], b[1], buf), bands))
                c1 = [sum(x[k] for x in cs_) for k in range(4)]
            c = [c[k] + c1[k] for k in range(4)]
        times.append(time.perf_counter() - t0)
        rays = c[0] + c[1]
    check = None
    if (y0, y1) == (0, H):
        check = bool(c[0] == cs["closest_rays"] and c[1] == cs["any_rays"] and
                     c[2] == cs["nodes_closest"] + cs["nodes_any"] and c[3] == cs["tris_closest"] + cs["tris_any"])
    return {"value": round(rays / float(np.median(times)) / 1e6, 3), "unit": "Mray/s", "cores": threads, "kind": "port",
            "single_thread_value": round(rate / 1e6, 3),          # one thread, 8 pixel rows through the image centre
            "sample": f"rows {y0}..{y1} of {H}, the {len(rvs)} frames of one step ({rays} rays, same frames and CWBVH as the GPU step)",
            "visit_counters_match_gpu": check}


def frame_loop_block(ctx, name, W, H, frames=60):
    """What INTEGRATION.md section 3 prescribes per DISPLAYED frame in place of Scene::Render's three passes (Scene.h:1208-1230: one sample,
    a copy pass, the output pass): crt_render_frame (one sample per pixel) + the tone-mapped RGBA8 image — left in device memory
    (crt_resolve_device: where the reference's output pass leaves it, the default framebuffer) or copied into host memory (crt_resolve).
    ms per iteration and its parts; the bytes of the last image against the oracle's resolve of the oracle's sum on 16 rows."""
    import numpy as np
    import caitlynrenderer_amd as cr
    args = ctx.args
    data, cam, label, _ = build_workload(name, "sbvh" if args.builder == "auto" else args.builder, args.convert, "lambert")
    scene = cr.Scene(data, W, H, 1)
    scene.set_shard(0, 1, args.tile)
    rnd = cr.Rnd()
    rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(2 * frames + 8)]
    for r in rvs[:8]:                                   # warm: tile order measured, code objects loaded, staging buffers allocated
        scene.render_frame(*r, sync=False)
    scene.resolve(1.0 / 8)
    scene.reset()
    # (a) the image stays in HBM: render + resolve enqueued frame after frame, one wait at the end
    t0 = time.perf_counter()
    for i in range(frames):
        scene.render_frame(*rvs[8 + i], sync=False)
        scene.resolve_device(1.0 / (i + 1), sync=False)
    scene.sync()
    dev_ms = (time.perf_counter() - t0) / frames * 1e3
    # (b) the image in host memory every frame (the D2H copy and its wait are on the loop's critical path)
    t0 = time.perf_counter()
    img = None
    for i in range(frames):
        scene.render_frame(*rvs[8 + frames + i], sync=False)
        img = scene.resolve(1.0 / (frames + i + 1))
    host_ms = (time.perf_counter() - t0) / frames * 1e3
    # parts: the segment launch by its own events, the resolve pass as what is left of (a)
    scene.set_option("timing", 1)
    scene.set_option("timing_accumulate", 16)
    for r in rvs[:16]:
        scene.render_frame(*r, sync=False)
    scene.sync()
    lt = scene.launch_times()
    launch_ms = float(np.median(lt)) if len(lt) else None
    scene.set_option("timing_accumulate", 0)
    scene.set_option("timing", 0)
    out = {"workload": label.split(",")[0].replace("procedural tessellated Cornell ", "") + f" {W}x{H}: crt_render_frame + resolve per displayed frame",
           "ms_per_frame_image_in_hbm": round(dev_ms, 4), "ms_per_frame_image_in_host_memory": round(host_ms, 4),
           "segment_launch_ms": round(launch_ms, 4) if launch_ms else None,
           "untile_resolve_ms": round(dev_ms - launch_ms, 4) if launch_ms else None, "d2h_and_wait_ms": round(host_ms - dev_ms, 4),
           "fps_image_in_hbm": round(1e3 / dev_ms, 1)}
    if not args.no_oracle_check:
        try:
            from oracle import binding as ob
            n_total = 16 + 2 * frames         # frames in the sum the last image shows: 16 (parts) were rendered after it — so render the check apart
            scene.reset()
            for r in rvs[:2]:
                scene.render_frame(*r, sync=False)
            got = scene.resolve(0.5)
            y0 = (H // 2 - 8) // 8 * 8
            orc = ob.Oracle(data, W, H, 1, cam)
            rows = np.zeros((H, W, 3), np.float32)
            for rx, ry in rvs[:2]:
                orc.render_rows(rx, ry, y0, y0 + 16, rows)
            want = ob.resolve(rows[y0:y0 + 16], 0.5)
            out["rgba_rows_match_oracle"] = bool(np.array_equal(got[y0:y0 + 16], want) and want[..., :3].max() > 0)
        except Exception as e:
            log(f"[bench] frame-loop oracle check did not run: {e!r}")
            out["rgba_rows_match_oracle"] = None
    scene.close()
    return out


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
    if args.gpus > 1 and "RANK" not in os.environ and not args.one_process:
        sys.exit(self_launch(args))

    # stdout carries exactly ONE JSON line: anything a library prints there (RCCL's version banner at
    # communicator creation, for one) is sent to stderr instead
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if (args.gpus == 1 and "RANK" not in os.environ and args.workload == "auto" and args.accel == "cwbvh"
            and not (args.dry_run or args.no_live_pmc or args.option)):
        # hardware counters of this very run, from child processes, before this process makes its first GPU call
        live_pmc([("mesh1m", args.depth, args.spp or 4)] + ([] if args.no_extra else [("mesh1m", 4, 4), ("cornell", 1, 1)])
                 + ([] if args.no_extra or args.no_hbm_resident else [(HBM_RESIDENT, 1, 4, "--device-built", "sah"), (HBM_RESIDENT, 4, 4, "--device-built", "sah")]))

    ctx = Ctx(args)
    if ctx.world != args.gpus and not args.one_process:
        args.gpus = ctx.world
    if ctx.world > 1:
        # launched under torch.distributed.run by someone else: the ranks of this node share its cores for the host SBVH build
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", ctx.world))
        os.environ.setdefault("CRT_BUILD_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, local_world))))
    ctx.init()
    if not args.dry_run:
        import __graft_entry__ as g
        g.build()

    N = len(ctx.one_proc_ids) if ctx.one_proc_ids else ctx.world
    auto = args.workload == "auto"
    if N == 1:
        name = "mesh1m" if auto else args.workload      # configs[2]: the workload BASELINE.json's targets are quoted on
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
        if N > 1 and args.builder == "auto" and not args.device_built and name != "cornell":
            args.device_built = "sah"          # every rank's crt_scene_create builds the tree on its own GPU: no host SBVH build in the N > 1 path
        head = run_block(ctx, name, W, H, args.depth, spp, True, N == 1 and not args.device_built, scaling, device_built=args.device_built)
        extra = {}
        if auto and not args.no_extra and args.accel == "cwbvh":
            if N == 1:
                extra["cornell"] = run_block(ctx, "cornell", 1920, 1080, 1, 1, True, False, "weak")
                extra["gpu_tree"] = run_block(ctx, "mesh1m", 1920, 1080, 1, 4, True, False, "weak", device_built="sah")
                extra["d2"] = run_block(ctx, "mesh1m", 1920, 1080, 2, 4, True, False, "weak")      # the other reading of the metric's "primary + 1 bounce": two path segments
                extra["incoherent"] = run_block(ctx, "mesh1m", 1920, 1080, 4, 4, True, False, "weak")
                extra["incoherent_disney"] = run_block(ctx, "mesh1m", 1920, 1080, 4, 4, True, False, "weak", materials="disney")
                extra["scale_base"] = run_block(ctx, "mesh1m", 3840, 2160, 1, 4, True, False, "strong")
                # the reference's own published claims (README.md:21-22), measured on this GPU: "CWBVH 2 to 4 times faster than SBVH" = the same
                # frames through the shipped shader's BVH2 walk (accel bvh2) over the SBVH; "SBVH 20-30 % faster than SAH BVH" = that walk
                # over the SAH-only tree (object splits, no spatial splits)
                extra["bvh2_cornell"] = run_block(ctx, "cornell", 1920, 1080, 1, 1, True, False, "weak", accel="bvh2")
                extra["bvh2_mesh1m"] = run_block(ctx, "mesh1m", 1920, 1080, 1, 4, True, False, "weak", accel="bvh2")
                extra["bvh2_mesh1m_sah"] = run_block(ctx, "mesh1m", 1920, 1080, 1, 4, True, False, "weak", accel="bvh2", sbvh_flags=1)
                extra["frame_loop_cornell"] = frame_loop_block(ctx, "cornell", 1920, 1080)
                extra["frame_loop_mesh1m"] = frame_loop_block(ctx, "mesh1m", 1920, 1080)
                if not args.no_hbm_resident:
                    # > 256 MiB of nodes + records: the one block whose `traffic` is HBM traffic.  Built on the GPU (binned SAH): the
                    # reference's host builder would take minutes at this size.
                    saved = args.steps
                    args.steps = max(3, args.steps // 5)
                    extra["hbm_resident"] = run_block(ctx, HBM_RESIDENT, 1920, 1080, 1, 4, True, False, "weak", device_built="sah")
                    extra["hbm_resident_d4"] = run_block(ctx, HBM_RESIDENT, 1920, 1080, 4, 4, True, False, "weak", device_built="sah")
                    args.steps = saved
            elif args.scaling == "strong":
                ctx.barrier()
                extra["n1_same_workload"] = run_block(ctx, name, W, H, args.depth, spp, False, False, "strong", device_built=args.device_built)      # on the same tree
                ctx.barrier()

    if ctx.rank == 0:
        out = {"metric": METRIC.replace("1920x1080", f"{W}x{H}"), "value": head["value"], "unit": head["unit"], "n_gpus": N, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": head["scaling"],
               "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": head["config"], "roofline": head["roofline"]}
        for k in ("step_ms_spread", "cpu_baseline", "gather_ms", "rank_device_ms_per_step", "sum_rows_match_oracle", "collective"):
            if k in head:
                out[k] = head[k]
        if head.get("dry_run"):
            out["dry_run"] = True
        n1 = extra.pop("n1_same_workload", None)
        if n1:
            # the same frame rendered by rank 0 alone in this job: what N ranks are measured against
            out["n1_same_workload"] = {"value": n1["value"], "ms_per_step": n1["ms_per_step"]}
            out["scaling_efficiency"] = round(head["value"] / (N * n1["value"]), 4)
        ex = {k: (compact(v) if "roofline" in v else v) for k, v in extra.items() if v is not None}
        if ex.get("bvh2_mesh1m") and ex.get("bvh2_cornell"):
            # README.md:21-22 next to this GPU's numbers (throughput ratios on the same frames; the reference quotes none of its own hardware)
            claims = {"readme": "README.md:21-22: 'SBVH 20-30 % faster than SAH BVH', 'CWBVH 2 to 4 times faster than SBVH'",
                      "cwbvh_over_bvh2_mesh1m": round(head["value"] / ex["bvh2_mesh1m"]["value"], 3),
                      "cwbvh_over_bvh2_cornell": round(ex["cornell"]["value"] / ex["bvh2_cornell"]["value"], 3) if ex.get("cornell") else None}
            if ex.get("bvh2_mesh1m_sah"):
                claims["sbvh_over_sah_bvh2_walk"] = round(ex["bvh2_mesh1m"]["value"] / ex["bvh2_mesh1m_sah"]["value"], 3)
                a, b = extra["bvh2_mesh1m"]["roofline"]["counters"], extra["bvh2_mesh1m_sah"]["roofline"]["counters"]
                claims["sah_over_sbvh_node_visits"] = round((b["nodes_closest"] + b["nodes_any"]) / max(1, a["nodes_closest"] + a["nodes_any"]), 3)
            out["reference_claims"] = claims
        if ex:
            out["extras"] = ex
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out, separators=(",", ":")) + "\n").encode())
    ctx.close()


HBM_RESIDENT = "mesh520"      # tessellation n = 520: 8,112,002 triangles, 16.2 M BVH2 nodes (below the 2^24 a float link can index)


def compact(b):
    """An extra block of the line: what it is, its throughput and its roofline figures, nothing else (README.md explains the fields)."""
    r, c = b["roofline"], b["config"]
    e = {"workload": c["workload"].split(",")[0].replace("procedural tessellated Cornell ", "") + f" {c['resolution']} d{c['path_segments']} spp{c['spp_per_step']}"
                     + (" disney" if "Disney" in c["workload"] else "") + (" gpu-built" if "device_build" in c else "") + (" bvh2" if r.get("accel") == "bvh2" else "")
         + (" sah-only" if "SAH BVH" in c["workload"] else ""),
         "value": b["value"], "ms_per_step": b["ms_per_step"], "launch_ms": r["launch_ms"], "samples_per_launch": r["samples_per_launch"],
         "frac": r["frac"]}
    if b.get("step_ms_spread"):
        e["step_ms"] = [b["step_ms_spread"]["min"], b["step_ms_spread"]["median"], b["step_ms_spread"]["max"]]      # min, median, max (README.md: how they are measured)
    for k in ("frac_executed", "traffic", "traffic_gbps", "hbm_frac", "issue_busy", "lane_util", "counter_frac", "non_traversal_share", "l2_hit_rate"):
        if r.get(k) is not None:
            e[k] = r[k]
    if b.get("sum_rows_match_oracle") is not None:
        e["sum_rows_match_oracle"] = b["sum_rows_match_oracle"]
    if c.get("streams", 1) > 1:
        e["streams"] = c["streams"]       # tile shards side by side: launch_ms is the step's wall time / launches per shard
    if "device_build" in c:
        e["device_build"] = c["device_build"]
    if c["workload"].find(f"n={HBM_RESIDENT[4:]}:") >= 0:
        e["scene_mb"] = round((80 * c["n_nodes8"] + 48 * c["n_tris8"]) / 1e6, 1)      # this scene does not fit the Infinity Cache: its traffic is HBM traffic
    return e


if __name__ == "__main__":
    main()
