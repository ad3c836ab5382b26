"""Per-rank cost of a sharded frame, measured on ONE GPU: rank world // 2 of `world` renders its tiles of the 1 M-triangle frame
through crt_render_frames, and the time per frame is set against perfect division of the whole frame's.  Columns: the samples of a
launch one after the other in each wave (wave_samples 0), side by side on the waves of a workgroup (1), what the library picks
by itself (2), and four samples of a 4 x 4 pixel quadrant in the lanes of one wave (3).  The sums of all are compared bit for bit first.

    python tools/shard_times.py [WxH] [depth ...]        (default 3840x2160, depths 1 4)

Numbers of this script are quoted in DESIGN.md §10 and profiles/r02_experiments.md §10."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as g
g.build()
import caitlynrenderer_amd as cr
from caitlynrenderer_amd.meshgen import tessellated_cornell

args = sys.argv[1:]
W, H = (int(x) for x in (args[0] if args and "x" in args[0] else "3840x2160").split("x"))
depths = [int(a) for a in args if "x" not in a] or [1, 4]
base, cam = g._cornell()
data = cr.SceneData.build(tessellated_cornell(base, 183), cam)
rnd = cr.Rnd()
rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(160)]

for depth in depths:
    sums = []
    for ws in (0, 1, 2, 3):
        scene = cr.Scene(data, 640, 360, depth)
        scene.set_option("wave_samples", ws)
        scene.set_shard(1, 3, 16)
        for b in (8, 5, 3, 2, 1):
            scene.render_frames(rvs[:b], sync=False)
        scene.sync()
        sums.append(scene.read_sum().copy())
        scene.close()
    same = all(np.array_equal(sums[0].view(np.uint32), s.view(np.uint32)) for s in sums[1:])
    print(f"depth {depth}: sums of the four forms identical: {same}", flush=True)
    assert same
    whole = {}
    for world in (1, 2, 4, 8):
        cells = []
        for ws in (0, 1, 3, 2):
            scene = cr.Scene(data, W, H, depth)
            scene.set_option("wave_samples", ws)
            if world > 1:
                scene.set_shard(world // 2, world, 16)
            t_end = time.perf_counter() + 0.3           # clocks up, tile costs measured and adopted
            while time.perf_counter() < t_end:
                scene.render_frames(rvs[:4], sync=False); scene.sync()
            for b in (4, 8):
                n = 160 if depth == 1 else 48
                best = 1e9
                for _ in range(3):
                    scene.sync(); t0 = time.perf_counter()
                    for i in range(0, n, b):
                        scene.render_frames(rvs[i:i + b], sync=False)
                    scene.sync(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
                if world == 1 and ws == 2:
                    whole[b] = best
                cells.append((ws, b, best))
            scene.close()
        line = "  ".join(f"ws{ws} x{b}: {t:.4f}" for ws, b, t in cells)
        eff = "  ".join(f"x{b}: {100 * whole[b] / world / t:.0f} %" for ws, b, t in cells if ws == 2) if whole else ""
        print(f"{W}x{H} depth {depth} rank {world // 2} of {world}, ms per frame: {line}   | picked form vs whole / {world}: {eff}", flush=True)
