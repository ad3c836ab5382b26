#!/usr/bin/env python3
"""Option "streams" measured: k tile shards of the frame (or of one rank's shard of it) side by side on k streams of the one GPU.

  python tools/streams_probe.py            # the table of profiles/r03_streams_probe.txt

Per configuration: milliseconds per 4-sample step (crt_render_frames, queued asynchronously, best of 3 x 100 steps after 0.3 s of
settling), launch events off.  Whole frames at 1920x1080 (1 and 4 segments, 1 M and 8 M triangles, the Cornell box at 1 sample per
step) and rank r of w of the 3840x2160 frame — what one rank of `bench.py --gpus w` renders — in % of perfect division."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    import caitlynrenderer_amd as cr
    rvs = [(0.6591631, 0.910802), (0.13842908, 0.1292837), (0.1558374, 0.07157449), (0.045871824, 0.9201364)]

    def run(make, W, H, depth, rank, world, streams, spp=4):
        sc = make(W, H, depth)
        if world > 1:
            sc.set_shard(rank, world, 16)
        if streams > 1:
            sc.set_option("streams", streams)

        def step():
            if spp > 1:
                sc.render_frames(rvs[:spp], sync=False)
            else:
                sc.render_frame(*rvs[0], sync=False)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            for _ in range(10):
                step()
            sc.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(100):
                step()
            sc.sync()
            best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
        sc.close()
        return best

    def host_built(name):
        data, cam, label, _ = bench.build_workload(name)

        def make(W, H, depth):
            sc = cr.Scene(data, W, H, depth)
            sc.update(cam)
            return sc
        return make, label

    def device_built(name):
        mesh, cam = bench.source_mesh(name)

        def make(W, H, depth):
            sc = cr.Scene(cr.SceneData.for_device_build(mesh, cam, builder="sah"), W, H, depth)
            sc.update(cam)
            return sc
        return make, f"{name} (GPU-built SAH tree)"

    print("whole frames, 1920x1080, ms per step (4 samples; Cornell: 1 sample) for 1 / 2 / 3 streams")
    for (mk, depth, spp) in ((host_built("mesh1m"), 1, 4), (host_built("mesh1m"), 2, 4), (host_built("mesh1m"), 4, 4),
                             (device_built(bench.HBM_RESIDENT), 4, 4), (host_built("cornell"), 1, 1), (host_built("cornell"), 3, 1)):
        make, label = mk
        t = [run(make, 1920, 1080, depth, 0, 1, k, spp) for k in (1, 2, 3)]
        print(f"  {label.split(',')[0][:58]:58s} {depth} segment(s): {t[0]:.4f} / {t[1]:.4f} / {t[2]:.4f}   ({t[0] / t[1] * 100 - 100:+.1f} % / {t[0] / t[2] * 100 - 100:+.1f} %)", flush=True)
    make, label = host_built("mesh1m")
    print("one rank's shard of the 3840x2160 frame of the 1 M-triangle mesh: ms per 4-sample step, 1 -> 2 streams (% of perfect division of the whole-frame time)")
    for depth in (1, 4):
        whole = run(make, 3840, 2160, depth, 0, 1, 1)
        print(f"  {depth} segment(s): whole frame {whole:.4f}", flush=True)
        for world, rank in ((2, 1), (4, 2), (8, 4)):
            a, b = run(make, 3840, 2160, depth, rank, world, 1), run(make, 3840, 2160, depth, rank, world, 2)
            print(f"     rank {rank} of {world}: {a:.4f} ({whole / world / a * 100:.0f} %) -> {b:.4f} ({whole / world / b * 100:.0f} %)", flush=True)


if __name__ == "__main__":
    main()
