#!/usr/bin/env python3
"""First-call cost of crt_scene_create(... CRT_BUILD_SAH) at 1,004,672 triangles, attributed.

usage: python tools/build_probe.py [variant]      (no argument: runs every variant in a FRESH child process each, prints a table)

Variants (each a fresh process, so every one starts with no code object loaded and no HIP context):
  cold        the device-built scene is the first thing the process asks of the library
  after_frame a host-built Cornell scene is created and renders one frame first (what bench.py's gpu_tree block sees: the traversal
              kernels' code object is loaded, the builders' is not)
  warmup      crt_warmup() first (loads every code object of the library and the builders' library kernels), then the scene
  tiny_first  a 32-triangle device-built scene first (every builder kernel has run once), then the 1 M one
Each prints upload / BVH2 / CWBVH device ms and the wall time of four consecutive creations.
"""
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
VARIANTS = ("cold", "after_frame", "warmup", "tiny_first")


def child(variant):
    import __graft_entry__ as g
    g.build()
    import caitlynrenderer_amd as cr
    from caitlynrenderer_amd.meshgen import tessellated_cornell
    base, cam = g._cornell()
    mesh = tessellated_cornell(base, 183)
    dd = cr.SceneData.for_device_build(mesh, cam, builder="sah")
    t_pre = time.perf_counter()
    if variant == "after_frame":
        s0 = cr.Scene(cr.SceneData.build(base, cam), 1920, 1080, 1)
        s0.render_frame(0.5, 0.5)
        s0.close()
    elif variant == "warmup":
        cr.warmup()
    elif variant == "tiny_first":
        s0 = cr.Scene(cr.SceneData.for_device_build(base, cam, builder="sah"), 64, 64, 1)
        s0.close()
    pre_ms = (time.perf_counter() - t_pre) * 1e3
    print(f"{variant}: preamble {pre_ms:.2f} ms", flush=True)
    for rep in range(4):
        t0 = time.perf_counter()
        s = cr.Scene(dd, 1920, 1080, 1)
        t1 = time.perf_counter()
        s.render_frame(0.5, 0.5)
        t2 = time.perf_counter()
        i = s.bvh_info()
        print(f"{variant} call {rep + 1}: scene_create wall {(t1 - t0) * 1e3:7.2f} ms  (upload {i['build_upload_ms']:6.2f}, BVH2 {i['build_lbvh_device_ms']:6.2f}, "
              f"CWBVH {i['build_convert_device_ms']:5.2f}, library total {i['build_wall_ms']:6.2f}); first frame {(t2 - t1) * 1e3:6.2f} ms", flush=True)
        s.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for v in VARIANTS:
            subprocess.run([sys.executable, os.path.abspath(__file__), v], check=False)
