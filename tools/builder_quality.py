#!/usr/bin/env python3
"""Tree quality and build time of the BVH2 builders on the 1,004,672-triangle mesh: node8 visits per primary / shadow ray of a
1920x1080 frame (the counting kernels), frame time, device build time.   usage: tools/builder_quality.py [builder ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.build()
import caitlynrenderer_amd as cr
from caitlynrenderer_amd.meshgen import tessellated_cornell

base, cam = g._cornell()
mesh = tessellated_cornell(base, 183)
builders = sys.argv[1:] or ["sbvh", "sah", "lbvh", "ploc4", "ploc16", "ploc64"]
rnd = cr.Rnd()
rvs = [(rnd.randf2(), rnd.randf2()) for _ in range(60)]
for b in builders:
    t0 = time.time()
    if b == "sbvh":
        data = cr.SceneData.build(mesh, cam, builder="sbvh", convert="device")
        scene = cr.Scene(data, 1920, 1080, 1)
        build = "host %.2f s" % (time.time() - t0)
    else:
        cr.Scene(cr.SceneData.for_device_build(mesh, cam, builder=b), 64, 64, 1).close()       # warm (code objects, allocator)
        t0 = time.time()
        scene = cr.Scene(cr.SceneData.for_device_build(mesh, cam, builder=b), 1920, 1080, 1)
        i = scene.bvh_info()
        build = "scene create %.2f ms wall, BVH2 %.2f ms + CWBVH %.2f ms device, BVH2 depth %d" % ((time.time() - t0) * 1e3, i["build_lbvh_device_ms"],
                                                                                                  i["build_convert_device_ms"], i["bvh2_depth"])
    info = scene.bvh_info()
    scene.set_option("count_visits", 1)
    scene.render_frame(*rvs[0])
    st = scene.frame_stats()
    scene.set_option("count_visits", 0)
    scene.set_option("timing", 0)
    for rv in rvs[:10]:
        scene.render_frame(*rv, sync=False)
    scene.sync()
    t0 = time.perf_counter()
    for rv in rvs[10:]:
        scene.render_frame(*rv, sync=False)
    scene.sync()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    rays = st["closest_rays"] + st["any_rays"]
    print(f"{b:8s} {info['n_nodes8']:7d} node8 depth {info['max_depth8']:2d} | nodes/primary ray {st['nodes_closest'] / st['closest_rays']:.3f}, tris {st['tris_closest'] / st['closest_rays']:.3f}; "
          f"nodes/shadow ray {st['nodes_any'] / st['any_rays']:.3f}, tris {st['tris_any'] / st['any_rays']:.3f} | frame {ms:.4f} ms = {rays / ms / 1e3:.0f} Mray/s | {build}", flush=True)
    scene.close()
