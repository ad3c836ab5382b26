"""Build-side timings on the GPU box: LBVH build, CWBVH conversion (device vs host), scene creation at 1,004,672 triangles.
usage: python tools/build_times.py   (profiles/r01_gpu_builders_* come from this under rocprofv3 --kernel-trace --stats)"""
import os, time, numpy as np, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
g.build()
import caitlynrenderer_amd as cr
from caitlynrenderer_amd.meshgen import tessellated_cornell
base, cam = g._cornell()
mesh = tessellated_cornell(base, 183)
for builder in ("lbvh", "lbvh", "lbvh", "sbvh"):      # the first GPU build pays for code-object loading
    t0 = time.time(); sb = cr.SBVH(mesh.triangles, mesh.vertices, builder=builder); t1 = time.time()
    print(builder, "build %.3fs" % (t1 - t0), "(device, total) ms =", sb.build_ms, "nodes", sb.flat_nodes.shape[0], flush=True)
    for rep in range(2):
        t0 = time.time(); d = cr.CWBVH().convert(sb, device=True); t1 = time.time()
        print("  device convert %.1f ms wall, (device, total) ms =" % ((t1 - t0) * 1e3), d.convert_ms, "node8", d.nodes.shape[0], "depth", d.depth, flush=True)
    t0 = time.time(); h = cr.CWBVH().convert(sb); t1 = time.time()
    print("  host convert %.1f ms" % ((t1 - t0) * 1e3), "identical", np.array_equal(h.nodes, d.nodes) and np.array_equal(h.tri_slots, d.tri_slots), flush=True)
print("--- everything on the device: crt_scene_create with CRT_BUILD_LBVH_ON_DEVICE (input arrays -> first frame)", flush=True)
for builder in ("lbvh", "sah"):
  dd = cr.SceneData.for_device_build(mesh, cam, builder=builder)
  print("  builder", builder, flush=True)
  for rep in range(4):
      t0 = time.time(); s = cr.Scene(dd, 1920, 1080, 1); t1 = time.time()
      s.render_frame(0.5, 0.5); t2 = time.time()
      i = s.bvh_info()
      print("  device-built scene create %.2f ms wall (upload %.2f, BVH2 %.2f device, CWBVH %.2f device, library total %.2f), first frame %.2f ms -> arrays to first frame %.2f ms; %d node8, depth %d, BVH2 depth %d"
            % ((t1 - t0) * 1e3, i["build_upload_ms"], i["build_lbvh_device_ms"], i["build_convert_device_ms"], i["build_wall_ms"], (t2 - t1) * 1e3, (t2 - t0) * 1e3,
               i["n_nodes8"], i["max_depth8"], i["bvh2_depth"]), flush=True)
      s.close()
print("--- the same tree through the host-array entry points", flush=True)
data = cr.SceneData.build(mesh, cam, builder="lbvh", convert="device")
for rep in range(2):
    t0 = time.time(); s = cr.Scene(data, 1920, 1080, 1); t1 = time.time()
    s.render_frame(0.5, 0.5); t2 = time.time()
    print("scene create %.1f ms, first frame %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    s.close()
