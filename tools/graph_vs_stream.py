"""Device time per frame: frames queued on the stream vs the same frames replayed as one captured hipGraph.
usage (GPU box): python tools/graph_vs_stream.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as g
g.build()
import caitlynrenderer_amd as cr
from caitlynrenderer_amd._lib import lib, check
from caitlynrenderer_amd.meshgen import tessellated_cornell
base, cam = g._cornell()
rnd = cr.Rnd()
rxy = np.array([rnd.randf2() for _ in range(2 * 16)], np.float32)
for label, mesh, depth in (("cornell d1", base, 1), ("cornell d3", base, 3), ("tess40 d1", tessellated_cornell(base, 40), 1), ("tess40 d4", tessellated_cornell(base, 40), 4)):
    data = cr.SceneData.build(mesh, cam)
    s = cr.Scene(data, 1920, 1080, depth)
    a, b = C.c_float(), C.c_float()
    check(lib().crt_debug_time_graph(s._h, 16, rxy.ctypes.data_as(C.c_void_p), 20, C.byref(a), C.byref(b)))
    print(f"{label}: stream {a.value * 1e3:.1f} us/frame, graph {b.value * 1e3:.1f} us/frame", flush=True)
    s.close()
