import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import caitlynrenderer_amd as cr
from caitlynrenderer_amd.meshgen import tessellated_cornell
mesh, cam = g._cornell()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 183
m = tessellated_cornell(mesh, n)
data = cr.SceneData.build(m, cam)
W,H = 960,540
rnd = cr.Rnd(); rx, ry = rnd.randf2(), rnd.randf2()
scene = cr.Scene(data, W, H, 2)
t=time.time(); scene.render_frame(rx, ry); print("frame s", time.time()-t, scene.frame_stats())
rays = scene.debug_read_queue(0, 1)
print("bounce rays", len(rays), "nan d", np.isnan(rays["d"]).any(axis=1).sum(), "nan o", np.isnan(rays["o"]).any(axis=1).sum(), "zero d", (rays["d"]==0).all(axis=1).sum(), "any zero comp", (rays["d"]==0).any(axis=1).sum())
t=time.time(); hits, st = scene.trace(rays, stats=True); print("trace s", time.time()-t)
nodes = st["nodes"].astype(np.int64); tris = st["tris"].astype(np.int64)
print("nodes mean/max", nodes.mean(), nodes.max(), "tris mean/max", tris.mean(), tris.max())
print("percentiles nodes", np.percentile(nodes,[50,90,99,99.9,99.99]), "tris", np.percentile(tris,[50,90,99,99.9,99.99]))
worst = np.argsort(-tris)[:8]
for i in worst: print(rays[i], nodes[i], tris[i], hits[i])
