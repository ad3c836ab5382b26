import sys, os, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import caitlynrenderer_amd as cr
from caitlynrenderer_amd.meshgen import tessellated_cornell
from oracle import binding as ob
mesh, cam = g._cornell()
m = tessellated_cornell(mesh, 40)
data = cr.SceneData.build(m, cam)
W,H = 1920,1080
rnd = cr.Rnd(); rx, ry = rnd.randf2(), rnd.randf2()
scene = cr.Scene(data, W, H, 1)
scene.set_option("count_visits", 1)
scene.render_frame(rx, ry)
st = scene.frame_stats()
orc = ob.Oracle(data, W, H, 1, cam)
ref, cnt = orc.render_frame(rx, ry, threads=16)
print("gpu", st["closest_rays"], st["any_rays"], st["nodes_closest"], st["tris_closest"], st["nodes_any"], st["tris_any"])
print("cpu", cnt, "nodes sum gpu", st["nodes_closest"]+st["nodes_any"], "tris sum gpu", st["tris_closest"]+st["tris_any"])
out = scene.read_sum()
diff = (out.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
print("pixels differing", diff.sum(), "max abs", np.abs(out-ref).max())
rays = orc.primary_rays(rx, ry, jitter=True)
hg, sg = scene.trace(rays.astype(cr.RAY_DT), stats=True)
hc, sc = orc.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, stats=True, threads=16)
print("primary hit mismatches tri", (hg["tri"]!=hc["tri"]).sum(), "t", (hg["t"].view(np.uint32)!=hc["t"].view(np.uint32)).sum(), "nodes", (sg["nodes"]!=sc["nodes"]).sum(), "tris", (sg["tris"]!=sc["tris"]).sum())
print("sum nodes prim", sg["nodes"].astype(np.int64).sum(), sc["nodes"].astype(np.int64).sum())
ys, xs = np.nonzero(diff)
for y, x in list(zip(ys, xs))[:10]:
    print(y, x, out[y, x], ref[y, x], hc[y*W+x])
