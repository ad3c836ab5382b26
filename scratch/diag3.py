import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import caitlynrenderer_amd as cr
from caitlynrenderer_amd.meshgen import tessellated_cornell
mesh, cam = g._cornell()
m = tessellated_cornell(mesh, 183)
data = cr.SceneData.build(m, cam)
W,H = 1920,1080
rnd = cr.Rnd(); rx, ry = rnd.randf2(), rnd.randf2()
scene = cr.Scene(data, W, H, 2)
scene.render_frame(rx, ry)
rays = scene.debug_read_queue(0, 1)
print("bounce rays", len(rays))
lo, hi = m.vertices.min(0), m.vertices.max(0)
def morton3(q, bits):
    x,y,z = [q[:,i].astype(np.uint64) for i in range(3)]
    code = np.zeros(len(q), np.uint64)
    for b in range(bits):
        code |= ((x>>np.uint64(b))&np.uint64(1))<<np.uint64(3*b+2) | ((y>>np.uint64(b))&np.uint64(1))<<np.uint64(3*b+1) | ((z>>np.uint64(b))&np.uint64(1))<<np.uint64(3*b)
    return code
def keys(bits):
    q = np.clip(((rays["o"]-lo)/(hi-lo+1e-6)*(1<<bits)).astype(np.int64), 0, (1<<bits)-1)
    return morton3(q, bits)
octant = ((rays["d"][:,0]<0).astype(np.uint64)<<np.uint64(2)) | ((rays["d"][:,1]<0).astype(np.uint64)<<np.uint64(1)) | (rays["d"][:,2]<0).astype(np.uint64)
# direction quantised finer: 3 bits per axis of direction on the unit cube
dq = np.clip(((rays["d"]/np.abs(rays["d"]).max(1,keepdims=True))*3.999+4).astype(np.int64),0,7)
dkey = morton3(dq,3)
orders = {
 "queue order": np.arange(len(rays)),
 "shuffled": np.random.default_rng(0).permutation(len(rays)),
 "octant": np.argsort(octant, kind="stable"),
 "octant,morton5": np.argsort((octant<<np.uint64(15))|keys(5), kind="stable"),
 "morton4,octant": np.argsort((keys(4)<<np.uint64(3))|octant, kind="stable"),
 "morton3,dir9": np.argsort((keys(3)<<np.uint64(9))|dkey, kind="stable"),
 "dir9,morton4": np.argsort((dkey<<np.uint64(12))|keys(4), kind="stable"),
 "morton6,octant": np.argsort((keys(6)<<np.uint64(3))|octant, kind="stable"),
}
from oracle import binding as ob
prim = ob.Oracle(data, W, H, 1, cam).primary_rays(rx, ry, jitter=True).astype(cr.RAY_DT)
d_ph = torch.empty((len(prim),16), dtype=torch.uint8, device="cuda")
d_pr = torch.from_numpy(prim.view(np.uint8).reshape(-1,32)).cuda()
ts=[]
for _ in range(6):
    scene.trace_device(d_pr.data_ptr(), len(prim), d_ph.data_ptr(), cr.CRT_TRACE_CLOSEST); ts.append(scene.frame_stats()["ms_trace_closest"])
print(f"primary rays       {np.median(ts[1:]):.4f} ms  {len(prim)/np.median(ts[1:])/1e3:.1f} Mray/s")
d_hits = torch.empty((len(rays),16), dtype=torch.uint8, device="cuda")
for name, idx in orders.items():
    r = np.ascontiguousarray(rays[idx])
    d_rays = torch.from_numpy(r.view(np.uint8).reshape(-1,32)).cuda()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(6):
        scene.trace_device(d_rays.data_ptr(), len(r), d_hits.data_ptr(), cr.CRT_TRACE_CLOSEST)
        ts.append(scene.frame_stats()["ms_trace_closest"])
    print(f"{name:18s} {np.median(ts[1:]):.4f} ms  {len(r)/np.median(ts[1:])/1e3:.1f} Mray/s")
