#!/usr/bin/env python3
"""Lane utilisation of the traversal loops, per block, from the counting kernels (crt_frame_stats.wave_steps_*):
lane visits / (64 x wave-level executions of the block).   usage: tools/lane_util.py [workload] [depth] [NAME=INT ...] [lanes]
`lanes`: count ONE 4-sample step in the form the bench's launches have (count_visits 2: four samples of a 4x4-pixel quadrant in the lanes of a wave)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import caitlynrenderer_amd as cr

name = sys.argv[1] if len(sys.argv) > 1 else "mesh1m"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
data, cam, label, _ = bench.build_workload(name)
scene = cr.Scene(data, 1920, 1080, depth)
lanes_form = "lanes" in sys.argv[3:]
for kv in [a for a in sys.argv[3:] if a != "lanes"]:
    k, v = kv.split("=")
    scene.set_option(k, int(v))
# two frames first: the bins of the bounce rays (option ray_bins) take their places from the previous frame's counts
for _ in range(2):
    scene.render_frame(0.6591631174087524, 0.9108020067214966)
if lanes_form:
    rnd = cr.Rnd()
    scene.set_option("count_visits", 2)
    scene.render_frames([(rnd.randf2(), rnd.randf2()) for _ in range(4)])
else:
    scene.set_option("count_visits", 1)
    scene.render_frame(0.6591631174087524, 0.9108020067214966)
st = scene.frame_stats()
print(f"{label}; 1920x1080, {depth} segment(s), options {sys.argv[3:]}")
print(f"  closest-hit rays {st['closest_rays']}, any-hit rays {st['any_rays']}; uniform node steps: {st['nodes_closest_uniform']} of {st['nodes_closest']} closest-hit visits, "
      f"{st['nodes_any_uniform']} of {st['nodes_any']} any-hit visits" + ("  [one 4-sample step in the lanes form]" if lanes_form else ""))
for what, v, w in (("closest-hit node block", "nodes_closest", "wave_steps_closest_nodes"), ("closest-hit triangle block", "tris_closest", "wave_steps_closest_tris"),
                   ("any-hit node block", "nodes_any", "wave_steps_any_nodes"), ("any-hit triangle block", "tris_any", "wave_steps_any_tris")):
    if st[w]:
        print(f"  {what:28s} {st[v]:12d} lane visits / {st[w]:10d} wave steps = {st[v] / st[w]:5.1f} lanes of 64 = {100.0 * st[v] / (64 * st[w]):5.1f} %")
