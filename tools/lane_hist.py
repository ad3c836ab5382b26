#!/usr/bin/env python3
"""Enabled lanes per node step of the counting kernels (crt_debug_step_hist): the distribution behind the lane-utilisation figures.

usage: python tools/lane_hist.py [workload] [depth]      e.g. mesh1m 4
Prints, for closest-hit and any-hit walks: node steps, mean enabled lanes, and the share of steps — and of the enabled-lane visits — that ran
with at most 32 / 16 / 8 lanes: what a scheme that gives the remaining rays 2 / 4 / 8 lanes each could address.
"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
g.build()
import caitlynrenderer_amd as cr
sys.argv += ["mesh1m", "4"][len(sys.argv) - 1:]
name, depth = sys.argv[1], int(sys.argv[2])
import bench
data, cam, label, _ = bench.build_workload(name)
scene = cr.Scene(data, 1920, 1080, depth)
scene.set_option("lanes_per_ray", 1)
scene.set_option("count_visits", 1)
rnd = cr.Rnd()
scene.debug_step_hist()                       # start
for _ in range(2):
    scene.render_frame(rnd.randf2(), rnd.randf2())
closest, anyh = scene.debug_step_hist()
scene.debug_step_hist(stop=True)
print(f"{label}; 1920x1080, {depth} segment(s), 2 counting frames, one lane per ray")
k = np.arange(65)
for nm, h in (("closest-hit", closest), ("any-hit", anyh)):
    h = h.astype(np.float64)
    steps, lanes = h.sum(), (h * k).sum()
    if steps == 0:
        continue
    line = f"{nm:12s} node steps {int(steps):10d}  mean enabled lanes {lanes / steps:5.1f} ({lanes / steps / 64:.1%})"
    for lim in (32, 16, 8):
        line += f" | <= {lim:2d} lanes: {h[:lim + 1].sum() / steps:5.1%} of the steps, {(h * k)[:lim + 1].sum() / lanes:5.1%} of the visits"
    print(line)
    # what K lanes per ray would save: a step with n <= 64 / K lanes costs c_K instead of 230 instructions
    base = steps * 230.0
    for K, cK in ((2, 116.0), (4, 89.0)):
        lim = 64 // K
        cost = h[:lim + 1].sum() * cK + h[lim + 1:].sum() * 230.0
        print(f"{'':12s} {K} lanes per ray when <= {lim} lanes are busy (node step {cK:.0f} instead of 230 instructions): node-step instructions x {cost / base:.3f}")
scene.close()
