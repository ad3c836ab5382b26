#!/usr/bin/env python3
"""Enabled lanes per node step of the counting kernels (crt_debug_step_hist): the distribution behind the lane-utilisation figures.

usage: python tools/lane_hist.py [workload] [depth] [distinct]      e.g. mesh1m 4; `distinct`: node steps by distinct nodes in the wave (timed launch form)
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
distinct = "distinct" in sys.argv[1:]
sys.argv = [a for a in sys.argv if a != "distinct"]
sys.argv += ["mesh1m", "4"][len(sys.argv) - 1:]
name, depth = sys.argv[1], int(sys.argv[2])
import bench
data, cam, label, _ = bench.build_workload(name)
scene = cr.Scene(data, 1920, 1080, depth)
rnd = cr.Rnd()
if distinct:
    # `distinct`: ONE 4-sample step in the form the bench's launches have (count_visits 2), node steps by the number of DISTINCT (node, octant)
    # keys among the step's enabled lanes: 1 = a uniform step (scalar-unit node test, 124 + ~45 instructions), n > 1 = a general step (214 + ~45
    # and five vector loads per lane) — and what a PACKET walk would execute instead: one scalar-unit step per distinct node, n x (124 + ~60)
    scene.render_frame(rnd.randf2(), rnd.randf2())
    scene.set_option("step_hist_mode", 1)
    scene.set_option("count_visits", 2)
    scene.debug_step_hist()
    scene.render_frames([(rnd.randf2(), rnd.randf2()) for _ in range(4)])
    closest, anyh = scene.debug_step_hist()
    scene.debug_step_hist(stop=True)
    print(f"{label}; 1920x1080, {depth} segment(s), one 4-sample step in the lanes form: node steps by distinct nodes in the wave")
    k = np.arange(65)
    for nm, h, uni in (("closest-hit", closest, 124), ("any-hit", anyh, 120)):
        h = h.astype(np.float64)
        steps = h.sum()
        if steps == 0:
            continue
        now = h[1] * (uni + 45.0) + h[2:].sum() * (214.0 + 45.0)
        packet = (h * k).sum() * (uni + 60.0)
        print(f"{nm:12s} node steps {int(steps):9d}: uniform {h[1] / steps:5.1%}, 2 nodes {h[2] / steps:5.1%}, 3 {h[3] / steps:5.1%}, 4 {h[4] / steps:5.1%}, 5-8 {h[5:9].sum() / steps:5.1%}, "
              f"9-16 {h[9:17].sum() / steps:5.1%}, more {h[17:].sum() / steps:5.1%}; mean distinct nodes per non-uniform step {(h * k)[2:].sum() / max(1.0, h[2:].sum()):.2f}")
        print(f"{'':12s} node-step vector instructions now ~{now / 1e6:.1f} M, as one scalar-unit step per distinct node ~{packet / 1e6:.1f} M (x {packet / now:.2f}); "
              f"pairs only (2 distinct -> two scalar-unit steps): x {(h[1] * (uni + 45.0) + h[2] * 2 * (uni + 60.0) + h[3:].sum() * 259.0) / now:.3f}")
    scene.close()
    sys.exit(0)
scene.set_option("lanes_per_ray", 1)
scene.set_option("count_visits", 1)
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
