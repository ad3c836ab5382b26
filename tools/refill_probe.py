#!/usr/bin/env python3
"""What lane refill is worth on COHERENT rays, measured on the explicit-ray-buffer kernel (k_trace: 256-ray pools per wave,
traverse_pool).  refill_min = 64 means "refill only when every lane is idle", i.e. four lock-step 64-ray batches per pool — the way
the fused segment kernel walks its first-segment rays; smaller values let a lane whose ray has finished take the pool's next ray.

  python tools/refill_probe.py [workload] [WxH]

Rays: the jittered primary rays of one frame in 8x8-block order (a wave's 64 rays = one pixel block), and the first segment's NEE
shadow rays in queue order (compacted, as the shadow-queue form emits them)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    import torch
    import caitlynrenderer_amd as cr
    from caitlynrenderer_amd import RAY_DT, HIT_DT, CRT_TRACE_CLOSEST, CRT_TRACE_ANY
    name = sys.argv[1] if len(sys.argv) > 1 else "mesh1m"
    W, H = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
    data, cam, label, _ = bench.build_workload(name)
    sc = cr.Scene(data, W, H, max_depth=2)
    sc.update(cam)
    rv = (0.6591631, 0.910802)
    # the frame's own rays: primary rays through the path-ray queue's debug read-out are not available for segment 0, so they come
    # from the shadow-queue form's counters run: render once with the queue form and read the shadow rays back
    sc.set_option("inplace_shadow", 0)
    sc.render_frame(*rv)
    shadow = sc.debug_read_queue(2, 0)
    bounce = sc.debug_read_queue(0, 1)           # the rays entering the second segment: cosine-distributed off every surface
    sc.set_option("inplace_shadow", 1)
    # primary rays: generated on the host by the same camera arithmetic (oracle-free: pinhole through pixel centres is enough for a
    # timing probe; jitter would move each ray by less than a pixel)
    c = cam.c if hasattr(cam, "c") else cam
    pos = np.array([c.position[k] for k in range(3)], np.float32)
    right = np.array([c.right[k] for k in range(3)], np.float32)
    up = np.array([c.up[k] for k in range(3)], np.float32)
    fwd = np.array([c.forward[k] for k in range(3)], np.float32)
    tan = np.float32(np.tan(np.float32(c.fov) * np.float32(0.5)))
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    dx = (2 * (xs + 0.5) / W - 1) * tan * np.float32(W / H)
    dy = (2 * (ys + 0.5) / H - 1) * tan
    d = right[None, None] * dx[..., None] + up[None, None] * dy[..., None] + fwd[None, None]
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    Hc, Wc = H // 8 * 8, W // 8 * 8
    order = np.arange(H * W).reshape(H, W)[:Hc, :Wc].reshape(Hc // 8, 8, Wc // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
    rays = np.zeros(order.size, RAY_DT)
    names = RAY_DT.names
    flat = d.reshape(-1, 3)[order]
    raw = np.zeros((order.size, 8), np.float32)
    raw[:, 0:3] = pos
    raw[:, 3] = 1e30
    raw[:, 4:7] = flat
    rays = raw.view(RAY_DT).reshape(-1)
    print(f"{label}; {W}x{H}: {rays.size} primary rays (8x8-block order), {shadow.size} shadow rays, {bounce.size} bounce rays (queue order); ray record fields {names}")
    for tag, r, mode in (("closest, primary rays", rays, CRT_TRACE_CLOSEST), ("any-hit, first-segment shadow rays", shadow, CRT_TRACE_ANY),
                         ("closest, bounce rays", bounce, CRT_TRACE_CLOSEST)):
        n = r.size
        d_r = torch.from_numpy(np.ascontiguousarray(r).view(np.uint8).copy()).cuda()
        d_h = torch.empty(n * HIT_DT.itemsize, dtype=torch.uint8, device="cuda")
        base = None
        for pool, refill in ((256, 64), (256, 16), (256, 8), (128, 64), (128, 16), (128, 8), (64, 64), (256, 16), (128, 16), (64, 64)):
            sc.set_option("trace_pool", pool)
            sc.set_option("refill_min", refill)
            t_settle = time.perf_counter()            # clocks up, caches warm (bench.py --settle-ms)
            while time.perf_counter() - t_settle < 0.25:
                for _ in range(10):
                    sc.trace_device(d_r.data_ptr(), n, d_h.data_ptr(), mode, sync=False)
                sc.sync()
            reps = 200
            t0 = time.perf_counter()
            for _ in range(reps):
                sc.trace_device(d_r.data_ptr(), n, d_h.data_ptr(), mode, sync=False)
            sc.sync()
            ms = (time.perf_counter() - t0) / reps * 1e3
            h = d_h.cpu().numpy().view(HIT_DT)
            chk = int(np.bitwise_xor.reduce(h.view(np.uint32).reshape(-1)))
            if base is None:
                base = (ms, chk)
            print(f"  {tag}: pool {pool:3d} refill_min {refill:2d}: {ms:.4f} ms = {n / ms / 1e3:8.1f} Mray/s  ({base[0] / ms:.3f} x the 256-ray lock-step pools), hits identical: {chk == base[1]}")
    sc.close()


if __name__ == "__main__":
    main()
