"""Tile sharding + read-back gather, world_size 2 over gloo on the CPU (the N>1 path of bench.py).
Rendering in these CPU tests is done by the oracle — the product has no CPU path."""
import os
import socket

import numpy as np
import pytest


def test_tile_order_is_a_partition(cr):
    from caitlynrenderer_amd import tiles
    order = tiles.tile_order(1920, 1080, 64)
    assert len(order) == 30 * 17 and len(set(order)) == len(order)
    for world in (1, 2, 3, 8):
        parts = [tiles.local_tiles(1920, 1080, 64, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == sorted(order)
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    dy, dx = tiles.pixel_grid(64)
    assert sorted(zip(dy.tolist(), dx.tolist())) == [(y, x) for y in range(64) for x in range(64)]
    # one 64-lane wave = one 8x8 pixel block
    assert dy[:64].max() == 7 and dx[:64].max() == 7


def test_library_deals_the_same_tiles_without_a_gpu(cr):
    """The multi-GPU bookkeeping of libcrt.so itself (crt_shard_tiles, the arithmetic crt_set_shard / crt_set_devices / option "streams"
    deal their tiles with) on no GPU at all: 8 ranks, and 8 devices behind one handle, get exactly the Python mirror's lists — a
    partition of the 4K frame of BASELINE configs[4] —, and a shard split again over streams is that shard's list dealt round-robin."""
    from caitlynrenderer_amd import tiles
    for (W, H, T) in ((3840, 2160, 16), (3840, 2160, 64), (250, 140, 16), (8, 8, 8)):
        order = tiles.tile_order(W, H, T)
        for world in (1, 2, 8):
            per_rank = [tiles.shard_tiles_of_library(W, H, T, r, world) for r in range(world)]              # crt_set_shard(r, world)
            per_dev = [tiles.shard_tiles_of_library(W, H, T, 0, 1, k, world) for k in range(world)]         # crt_set_devices, device k of `world`
            assert per_rank == per_dev == [tiles.local_tiles(W, H, T, r, world) for r in range(world)]
            assert sorted(sum(per_rank, [])) == sorted(order)
            assert max(map(len, per_rank)) - min(map(len, per_rank)) <= 1
        # option "streams" = 3 on rank 5 of 8: stream j renders tiles j, j + 3, ... of that rank's own list
        mine = tiles.local_tiles(W, H, T, 5, 8)
        parts = [tiles.shard_tiles_of_library(W, H, T, 5, 8, j, 3) for j in range(3)]
        assert parts == [mine[j::3] for j in range(3)]
    import pytest
    with pytest.raises(cr.CrtError):
        tiles.shard_tiles_of_library(64, 64, 12)           # tile not a multiple of 8
    with pytest.raises(cr.CrtError):
        tiles.shard_tiles_of_library(64, 64, 16, 2, 2)     # rank >= world
    with pytest.raises(cr.CrtError):
        tiles.shard_tiles_of_library(64, 64, 16, 0, 1, 3, 3)


def test_untile_roundtrip(cr):
    from caitlynrenderer_amd import tiles
    W, H, T = 100, 70, 16
    rng = np.random.default_rng(0)
    full = rng.random((H, W, 3)).astype(np.float32)
    frame = np.zeros_like(full)
    for r in range(3):
        tl = tiles.local_tiles(W, H, T, r, 3)
        dy, dx = tiles.pixel_grid(T)
        packed = np.zeros((len(tl), T * T, 3), np.float32)
        for t, (tx, ty) in enumerate(tl):
            py, px = ty * T + dy, tx * T + dx
            ok = (py < H) & (px < W)
            packed[t][ok] = full[py[ok], px[ok]]
        tiles.untile_into(frame, packed, tl, T)
    assert np.array_equal(frame, full)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, T, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import json
    import torch
    import torch.distributed as dist
    import caitlynrenderer_amd as cr
    from caitlynrenderer_amd import tiles
    from oracle import binding as ob
    import __graft_entry__ as g
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mesh, cam = g._cornell()
    data = cr.SceneData.build(mesh, cam)
    o = ob.Oracle(data, W, H, 1, cam)
    # each rank renders only the pixel rows its tiles touch, then packs its own tiles
    full = np.zeros((H, W, 3), np.float32)
    tl = tiles.local_tiles(W, H, T, rank, world)
    for ty in sorted({ty for _, ty in tl}):
        o.render_rows(0.6591631, 0.910802, ty * T, min(H, (ty + 1) * T), full)
    dy, dx = tiles.pixel_grid(T)
    packed = np.zeros((len(tl), T * T, 3), np.float32)
    for t, (tx, ty) in enumerate(tl):
        py, px = ty * T + dy, tx * T + dx
        ok = (py < H) & (px < W)
        packed[t][ok] = full[py[ok], px[ok]]
    frame = tiles.gather_frame(torch.from_numpy(packed.ravel()), W, H, T, rank, world)
    np.save(os.path.join(out_dir, f"frame{rank}.npy"), frame)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gather_equals_single_process(cr, ob, cornell, cornell_data, tmp_path):
    import torch.multiprocessing as mp
    W, H, T = 96, 56, 16
    port = _free_port()
    mp.spawn(_worker, args=(2, port, W, H, T, str(tmp_path)), nprocs=2, join=True)
    o = ob.Oracle(cornell_data, W, H, 1, cornell[1])
    want, _ = o.render_frame(0.6591631, 0.910802)
    for r in range(2):
        got = np.load(tmp_path / f"frame{r}.npy")
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
