"""Framebuffer tile sharding across GPUs and the read-back gather.

No reference counterpart: the reference is one OpenGL context.  Pixels are independent, so the
frame is cut into tile x tile squares, dealt round-robin in Morton order to the ranks (the Cornell
view misses 44 % of a 16:9 frame, so contiguous bands would be unbalanced), every rank accumulates
its own tiles locally, and the only communication is one all-gather of the packed RGB32F tile
buffers at read-back (RCCL over xGMI when the process group is "nccl"; "gloo" in the CPU tests).
`tile_order` must stay identical to build_shard() in csrc/crt_device.cpp.
"""
import numpy as np


def _spread(v):
    v = v & 0xFFFF
    v = (v | (v << 8)) & 0x00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v


def tile_order(width, height, tile):
    """All tiles of the frame as (tx, ty), sorted by Morton code."""
    nx, ny = (width + tile - 1) // tile, (height + tile - 1) // tile
    items = sorted((_spread(x) | (_spread(y) << 1), x, y) for y in range(ny) for x in range(nx))
    return [(x, y) for _, x, y in items]


def local_tiles(width, height, tile, rank, world):
    return tile_order(width, height, tile)[rank::world]


def shard_tiles_of_library(width, height, tile, rank=0, world=1, device=0, n_devices=1):
    """crt_shard_tiles: the (tx, ty) list libcrt.so itself deals to logical device `device` of `n_devices` inside shard `rank` of `world`
    (host arithmetic, no GPU) — what `local_tiles` must agree with."""
    import ctypes as C
    from . import _lib
    n = C.c_size_t()
    _lib.check(_lib.lib().crt_shard_tiles(width, height, tile, rank, world, device, n_devices, None, 0, C.byref(n)))
    xy = np.zeros((n.value, 2), np.uint32)
    _lib.check(_lib.lib().crt_shard_tiles(width, height, tile, rank, world, device, n_devices, xy.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
    return [(int(x), int(y)) for x, y in xy]


def max_local_tiles(width, height, tile, world):
    n = len(tile_order(width, height, tile))
    return (n + world - 1) // world


def pixel_grid(tile):
    """(dy, dx) of the in-tile pixel order: 8x8 blocks row-major, pixels row-major inside a block."""
    j = np.arange(tile * tile)
    blk, k = j >> 6, j & 63
    bpr = tile >> 3
    dx = (blk % bpr) * 8 + (k & 7)
    dy = (blk // bpr) * 8 + (k >> 3)
    return dy, dx


def untile_into(frame, packed, tiles, tile):
    """Scatter one rank's packed (n_tiles, tile*tile, 3) buffer into frame (H, W, 3)."""
    H, W = frame.shape[0], frame.shape[1]
    dy, dx = pixel_grid(tile)
    packed = np.asarray(packed).reshape(-1, tile * tile, 3)
    for t, (tx, ty) in enumerate(tiles):
        py, px = ty * tile + dy, tx * tile + dx
        ok = (py < H) & (px < W)
        frame[py[ok], px[ok]] = packed[t][ok]
    return frame


def gather_frame(local_packed, width, height, tile, rank, world, group=None):
    """All-gather the packed tile buffers and un-tile; returns the full (H, W, 3) frame on every rank.

    local_packed: torch tensor (n_local_tiles*tile*tile*3,) on the device the process group uses.
    """
    import torch
    import torch.distributed as dist
    cap = max_local_tiles(width, height, tile, world) * tile * tile * 3
    send = local_packed.new_zeros(cap)
    send[: local_packed.numel()] = local_packed
    recv = local_packed.new_empty(world * cap)
    dist.all_gather_into_tensor(recv, send, group=group)
    frame = np.zeros((height, width, 3), np.float32)
    host = recv.cpu().numpy().reshape(world, cap)
    for r in range(world):
        tl = local_tiles(width, height, tile, r, world)
        untile_into(frame, host[r][: len(tl) * tile * tile * 3], tl, tile)
    return frame


def gather_packed_to_root(local_packed, recv, world, group=None, dst=0):
    """Read-back collective of bench.py: every rank sends its (padded) packed tile buffer to rank `dst`
    (RCCL grouped send/recv: each peer's slice crosses its own xGMI link once).  `recv` is a
    (world * local_packed.numel()) tensor on the root, ignored elsewhere."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    if rank == dst:
        parts = list(recv.view(world, -1).unbind(0))
        dist.gather(local_packed, parts, dst=dst, group=group)
    else:
        dist.gather(local_packed, None, dst=dst, group=group)


def untile_gathered(recv_host, width, height, tile, world):
    """Root side: (world, cap) packed buffers -> (H, W, 3) frame."""
    frame = np.zeros((height, width, 3), np.float32)
    cap = recv_host.shape[1]
    for r in range(world):
        tl = local_tiles(width, height, tile, r, world)
        untile_into(frame, recv_host[r][: len(tl) * tile * tile * 3], tl, tile)
    return frame
