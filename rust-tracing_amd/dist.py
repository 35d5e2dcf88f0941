"""Frame-end exchange for the multi-GPU path: one process per GPU, tiles dealt round-robin, ONE gather.

The render itself needs no communication (every pixel is independent and its random stream is keyed by
(seed, pixel, sample), so a tile renders the same on any GPU).  At frame end each rank contributes its compact
tile buffer (RT_OUT_TILES layout, include/rt_amd.h) and rank 0 receives them back to back:

    [ shard 0 tiles | shard 1 tiles | ... ]     each padded to the size of shard 0, the largest

`torch.distributed.gather` over the "nccl" backend is RCCL on ROCm: on an 8-GPU MI355X node the seven incoming
buffers arrive over seven direct xGMI links.  The same function runs over "gloo" on CPU tensors, which is how
the partition / gather / reassembly logic is tested without GPUs (tests/test_sharding_gloo.py).
"""
from __future__ import annotations

import importlib

_rt = importlib.import_module(__name__.rsplit(".", 1)[0])


def shard_stride(width: int, height: int, world: int) -> int:
    """Doubles per rank in the gathered buffer: shard 0 owns ceil(tiles / world) tiles, nobody owns more."""
    tiles_x = (width + _rt.RT_TILE_W - 1) // _rt.RT_TILE_W
    tiles_y = (height + _rt.RT_TILE_H - 1) // _rt.RT_TILE_H
    local = (tiles_x * tiles_y + world - 1) // world
    return local * _rt.RT_TILE_W * _rt.RT_TILE_H * 3


def gather_tiles(local_tiles, gathered, rank: int, world: int, dst: int = 0):
    """local_tiles: 1-D float64 tensor of shard_stride() elements on every rank.
    gathered: on `dst`, a 1-D tensor of world * shard_stride() elements (ignored elsewhere)."""
    import torch.distributed as dist
    if world == 1:
        if gathered is not None and gathered.data_ptr() != local_tiles.data_ptr():
            gathered.copy_(local_tiles)
        return
    if rank == dst:
        n = local_tiles.numel()
        dist.gather(local_tiles, [gathered[r * n:(r + 1) * n] for r in range(world)], dst=dst)
    else:
        dist.gather(local_tiles, None, dst=dst)
