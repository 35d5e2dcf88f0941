"""One rank of the CPU rehearsal of the multi-GPU path (tests/test_sharding_gloo.py): renders its shard of the
frame's tiles with the CPU oracle (standing in for the GPU), then takes part in the frame-end gather through the
same rust-tracing_amd/dist.py code bench.py uses, over gloo."""
import importlib
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
import torch.distributed as dist

import oracle_lib
import scene_cases

rt = importlib.import_module("rust-tracing_amd")
rtdist = importlib.import_module("rust-tracing_amd.dist")


def main():
    case, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hs = scene_cases.build(rt, case)
    w, h = hs.width, hs.height
    stride = rtdist.shard_stride(w, h, world)
    tiles = torch.zeros(stride, dtype=torch.float64)
    params = rt.render_params(seed=11, shard_index=rank, shard_count=world, out_layout=rt.RT_OUT_TILES)
    n = int(oracle_lib.lib().orc_out_size(w, h, rt.RT_OUT_TILES, rank, world))
    oracle_lib.render(hs, params, threads=2, out=tiles.numpy()[:n])
    gathered = torch.zeros(stride * world, dtype=torch.float64) if rank == 0 else None
    rtdist.gather_tiles(tiles, gathered, rank, world)
    if rank == 0:
        np.save(out_path, gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
