"""The N > 1 path without GPUs: world_size-2 (and 3) process groups over gloo.  Every rank renders its round-robin
share of the 8x8 tiles, rank 0 gathers the tile buffers (rust-tracing_amd/dist.py, the code bench.py runs over
RCCL) and the reassembled frame must be the single-process frame, bit for bit."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import scene_cases

ROOT = Path(__file__).resolve().parent.parent


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def tiles_to_frame(gathered, w, h, world, stride):
    """numpy statement of rt_tiles_to_frame_device (include/rt_amd.h, RT_OUT_TILES)."""
    tiles_x = (w + 7) // 8
    frame = np.zeros((h, w, 3))
    for j in range(h):
        for i in range(w):
            k = (j // 8) * tiles_x + i // 8
            shard, lt = k % world, k // world
            src = shard * stride + ((lt * 8 + j % 8) * 8 + i % 8) * 3
            frame[j, i] = gathered[src:src + 3]
    return frame.reshape(-1)


@pytest.mark.parametrize("world", [2, 3])
def test_tile_shards_gather_into_the_single_process_frame(rt, oracle, tmp_path, world):
    case = "ragged_random_balls_53x29_4spp"
    out = tmp_path / "gathered.npy"
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "_gloo_worker.py"), case, str(out)], env=env))
    try:
        for p in procs:
            assert p.wait(timeout=180) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    import importlib
    rtdist = importlib.import_module("rust-tracing_amd.dist")
    hs = scene_cases.build(rt, case)
    w, h = hs.width, hs.height
    stride = rtdist.shard_stride(w, h, world)
    assert stride == rt.amd_lib().rt_out_size(w, h, rt.RT_OUT_TILES, 0, world)
    gathered = np.load(out)
    assert gathered.size == stride * world
    whole = oracle.render(hs, rt.render_params(seed=11))
    frame = tiles_to_frame(gathered, w, h, world, stride)
    assert np.array_equal(frame.view(np.uint64), whole.view(np.uint64))
    # every tile went to exactly one shard: shard sizes add up to the tile count
    tiles = ((w + 7) // 8) * ((h + 7) // 8)
    assert sum(rt.out_size(w, h, rt.RT_OUT_TILES, r, world) for r in range(world)) == tiles * 64 * 3


def test_bench_starts_its_own_ranks_when_asked_for_more_than_one_gpu(rt):
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): the script decides from its arguments alone, before touching
    any GPU, to run torch.distributed.run with two ranks of itself as a child.  Without a GPU every rank then stops at the renderer's
    'no HIP device' error — which is what this CPU test can see of it (the GPU suite runs the same command to the end)."""
    if rt.amd_lib().rt_device_count() > 0:
        pytest.skip("covered by tests/test_gpu_cli.py on a GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "c1", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))
    assert r.returncode != 0
    # (the launcher stops the other rank as soon as one has failed: the message appears once or twice)
    assert "bench.py rank" in r.stderr and "no HIP device for local rank 0" in r.stderr and "ChildFailedError" in r.stderr, r.stderr[-1500:]
    assert "must be launched with" not in r.stderr
