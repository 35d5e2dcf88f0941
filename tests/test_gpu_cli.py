"""The command-line renderer (rust-tracing_amd/host/main.cpp, the reference's `main`): runs on the GPU, writes the PNG
the Python path produces for the same scene and seeds, single-pass or progressive."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rtrace_writes_the_same_png_as_the_library_path(rt, gpu, tmp_path):
    from PIL import Image
    exe = rt.LIB_DIR / "rtrace"
    assert exe.exists(), "run build() first"
    args = ["-s", "6", "--width", "96", "--spp", "24", "--depth", "8", "--seed", "5", "--scene-seed", "1"]
    out1 = tmp_path / "single"
    r = subprocess.run([str(exe), *args, "-o", str(out1)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Building BVH" in r.stdout and "Render time" in r.stdout and "PNG encoding" in r.stdout  # the reference's three timings
    out2 = tmp_path / "progressive"
    r = subprocess.run([str(exe), *args, "--progressive", "7", "-o", str(out2)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    a = np.asarray(Image.open(str(out1) + ".png").convert("RGB"))
    b = np.asarray(Image.open(str(out2) + ".png").convert("RGB"))
    assert a.shape == (96, 96, 3) and np.array_equal(a, b)
    hs = rt.HostScene(6, scene_seed=1, width=96, spp=24, depth=8)
    sums = rt.DeviceScene(hs).render(rt.render_params(seed=5))
    assert np.array_equal(rt.resolve_rgb8_host(96, 96, 24, sums), a)
    # -l/--live needs a window system: refused, not ignored
    r = subprocess.run([str(exe), "-l"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "live" in r.stderr


def test_bench_launches_its_own_ranks_from_a_bare_shell(rt, gpu):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts torch.distributed.run itself (as a child, before
    anything touches the GPU) — here over gloo, both ranks on the box's one GPU — and prints one line for n_gpus = 2."""
    import json
    import os
    import sys
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "c1", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "strong"
    assert "gloo" in line["gather"] and line["config"]["workload"].startswith("random-spheres 400x225")


def test_bench_with_more_ranks_than_gpus_fails_loudly_and_soon(rt, gpu):
    """`python bench.py --gpus 2 --backend nccl` on a box with ONE GPU: the rank without a device says so on stderr and the whole
    launch ends with a non-zero exit code well inside three minutes — it must not sit in a rendezvous or an RCCL bootstrap.  (What the
    first contact with a real multi-GPU node would look like if a device were missing or refused.)"""
    import os
    import sys
    import time
    if gpu > 1:
        pytest.skip("this box has a second GPU: the launch would be a real two-rank run")
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "nccl", "--workload", "c1", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=str(root))
    took = time.time() - t0
    assert r.returncode != 0, r.stdout[-2000:]
    assert took < 180, f"took {took:.0f} s"
    assert "bench.py rank 1: no HIP device for local rank 1" in r.stderr, r.stderr[-3000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "a result line was printed by a launch that failed"
