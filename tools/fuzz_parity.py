#!/usr/bin/env python3
"""Long fuzz run of tests/custom_scenes.py::random_scene: GPU (both walks) against the oracle, bit for bit.
Usage: python tools/fuzz_parity.py [first_seed] [count]"""
import importlib, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
rt = importlib.import_module("rust-tracing_amd")
import custom_scenes, scene_cases, oracle_lib

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 500)
cam = scene_cases.build(rt, "ragged_cornell_37x37_4spp")
lib = rt.amd_lib()
bad = 0
for seed in range(first, first + count):
    scene = custom_scenes.random_scene(cam, seed)
    for render_seed in (5, 6):
        params = rt.render_params(seed=render_seed)
        want = oracle_lib.render(scene, params)
        for ordered in (2, 0):
            lib.rt_debug_set_traversal(ordered, 0)
            got = rt.DeviceScene(scene).render(params)
            if not np.array_equal(got.view(np.uint64), want.view(np.uint64)):
                bad += 1
                print("MISMATCH scene", seed, "render seed", render_seed, "ordered", ordered, int((got.view(np.uint64) != want.view(np.uint64)).sum()), "values", flush=True)
    if seed % 100 == 99:
        print("...", seed + 1 - first, "scenes,", bad, "mismatches", flush=True)
lib.rt_debug_set_traversal(1, 0)
print("done:", count, "scenes,", bad, "mismatches")
sys.exit(1 if bad else 0)
