#!/usr/bin/env python3
"""Long fuzz run of tests/custom_scenes.py::random_scene: GPU (the reference-order walk, the library's own trees with two and with
four children per record, with the default and with odd settings of the walk's shortcuts) against the oracle, bit for bit.
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
bad = 0
for seed in range(first, first + count):
    scene = custom_scenes.random_scene(cam, seed)
    for render_seed in (5, 6):
        params = rt.render_params(seed=render_seed)
        want = oracle_lib.render(scene, params)
        variants = [dict(walk=rt.RT_WALK_REFERENCE_ORDER), dict(walk=rt.RT_WALK_OWN_TREES, wide=0), dict(walk=rt.RT_WALK_OWN_TREES, wide=1),
                    dict(walk=rt.RT_WALK_OWN_TREES, wide=1, flat_max=seed % 9, leaf_max=1 + seed % 8, start_shortcut=seed % 2, defer_instances=(seed >> 1) % 2,
                         seq_lookahead=(seed >> 2) % 2, slow_min=1 + seed % 5, slow_age=seed % 40, quad_filter=(seed >> 3) % 2, medium_first=(seed >> 4) % 2)]
        for opts in variants:
            got = rt.DeviceScene(scene, **opts).render(params)
            if not np.array_equal(got.view(np.uint64), want.view(np.uint64)):
                bad += 1
                print("MISMATCH scene", seed, "render seed", render_seed, opts, int((got.view(np.uint64) != want.view(np.uint64)).sum()), "values", flush=True)
    if seed % 100 == 99:
        print("...", seed + 1 - first, "scenes,", bad, "mismatches", flush=True)
print("done:", count, "scenes,", bad, "mismatches")
sys.exit(1 if bad else 0)
