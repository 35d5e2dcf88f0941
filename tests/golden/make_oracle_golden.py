#!/usr/bin/env python3
"""Generates tests/golden/oracle_frames.json: for every case of tests/scene_cases.py the SHA-256 of the oracle's
f64 sum buffer, of the resolved RGB8 image, a few probe pixels (bit patterns) and the event counters.

These are REGRESSION vectors for this repo's own seeded pipeline (host scene builder -> oracle); they do not pin
the oracle to the reference (tests/test_reference_pins.py does that, statistically).  Re-run after any deliberate
change of the normative definitions in include/rt_amd.h:   python tests/golden/make_oracle_golden.py
"""
import hashlib
import importlib
import json
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE.parent))
import numpy as np
import oracle_lib
import scene_cases

rt = importlib.import_module("rust-tracing_amd")
SEED = 20231003


def record(name):
    hs = scene_cases.build(rt, name)
    out, cnt = oracle_lib.render(hs, rt.render_params(seed=SEED), want_counters=True)
    spp = hs.camera.samples_per_pixel
    rgb = rt.resolve_rgb8_host(hs.width, hs.height, spp, out)
    rng = np.random.default_rng(0)
    probes = sorted(set(int(i) for i in rng.integers(0, out.size, 12)))
    return {"width": hs.width, "height": hs.height, "spp": spp, "seed": SEED,
            "sums_sha256": hashlib.sha256(out.tobytes()).hexdigest(),
            "rgb8_sha256": hashlib.sha256(rgb.tobytes()).hexdigest(),
            "mean_radiance": float(out.mean() / spp),
            "probes": {str(i): int(out.view(np.uint64)[i]) for i in probes},
            "counters": {k: cnt[k] for k in ("samples", "rays", "rng_draws", "sphere_tests", "quad_tests", "medium_visits",
                                             "noise_evals", "image_lookups", "node_visits")}}


def main():
    result = {"note": "oracle outputs, reference-faithful mode; see make_oracle_golden.py", "cases": {}}
    for name in scene_cases.CASES:
        result["cases"][name] = record(name)
        print(name, result["cases"][name]["sums_sha256"][:16], result["cases"][name]["mean_radiance"])
    (HERE / "oracle_frames.json").write_text(json.dumps(result, indent=1) + "\n")


if __name__ == "__main__":
    main()
