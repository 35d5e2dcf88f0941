"""The oracle against the committed golden vectors (tests/golden/oracle_frames.json) — guards the seeded pipeline
(host scene build -> describe -> oracle) against silent drift on the CPU, and the GPU against the same vectors
without running the oracle at all."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

import scene_cases

GOLD = json.loads((Path(__file__).parent / "golden" / "oracle_frames.json").read_text())["cases"]


def check(rt, hs, out, g, counters=None):
    assert (hs.width, hs.height, hs.camera.samples_per_pixel) == (g["width"], g["height"], g["spp"])
    bits = out.view(np.uint64)
    for i, want in g["probes"].items():
        assert int(bits[int(i)]) == want, f"probe {i}"
    assert hashlib.sha256(out.tobytes()).hexdigest() == g["sums_sha256"]
    rgb = rt.resolve_rgb8_host(hs.width, hs.height, g["spp"], out)
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == g["rgb8_sha256"]
    if counters:
        for k, v in g["counters"].items():
            assert counters[k] == v, k


@pytest.mark.parametrize("name", list(GOLD))
def test_oracle_matches_golden(rt, oracle, name):
    hs = scene_cases.build(rt, name)
    out, cnt = oracle.render(hs, rt.render_params(seed=GOLD[name]["seed"]), want_counters=True)
    check(rt, hs, out, GOLD[name], cnt)


def test_oracle_is_thread_count_independent(rt, oracle):
    hs = scene_cases.build(rt, "c2_random_balls_96x64_8spp_d50")
    p = rt.render_params(seed=1)
    a = oracle.render(hs, p, threads=1); b = oracle.render(hs, p, threads=7)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("wide", [0, 1])
@pytest.mark.parametrize("name", list(GOLD))
def test_gpu_matches_golden(rt, gpu, name, wide):
    """(wide: the library's own trees with two or with four children per record, rt_scene_options.wide)"""
    hs = scene_cases.build(rt, name)
    out = rt.DeviceScene(hs, wide=wide).render(rt.render_params(seed=GOLD[name]["seed"]))
    check(rt, hs, out, GOLD[name])
