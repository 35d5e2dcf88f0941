"""The oracle's fixed transcendental algorithms (include/rt_amd.h "Arithmetic") against the platform libm, and the
per-path generator's basic properties.  (The GPU's copies are compared with these bit for bit in
tests/test_gpu_parity.py.)"""
import math

import numpy as np


def ulps(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.spacing(np.maximum(np.abs(b), np.finfo(np.float64).tiny))


def test_log(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.integers(1, 1 << 53, 50000).astype(np.float64) * 2.0 ** -53,
                        rng.uniform(1e-300, 1e300, 1000), [0.5, 1.0, 2.0, math.sqrt(0.5), 5e-324, 2.0 ** -53]])
    got = np.array([L.orc_log(float(v)) for v in x])
    assert ulps(got, np.log(x)).max() <= 1.0
    assert L.orc_log(0.0) == -math.inf and L.orc_log(1.0) == 0.0 and math.isnan(L.orc_log(-1.0))


def test_sin(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-5000, 5000, 50000), rng.uniform(-1, 1, 5000), [0.0, 1e-300, math.pi, 1e5, -1e5]])
    got = np.array([L.orc_sin(float(v)) for v in x])
    want = np.sin(x)
    assert np.abs(got - want).max() <= 2.3e-16          # absolute: what a colour needs
    big = np.abs(want) > 1e-3
    assert ulps(got[big], want[big]).max() <= 2.0
    assert math.isnan(L.orc_sin(math.inf)) and math.isnan(L.orc_sin(2e6))


def test_acos_atan2_pow5(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-1, 1, 50000), [-1.0, 1.0, 0.0, 0.5, -0.5, 1 - 1e-16, -1 + 1e-16]])
    got = np.array([L.orc_acos(float(v)) for v in x])
    assert ulps(got, np.arccos(x)).max() <= 2.0
    y = rng.uniform(-1, 1, 50000); xx = rng.uniform(-1, 1, 50000)
    got = np.array([L.orc_atan2(float(a), float(b)) for a, b in zip(y, xx)])
    assert ulps(got, np.arctan2(y, xx)).max() <= 2.0
    for a, b in [(0.0, 1.0), (-0.0, 1.0), (0.0, -1.0), (-0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (0.0, 0.0)]:
        assert L.orc_atan2(a, b) == math.atan2(a, b) and math.copysign(1, L.orc_atan2(a, b)) == math.copysign(1, math.atan2(a, b))
    p = rng.uniform(0, 1, 20000)
    got = np.array([L.orc_pow5(float(v)) for v in p])
    assert ulps(got, p ** 5).max() <= 3.0


def test_rng_definition_and_statistics(oracle):
    L = oracle.lib()
    mask = (1 << 64) - 1

    def mix64(z):
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        return z ^ (z >> 31)
    G = 0x9E3779B97F4A7C15
    for seed, pixel, sample, n in [(1, 0, 0, 0), (20231003, 959999, 499, 13), (mask, 2 ** 31, 9999, 1000)]:
        key = mix64(mix64((seed + G) & mask) ^ ((pixel << 32) | sample))
        assert L.orc_rng_key(seed, pixel, sample) == key                 # rt_amd.h "RNG", line by line
        sx, sy = key, mix64((key + G) & mask)                            # RomuDuoJr started from the key
        for _ in range(n + 1):
            x = sx
            sx = (0xD3833E804F4C574B * sy) & mask
            sy = (sy - x) & mask
            sy = ((sy << 27) | (sy >> 37)) & mask
        assert L.orc_rng_draw(key, n) == x
        assert L.orc_kat_random(key, n) == (x >> 11) * 2.0 ** -53
        v12 = np.array([(x >> 12) | 0x3FF0000000000000], dtype=np.uint64).view(np.float64)[0]
        assert L.orc_kat_gen_range(key, n, -1.0, 1.0) == (v12 - 1.0) * 2.0 + -1.0
    # distinct (pixel, sample) -> distinct streams; uniformity of the 53-bit draws
    keys = {L.orc_rng_key(1, p, s) for p in range(200) for s in range(50)}
    assert len(keys) == 200 * 50
    u = np.array([L.orc_kat_random(L.orc_rng_key(7, p, 0), n) for p in range(2000) for n in range(10)])
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    hist = np.histogram(u, bins=20, range=(0, 1))[0]
    assert ((hist - 1000) ** 2 / 1000).sum() < 45.0  # chi-square, 19 dof
    # lag-1 correlation within a stream and across neighbouring pixels
    a = np.array([L.orc_kat_random(L.orc_rng_key(7, p, 0), 0) for p in range(5000)])
    b = np.array([L.orc_kat_random(L.orc_rng_key(7, p + 1, 0), 0) for p in range(5000)])
    c = np.array([L.orc_kat_random(L.orc_rng_key(7, p, 0), 1) for p in range(5000)])
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.05 and abs(np.corrcoef(a, c)[0, 1]) < 0.05
