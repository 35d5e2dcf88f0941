#!/usr/bin/env python3
"""A BASELINE configuration rendered IN FULL on the GPU and by the CPU oracle, compared bit for bit (u64 view of every f64 sum).
The test suite compares at sizes the oracle finishes in seconds and checks the full sizes through properties; this tool spends
the minutes the direct comparison costs (C2 in full: 480 M samples, about three minutes of 16 host cores in the
reference-faithful mode).  Usage: python tools/full_config_parity.py c2 [--spp N] [--mode reference|tight]"""
import argparse, importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
rt = importlib.import_module("rust-tracing_amd")
import bench, oracle_lib

ap = argparse.ArgumentParser()
ap.add_argument("workload", choices=sorted(bench.WORKLOADS))
ap.add_argument("--spp", type=int, default=0, help="the first N samples per pixel only (default: the configuration's own count)")
ap.add_argument("--mode", choices=("reference", "tight"), default="reference",
                help="the oracle's box test: the reference's own, or the narrowing one (same image, 3-4x faster)")
args = ap.parse_args()
wl = bench.WORKLOADS[args.workload]
spp = min(args.spp, wl["spp"]) if args.spp > 0 else wl["spp"]
hs = rt.HostScene(wl["scene"], scene_seed=bench.SCENE_SEED, width=wl["width"], aspect=wl["aspect"], spp=wl["spp"], depth=wl["depth"],
                  earth_image=wl.get("earth_image"))
params = rt.render_params(seed=bench.RENDER_SEED, sample_end=spp)
t = time.perf_counter()
got = rt.DeviceScene(hs).render(params)
t_gpu = time.perf_counter() - t
threads = oracle_lib.default_threads()
t = time.perf_counter()
want = oracle_lib.render(hs, params, threads=threads,
                         aabb_mode=oracle_lib.ORC_AABB_TIGHT if args.mode == "tight" else oracle_lib.ORC_AABB_REFERENCE)
t_cpu = time.perf_counter() - t
bad = int((got.view(np.uint64) != want.view(np.uint64)).sum())
samples = hs.width * hs.height * spp
print(f"{args.workload} {wl['name']}: {hs.width}x{hs.height}, {spp} of {wl['spp']} spp = {samples / 1e6:.1f} M samples; "
      f"GPU {t_gpu:.2f} s (scene upload and read-back included), oracle ({args.mode} box test, {threads} threads) {t_cpu:.1f} s; "
      f"{got.size} f64 sums compared, {bad} differ")
sys.exit(1 if bad else 0)
