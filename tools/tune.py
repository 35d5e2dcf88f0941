#!/usr/bin/env python3
"""Scheduler-knob sweep for the render kernel (speed only; results never change).  Interleaved rounds in one
process, median of the kernel times.  Usage: python tools/tune.py [--spp N] [--workload c2] [--bvh reference]"""
import argparse, importlib, itertools, json, statistics, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
rt = importlib.import_module("rust-tracing_amd")
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--workload", default="c2")
ap.add_argument("--bvh", default="reference")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--grid", default="prim=8,16;other=16;shade=32,48;box=32,40,48;lds=0,1")
args = ap.parse_args()
wl = dict(bench.WORKLOADS[args.workload]); wl.pop("name")
hs = rt.HostScene(wl["scene"], scene_seed=1, width=wl["width"], aspect=wl["aspect"], spp=args.spp, depth=wl["depth"],
                  earth_image=wl.get("earth_image"), bvh=args.bvh)
ds = rt.DeviceScene(hs)
frame = torch.zeros(hs.width * hs.height * 3, dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream()
axes = {}
for part in args.grid.split(";"):
    k, v = part.split("="); axes[k] = [int(x) for x in v.split(",")]
combos = list(itertools.product(axes["prim"], axes["other"], axes["shade"], axes["box"], axes.get("lds", [-1]), axes.get("new", [-1])))
times = {c: [] for c in combos}
lib = rt.amd_lib()
params = rt.render_params(seed=1)
ds.render_device(params, frame.data_ptr(), stream.cuda_stream); torch.cuda.synchronize()
for r in range(args.rounds):
    for c in combos:
        lib.rt_debug_set_tuning(*c)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream); ds.render_device(params, frame.data_ptr(), stream.cuda_stream); e1.record(stream)
        torch.cuda.synchronize()
        times[c].append(e0.elapsed_time(e1))
msamples = hs.width * hs.height * args.spp / 1e6
rows = sorted(((statistics.median(v), c) for c, v in times.items()))
for t, c in rows[:12]:
    print(f"prim={c[0]:3d} other={c[1]:3d} shade={c[2]:3d} box={c[3]:2d} lds={c[4]:2d} new={c[5]:3d}  {t:8.3f} ms  {msamples / t * 1e3:8.1f} Msamples/s")
print("worst:", rows[-1])
