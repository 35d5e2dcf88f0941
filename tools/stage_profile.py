#!/usr/bin/env python3
"""Where do the render kernel's waves spend their cycles?  Runs the instrumented kernel once and prints, per
scheduler stage, rounds, mean active lanes per round and the share of wave cycles."""
import argparse, importlib, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
rt = importlib.import_module("rust-tracing_amd")
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--workload", default="c2")
ap.add_argument("--bvh", default="reference")
args = ap.parse_args()
wl = dict(bench.WORKLOADS[args.workload]); wl.pop("name")
hs = rt.HostScene(wl["scene"], scene_seed=1, width=wl["width"], aspect=wl["aspect"], spp=args.spp, depth=wl["depth"],
                  earth_image=wl.get("earth_image"), bvh=args.bvh)
ds = rt.DeviceScene(hs)
print("scene stats", ds.stats())
frame = torch.zeros(hs.width * hs.height * 3, dtype=torch.float64, device="cuda")
cnt = ds.render_device_counted(rt.render_params(seed=1), frame.data_ptr(), torch.cuda.current_stream().cuda_stream)
prof = rt.debug_stage_profile()
total = sum(v["cycles"] for v in prof.values())
n = cnt["samples"]
print({k: round(v / n, 3) for k, v in cnt.items()})
for k, v in prof.items():
    if "." in k:
        print(f"  {k:15s} cycles {100 * v['cycles'] / total:5.1f} %")
        continue
    print(f"{k:7s} rounds per 64 samples {v['rounds'] * 64 / n:9.2f}  mean active lanes {v['mean_active_lanes']:5.1f}  cycles {100 * v['cycles'] / total:5.1f} %"
          f"  cycles/round {v['cycles'] / max(1, v['rounds']):8.1f}")
