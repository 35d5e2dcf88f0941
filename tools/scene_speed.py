#!/usr/bin/env python3
"""Msamples/s of any of the nine scenes at its in-code camera, both walks.  Usage: python tools/scene_speed.py [scene ...] [--spp N]"""
import argparse, importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
rt = importlib.import_module("rust-tracing_amd")
ap = argparse.ArgumentParser()
ap.add_argument("scenes", nargs="*", type=int, default=list(range(9)))
ap.add_argument("--spp", type=int, default=64)
args = ap.parse_args()
names = ["random_balls", "two_spheres", "earth", "two_perlin_spheres", "quads", "simple_light", "cornell_box", "cornell_smoke", "final_scene"]
for scene in args.scenes:
    hs = rt.HostScene(scene, spp=args.spp, earth_image="synthetic:1024x512")
    out = []
    for ordered in (2, 0):
        rt.amd_lib().rt_debug_set_traversal(ordered, 0)
        ds = rt.DeviceScene(hs)
        frame = torch.zeros(hs.width * hs.height * 3, dtype=torch.float64, device="cuda")
        s = torch.cuda.current_stream(); p = rt.render_params(seed=1)
        ds.render_device(p, frame.data_ptr(), s.cuda_stream); torch.cuda.synchronize()
        t = time.perf_counter(); ds.render_device(p, frame.data_ptr(), s.cuda_stream); torch.cuda.synchronize(); dt = time.perf_counter() - t
        out.append(f"{'own trees' if ds.stats()['ordered'] else 'reference order'}: {hs.width * hs.height * args.spp / dt / 1e6:7.0f} Msamples/s")
    print(f"{scene} {names[scene]:20s} {hs.width}x{hs.height}x{args.spp} depth {hs.camera.max_depth}:  " + "   ".join(out))
rt.amd_lib().rt_debug_set_traversal(1, 0)
