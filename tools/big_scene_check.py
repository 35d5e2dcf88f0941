#!/usr/bin/env python3
"""Sanity / speed check of the ordered walk on scenes far larger than the LDS: n random spheres as one flat list."""
import importlib, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
rt = importlib.import_module("rust-tracing_amd")
import custom_scenes, scene_cases
cam_src = rt.HostScene(0, width=512, aspect=1.0, spp=16, depth=8)
for n in (int(a) for a in (sys.argv[1:] or ["20000", "100000"])):
    sc = custom_scenes.many_spheres_scene(cam_src, n)
    sc.camera.samples_per_pixel = 16
    t = time.time(); ds = rt.DeviceScene(sc); t_create = time.time() - t
    st = ds.stats()
    frame = torch.zeros(sc.width * sc.height * 3, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream(); p = rt.render_params(seed=1)
    ds.render_device(p, frame.data_ptr(), s.cuda_stream); torch.cuda.synchronize()
    t = time.time(); ds.render_device(p, frame.data_ptr(), s.cuda_stream); torch.cuda.synchronize(); dt = time.time() - t
    cnt = ds.render_device_counted(rt.render_params(seed=1, sample_end=2), frame.data_ptr(), s.cuda_stream)
    per = {k: round(v / cnt["samples"], 2) for k, v in cnt.items() if k in ("rays", "node_visits", "sphere_tests")}
    print(f"{n} spheres: create {t_create:.2f} s, ordered={st['ordered']} lds_nodes={st['lds_nodes']} stack={st['stack_entries']}; "
          f"{sc.width}x{sc.height}x16: {dt * 1e3:.1f} ms = {sc.width * sc.height * 16 / dt / 1e6:.0f} Msamples/s; per sample {per}; "
          f"finite={bool(torch.isfinite(frame).all())}")
