#!/usr/bin/env python3
"""Renders final_scene at the reference's in-code settings (800x800, depth 40, src/main.rs:508-644) for a few scene
seeds and writes the PNGs and the 12x12 linear block means under gpurun_out/ — the measurement behind the final_scene
row of tests/test_reference_pins.py (how far the published screenshot lies from our renders, against how far our renders
lie from one another)."""
import importlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
rt = importlib.import_module("rust-tracing_amd")
from test_reference_pins import block_means  # noqa: E402


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    seeds = [int(s) for s in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3]
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    result = {"spp": spp, "seeds": {}}
    for seed in seeds:
        hs = rt.HostScene(8, scene_seed=seed, spp=spp, earth_image=str(ROOT / "assets" / "earth-large.jpg"))
        sums = rt.DeviceScene(hs).render(rt.render_params(seed=7))
        lin = np.clip(sums.reshape(hs.height, hs.width, 3) / spp, 0.0, 0.999 ** 2.2)
        result["seeds"][str(seed)] = {"mean_linear": lin.reshape(-1, 3).mean(axis=0).tolist(),
                                      "blocks_linear": np.round(block_means(lin, 12), 6).tolist()}
        try:
            from PIL import Image
            Image.fromarray(np.clip(256 * lin ** (1 / 2.2), 0, 255).astype(np.uint8)).save(out / f"final_scene_seed{seed}.png")
        except ImportError:
            pass
        print("seed", seed, "mean", result["seeds"][str(seed)]["mean_linear"], flush=True)
    (out / "final_scene_pin.json").write_text(json.dumps(result))


if __name__ == "__main__":
    main()
