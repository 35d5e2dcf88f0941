#!/usr/bin/env python3
"""Generates tests/golden/reference_screenshot_stats.json from the reference's published screenshots.

The reference (Husenap/rust-tracing) has no tests or golden vectors; the only outputs it publishes are the
PNGs under screenshots/ (README.md:20-37).  Two of them show scenes with no build-time randomness, rendered by the committed code at the committed
camera settings, so a converged render of the same scene must reproduce them up to Monte-Carlo noise:
    cornell_box.png    600x600    src/main.rs:344-421
    cornell_smoke.png  600x600    src/main.rs:423-506
A third shows a scene whose only build-time randomness is the Perlin tables of its marble texture, on a black
background (no dependence on the sky), at the committed settings:
    simple_light.png   600x337    src/main.rs:296-342
Different tables move the marble's veins, not its average: the image mean is reproduced to a fraction of a per cent
whatever the tables, and the means over a 3x3 grid to a few per cent — which pins Sphere::hit, the sphere and quad lights, the
noise texture and the 16:9 camera, none of which the Cornell scenes contain.
A fourth shows final_scene, whose build-time randomness is confined to parts of the frame: the 400 box heights of the ground (the lower
third), the 1000 small spheres of the white cube and the Perlin tables of the marble sphere.  Everything else in it — the quad light, the fog
filling the room, the motion-blurred sphere, the earth (assets/earth-large.jpg, which the repo holds), the fuzzy metal sphere — is fixed by the
committed code, and those blocks of the frame must come out the same whatever the scene seed:
    final_scene.png    800x800    src/main.rs:508-639
It is the only screenshot that exercises Metal, Dielectric, the moving sphere, both ConstantMediums, Translate/RotateY and ImageTexture.
(earth.png, perlin.png and random_balls.png were rendered by an older revision with a gradient sky, like checker.png
below: none of them is used.)
(checker.png shows the deterministic two_spheres scene too, but it was rendered by an older revision with a
gradient sky: its sky pixels are (229,240,255), whereas the committed constant background (0.7,0.8,1.0),
src/main.rs:163, encodes to (217,231,255).  Geometry agrees, colours cannot, so it is not used as a pin.)
This script reduces each to a GRID x GRID table of block means — both of the sRGB bytes and of the
linearised values (((byte + 0.5)/256)^2.2: the centre of the interval of linear values that
color_to_rgb, src/color.rs:12-19, maps to that byte — `(256 * x^(1/2.2)) as u8` truncates) — which is the
fixture the tests compare against.  It is DATA derived from the reference's images, not reference source.

Run in the build container (needs /root/reference); the GPU box only sees the committed JSON.
"""
import json
import sys
from pathlib import Path

import numpy as np
from PIL import Image

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent / "reference_screenshot_stats.json"
GRID = 12

SHOTS = {
    "cornell_box": {"file": "screenshots/cornell_box.png", "scene": 6, "source": "src/main.rs:344-421"},
    "cornell_smoke": {"file": "screenshots/cornell_smoke.png", "scene": 7, "source": "src/main.rs:423-506"},
    "simple_light": {"file": "screenshots/simple_light.png", "scene": 5, "source": "src/main.rs:296-342"},
    "final_scene": {"file": "screenshots/final_scene.png", "scene": 8, "source": "src/main.rs:508-639"},
}


def block_means(a, grid):
    h, w, c = a.shape
    ys = [round(k * h / grid) for k in range(grid + 1)]
    xs = [round(k * w / grid) for k in range(grid + 1)]
    out = np.zeros((grid, grid, c))
    for r in range(grid):
        for q in range(grid):
            out[r, q] = a[ys[r]:ys[r + 1], xs[q]:xs[q + 1]].reshape(-1, c).mean(axis=0)
    return out


def main():
    result = {"grid": GRID, "note": "block means of the reference's screenshots; rows top to bottom", "shots": {}}
    for name, info in SHOTS.items():
        img = np.asarray(Image.open(REF / info["file"]).convert("RGB"), dtype=np.float64)
        lin = ((img + 0.5) / 256.0) ** 2.2
        result["shots"][name] = {
            "file": info["file"], "scene": info["scene"], "source": info["source"],
            "width": int(img.shape[1]), "height": int(img.shape[0]),
            "mean_srgb": [round(float(x), 4) for x in img.reshape(-1, 3).mean(axis=0)],
            "mean_linear": [round(float(x), 6) for x in lin.reshape(-1, 3).mean(axis=0)],
            "blocks_srgb": np.round(block_means(img, GRID), 3).tolist(),
            "blocks_linear": np.round(block_means(lin, GRID), 6).tolist(),
        }
    OUT.write_text(json.dumps(result, indent=1) + "\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
