#!/usr/bin/env python3
"""Generates tests/golden/reference_edge_pins.npz: where the luminance edges of two more reference screenshots lie.

    screenshots/checker.png   scene 1, two_spheres   (src/main.rs:140-173): CheckerTexture 0.32 on two radius-10 spheres
    screenshots/earth.png     scene 2, earth         (src/main.rs:175-203): ImageTexture on a radius-2 sphere

Both were rendered by an OLDER revision of the reference than the committed one — a gradient sky where the committed code has a
constant one (src/renderer.rs:152-153), and texels without the committed powf(2.2) (src/color.rs:21-26) — so their SHADING pins
nothing.  Their GEOMETRY does: the cell boundaries of the checker (scale, the i32 parity of the three floors, src/texture.rs:60-69),
the silhouettes of the spheres, the continents' outlines (sphere UV mapping and the 1 - v flip, src/sphere.rs:48-52,
src/texture.rs:83-92) and the 16:9 / vfov 20 camera at (13,2,3) resp. (12,0,0) are the same in both revisions, and an edge is where
it is whatever the colours on its two sides.  So only edge POSITIONS are kept, and kept THIN: per direction and sign, the pixels where
the central difference of the 3x3-filtered luminance exceeds HI = 45 (strong) / LO = 25 (weak) levels AND is a local maximum along its
own axis (the ridge of the edge, one or two pixels wide — the Cornell pins' thresholded bands are three to four wide, which on a
checker whose cells are a few pixels apart would let almost anything pass).  A strong ridge pixel of one image must lie within one pixel
of a weak ridge pixel of the same direction and sign of the other, both ways (tests/test_reference_pins.py).  Measured with the CPU
oracle at 24 spp: checker 36 278 / 34 819 strong ridge pixels, 100 % found both ways; a checker scale of 0.325 instead of 0.32 leaves
23 %, a frame shifted by two pixels 87 %, by one pixel 99.85 % (the tolerance); earth 2 815 / 4 055, 99.8 % both ways, 72 % after a
two-pixel shift.  (The old sky and the old texel curve change an edge's contrast, not its place: hence the low thresholds.)

Data derived from the reference's published images, not reference source.  Run in the build container (needs /root/reference)."""
import sys
from pathlib import Path

import numpy as np
from PIL import Image

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_reference_pixel_pins import gradients  # noqa: E402

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent / "reference_edge_pins.npz"
HI, LO = 45.0, 25.0


def ridges(srgb, threshold):
    """per direction and sign (+x, -x, +y, -y): gradient beyond the threshold and a local maximum of its magnitude along its axis"""
    gx, gy = gradients(srgb)
    ax, ay = np.abs(gx), np.abs(gy)
    rx, ry = np.zeros_like(ax, dtype=bool), np.zeros_like(ay, dtype=bool)
    rx[:, 1:-1] = (ax[:, 1:-1] >= ax[:, :-2]) & (ax[:, 1:-1] >= ax[:, 2:])
    ry[1:-1, :] = (ay[1:-1, :] >= ay[:-2, :]) & (ay[1:-1, :] >= ay[2:, :])
    return np.stack([(gx > threshold) & rx, (gx < -threshold) & rx, (gy > threshold) & ry, (gy < -threshold) & ry])


def main():
    out = {"hi": HI, "lo": LO}
    for name, scene in (("checker", 1), ("earth", 2)):
        img = np.asarray(Image.open(REF / "screenshots" / f"{name}.png").convert("RGB"), dtype=np.float64)
        assert img.shape == (675, 1200, 3)
        out[f"{name}_scene"] = scene
        out[f"{name}_strong"] = np.packbits(ridges(img, HI))
        out[f"{name}_weak"] = np.packbits(ridges(img, LO))
        print(name, "strong ridge pixels", int(ridges(img, HI).sum()), "weak", int(ridges(img, LO).sum()))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
