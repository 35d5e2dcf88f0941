#!/usr/bin/env python3
"""Generates tests/golden/reference_pixel_pins.npz: pixel-level pins from the two reference screenshots whose scenes have no
build-time randomness and which the committed code renders at the committed settings (600x600, depth 8):
    screenshots/cornell_box.png    src/main.rs:344-421
    screenshots/cornell_smoke.png  src/main.rs:423-506

Two reductions of each PNG (data derived from the reference's published images, not reference source):

  box7     the sRGB bytes after a 7x7 box filter, in quarter levels (uint16), on every second pixel of every second row.  Why 7x7 and not 3x3: the screenshots are
           Monte-Carlo renders themselves — the residual of cornell_box.png against its own 5x5 mean in a flat stretch of the back wall
           is 5.7 levels rms per pixel (a 1024 spp render of ours shows 11.6 there, so the screenshot holds about 4096 spp, the in-code
           setting) — and after a 3x3 filter 1.9 levels of that remain: no render, however converged, agrees with it to +-3 levels on 99 %
           of the pixels.  After 7x7, 0.8 remain, and an edge of contrast C displaced by one pixel still moves the filtered value by C / 7.
  edges    where the luminance (Rec. 709 weights) of the 3x3-filtered image changes by more than HI = 60 levels over two pixels, per
           direction and sign (the silhouettes of the boxes, the rectangle of the light, the room's corners) — and by more than LO = 35:
           a strong edge of one image must lie within one pixel of an edge of the same direction and sign of the other, both ways.

Run in the build container (needs /root/reference); the GPU box only sees the committed .npz.
"""
import sys
from pathlib import Path

import numpy as np
from PIL import Image
from scipy.ndimage import uniform_filter

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent / "reference_pixel_pins.npz"
HI, LO = 60.0, 35.0
LUM = np.array([0.2126, 0.7152, 0.0722])


def box(a, k):
    return uniform_filter(a, size=(k, k, 1), mode="nearest")


def gradients(srgb):
    """central differences of the 3x3-filtered luminance along x and y (zero on the frame's border)"""
    lum = box(srgb, 3) @ LUM
    gx, gy = np.zeros_like(lum), np.zeros_like(lum)
    gx[:, 1:-1] = lum[:, 2:] - lum[:, :-2]
    gy[1:-1, :] = lum[2:, :] - lum[:-2, :]
    return gx, gy


def edge_masks(srgb, threshold):
    gx, gy = gradients(srgb)
    return np.stack([gx > threshold, gx < -threshold, gy > threshold, gy < -threshold])


def main():
    out = {"hi": HI, "lo": LO}
    for name, scene in (("cornell_box", 6), ("cornell_smoke", 7)):
        img = np.asarray(Image.open(REF / "screenshots" / f"{name}.png").convert("RGB"), dtype=np.float64)
        assert img.shape == (600, 600, 3)
        out[f"{name}_scene"] = scene
        out[f"{name}_box7"] = np.round(box(img, 7)[::2, ::2] * 4.0).astype(np.uint16)
        out[f"{name}_strong"] = np.packbits(edge_masks(img, HI))
        out[f"{name}_weak"] = np.packbits(edge_masks(img, LO))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
