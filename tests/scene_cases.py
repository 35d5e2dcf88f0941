"""Scene/camera configurations shared by the golden, parity and bench code.

Sizes are chosen so that the CPU oracle finishes each case in seconds."""

from pathlib import Path

EARTH_SMALL = "synthetic:256x128"
ASSETS = Path(__file__).resolve().parent.parent / "assets"  # the reference's own image-texture inputs (assets/README.md)

# name -> dict(scene=, width=, aspect=, spp=, depth=, earth_image=)
CASES = {
    # BASELINE.json configs[0]: the reference's own CPU-runnable case
    "c1_random_balls_400x225_10spp_d10": dict(scene=0, width=400, aspect=16.0 / 9.0, spp=10, depth=10),
    # the bench workload's scene (configs[1]) at a size the oracle can check
    "c2_random_balls_96x64_8spp_d50": dict(scene=0, width=96, aspect=1.5, spp=8, depth=50),
    "two_spheres_80x45_8spp": dict(scene=1, width=80, spp=8, depth=8),
    "earth_80x45_8spp": dict(scene=2, width=80, spp=8, depth=8, earth_image=EARTH_SMALL),
    "two_perlin_spheres_80x45_8spp": dict(scene=3, width=80, spp=8, depth=8),
    "quads_64x64_8spp": dict(scene=4, width=64, spp=8, depth=8),
    "simple_light_80x45_16spp": dict(scene=5, width=80, spp=16, depth=8),
    "c3_cornell_box_64x64_16spp_d50": dict(scene=6, width=64, spp=16, depth=50),
    "cornell_smoke_64x64_16spp": dict(scene=7, width=64, spp=16, depth=8),
    "c4_final_scene_64x64_8spp_d40": dict(scene=8, width=64, spp=8, depth=40, earth_image=EARTH_SMALL),
    # BASELINE.json configs[4] (the 8-GPU case: final_scene at depth 50) at a size the oracle can check
    "c5_final_scene_64x64_8spp_d50": dict(scene=8, width=64, spp=8, depth=50, earth_image=EARTH_SMALL),
    # image textures: a size that is no multiple of the device's 8x8 texel tiles; the reference's own JPEG asset
    "earth_ragged_image_80x45_8spp": dict(scene=2, width=80, spp=8, depth=8, earth_image="synthetic:250x123"),
    "earth_small_jpg_80x45_8spp": dict(scene=2, width=80, spp=8, depth=8, earth_image=str(ASSETS / "earth-small.jpg")),
    # ragged: neither dimension a multiple of the 8x8 tile
    "ragged_cornell_37x37_4spp": dict(scene=6, width=37, spp=4, depth=8),
    "ragged_random_balls_53x29_4spp": dict(scene=0, width=53, spp=4, depth=10),
}


def build(rt, name, **overrides):
    kw = dict(CASES[name])
    kw.update(overrides)
    return rt.HostScene(kw.pop("scene"), **kw)
