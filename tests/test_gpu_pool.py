"""The pool kernel (rt_pool_kernel.hip: ray compaction across stages — traversal lanes swap finished queries for fresh rays
through a slot pool in the LDS, service waves shade / end whole words at full width) renders the frames path_kernel renders,
bit for bit: every scene that lives in the LDS whole, spheres / quads + frames / every-feature instantiations, media,
textures, ragged sizes, chained sample ranges, tile shards."""
import numpy as np
import pytest

import custom_scenes
import scene_cases

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("name", [n for n in scene_cases.CASES if "final_scene" not in n])
def test_pool_kernel_matches_the_oracle(rt, oracle, gpu, name):
    hs = scene_cases.build(rt, name)
    params = rt.render_params(seed=11)
    want = oracle.render(hs, params)
    ds = rt.DeviceScene(hs, walk=rt.RT_WALK_OWN_TREES, pool=1)
    assert ds.stats()["ordered"] == 1 and ds.stats()["lds_nodes"] > 0
    got = ds.render(params)
    assert rt.debug_last_launch()["pool_slots"] >= 256, "the pool kernel did not run (silent fall-back to path_kernel)"
    assert (bits(got) == bits(want)).all()


def test_pool_kernel_on_hand_made_scenes(rt, oracle, gpu):
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    params = rt.render_params(seed=3)
    for scene in (custom_scenes.tie_scene(cam, 1), custom_scenes.media_scene(cam, 2), custom_scenes.nested_frames_scene(cam),
                  custom_scenes.single_sphere_scene(cam), custom_scenes.empty_frame_scene(cam), custom_scenes.many_spheres_scene(cam, 300)):
        want = oracle.render(scene, params)
        got = rt.DeviceScene(scene, walk=rt.RT_WALK_OWN_TREES, pool=1).render(params)
        assert rt.debug_last_launch()["pool_slots"] >= 256
        assert (bits(got) == bits(want)).all()


def test_pool_kernel_shards_and_sample_ranges(rt, gpu):
    """Chained sample ranges over several launches (a 1 MiB sample buffer) and three tile shards give the frame one launch gives."""
    hs = scene_cases.build(rt, "c2_random_balls_96x64_8spp_d50")
    whole = rt.DeviceScene(hs, pool=0).render(rt.render_params(seed=2))
    ds = rt.DeviceScene(hs, pool=1, sample_buffer_bytes=1 << 20)
    part = ds.render(rt.render_params(seed=2, sample_end=3))
    import ctypes as C
    p2 = rt.render_params(seed=2, sample_begin=3, accumulate=True)
    assert rt.amd_lib().rt_render(ds._handle, C.byref(hs.camera), C.byref(p2), part.ctypes.data_as(C.POINTER(C.c_double))) == 0
    assert (bits(part) == bits(whole)).all()
    frame = np.zeros_like(whole)
    for r in range(3):
        tiles = ds.render(rt.render_params(seed=2, shard_index=r, shard_count=3))
        frame += tiles  # (RT_OUT_FRAME: each shard writes its own pixels, zeros elsewhere)
    assert (bits(frame) == bits(whole)).all()


def test_pool_kernel_full_frames_equal_path_kernel(rt, gpu):
    """BASELINE configs[1] and [2] at their full image sizes (24 / 64 of their spp): 23 M and 23 M paths through the pool —
    every workgroup's pool fills and drains thousands of times — give path_kernel's frame, value for value."""
    import hashlib
    import torch

    def frame(hs, pool):
        ds = rt.DeviceScene(hs, pool=pool)
        out = torch.zeros(hs.width * hs.height * 3, dtype=torch.float64, device="cuda")
        ds.render_device(rt.render_params(seed=1), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()

    for scene, width, aspect, spp in ((0, 1200, 1.5, 24), (6, 600, 1.0, 64)):
        hs = rt.HostScene(scene, scene_seed=1, width=width, aspect=aspect, spp=spp, depth=50)
        assert frame(hs, 1) == frame(hs, 0)
