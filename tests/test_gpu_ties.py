"""GPU parity on hand-made scenes (tests/custom_scenes.py): exact ties, single primitives, empty frames — in both
traversal modes (the library's own trees walked nearest child first / the reference's tree in the reference's order)."""
import numpy as np
import pytest

import custom_scenes
import scene_cases

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


WALKS = {2: "RT_WALK_OWN_TREES", 1: "RT_WALK_AUTO", 0: "RT_WALK_REFERENCE_ORDER"}  # per-scene options (rt_scene_options.walk)


def check(rt, oracle, scene, what):
    params = rt.render_params(seed=3)
    want = oracle.render(scene, params)
    for ordered in (2, 0):  # 2: own trees wherever the scene allows it; 0: reference order
        for leaf, wide in (((1, 0), (2, 0), (4, 0), (8, 0), (1, 1), (3, 1), (8, 1)) if ordered else ((0, 0),)):  # (wide: four children per record)
            ds = rt.DeviceScene(scene, walk=getattr(rt, WALKS[ordered]), leaf_max=leaf, wide=wide)
            assert ds.stats()["ordered"] == (1 if ordered else 0)
            got = ds.render(params)
            bad = np.flatnonzero(bits(got) != bits(want))
            assert bad.size == 0, f"{what}: ordered={ordered} leaf={leaf} wide={wide}: {bad.size} of {want.size} values differ, first at {bad[:4]}"


@pytest.mark.parametrize("order", [0, 1, 2])
def test_exact_ties_resolve_as_the_reference_scan_does(rt, oracle, gpu, order):
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    check(rt, oracle, custom_scenes.tie_scene(cam, order), f"tie scene, order {order}")


def test_single_primitive_and_empty_frame(rt, oracle, gpu):
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    check(rt, oracle, custom_scenes.single_sphere_scene(cam), "single sphere")
    check(rt, oracle, custom_scenes.empty_frame_scene(cam), "empty frame")


@pytest.mark.parametrize("name", list(scene_cases.CASES))
def test_both_traversals_match_the_oracle(rt, oracle, gpu, name):
    """test_gpu_parity checks the default walk; here every scene that has two walks is rendered with each."""
    hs = scene_cases.build(rt, name)
    check(rt, oracle, hs, name)


@pytest.mark.parametrize("n, lds_nodes_expected", [(900, True), (4000, False)])
def test_scenes_larger_than_the_lds(rt, oracle, gpu, n, lds_nodes_expected):
    """The BASELINE scenes all fit the LDS whole.  900 spheres: the records fit, the sphere table does not (it is read
    from global memory); 4000: nothing fits, and the per-lane stacks switch to 4-byte entries."""
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    scene = custom_scenes.many_spheres_scene(cam, n)
    st = rt.DeviceScene(scene).stats()  # (the default choice: a scene this size gets the library's own trees)
    assert st["ordered"] == 1 and st["n_spheres"] == n + 1
    assert (st["lds_nodes"] > 0) == lds_nodes_expected and st["lds_bytes"] < 160 * 1024
    check(rt, oracle, scene, f"{n} spheres")


@pytest.mark.parametrize("order", [0, 1, 2])
def test_media_between_other_objects(rt, oracle, gpu, order):
    """Media bounded by a sphere, by a rotated cube and by a bare list, first / last / next to each other in the scan:
    each draws iff its boundary is crossed before the closest hit found so far in the reference's scan."""
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    scene = custom_scenes.media_scene(cam, order)
    check(rt, oracle, scene, f"media scene, order {order}")


def test_medium_inside_a_frame_keeps_the_reference_walk(rt, oracle, gpu):
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    scene = custom_scenes.media_scene(cam, 0, nested=True)
    assert rt.DeviceScene(scene, walk=rt.RT_WALK_OWN_TREES).stats()["ordered"] == 0
    params = rt.render_params(seed=3)
    want = oracle.render(scene, params)
    got = rt.DeviceScene(scene).render(params)
    assert (bits(got) == bits(want)).all()


def test_too_many_sequence_steps_fall_back_to_the_reference_walk(rt, oracle, gpu):
    """81 steps (> ORDERED_MAX_STEPS): the ordered layout is turned down even when asked for; the reference-order walk
    must see every medium's own boundary sphere."""
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    scene = custom_scenes.many_media_scene(cam, 40)
    params = rt.render_params(seed=3)
    want = oracle.render(scene, params)
    for ordered in (2, 1, 0):
        ds = rt.DeviceScene(scene, walk=getattr(rt, WALKS[ordered]))
        assert ds.stats()["ordered"] == 0
        assert (bits(ds.render(params)) == bits(want)).all(), f"ordered={ordered}"
    check(rt, oracle, custom_scenes.many_media_scene(cam, 20), "41 steps")


def test_frames_inside_frames(rt, oracle, gpu):
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    scene = custom_scenes.nested_frames_scene(cam)
    assert rt.DeviceScene(scene).stats()["max_instance_depth"] >= 3
    check(rt, oracle, scene, "nested frames")


@pytest.mark.parametrize("spheres", [True, False])
@pytest.mark.parametrize("n", [3, 20, 28, 29, 50])  # 28 cubes: 32 instances with the nested ones, 29: 33
def test_many_instances(rt, oracle, gpu, n, spheres):
    """Up to 32 instances of the world frame are noted as bits and walked after the world's own tree (quads + frames kernel:
    spheres=False); more than 32, or any number on the every-feature kernel, are entered where the walk meets them: all against
    the oracle, in both walks."""
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    scene = custom_scenes.many_instances_scene(cam, n, spheres=spheres)
    st = rt.DeviceScene(scene).stats()
    assert st["n_instances"] >= n and (st["n_spheres"] > 0) == spheres
    check(rt, oracle, scene, f"{n} instances, spheres={spheres}")


def test_random_object_graphs(rt, oracle, gpu):
    """Fuzz: 40 random scenes (tests/custom_scenes.py::random_scene), each rendered by whichever walks it supports."""
    lib = rt.amd_lib()
    cam = scene_cases.build(rt, "ragged_cornell_37x37_4spp")
    params = rt.render_params(seed=5)
    ordered_seen = 0
    for seed in range(40):
        scene = custom_scenes.random_scene(cam, seed)
        want = oracle.render(scene, params)
        for ordered in (2, 3, 0):  # (3: own trees with four children per record)
            ds = rt.DeviceScene(scene, walk=getattr(rt, WALKS[min(ordered, 2)]), wide=1 if ordered == 3 else 0)
            ordered_seen += ds.stats()["ordered"] if ordered != 3 else 0
            got = ds.render(params)
            bad = np.flatnonzero(bits(got) != bits(want))
            assert bad.size == 0, f"random scene {seed}, ordered={ds.stats()['ordered']}: {bad.size} of {want.size} values differ"
    assert ordered_seen >= 30


def test_default_choice_of_the_walk(rt, gpu):
    """By default a scene gets the library's own trees unless the reference-order walk measured faster for its kind
    (DESIGN.md "Ordered layout"): a single primitive."""
    want = {"c1_random_balls_400x225_10spp_d10": 1, "two_spheres_80x45_8spp": 1, "earth_80x45_8spp": 0, "two_perlin_spheres_80x45_8spp": 1,
            "quads_64x64_8spp": 1, "simple_light_80x45_16spp": 1, "c3_cornell_box_64x64_16spp_d50": 1, "cornell_smoke_64x64_16spp": 1,
            "c4_final_scene_64x64_8spp_d40": 1}
    for name, ordered in want.items():
        assert rt.DeviceScene(scene_cases.build(rt, name)).stats()["ordered"] == ordered, name


@pytest.mark.parametrize("shortcuts", [(0, 0, 0, 0, 1, 0, 1, 0), (8, 0, 1, 1, 64, 3, 0, 1), (0, 1, 0, 1, 2, 1000, 1, 0), (3, 1, 1, 0, 4, 32, 1, 1), (8, 1, 1, 1, 4, 32, 0, 0),
                                       (8, 1, 1, 0, 4, 32, 1, 0)])
def test_walk_shortcuts_never_change_the_image(rt, oracle, gpu, shortcuts):
    """Flat leaves for small frames, the start shortcut, instances walked last, the sequence look-ahead, the f32 filter in front of a
    flat leaf's quads, the draw of a sphere-bounded medium made ahead for a ray that starts inside it: every combination
    that differs from the default (8, 1, 1, 1, 4, 32, 1, 1 — which every other test runs; set per scene through rt_scene_options), noise-texture hits waiting in the shade stage included, renders the oracle's image bit for bit."""
    lib = rt.amd_lib()
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    scenes = [(name, scene_cases.build(rt, name)) for name in ("c1_random_balls_400x225_10spp_d10", "two_spheres_80x45_8spp", "quads_64x64_8spp",
              "two_perlin_spheres_80x45_8spp", "simple_light_80x45_16spp", "c3_cornell_box_64x64_16spp_d50", "cornell_smoke_64x64_16spp", "c4_final_scene_64x64_8spp_d40")]
    scenes += [("nested frames", custom_scenes.nested_frames_scene(cam)), ("media 0", custom_scenes.media_scene(cam, 0)),
               ("media 1", custom_scenes.media_scene(cam, 1)), ("media 2", custom_scenes.media_scene(cam, 2)), ("ties 2", custom_scenes.tie_scene(cam, 2))]
    params = rt.render_params(seed=3)
    flat_max, start_shortcut, defer_instances, seq_lookahead, slow_min, slow_age, quad_filter, medium_first = shortcuts
    for name, hs in scenes:
        want = oracle.render(hs, params)
        got = rt.DeviceScene(hs, flat_max=flat_max, start_shortcut=start_shortcut, defer_instances=defer_instances,
                             seq_lookahead=seq_lookahead, slow_min=slow_min, slow_age=slow_age, quad_filter=quad_filter,
                             medium_first=medium_first).render(params)
        bad = np.flatnonzero(bits(got) != bits(want))
        assert bad.size == 0, f"{name}, shortcuts {shortcuts}: {bad.size} of {want.size} values differ"


@pytest.mark.parametrize("n_lambertian", [65532, 65533, 66000])
def test_more_materials_than_a_parked_index_can_name(rt, oracle, gpu, n_lambertian):
    """Parked attenuations are 16-bit material indices; a scene with 65 534 materials still parks indices, one with 65 535 (the table's
    Color::ONE entry would get the index that marks a parked colour) or 66 002 (most of them unused, the used ones spread over the whole
    range) is rendered by the kernels that park colours instead — same image as the oracle's, in every walk."""
    import random
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    rnd = random.Random(3)
    s = custom_scenes.CustomScene(cam, spp=4, depth=12, background=(0.7, 0.8, 1.0))
    mats = [s.lambertian(rnd.random(), rnd.random(), rnd.random()) for _ in range(n_lambertian)]
    mats += [s.metal(0.9, 0.8, 0.7, 0.1), s.dielectric(1.5)]
    n = len(mats)
    pick = [mats[0], mats[1], mats[65530], mats[65531], mats[n // 2], mats[n - 4], mats[n - 3], mats[n - 2], mats[n - 1]]
    items = [s.sphere((0.0, -1003.0, 0.0), 1000.0, mats[40000])]
    for k in range(40):
        items.append(s.sphere((rnd.uniform(-4, 4), rnd.uniform(-2.5, 1.5), rnd.uniform(-4, 2)), rnd.uniform(0.3, 0.8), pick[k % len(pick)]))
    items.append(s.quad((-5.0, -3.0, -5.0), (10.0, 0.0, 0.0), (0.0, 6.0, 0.0), mats[n - 5]))
    scene = s.finish(s.list(items))
    params = rt.render_params(seed=3)
    want = oracle.render(scene, params)
    for opts in (dict(walk=rt.RT_WALK_REFERENCE_ORDER), dict(walk=rt.RT_WALK_OWN_TREES, wide=0), dict(walk=rt.RT_WALK_OWN_TREES, wide=1)):
        got = rt.DeviceScene(scene, **opts).render(params)
        assert (bits(got) == bits(want)).all(), opts


@pytest.mark.parametrize("n,level", [(300, 3), (1200, 1), (6000, 0)])
def test_scenes_at_every_lds_level_in_both_record_forms(rt, oracle, gpu, n, level):
    """n spheres as one flat list: everything in the LDS (300), only the records (1200), nothing (6000) — each walked through records of
    two and of four children, and in the reference's order."""
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    scene = custom_scenes.many_spheres_scene(cam, n)
    params = rt.render_params(seed=3)
    want = oracle.render(scene, params)
    for opts in (dict(walk=rt.RT_WALK_OWN_TREES, wide=0), dict(walk=rt.RT_WALK_OWN_TREES, wide=1), dict(walk=rt.RT_WALK_REFERENCE_ORDER)):
        ds = rt.DeviceScene(scene, **opts)
        if opts.get("wide") == 1:
            assert rt.debug_last_launch is not None and (ds.stats()["lds_nodes"] > 0) == (level > 0), ds.stats()
        got = ds.render(params)
        assert (bits(got) == bits(want)).all(), (n, opts)
        if opts.get("wide") == 1:
            assert rt.debug_last_launch()["lds_level"] == level, rt.debug_last_launch()
