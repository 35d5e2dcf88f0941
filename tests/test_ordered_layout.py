"""The ordered layout (rust-tracing_amd/csrc/rt_ordered.hpp): the scene compiler's own trees for scenes whose
closest-hit queries do not depend on the visiting order.  CPU checks of the structure (no device needed); the walk
itself is checked bit for bit against the oracle in test_gpu_parity.py / test_gpu_ties.py."""
import numpy as np
import pytest

import custom_scenes
import scene_cases

KIND_INNER, KIND_SPHERES, KIND_QUADS, KIND_INSTANCE, KIND_EMPTY = 0, 1, 2, 3, 7


def f32_box(words):
    return words.view(np.float32).astype(np.float64).reshape(3, 2)  # rows x, y, z; columns lo, hi


def prim_bound(lay, kind, index):
    if kind == KIND_SPHERES:
        c, r, vec, moving = lay["spheres"][index, 0:3], abs(lay["spheres"][index, 3]), lay["spheres"][index, 4:7], lay["spheres"][index, 8]
        ends = [c] + ([c + vec] if moving else [])
        return np.min([e - r for e in ends], axis=0), np.max([e + r for e in ends], axis=0)
    q, u, v = lay["quads"][index, 0:3], lay["quads"][index, 3:6], lay["quads"][index, 6:9]
    corners = np.array([q, q + u, q + v, q + u + v])
    return corners.min(axis=0), corners.max(axis=0)


def to_parent(inst, lo, hi):
    """A frame's box seen from the enclosing frame (RotateY's corner loop, then Translate's shift)."""
    off, s, c, flags = inst[0:3], inst[3], inst[4], int(inst[6])
    pts = []
    for x in (lo[0], hi[0]):
        for y in (lo[1], hi[1]):
            for z in (lo[2], hi[2]):
                p = np.array([x, y, z])
                if flags & 2:
                    p = np.array([c * p[0] + s * p[2], p[1], -s * p[0] + c * p[2]])
                if flags & 1:
                    p = p + off
                pts.append(p)
    pts = np.array(pts)
    return pts.min(axis=0), pts.max(axis=0)


def walk(lay, node, seen, depth_left):
    """Bound of everything under record `node`; checks each child's box against it.  Returns (lo, hi, stack need)."""
    assert depth_left > 0, "tree deeper than the advertised stack"
    rec = lay["nodes"][node]
    lo_all, hi_all, need = np.full(3, np.inf), np.full(3, -np.inf), 0
    for slot in range(2):
        ref = int(rec[12 + slot])
        kind, count, index = ref >> 29, ((ref >> 26) & 7) + 1, ref & ((1 << 26) - 1)
        if kind == KIND_EMPTY:
            continue
        box = f32_box(rec[6 * slot:6 * slot + 6])
        if kind == KIND_INNER:
            lo, hi, sub = walk(lay, index, seen, depth_left - 1)
        elif kind == KIND_INSTANCE:
            inst = lay["instances"][index]
            ilo, ihi, sub = walk(lay, int(inst[7]), seen, depth_left - 2)
            lo, hi = to_parent(inst, ilo, ihi)
            sub += 1  # the frame-exit marker
        else:
            assert kind in (KIND_SPHERES, KIND_QUADS)
            bounds = [prim_bound(lay, kind, index + i) for i in range(count)]
            for i in range(count):
                key = (kind, index + i)
                assert key not in seen, "a primitive sits in two leaves"
                seen.add(key)
            seqs = lay["spheres"][index:index + count, 7] if kind == KIND_SPHERES else lay["quads"][index:index + count, 9]
            assert (np.diff(seqs) > 0).all(), "a leaf's primitives are kept in scan order"
            lo, hi, sub = np.min([b[0] for b in bounds], axis=0), np.max([b[1] for b in bounds], axis=0), 0
        assert (box[:, 0] <= lo).all() and (box[:, 1] >= hi).all(), (node, slot, box, lo, hi)
        # and not absurdly loose: within the quad padding / f32 rounding of the content
        slack = 1e-4 + 1e-6 * np.maximum(np.abs(lo), np.abs(hi))
        assert (box[:, 0] >= lo - slack).all() and (box[:, 1] <= hi + slack).all(), (node, slot, box, lo, hi)
        lo_all, hi_all, need = np.minimum(lo_all, lo), np.maximum(hi_all, hi), max(need, sub)
    return lo_all, hi_all, need + 1


def walk_steps(lay, seen):
    """Every step of the world frame's sequence: its tree(s) hold their primitives inside the step's box.  Returns the
    deepest stack need."""
    need = 0
    for step in lay["steps"]:
        kind, a, b = int(step[0]), int(step[1]), int(step[2])
        box = f32_box(step[4:10])
        if kind == 1:  # a medium bounded by one sphere: that sphere is in no leaf
            index = int(lay["media"][a])
            seen.add((KIND_SPHERES, index))
            lo, hi = prim_bound(lay, KIND_SPHERES, index)
            sub = 0
        else:
            lo, hi, sub = walk(lay, a if kind == 0 else b, seen, 64)
        assert (box[:, 0] <= lo).all() and (box[:, 1] >= hi).all(), (kind, box, lo, hi)
        need = max(need, sub)
    return need


@pytest.mark.parametrize("name", list(scene_cases.CASES))
def test_tree_holds_every_primitive_once_inside_its_boxes(rt, name):
    hs = scene_cases.build(rt, name)
    lay = rt.debug_ordered_layout(hs)
    assert lay["ordered"]
    seen = set()
    need = walk_steps(lay, seen)
    assert len(seen) == len(lay["spheres"]) + len(lay["quads"])
    assert need <= lay["stack_entries"] <= 32
    assert int(lay["steps"][0][0]) != 0 or lay["root"] == int(lay["steps"][0][1])
    # seq is a numbering of all primitives in the reference's scan order: a permutation of 0..n-1
    seqs = np.concatenate([lay["spheres"][:, 7], lay["quads"][:, 9]])
    assert sorted(seqs.astype(int)) == list(range(len(seqs)))


def test_scenes_with_media_become_sequences_in_scan_order(rt):
    """cornell_smoke: walls, then the two smoke boxes (each with a boundary tree), then the wall added after them;
    final_scene: whatever order the top-level BVH (built with scene_seed 1) scans its eleven leaves in."""
    lay = rt.debug_ordered_layout(scene_cases.build(rt, "cornell_smoke_64x64_16spp"))
    assert lay["ordered"] and [int(k) for k in lay["steps"][:, 0]] == [0, 2, 2, 0]
    lay = rt.debug_ordered_layout(scene_cases.build(rt, "c4_final_scene_64x64_8spp_d40"))
    kinds = [int(k) for k in lay["steps"][:, 0]]
    assert lay["ordered"] and kinds.count(1) == 2 and kinds.count(2) == 0 and 1 <= kinds.count(0) <= 3
    assert len(lay["spheres"]) == 1008 and len(lay["quads"]) == 2401


def test_a_medium_inside_a_frame_is_left_to_the_reference_walk(rt):
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    assert rt.debug_ordered_layout(custom_scenes.media_scene(cam, 0))["ordered"]
    lay = rt.debug_ordered_layout(custom_scenes.media_scene(cam, 0, nested=True))
    assert not lay["ordered"] and len(lay["nodes"]) == 0


def test_a_scene_the_ordered_layout_turns_down_is_left_untouched(rt):
    """More steps than the kernel's sequence table holds (ORDERED_MAX_STEPS): build_ordered gives up — and must not have
    moved any medium's boundary sphere by then (round-1 bug: first_node was rewritten before the size check)."""
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    scene = custom_scenes.many_media_scene(cam, 40)
    lay = rt.debug_ordered_layout(scene)
    assert not lay["ordered"] and len(lay["media"]) == 40
    for m, first in enumerate(lay["media"]):
        b = scene.spheres[scene.media[m].boundary.index]
        assert tuple(lay["spheres"][first][:4]) == (b.center.x, b.center.y, b.center.z, b.radius), m
    # a few steps fewer and the same kind of scene is taken
    assert rt.debug_ordered_layout(custom_scenes.many_media_scene(cam, 20))["ordered"]


def test_random_spheres_tree_is_shallow_and_tight(rt):
    """BASELINE configs 1/2: 485 spheres.  The SAH tree's total child-box area (what a random ray's visit count
    is proportional to) must stay well under the reference tree's."""
    hs = scene_cases.build(rt, "c2_random_balls_96x64_8spp_d50")
    lay = rt.debug_ordered_layout(hs)
    assert len(lay["spheres"]) == 485 and len(lay["nodes"]) <= 484
    assert lay["stack_entries"] <= 16

    def half_area(b):
        e = b[:, 1] - b[:, 0]
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0]

    ordered_area = sum(half_area(f32_box(rec[6 * s:6 * s + 6])) for rec in lay["nodes"] for s in range(2) if int(rec[12 + s]) >> 29 != KIND_EMPTY)
    threaded = rt.debug_compiled_nodes(hs, refit=True)
    ref_area = sum(half_area(np.array([[n.lo32[k], n.hi32[k]] for k in range(3)], dtype=np.float64)) for n in threaded if not n.no_bbox)
    assert ordered_area < 0.6 * ref_area, (ordered_area, ref_area)


def test_hand_made_scenes_compile(rt):
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    for scene in (custom_scenes.tie_scene(cam, 0), custom_scenes.single_sphere_scene(cam), custom_scenes.empty_frame_scene(cam),
                  custom_scenes.media_scene(cam, 0), custom_scenes.media_scene(cam, 1), custom_scenes.media_scene(cam, 2)):
        lay = rt.debug_ordered_layout(scene)
        assert lay["ordered"]
        seen = set()
        walk_steps(lay, seen)
        assert len(seen) == len(lay["spheres"]) + len(lay["quads"])
    lay = rt.debug_ordered_layout(custom_scenes.single_sphere_scene(cam))
    assert len(lay["nodes"]) == 1 and int(lay["nodes"][0][13]) >> 29 == KIND_EMPTY  # a root record with one child


def test_ties_depend_on_the_scan_order_in_the_oracle(rt, oracle):
    """Coincident copies: which one is seen is decided by the scan order (a later quad replaces, a later sphere does
    not) — so permuting the world list must change the image, and the GPU tests can tell a wrong tie rule."""
    cam = scene_cases.build(rt, "quads_64x64_8spp")
    imgs = [oracle.render(custom_scenes.tie_scene(cam, order), rt.render_params(seed=3)) for order in range(3)]
    assert (imgs[0] != imgs[1]).mean() > 0.05 and (imgs[0] != imgs[2]).mean() > 0.05  # (most of the view is sky)
    # and the tight box mode agrees with the reference's on every one of them
    for order in range(3):
        tight = oracle.render(custom_scenes.tie_scene(cam, order), rt.render_params(seed=3), aabb_mode=oracle.ORC_AABB_TIGHT)
        assert (tight.view(np.uint64) == imgs[order].view(np.uint64)).all()


def test_small_frames_keep_their_primitives_in_one_leaf(rt):
    """Cornell box (rt_ordered.hpp flat_max): the six walls are ONE leaf under the root, beside the record holding the two
    instances; each box's six faces are one leaf under its frame's root.  4 records in all (17 with a tree per frame).
    With the shortcut off (rt_scene_options.flat_max = 0) the trees are back."""
    hs = scene_cases.build(rt, "c3_cornell_box_64x64_16spp_d50")
    lay = rt.debug_ordered_layout(hs)
    assert lay["ordered"] and len(lay["nodes"]) == 4 and len(lay["quads"]) == 18 and len(lay["instances"]) == 2
    refs = lambda node: [int(lay["nodes"][node][12 + k]) for k in range(2)]
    kind, count = (lambda r: r >> 29), (lambda r: ((r >> 26) & 7) + 1)
    world = refs(lay["root"])
    assert kind(world[0]) == KIND_QUADS and count(world[0]) == 6 and kind(world[1]) == KIND_INNER
    assert sorted(kind(r) for r in refs(world[1] & ((1 << 26) - 1))) == [KIND_INSTANCE, KIND_INSTANCE]
    for inst in lay["instances"]:
        r = refs(int(inst[7]))
        assert kind(r[0]) == KIND_QUADS and count(r[0]) == 6 and kind(r[1]) == KIND_EMPTY
    # random-spheres (485 spheres) is a tree as before: only frames of <= 8 primitives go flat
    assert len(rt.debug_ordered_layout(scene_cases.build(rt, "c1_random_balls_400x225_10spp_d10"))["nodes"]) == 484
    assert len(rt.debug_ordered_layout(hs, flat_max=0)["nodes"]) == 17


def test_four_child_records_hold_every_primitive_once(rt):
    """rt_scene_options.wide: the binary trees collapsed to records of four children (rt_ordered.hpp widen) — every primitive in exactly
    one leaf, every child record's boxes inside its slot's box, empty slots closed to every ray; half the records, half the stack."""
    for name in scene_cases.CASES:
        hs = scene_cases.build(rt, name)
        for opts in ({}, {"flat_max": 0}, {"leaf_max": 4}):
            st = rt.debug_wide_layout(hs, **opts)
            media_boundaries = 2 if "final_scene" in name else 0  # (solved in place: in no leaf)
            assert st["violations"] == 0 and st["found"] == st["primitives"] - media_boundaries, (name, opts, st)
    st = rt.debug_wide_layout(scene_cases.build(rt, "c1_random_balls_400x225_10spp_d10"))
    assert st["records"] <= 0.5 * 484 and st["stack_entries"] <= 7 and st["deepest"] <= 7, st
