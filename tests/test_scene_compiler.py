"""The scene compiler inside librt_amd (object graph -> threaded records, box refit, f32 rounding), checked on the CPU
through rt_debug_compiled_nodes: the walk relies on these invariants to be allowed to skip a subtree."""
import numpy as np
import pytest

NK_INNER, NK_SPHERES, NK_QUADS, NK_INST_ENTER, NK_INST_EXIT, NK_MEDIUM_ENTER, NK_MEDIUM_EXIT, NK_MEDIUM_SPHERE = range(8)


def nodes_of(rt, scene, refit, **kw):
    hs = rt.HostScene(scene, spp=1, earth_image="synthetic:16x8", **kw)
    return hs, list(rt.debug_compiled_nodes(hs, refit))


@pytest.mark.parametrize("scene", range(9))
@pytest.mark.parametrize("bvh", ["reference", "sah"])
def test_threaded_links_and_containment(rt, scene, bvh):
    hs, nodes = nodes_of(rt, scene, True, bvh=bvh)
    n = len(nodes)
    assert n > 0
    for i, nd in enumerate(nodes):
        assert i < nd.skip <= n, (i, nd.skip)                     # links only go forward: the walk terminates
        # the f32 box the kernel tests contains the f64 box (outward rounding)
        for k in range(3):
            assert float(nd.lo32[k]) <= nd.lo[k] and nd.hi[k] <= float(nd.hi32[k])
        if nd.kind in (NK_SPHERES, NK_QUADS, NK_MEDIUM_SPHERE) and not nd.no_bbox:
            for k in range(3):                                     # a leaf's box contains its primitives
                assert nd.lo[k] <= nd.prim_lo[k] and nd.prim_hi[k] <= nd.hi[k], (i, k)

    # the box of an inner record contains the boxes of its direct children (same frame); a flat child (a quad) is
    # padded by 5e-5 on its thin axis (src/aabb.rs:35-53) and may stick out of a parent by that much: the parent
    # bounds the geometry, not the padding
    def check(i, end):
        k = i
        while k < end:
            nd = nodes[k]
            if nd.kind == NK_INNER:
                lo = np.array(nd.lo); hi = np.array(nd.hi)
                j = k + 1
                while j < nd.skip:
                    c = nodes[j]
                    if not c.no_bbox and not nd.no_bbox:
                        assert (lo <= np.array(c.lo) + 6e-5).all() and (np.array(c.hi) <= hi + 6e-5).all(), (k, j)
                    j = c.skip
                check(k + 1, nd.skip)
            elif nd.kind in (NK_INST_ENTER, NK_MEDIUM_ENTER):
                assert nodes[nd.skip - 1].kind in (NK_INST_EXIT, NK_MEDIUM_EXIT)   # brackets are balanced
                check(k + 1, nd.skip - 1)
            k = nd.skip
    check(0, n)
    d = hs.desc
    assert sum(nd.b for nd in nodes if nd.kind == NK_QUADS) >= d.n_quads
    assert sum(1 for nd in nodes if nd.kind in (NK_MEDIUM_ENTER, NK_MEDIUM_SPHERE)) == d.n_media


def test_refit_only_shrinks_boxes_and_fixes_the_origin_spanning_cubes(rt):
    hs, tight = nodes_of(rt, 8, True)
    _, loose = nodes_of(rt, 8, False)
    assert len(tight) == len(loose)
    spans_origin = lambda nd: all(nd.lo[k] <= 0.0 <= nd.hi[k] for k in range(3))
    shrunk = 0
    for a, b in zip(tight, loose):
        assert (a.kind, a.skip, a.a, a.b, a.no_bbox) == (b.kind, b.skip, b.a, b.b, b.no_bbox)
        if a.no_bbox:
            continue
        for k in range(3):
            assert b.lo[k] <= a.lo[k] and a.hi[k] <= b.hi[k]
        shrunk += a.hi[0] - a.lo[0] < b.hi[0] - b.lo[0]
    cubes_loose = [b for b in loose if b.kind == NK_QUADS and b.b == 6]
    cubes_tight = [a for a in tight if a.kind == NK_QUADS and a.b == 6]
    assert len(cubes_loose) == 400
    # the reference's cube lists all start from the all-zero default box (src/hittable.rs:50-57) ...
    assert all(spans_origin(b) for b in cubes_loose)
    # ... the refitted ones are the cubes themselves: 100 wide, and only the four around the origin touch it
    assert sum(spans_origin(a) for a in cubes_tight) <= 4
    assert all(abs((a.hi[0] - a.lo[0]) - 100.0) < 1e-6 for a in cubes_tight)
    assert shrunk > 400


def test_sphere_bounded_media_become_one_record(rt):
    _, nodes = nodes_of(rt, 8, True)            # final_scene: both media are bounded by a Sphere (src/main.rs:565-587)
    assert sum(nd.kind == NK_MEDIUM_SPHERE for nd in nodes) == 2 and not any(nd.kind == NK_MEDIUM_ENTER for nd in nodes)
    _, smoke = nodes_of(rt, 7, True)            # cornell_smoke: boundaries are Translate(RotateY(cube)) (src/main.rs:469-489)
    assert sum(nd.kind == NK_MEDIUM_ENTER for nd in smoke) == 2 and sum(nd.kind == NK_INST_ENTER for nd in smoke) == 2
