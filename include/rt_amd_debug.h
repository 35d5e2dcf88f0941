/* rt_amd_debug.h — test and tuning hooks of librt_amd.  NOT part of the drop-in boundary (include/rt_amd.h): a host
 * renderer never needs these.  tests/, tools/ and bench.py's work counters use them.
 *
 * The two setters change PROCESS-WIDE defaults (under a mutex): they exist for A/B runs in tests and tools; product code
 * passes rt_scene_options to rt_scene_create_ex instead. */
#ifndef RT_AMD_DEBUG_H
#define RT_AMD_DEBUG_H

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: evaluates one of the device-side scalar functions of the normative arithmetic over host arrays
 * (out[i] = f(a[i], b[i])), so that tests can compare the GPU's results with the oracle's bit for bit.
 * For the two RNG ops, a[i] and b[i] carry the BIT PATTERNS of the 64-bit stream key and of the draw number
 * (0-based, as in "draw n" of the RNG definition below). */
typedef enum rt_debug_op {
    RT_DEBUG_LOG = 1, RT_DEBUG_SIN = 2, RT_DEBUG_ACOS = 3, RT_DEBUG_ATAN2 = 4 /* atan2(a, b) */, RT_DEBUG_POW5 = 5,
    RT_DEBUG_SQRT = 6, RT_DEBUG_DIV = 7 /* a / b */, RT_DEBUG_MUL_ADD = 8 /* a * b + a, two roundings */,
    RT_DEBUG_RNG_RANDOM = 9, RT_DEBUG_RNG_RANGE = 10 /* gen_range(-1.0..1.0) */,
    RT_DEBUG_F32_ABOVE = 11, RT_DEBUG_F32_BELOW = 12 /* the ordered walk's outward f32 conversions of an interval end (as doubles) */,
    RT_DEBUG_RNG_UNNEXT = 13 /* random() number `b` of stream `a` after two draws too many were made and taken back (Rng::unnext) */
} rt_debug_op;
int rt_debug_eval(int32_t op, int64_t n, const double *a, const double *b, double *out, int device);

/* Test hook: runs the kernel's conservative f32 box test and the exact f64 slab test on n (ray, box) pairs —
 * rays[i] = (origin xyz, direction xyz), boxes[i] = (lo xyz, hi xyz), interval (tmin, tmax) — and reports, per pair,
 * whether each test enters the box: out_f32_hit bit 0 = the reference-order walk's test, bits 1 and 2 = the ordered walk's
 * pair test with the box in slot 0 / slot 1.  The f32 tests must enter wherever the exact one does (tests/test_gpu_parity.py). */
int rt_debug_box_tests(int64_t n, const double *rays, const double *boxes, double tmin, double tmax,
                       uint8_t *out_exact_hit, uint8_t *out_f32_hit, int device);

/* Test hook: the quad stage's conservative f32 filter (rt_device_scene.h quad_pair_keep) and the exact f64 Quad::hit
 * (src/quad.rs:96-127) on n (ray, quad) pairs — rays[i] = (origin xyz, direction xyz), quads[i] = (Q xyz, u xyz, v xyz), the
 * derived fields as Quad::new computes them (src/quad.rs:24-27), interval [tmin, tmax].  out_exact_hit[i] = 1: the exact test
 * accepts; out_keep[i] bit 0 / bit 1: the filter keeps the quad in the first / second slot of its pair record; bit 2 / bit 3: it
 * claims alpha and beta certainly inside [0, 1] (the exact test then skips them); bit 4: the exact alpha and beta ARE inside (1 also
 * where the exact test ends before it gets to them).  The filter must keep whatever the exact test accepts, and may claim "inside"
 * only where bit 4 is set (tests/test_gpu_parity.py). */
int rt_debug_quad_filter_tests(int64_t n, const double *rays, const double *quads, double tmin, double tmax, uint8_t *out_exact_hit,
                               uint8_t *out_keep, int device);

/* Test hook (no GPU needed): runs the scene compiler and returns the records the device would walk — the f64 box each
 * carries after refitting (refit != 0) or as the reference has it (refit == 0), the outward-rounded f32 box actually
 * tested, the threaded links, and the bound of the record's own primitives — so that tests can check the compiler's
 * invariants (links, containment) on the CPU.  out_nodes may be NULL to query the count. */
typedef struct rt_debug_node {
    double lo[3], hi[3];           /* box of the record (f64) */
    float lo32[3], hi32[3];        /* what the kernel tests */
    double prim_lo[3], prim_hi[3]; /* bound of the leaf's own primitives (+inf/-inf if none) */
    uint32_t skip, kind, no_bbox, a, b, _pad;
} rt_debug_node;
int rt_debug_compiled_nodes(const rt_scene_desc *desc, int32_t refit, rt_debug_node *out_nodes, int64_t capacity,
                            int64_t *out_count);

/* Test / tuning hook: how scenes created from now on are walked.  ordered = 2: the library's own trees, nearest child
 * first — with media, a sequence of trees and media in the reference's scan order (DESIGN.md "Ordered layout"; a medium
 * inside a Translate / RotateY frame keeps the other walk); 0: every scene walks the reference's tree in the reference's
 * order; 1 (default): as 2, except for scenes measured faster the other way (a single primitive).
 * leaf_max > 0: primitives per leaf of those trees at most; 0: back to the default.  Negative: keep.
 * Affects speed only, never results. */
int rt_debug_set_traversal(int32_t ordered, int32_t leaf_max);

/* Test / tuning hook: which shortcuts the ordered walk of scenes created / rendered from now on takes (negative: keep).
 * flat_max: a frame of at most this many primitives keeps them in one leaf under its root (0: off; default 8) — scene creation;
 * start_shortcut: a query starts with the primitives of a leaf under the root that spans the scene (0 / 1) — per render;
 * defer_instances: the world frame's instances are walked after its own tree (0 / 1) — per render;
 * seq_lookahead: a query looks ahead at the later steps of the world's sequence when it starts (0 / 1) — per render;
 * slow_min, slow_age: hits on a noise texture wait in the shade stage for slow_min of their kind, at most slow_age shade rounds
 * (slow_min 1: nobody waits; defaults 4, 32) — per render.
 * Affects speed only, never results. */
int rt_debug_set_walk_shortcuts(int32_t flat_max, int32_t start_shortcut, int32_t defer_instances, int32_t seq_lookahead,
                                int32_t slow_min, int32_t slow_age);

/* Test hook: the ordered layout the scene compiler builds for `desc` (no device needed).  Set the cap_* fields and
 * the pointers (any may be null: only the counts are returned then).
 * nodes: 16 words per record = two boxes as 6 floats (x.lo, x.hi, y.lo, y.hi, z.lo, z.hi), two child references
 * (kind << 29 | (count - 1) << 26 | index; kind 0 record, 1 spheres, 2 quads, 3 instance, 7 empty), 2 unused.
 * spheres: 9 doubles = center, radius, center_vec, seq, is_moving.  quads: 10 = q, u, v, seq.
 * instances: 8 = offset, sin, cos, parent, flags (1 translate, 2 rotate), root record.
 * steps: the world frame's sequence, 28 words per step = kind (0 tree, 1 medium bounded by one sphere, 2 medium with a
 * boundary tree), a (tree: root record; medium: its index), b (boundary tree's root), moving, box as 6 floats, 2 unused,
 * then as doubles: the boundary sphere's center (3), radius, center_vec (3), and the medium's neg_inv_density.
 * media: per medium, the index of its boundary sphere (kind 1 steps). `root` is the first step's tree. */
typedef struct rt_debug_ordered {
    int64_t cap_nodes, cap_spheres, cap_quads, cap_instances, cap_steps, cap_media;
    int64_t n_nodes, n_spheres, n_quads, n_instances, n_steps, n_media;
    uint32_t ordered, root, stack_entries, _pad;
    uint32_t *nodes;
    double *spheres, *quads, *instances;
    uint32_t *steps, *media;
} rt_debug_ordered;
int rt_debug_ordered_layout(const rt_scene_desc *desc, rt_debug_ordered *io);
/* ... with the tree options of `options` (leaf_max, flat_max; NULL: the process defaults) */
int rt_debug_ordered_layout_ex(const rt_scene_desc *desc, const rt_scene_options *options, rt_debug_ordered *io);

/* Structure check of the four-child layout (rt_scene_options.wide), on the CPU: out[0] records, out[1] primitives found in leaves,
 * out[2] primitives in the scene's tables (a medium's boundary sphere solved in place is in no leaf), out[3] stack entries a walk needs,
 * out[4] violations (a primitive in no leaf or in two, a child record's boxes outside its slot's box, a bad reference, an empty slot
 * whose box a ray could enter), out[5] records on the longest chain. */
int rt_debug_wide_layout(const rt_scene_desc *desc, const rt_scene_options *options, uint64_t out[6]);

/* How the calling thread's last render was launched: out[0] = 0 (reserved), out[1] = LDS level, out[2] = workgroup threads,
 * out[3] = workgroups. */
int rt_debug_last_launch(uint32_t out[4]);

/* Profiling hook: where the last rt_render_device_counted call's waves spent their time.  For each scheduler stage
 * (box, sphere, quad, other, shade, new-job): rounds run, lanes active summed over those rounds, shader cycles (s_memtime)
 * summed over waves; then six parts of the shade / path-end rounds, cycles only (hit rebuild, unit-sphere rejection sampling,
 * texture, material + next ray, attenuation products + sample store, job hand-out + camera ray — the stage slots keep the
 * remainder: scheduling and the start of the next query). */
int rt_debug_stage_profile(uint64_t out[36]);

/* Tuning hook: the wave scheduler's knobs (DESIGN.md "Scheduler").  A deferred stage runs once th/64 of a wave's
 * live lanes wait for it (th_new: the path-end / next-job stage); the box loop keeps running while th_box/64 of them are in
 * it; use_lds = 0 forces the
 * scene to be gathered from global memory even when it fits the LDS.  A negative threshold restores the built-in
 * per-scene-class preset; a negative use_lds keeps the current setting.
 * Affects speed only, never results.  Process-wide; not for concurrent use with renders. */
int rt_debug_set_tuning(int32_t th_prim, int32_t th_other, int32_t th_shade, int32_t th_box, int32_t use_lds,
                        int32_t th_new);


#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_DEBUG_H */
