// rt_kernel.hip — the device code of librt_amd for MI355X (gfx950): the render megakernel and the small frame-end kernels.
//
// path_kernel is a persistent, wave-scheduled stage machine.  A work unit ("job") is one camera path = (pixel, sample);
// lanes are not tied to pixels: every wave pulls job ranges from a device-side counter and a lane that finishes a path takes
// the next job, so the grid is sized to the machine, not to the image.  Per path a lane generates the camera ray
// (src/camera.rs:112-137), walks the scene — the library's own SAH trees nearest child first (rt_ordered.hpp) or the
// threaded records in the reference's order (rt_layout.h): box test, sphere / quad intersection, frame changes, constant
// media — shades the closest hit with the reference's five materials and four textures (src/material.rs, src/texture.rs)
// and loops over bounces (ray_color, src/renderer.rs:139-155, made iterative).  All arithmetic that reaches a result is f64
// in the reference's operation order with no FMA contraction, so the per-pixel sums equal the CPU restatement's bit for bit.
//
// There is no dense contraction anywhere on this path: no MFMA.  The bound is VALU issue under divergence (DESIGN.md
// "Roofline").  Host side: rt_api.cpp (C ABI, scene upload, launches), rt_debug.cpp (test hooks), rt_gather.cpp (RCCL).
#include "rt_device_scene.h"

// =====================================================================================================
// Device side
// =====================================================================================================
namespace {

// ---- the render kernel: a wave-scheduled stage machine ----------------------------------------------------
//
// Work unit ("job") = one camera path = (pixel, sample).  Lanes are not tied to pixels: a lane that finishes a
// path takes the next job of the wave's current job range, so no lane waits for a neighbour's longer path.
// A path's colour goes to a sample buffer indexed by job; sum_samples_kernel then adds each pixel's samples in
// order s = 0, 1, 2, ... — the reference's sequential `avg_color += new_color` (src/renderer.rs:35-40) — so
// the result does not depend on which lane traced what, or when.
//
// Every lane is in one of a few stages (box test / sphere test / quad test / frame change or medium step /
// shade+next ray).  Each scheduler round the wave counts its lanes per stage with ballots and runs ONE stage for
// all the lanes in it: the rare stages are deferred until enough lanes have queued up for them, so that every
// instruction stream the wave issues has most of its 64 lanes active.
// A round costs the same for any number of lanes, so what keeps the lanes of a wave TOGETHER is nearly free and what makes a
// lane wait is dear: a query starts with the primitives of a leaf that spans the scene (start shortcut: one round of the
// primitive stage for all the lanes a shade round has just served), small frames keep their primitives in one leaf (flat
// leaves, rt_ordered.hpp), the world's instances are walked after its own tree (no frame change back in between), a query
// that cannot reach a later step of the world's sequence ends it at its start.
// (a lane waiting for the path-end stage holds ST_NEWJOB + Terminal: what its path ended on)
enum Stage : uint32_t { ST_BOX = 0, ST_SPHERE = 1, ST_QUAD = 2, ST_OTHER = 3, ST_SHADE = 4, ST_NEWJOB = 5, ST_DONE = 9 };
// what a path ended on (ST_NEWJOB multiplies the parked attenuations back onto it): the background, Color::ONE (a light:
// its emitted colour is the last thing parked), nothing (absorbed / depth exhausted); STORED: no path to finish
enum Terminal : uint32_t { TERM_BACKGROUND = 0, TERM_ONE = 1, TERM_ZERO = 2, TERM_STORED = 3 };

#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 3 // waves per SIMD the register allocator must leave room for in the kernels that gather the scene from global memory.
// (Measured again in round 3 with 4: the every-feature kernel then spills 63-75 registers instead of 0-2 — none in the box loop — and
// final_scene runs 2 % faster, 1114 -> 1136 Msamples/s, but the spills' scratch traffic takes its HBM bytes from 476 GB to 2.4 TB per
// frame: not worth it.  The LDS-resident every-feature kernels, whose scenes are small, do run four waves: rt_kernels.h.)
#endif

// LDS: 0 = scene gathered from global memory; 1 = node table in LDS; 2 = + sphere table; 3 = + quad table
// ORDERED: walk the compiler's own trees nearest child first (scenes without a ConstantMedium), else the threaded
// records in the reference's order
// AUX: the small tables (materials, textures, frames, media, Perlin) are copied into the LDS too — for scenes whose big
// tables do not fit there, so that a hit's material -> texture -> noise chain is not three trips to memory
// WIDE: the ordered walk's records hold four children (rt_layout.h ONode4) instead of two
template <bool COUNT, int LDS, int THREADS, uint32_t FEAT, bool ORDERED, bool AUX = false, bool WIDE = false>
__global__ __launch_bounds__(THREADS, LDS ? 1 : RT_MIN_WAVES) void path_kernel(const KParams P) {
    constexpr bool HAS_SPHERES = (FEAT & F_SPHERES) != 0, HAS_QUADS = (FEAT & F_QUADS) != 0, HAS_FRAMES = (FEAT & F_FRAMES) != 0,
                   HAS_MEDIA = (FEAT & F_MEDIA) != 0, HAS_TEXTURES = (FEAT & F_TEXTURES) != 0;
    constexpr bool HAS_OTHER = HAS_FRAMES || HAS_MEDIA;
    // the quads of a multi-quad leaf go through the conservative f32 filter first (rt_device_scene.h quad_pair_keep): scenes that live
    // in the LDS whole (their filter records with them)
    constexpr bool QFILT = ORDERED && LDS == 3 && HAS_QUADS;
    // parked attenuations loaded per trip when a path ends: 4 where registers allow (the general kernels at 128 registers
    // already spill; their 256-thread form has 168)
    constexpr uint32_t CHAIN = (HAS_TEXTURES && LDS != 0) ? 1u : 4u;
    const double INF = __builtin_inf();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    if constexpr (LDS != 0) { // (the per-lane stacks of the ordered walk follow the image)
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw);
        for (uint32_t k = threadIdx.x; k < P.lds_image_bytes / 16u; k += THREADS) dst[k] = P.lds_image[k];
        __syncthreads();
    }
    // the world frame's sequence (scenes with media) sits behind the stacks
    const OSeq *const seq_tab = reinterpret_cast<const OSeq *>(lds_raw + P.lds_seq_off);
    if constexpr (ORDERED && HAS_MEDIA) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw + P.lds_seq_off);
        const uint4 *src = reinterpret_cast<const uint4 *>(P.oseq);
        for (uint32_t k = threadIdx.x; k < P.n_oseq * (uint32_t)(sizeof(OSeq) / 16u); k += THREADS) dst[k] = src[k];
        __syncthreads();
    }
    const Sphere *const sphere_tab = LDS >= 2 ? reinterpret_cast<const Sphere *>(lds_raw + P.lds_off_spheres) : P.spheres;
    const Quad *const quad_tab = LDS == 3 ? reinterpret_cast<const Quad *>(lds_raw + P.lds_off_quads) : P.quads;
    if constexpr (AUX) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw + P.lds_aux_off);
        for (uint32_t k = threadIdx.x; k < P.aux_bytes / 16u; k += THREADS) dst[k] = P.aux_image[k];
        __syncthreads();
    }
    const DMaterial *const mats_tab = AUX ? reinterpret_cast<const DMaterial *>(lds_raw + P.lds_aux_off + P.aux_off_mats) : P.mats;
    const rt_texture *const texs_tab = AUX ? reinterpret_cast<const rt_texture *>(lds_raw + P.lds_aux_off + P.aux_off_texs) : P.texs;
    const Instance *const inst_tab = AUX ? reinterpret_cast<const Instance *>(lds_raw + P.lds_aux_off + P.aux_off_insts) : P.insts;
    const Medium *const media_tab = AUX ? reinterpret_cast<const Medium *>(lds_raw + P.lds_aux_off + P.aux_off_media) : P.media;
    const rt_perlin *const perlin_tab = AUX ? reinterpret_cast<const rt_perlin *>(lds_raw + P.lds_aux_off + P.aux_off_perlins) : P.perlins;
    // the world-frame ray of a lane while it walks inside an instance (Translate / RotateY subtree) is parked in the LDS where
    // the scene leaves room (Cornell: a frame exit waited for six loads from memory, a fifth of the frame's cycles went there),
    // else in global memory (touched 0.3-2.6 times per sample)
    // (in the kernels that are short of registers the slot addresses are formed where they are used, from a lane index the optimiser
    // cannot look behind: hoisted out of the main loop the twelve of them — six per memory — sat in 24 registers from the kernel's first
    // instruction to its last, or in scratch: every-feature kernel 168 registers + 20-42 spilled -> 167-168 + 0-2)
    constexpr bool SLOTS_AT_USE = HAS_TEXTURES || HAS_MEDIA; // (the other kernels have the registers and change frames often: Cornell
    // 2448 Msamples/s with the addresses kept, 2401-2425 with them formed at use; final_scene, 0.3 frame changes per sample: 1090 / 1117)
    const bool world_in_lds = P.lds_world_off != 0xffffffffu;
    double *const world_lds = reinterpret_cast<double *>(lds_raw + (world_in_lds ? P.lds_world_off : 0u)) + threadIdx.x;
    double *const world_glb = P.world_slots + gtid;
    auto slot_lane = [&]() -> uint32_t {
        uint32_t t = threadIdx.x;
        asm volatile("" : "+v"(t));
        return t;
    };
    auto park_world_ray = [&](V3 po, V3 pd) {
        if constexpr (SLOTS_AT_USE) {
            const uint32_t t = slot_lane();
            if (P.lds_world_off != 0xffffffffu) { // (two branches, not one selected pointer: each keeps its address space)
                double *const w = reinterpret_cast<double *>(lds_raw + P.lds_world_off) + t;
                w[0] = po.x; w[THREADS] = po.y; w[2 * THREADS] = po.z;
                w[3 * THREADS] = pd.x; w[4 * THREADS] = pd.y; w[5 * THREADS] = pd.z;
            } else {
                double *const w = P.world_slots + (blockIdx.x * blockDim.x + t);
                const size_t ws = P.n_threads;
                w[0] = po.x; w[ws] = po.y; w[2 * ws] = po.z;
                w[3 * ws] = pd.x; w[4 * ws] = pd.y; w[5 * ws] = pd.z;
            }
        } else if (world_in_lds) {
            world_lds[0] = po.x; world_lds[THREADS] = po.y; world_lds[2 * THREADS] = po.z;
            world_lds[3 * THREADS] = pd.x; world_lds[4 * THREADS] = pd.y; world_lds[5 * THREADS] = pd.z;
        } else {
            const size_t ws = P.n_threads;
            world_glb[0] = po.x; world_glb[ws] = po.y; world_glb[2 * ws] = po.z;
            world_glb[3 * ws] = pd.x; world_glb[4 * ws] = pd.y; world_glb[5 * ws] = pd.z;
        }
    };
    auto restore_world_ray = [&](V3 &ro, V3 &rd) {
        if constexpr (SLOTS_AT_USE) {
            const uint32_t t = slot_lane();
            if (P.lds_world_off != 0xffffffffu) {
                const double *const w = reinterpret_cast<const double *>(lds_raw + P.lds_world_off) + t;
                ro = v3(w[0], w[THREADS], w[2 * THREADS]);
                rd = v3(w[3 * THREADS], w[4 * THREADS], w[5 * THREADS]);
            } else {
                const double *const w = P.world_slots + (blockIdx.x * blockDim.x + t);
                const size_t ws = P.n_threads;
                ro = v3(w[0], w[ws], w[2 * ws]);
                rd = v3(w[3 * ws], w[4 * ws], w[5 * ws]);
            }
        } else if (world_in_lds) {
            ro = v3(world_lds[0], world_lds[THREADS], world_lds[2 * THREADS]);
            rd = v3(world_lds[3 * THREADS], world_lds[4 * THREADS], world_lds[5 * THREADS]);
        } else {
            const size_t ws = P.n_threads;
            ro = v3(world_glb[0], world_glb[ws], world_glb[2 * ws]);
            rd = v3(world_glb[3 * ws], world_glb[4 * ws], world_glb[5 * ws]);
        }
    };

    Counts cn{};
    Rng rng;
    rng.x = 0; rng.y = 0;

    // ---- per-lane path state ----
    V3 o = v3(0, 0, 0), d = v3(0, 0, 1); // current-frame ray
    double a = 1.0, time = 0.0;          // |d|^2 (Sphere::hit's `a`), ray time
    // f32 copies for the conservative box test: origin, 1/d, and the bound E on what rounding the origin to f32
    // can move a slab distance (see the box stage); `degenerate`: some 1/d or E is not finite -> enter every box
    std::conditional_t<ORDERED, RayPair32, Ray32> r32;
    if constexpr (ORDERED) r32 = make_ray_pair32(o, d, LDS != 0 ? P.lds_off_node_b : (WIDE ? 32u : 16u), P.box_extent); else r32 = make_ray32(o, d);
    float tmin32 = 0, tmax32 = 0;
    uint32_t job = 0;
    int32_t depth = 0;
    uint32_t n_att = 0;
    // ---- per-lane traversal state ----
    double cur_tmin = 0.001, cur_tmax = INF; // interval tests run against (ray_color's, or a medium boundary query's)
    double best_t = INF, med_t1 = 0.0;
    // (ordered walk, media) the draw of the sphere-bounded medium at the NEXT step of the sequence, made before the tree in front of it is
    // walked — for a ray that starts inside the ball (see where a query starts); NaN: none pending
    double pre_hd = __builtin_nan("");
    uint32_t best_prim = PRIM_NONE;
    int32_t best_inst = -1, cur_inst = -1;
    uint32_t node = 0, prim_cur = 0, prim_end = 0;
    uint32_t mode = 0; // ConstantMedium: 0 outside, 1 first boundary query, 2 second; bit 8: boundary was hit
    uint32_t stage = ST_NEWJOB + TERM_STORED;
    // ---- ordered walk: a stack of children set aside (one entry per level at most), in the LDS, [level][thread] ----
    // An entry is what `node` shall hold when the entry's turn comes: an inner record's index, or — for a LEAF child — its
    // parent's index with the "skip the other child" bit, so that the leaf's box is tested again, against the interval as
    // it has shrunk by then, before its primitive is (a sphere test costs 2-3 box rounds); S_EXIT: leave the current frame.
    using StackT = std::conditional_t<LDS != 0, uint16_t, uint32_t>;
    constexpr uint32_t S_EXIT = LDS != 0 ? 0xffffu : 0xffffffffu;
    constexpr uint32_t SKIP_CHILD0 = LDS != 0 ? 0x4000u : 0x40000000u, SKIP_CHILD1 = LDS != 0 ? 0x8000u : 0x80000000u,
                       NODE_INDEX = SKIP_CHILD0 - 1u;
    // wide records: an entry is a record's index and, above it, the mask of its children still to look at (all four for a record
    // the walk comes to for the first time)
    constexpr uint32_t W_SHIFT = LDS != 0 ? 12u : 26u, W_INDEX = (1u << W_SHIFT) - 1u, W_FULL = WIDE ? (0xfu << W_SHIFT) : 0u;
    constexpr uint32_t NODE_FRAME_EXIT = 0xffffffffu, NODE_SEQ_NEXT = 0xfffffffeu; // ST_OTHER: leave the current frame / take the next step of the world's sequence
    StackT *const stack = reinterpret_cast<StackT *>(lds_raw + P.lds_stack_off) + threadIdx.x;
    uint32_t sp = 0;
    uint32_t seq_pc = 0; // (scenes with media) the next step of the world frame's sequence
    const uint32_t first_node = ORDERED ? (P.o_root == 0xfffffffeu ? P.o_root : (P.o_root | W_FULL)) : 0u;
    // what a lane does next in an ordered walk: go to `ref` if it has one, else take the last child set aside
    // (written as selects of VALUES: when the branches assign different variables the optimiser turns them into one store
    // through a selected address, and the variables end up in scratch memory — in the hottest loop of the kernel)
    // Instances of the world frame are not entered where the walk meets them: the lane notes them (`deferred`, one bit per instance,
    // set in the box round) and walks them when the world's own tree is done (the scheduler sends it there), one after the other.  The closest hit is a minimum (ties: wins_tie), so the order
    // of the visits is free; what it buys: no frame change BACK to the world between them (the world ray is only restored when the
    // query's result is used), i.e. one ST_OTHER round per instance instead of two — a sixth of Cornell's cycles went into those rounds.
    // (Instances inside an instance are entered on the spot, with an S_EXIT entry on the stack, as before.)
    constexpr uint32_t NODE_DEFERRED = 0x80000000u; // ST_OTHER: enter instance (node & 31) from the world frame
    // (Kernels for scenes with media are left as they were: final_scene has one instance, behind the last step of its sequence — nothing
    // to gain, and the two instructions per box round that note the instances cost it 3 %.)
    constexpr bool DEFER = HAS_FRAMES && ORDERED && !HAS_MEDIA;
    const bool defer = DEFER && P.defer_instances != 0;
    uint32_t deferred = 0;
    constexpr uint32_t SEQ_JUMP = 0x80000000u; // in seq_pc while a tree is walked: see the look-ahead where a query starts
    auto o_next = [&](bool have, uint32_t ref) {
        uint32_t new_stage, new_node = node, new_cur = prim_cur, new_end = prim_end;
        if (have) {
            // OrderedKind INNER / SPHERES / QUADS / INSTANCE = 0 / 1 / 2 / 3 = Stage ST_BOX / ST_SPHERE / ST_QUAD / ST_OTHER
            const uint32_t kind = ref >> OREF_KIND_SHIFT, index = ref & OREF_INDEX_MASK;
            const bool leaf = kind == OK_SPHERES || kind == OK_QUADS;
            new_stage = kind;
            new_node = leaf ? node : (kind == OK_INNER ? (index | W_FULL) : index); // (an inner record is come to with all four children to look at)
            new_cur = leaf ? index : prim_cur;
            new_end = leaf ? index + ((ref >> OREF_COUNT_SHIFT) & OREF_COUNT_MASK) + 1u : prim_end;
        } else if (sp != 0) { // the last child set aside
            sp--;
            const uint32_t e = stack[sp * THREADS];
            const bool leave_frame = HAS_FRAMES && e == S_EXIT;
            new_node = leave_frame ? NODE_FRAME_EXIT : e;
            new_stage = leave_frame ? (uint32_t)ST_OTHER : (uint32_t)ST_BOX;
        } else { // this tree is done (but for the instances it met: see the scheduler)
            new_stage = ST_SHADE;
            if constexpr (HAS_MEDIA) { // ... the world's sequence may go on; a boundary query reports to its medium
                const bool more = seq_pc < P.n_oseq || (mode & 3u) != 0;
                new_node = more ? NODE_SEQ_NEXT : node;
                new_stage = more ? (uint32_t)ST_OTHER : (uint32_t)ST_SHADE;
                if constexpr (ORDERED) {
                    // the look-ahead found that the next step this ray can reach is another tree (SEQ_JUMP | its index): on with that
                    // tree's root at once, not through a round of ST_OTHER that would only have looked the root up
                    if ((seq_pc & SEQ_JUMP) != 0u) {
                        const uint32_t k = seq_pc & ~SEQ_JUMP;
                        bool jump = true;
                        if (HAS_SPHERES && pre_hd == pre_hd) {
                            // step k - 1 is the medium whose draw was made ahead.  If the tree in front of it hit nothing within the clipped
                            // interval, the draw's verdict stands as it was found — the candidate, or nothing — and the walk goes on with the
                            // tree behind the medium; if it hit something, the medium's own step decides (the reference's predicate with that hit)
                            const bool tree_hit = best_prim != PRIM_NONE && (best_prim & PRIM_KIND_MASK) != PRIM_MEDIUM;
                            if (tree_hit) {
                                jump = false;
                                seq_pc = k - 1u;
                                new_node = NODE_SEQ_NEXT;
                                new_stage = ST_OTHER;
                            } else {
                                if (med_t1 == med_t1) { best_t = med_t1; best_prim = PRIM_MEDIUM | seq_tab[k - 1u].a; best_inst = cur_inst; }
                                cur_tmax = best_t;
                                tmax32 = f32_above(cur_tmax);
                                pre_hd = __builtin_nan("");
                            }
                        }
                        if (jump) {
                            new_node = seq_tab[k].a | W_FULL;
                            new_stage = ST_BOX;
                            seq_pc = k + 1u;
                        }
                    }
                }
            }
        }
        stage = new_stage; node = new_node; prim_cur = new_cur; prim_end = new_end;
    };
    // ties (ordered walk): two primitives hit at exactly the same t.  The reference scans in a fixed order and keeps the
    // first unless a later one passes its interval test: Sphere::hit wants t < closest (Interval::surrounds,
    // src/sphere.rs:70-76), Quad::hit t <= closest (Interval::contains, src/quad.rs:105-107) — so of two primitives
    // at the same t the later one in scan order wins iff it is a quad.  That relation is a total order (a later quad beats
    // everything before it, a sphere nothing), so applying it pairwise in ANY visiting order ends with the primitive the
    // reference's scan ends with.  `seq` is the scan position.
    auto wins_tie = [&](uint32_t my_seq, bool i_am_quad) -> bool {
        // (a medium hit is always earlier in the scan than whatever is tested now: the sequence runs in scan order)
        if (HAS_MEDIA && (best_prim & PRIM_KIND_MASK) == PRIM_MEDIUM) return i_am_quad;
        const bool best_is_quad = (best_prim & PRIM_KIND_MASK) == PRIM_QUAD;
        const uint32_t bi = best_prim & PRIM_INDEX_MASK;
        const uint32_t best_seq = best_is_quad ? quad_tab[bi].seq : (sphere_tab[bi].seq_moving >> 1);
        return my_seq > best_seq ? i_am_quad : !best_is_quad;
    };

    // Parked attenuations.  ray_color multiplies the attenuations back innermost first (src/renderer.rs:147-149), so a path parks one
    // per bounce and the path end multiplies them onto the terminal, last parked first.  For every material whose attenuation is a
    // constant — a SolidColor albedo, Metal — what is parked is the material's INDEX (16 bits): the path end reads the colour from the
    // material table (in the LDS wherever the small tables fit), the same values in the same order.  The last four indices live in two
    // registers used as a shift register (newest in the low half of ids[0]), so the path end finds them at fixed places; an index that
    // falls out of them goes to att_ids [level][thread] — 4 bytes per lane, consecutive lanes consecutive addresses.  A texture's value
    // (Checker / Image / Noise) is parked as the colour itself, in att_stack [level][component][thread], and ID_COLOUR in the index
    // stack says so (kernels with textures only; a scene with more than 65534 materials is rendered by those).  A level that a path does
    // not have reads the table's last entry, whose colour is Color::ONE: multiplying by it is the identity, bit for bit.
    // (Before: three 8-byte stores per bounce at a 24-byte lane stride and as many loads at the path end — 1.8x the bytes in partial
    // lines, 57 GB of the Cornell frame's 121 GB of writes.)
    constexpr uint32_t ID_COLOUR = 0xffffu;
    constexpr uint32_t IDW = 2u, IDS_IN_REGS = 2u * IDW;
    uint32_t ids[IDW] = {};
    // (element indices are 32 bits wide — launch_render checks the stacks have < 2^32 elements — and are formed from gtid where they
    // are used: a scalar base and one offset register instead of a pointer pair kept alive through the whole walk)
    uint32_t *const att_ids = P.att_ids;
    double *const att_col = P.att_stack;
    const uint32_t att_lanes = P.n_threads;
    auto park = [&](uint32_t id, V3 colour) {
        if constexpr (HAS_TEXTURES) {
            if (id == ID_COLOUR) {
                const uint32_t at = n_att * 3u * att_lanes + gtid;
                att_col[at] = colour.x; att_col[at + att_lanes] = colour.y; att_col[at + 2u * att_lanes] = colour.z;
            }
        }
        if (n_att >= IDS_IN_REGS) att_ids[(n_att - IDS_IN_REGS) * att_lanes + gtid] = ids[IDW - 1u] >> 16; // the oldest index in the registers makes room
    #pragma unroll
        for (uint32_t q = IDW - 1u; q > 0u; --q) ids[q] = __builtin_amdgcn_alignbit(ids[q], ids[q - 1u], 16);
        ids[0] = (ids[0] << 16) | id;
        n_att++;
    };
    auto parked_colour = [&](uint32_t id, uint32_t level) -> V3 {
        if constexpr (HAS_TEXTURES) {
            if (id == ID_COLOUR) {
                const uint32_t at = level * 3u * att_lanes + gtid;
                return v3(att_col[at], att_col[at + att_lanes], att_col[at + 2u * att_lanes]);
            }
        }
        return ld3(mats_tab[id].albedo);
    };

    // wave-uniform: the job range this wave currently owns
    uint32_t job_next = 0, job_end = 0, jobs_seen_left = P.n_jobs;
    bool jobs_left = true;

    const uint32_t n_nodes = P.n_nodes;

    auto refresh_ray32 = [&]() {
        if constexpr (ORDERED) r32 = make_ray_pair32(o, d, LDS != 0 ? P.lds_off_node_b : (WIDE ? 32u : 16u), P.box_extent);
        else r32 = make_ray32(o, d);
    };
    // (the ordered walk's test takes the interval rounded outward; the threaded one has the slack for either rounding)
    auto refresh_interval32 = [&]() {
        tmin32 = ORDERED ? f32_below(cur_tmin) : (float)cur_tmin;
        tmax32 = ORDERED ? f32_above(cur_tmax) : (float)cur_tmax;
    };

    // ---- ConstantMedium::hit (src/constant_medium.rs:33-71), shared by both walks ----
    // a medium bounded by one Sphere: the two boundary.hit calls solve the same quadratic (src/sphere.rs:58-83), first over
    // (-inf, inf), then over (t1 + 0.0001, inf)
    auto medium_sphere_hit = [&](uint32_t na, V3 center, V3 center_vec, bool moving, double radius, double neg_inv_density) {
        if (COUNT) { cn.medium_visits++; cn.sphere_tests++; }
        bool drew = false;
        if (moving) center = center + center_vec * time;
        const V3 oc = o - center;
        const double half_b = dot(oc, d);
        const double c = len2(oc) - radius * radius;
        const double discriminant = half_b * half_b - a * c;
        if (!(discriminant < 0.0)) {
            const double sqrtd = __builtin_sqrt(discriminant);
            const double root_a = (-half_b - sqrtd) / a, root_b = (-half_b + sqrtd) / a;
            // Interval::surrounds is strict at both ends (src/interval.rs:44-46)
            const bool a1 = -INF < root_a && root_a < INF, b1 = -INF < root_b && root_b < INF;
            if (a1 || b1) {
                const double t1 = a1 ? root_a : root_b;
                if (COUNT) cn.sphere_tests++;
                const double lo2 = t1 + 0.0001;
                const bool a2 = lo2 < root_a && root_a < INF, b2 = lo2 < root_b && root_b < INF;
                if (a2 || b2) {
                    const double t2 = a2 ? root_a : root_b;
                    double h1 = __builtin_fmax(t1, 0.001);
                    const double h2 = __builtin_fmin(t2, best_t);
                    if (h1 < h2) {
                        h1 = __builtin_fmax(h1, 0.0);
                        const double ray_length = __builtin_sqrt(len2(d));
                        const double distance_inside_boundary = (h2 - h1) * ray_length;
                        double hit_distance = pre_hd; // (the draw made where the query started, if one was)
                        if (!(pre_hd == pre_hd)) {
                            if (COUNT) cn.rng_draws++;
                            hit_distance = neg_inv_density * rt_log(rng.random());
                        }
                        drew = true;
                        if (hit_distance <= distance_inside_boundary) {
                            best_t = h1 + hit_distance / ray_length;
                            best_prim = PRIM_MEDIUM | na;
                            best_inst = cur_inst;
                            cur_tmax = best_t;
                        }
                    }
                }
            }
        }
        if constexpr (ORDERED) {
            // a draw made ahead that the reference does not make after all (the tree in between hit at exactly t_min): taken back
            if (pre_hd == pre_hd && !drew) { rng.unnext(); if (COUNT) cn.rng_draws--; }
            pre_hd = __builtin_nan("");
        }
    };
    // a boundary query (mode 1 or 2) of medium `na` has just ended; true: the second query has to run (interval set)
    auto medium_boundary_done = [&](uint32_t na, double neg_inv_density) -> bool {
        bool again = false;
        const bool sub_hit = (mode & 0x100u) != 0;
        if ((mode & 3u) == 1) {
            if (sub_hit) { // boundary.hit(r, (hit1.t + 0.0001, inf)) (src/constant_medium.rs:36-38)
                med_t1 = cur_tmax;
                mode = 2;
                cur_tmin = med_t1 + 0.0001;
                cur_tmax = INF;
                again = true;
            }
        } else if (sub_hit) { // src/constant_medium.rs:40-61
            double h1 = __builtin_fmax(med_t1, 0.001);
            const double h2 = __builtin_fmin(cur_tmax, best_t);
            if (h1 < h2) {
                h1 = __builtin_fmax(h1, 0.0);
                const double ray_length = __builtin_sqrt(len2(d));
                const double distance_inside_boundary = (h2 - h1) * ray_length;
                if (COUNT) cn.rng_draws++;
                const double hit_distance = neg_inv_density * rt_log(rng.random());
                if (hit_distance <= distance_inside_boundary) {
                    best_t = h1 + hit_distance / ray_length;
                    best_prim = PRIM_MEDIUM | na;
                    best_inst = cur_inst;
                }
            }
        }
        if (!again) {
            mode = 0;
            cur_tmin = 0.001;
            cur_tmax = best_t;
        }
        return again;
    };
    // (scenes with media) take the next steps of the world frame's sequence (rt_layout.h OSeq) until one needs a walk:
    auto seq_advance = [&]() {
        if constexpr (ORDERED && HAS_MEDIA) {
            // skip the steps whose box the ray cannot reach within (0.001, closest so far): a tree there holds
            // nothing closer, a medium there draws nothing (src/constant_medium.rs:40-44: t1 >= t2)
            stage = ST_SHADE;
            const bool pending = pre_hd == pre_hd; // (the next step is the medium whose draw was made ahead: never skipped)
            if (pending) cur_tmax = best_t;        // (the tree in front of it was walked to the draw's candidate only)
            bool first_step = true;
            while (seq_pc < P.n_oseq) {
                const OSeq *rec = &seq_tab[seq_pc];
                seq_pc++;
                float enter;
                bool miss0, miss1;
                box_pair_f32(opair_of_box(rec->box, r32), r32, f32_below(cur_tmin), f32_above(cur_tmax), miss0, miss1, enter, enter);
                if (miss0 && !(pending && first_step)) continue;
                first_step = false;
                if (rec->kind == OSEQ_TREE) {
                    node = rec->a | W_FULL; sp = 0; stage = ST_BOX;
                } else if (rec->kind == OSEQ_MEDIUM_SPHERE) {
                    medium_sphere_hit(rec->a, ld3(rec->center), ld3(rec->center_vec), rec->moving != 0, rec->radius, rec->neg_inv_density);
                    // (it may have lowered cur_tmax: the following steps are tested against that)
                    continue;
                } else { // boundary.hit(r, UNIVERSE) (src/constant_medium.rs:35)
                    if (COUNT) cn.medium_visits++;
                    mode = 1;
                    cur_tmin = -INF;
                    cur_tmax = INF;
                    node = rec->b | W_FULL; sp = 0; stage = ST_BOX;
                }
                break;
            }
        }
    };
    // ---- one record of the ordered walk (a box round's work for one lane) ----
    auto visit_record = [&]() {
        if constexpr (ORDERED) {
            // one record: both children's boxes; walk the nearer one, set the other aside
            const uint32_t nid = node & NODE_INDEX;
            const OPair nd = load_opair<LDS>(P, lds_raw, nid, r32.offx, r32.offy, r32.offz);
            if (COUNT) cn.node_visits++;
            float e0, e1;
            bool m0, m1;
            box_pair_f32(nd, r32, tmin32, tmax32, m0, m1, e0, e1);
            bool h0 = !m0, h1 = !m1;
            h0 = h0 & ((node & SKIP_CHILD0) == 0u) & (nd.c0 < (OK_EMPTY << OREF_KIND_SHIFT));
            h1 = h1 & ((node & SKIP_CHILD1) == 0u) & (nd.c1 < (OK_EMPTY << OREF_KIND_SHIFT));
            if constexpr (DEFER) {
                // (rare, so behind a branch the whole wave takes or skips: two instructions where no lane sees an instance)
                const uint32_t c_max = nd.c0 > nd.c1 ? nd.c0 : nd.c1;
                if (defer && __ballot(c_max >= (OK_INSTANCE << OREF_KIND_SHIFT)) != 0ull) {
                    const bool world = cur_inst < 0;
                    if (world && h0 && (nd.c0 >> OREF_KIND_SHIFT) == OK_INSTANCE) { deferred |= 1u << (nd.c0 & 31u); h0 = false; }
                    if (world && h1 && (nd.c1 >> OREF_KIND_SHIFT) == OK_INSTANCE) { deferred |= 1u << (nd.c1 & 31u); h1 = false; }
                }
            }
            const bool one_first = h1 && (!h0 || e1 < e0);
            if (h0 && h1) {
                const uint32_t far_ref = one_first ? nd.c0 : nd.c1;
                const uint32_t entry = far_ref < (1u << OREF_KIND_SHIFT) ? far_ref : (nid | (one_first ? SKIP_CHILD1 : SKIP_CHILD0));
                stack[sp * THREADS] = (StackT)entry;
                sp++;
            }
            o_next(h0 || h1, one_first ? nd.c1 : nd.c0);
        }
    };
    // ---- one WIDE record (rt_layout.h ONode4): four boxes at once, on with the nearest child that is hit; if others are hit too the
    // record itself is set aside with the mask of those — they are looked at again when its turn comes, against the interval as it has
    // shrunk by then (a child missed now is missed then: it leaves the mask for good)
    auto visit_wide = [&](bool any_degenerate) { // (any_degenerate: wave-uniform, some lane's ray is — rays do not change inside the box loop)
        if constexpr (ORDERED && WIDE) {
            const uint32_t nid = node & W_INDEX;
            const uint32_t todo = (node >> W_SHIFT) & 0xfu;
            const OQuad nd = load_oquad<LDS>(P, lds_raw, nid, r32.offx, r32.offy, r32.offz);
            if (COUNT) cn.node_visits++;
            float en[4], le[4];
            box_quad_f32(nd, r32, tmin32, tmax32, en, le);
            // Which children are entered, and which of them first — in integer arithmetic on the floats' bits (a compare-and-select per
            // child costs two instructions and the wait states between them; these cost one each).  A box is missed iff leave - enter is
            // negative (no NaN can arise here: planes and ray constants are finite, an empty slot gives -inf); the sign, spread over the
            // word, turns the child's key — where the ray enters it — into an all-ones NaN, which the minimum ignores and nothing equals;
            // so does the complement of the entry's mask for a child that is no longer to be looked at.
            uint32_t missed[4], key[4];
            uint32_t miss_bits = 0;
        #pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                missed[k] = (uint32_t)((int32_t)__float_as_uint(le[k] - en[k]) >> 31);
                miss_bits |= missed[k] & (1u << k);
                key[k] = __float_as_uint(en[k]) | missed[k] | (uint32_t)((int32_t)(~todo << (31u - k)) >> 31);
            }
            uint32_t hit = ~miss_bits & todo;
            if (any_degenerate) { // a ray with a zero or infinite direction component: every box that exists is entered
                if (r32.degenerate) {
                    hit = 0;
        #pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) hit |= nd.c[k] < (OK_EMPTY << OREF_KIND_SHIFT) ? (1u << k) : 0u;
                    hit &= todo;
        #pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) key[k] = (hit >> k & 1u) ? 0u : 0xffffffffu; // (in any order)
                }
            }
            if constexpr (DEFER) {
                // the world frame's instances are noted, not entered (see `deferred`); rare, so behind a wave-uniform branch
                const uint32_t c01 = nd.c[0] > nd.c[1] ? nd.c[0] : nd.c[1], c23 = nd.c[2] > nd.c[3] ? nd.c[2] : nd.c[3];
                if (defer && __ballot((c01 > c23 ? c01 : c23) >= (OK_INSTANCE << OREF_KIND_SHIFT)) != 0ull) {
                    if (cur_inst < 0) {
        #pragma unroll
                        for (uint32_t k = 0; k < 4u; ++k)
                            if ((hit >> k & 1u) && (nd.c[k] >> OREF_KIND_SHIFT) == OK_INSTANCE) { deferred |= 1u << (nd.c[k] & 31u); hit &= ~(1u << k); key[k] = 0xffffffffu; }
                    }
                }
            }
            // the nearest of the children that are entered (v_min3 / v_min return the operand that is not a NaN)
            float m012, nearest;
            asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m012) : "v"(__uint_as_float(key[0])), "v"(__uint_as_float(key[1])), "v"(__uint_as_float(key[2])));
            asm("v_min_f32 %0, %1, %2" : "=v"(nearest) : "v"(m012), "v"(__uint_as_float(key[3])));
            uint32_t t = 3u, ref = nd.c[3];
            if (__uint_as_float(key[2]) == nearest) { t = 2u; ref = nd.c[2]; }
            if (__uint_as_float(key[1]) == nearest) { t = 1u; ref = nd.c[1]; }
            if (__uint_as_float(key[0]) == nearest) { t = 0u; ref = nd.c[0]; }
            const uint32_t rest = hit & ~(1u << t);
            if (rest != 0u) {
                stack[sp * THREADS] = (StackT)(nid | (rest << W_SHIFT));
                sp++;
            }
            o_next(hit != 0u, ref);
        }
    };
    // COUNT only: per stage, rounds run / lanes active in them / shader cycles spent (wave-level, kept by lane 0)
    // slots 0-5: the stages; 6-7: parts of the shade stage (hit rebuild up to the material's first draw | unit-sphere
    // rejection sampling); the rest of a shade round (material evaluation, query start) stays in slot 4
    // The accumulators live in the LDS (one row of PROF_SLOTS x 3 per wave, after the stacks): in registers they would be
    // indexed dynamically and pushed to scratch, which distorts the very timings they record.
    unsigned long long *const prof = reinterpret_cast<unsigned long long *>(lds_raw + P.lds_prof_off) + (threadIdx.x >> 6) * (PROF_SLOTS * 3u);
    if (COUNT) {
        if (lane < PROF_SLOTS * 3u) prof[lane] = 0;
    }
    auto prof_add = [&](uint32_t slot, uint32_t what, unsigned long long v) { // what: 0 rounds, 1 lanes, 2 cycles
        if (lane == 0) prof[slot * 3u + what] += v;
    };
#define PROF_MARK(slot)                                                                                       \
    do {                                                                                                      \
        if (COUNT) {                                                                                          \
            const unsigned long long t_mark = __builtin_amdgcn_s_memtime();                                   \
            prof_add(slot, 2u, t_mark - t_prev);                                                              \
            t_prev = t_mark;                                                                                  \
        }                                                                                                     \
    } while (0)
    unsigned long long t_prev = COUNT ? __builtin_amdgcn_s_memtime() : 0;
    uint32_t prev_run = ST_NEWJOB;
    uint32_t slow_waited = 0; // (wave-uniform) shade rounds in a row that left lanes with a dear texture waiting

    for (;;) {
        if (COUNT) {
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();
            prof_add(prev_run, 2u, t_now - t_prev);
            t_prev = t_now;
        }
        // ---------------- scheduler: which stage has enough lanes queued? ----------------
        if constexpr (DEFER) {
            // the world's tree is done, but the walk met instances: the next of them
            const bool tree_done = stage == ST_SHADE;
            if (__ballot(tree_done && deferred != 0u) != 0ull) {
                if (tree_done && deferred != 0u) {
                    node = NODE_DEFERRED | (uint32_t)__builtin_ctz(deferred);
                    deferred &= deferred - 1u;
                    stage = ST_OTHER;
                }
            }
        }
        const uint32_t c_box = (uint32_t)__popcll(__ballot(stage == ST_BOX));
        const uint32_t c_sph = HAS_SPHERES ? (uint32_t)__popcll(__ballot(stage == ST_SPHERE)) : 0u;
        const uint32_t c_quad = HAS_QUADS ? (uint32_t)__popcll(__ballot(stage == ST_QUAD)) : 0u;
        const uint32_t c_oth = HAS_OTHER ? (uint32_t)__popcll(__ballot(stage == ST_OTHER)) : 0u;
        // a finished query that hit nothing needs no shading: the path ends on the background
        if (stage == ST_SHADE && best_prim == PRIM_NONE) stage = ST_NEWJOB + TERM_BACKGROUND;
        // th_new == 0: no separate path-end rounds — every shade round ends with the path-end block (for its own lanes that
        // just finished and any that were waiting), and the two queues count as one (measured better on final_scene)
        // (the five thresholds travel in ONE scalar register — th_pack: prim | other << 8 | shade << 16 | box << 24, th_new apart: as five
        // kernel arguments one of them was re-read from the argument segment, with a wait, in every pass through here)
        const uint32_t th_pack = P.th_pack;
        const uint32_t th_prim = th_pack & 0xffu, th_other = (th_pack >> 8) & 0xffu, th_shade = (th_pack >> 16) & 0xffu, th_box = th_pack >> 24;
        const bool merged = P.th_new == 0; // (the every-feature presets keep it merged: measured best on final_scene, tools/tune.py)
        const uint32_t n_shade = (uint32_t)__popcll(__ballot(stage == ST_SHADE));
        const uint32_t n_new = (uint32_t)__popcll(__ballot(stage - ST_NEWJOB < 4u));
        const uint32_t c_shade = merged ? n_shade + n_new : n_shade;
        const uint32_t c_new = merged ? 0u : n_new;
        const uint32_t live = c_box + c_sph + c_quad + c_oth + n_shade + n_new;
        if (live == 0) break;
        uint32_t run = ST_BOX, best_c = 0;
        // a deferred stage becomes runnable once its queue holds th/64 of the live lanes ...
        if (c_sph * 64u >= th_prim * live && c_sph > best_c) { run = ST_SPHERE; best_c = c_sph; }
        if (c_quad * 64u >= th_prim * live && c_quad > best_c) { run = ST_QUAD; best_c = c_quad; }
        if (c_oth * 64u >= th_other * live && c_oth > best_c) { run = ST_OTHER; best_c = c_oth; }
        if (c_shade * 64u >= th_shade * live && c_shade > best_c) { run = ST_SHADE; best_c = c_shade; }
        if (c_new * 64u >= P.th_new * live && c_new > best_c) { run = ST_NEWJOB; best_c = c_new; }
        if (best_c == 0 && c_box == 0) { // ... or when nothing else can run
            run = ST_SPHERE; best_c = c_sph;
            if (c_quad > best_c) { run = ST_QUAD; best_c = c_quad; }
            if (c_oth > best_c) { run = ST_OTHER; best_c = c_oth; }
            if (c_shade > best_c) { run = ST_SHADE; best_c = c_shade; }
            if (c_new > best_c) { run = ST_NEWJOB; best_c = c_new; }
        }

        if (COUNT) {
            prev_run = run;
            prof_add(run, 0u, 1);
            prof_add(run, 1u, run == ST_BOX ? c_box : run == ST_SPHERE ? c_sph : run == ST_QUAD ? c_quad : run == ST_OTHER ? c_oth : run == ST_SHADE ? c_shade : c_new);
        }
        if (run == ST_BOX) {
            // ---------------- box test + dispatch on the record kind ----------------
            // stays in this loop (one ballot per round) while enough of the wave's live lanes are walking boxes
            uint32_t in_box;
            const bool any_degenerate = ORDERED && WIDE && __ballot(r32.degenerate) != 0ull;
            do {
                if constexpr (ORDERED) {
                    if (stage == ST_BOX) { if constexpr (WIDE) visit_wide(any_degenerate); else visit_record(); }
                } else {
                    if (stage == ST_BOX) {
                        const NodeData nd = load_node<LDS>(P, lds_raw, node);
                        if (COUNT) cn.node_visits += (nd.packed & N32_NO_BBOX) ? 0u : 1u;
                        const bool miss = box_miss_f32(nd.lo, nd.hi, r32, tmin32, tmax32);
                        // dispatch on the record kind, branch-free: INNER (0) walks on to the next record, a leaf (1, 2)
                        // queues for its primitive stage and will continue at `skip`, anything else (>= 3) queues for ST_OTHER
                        const uint32_t kind = nd.packed & N32_KIND_MASK;
                        const uint32_t a_field = nd.packed >> N32_A_SHIFT;
                        const bool is_leaf = kind == NK_SPHERES || kind == NK_QUADS;
                        if (!miss && is_leaf) {
                            prim_cur = a_field;
                            prim_end = a_field + ((nd.packed >> N32_COUNT_SHIFT) & N32_COUNT_MASK);
                        }
                        // NodeKind INNER / SPHERES / QUADS = 0 / 1 / 2 = Stage ST_BOX / ST_SPHERE / ST_QUAD
                        const uint32_t hit_stage = (!HAS_OTHER || kind < 3u) ? kind : (uint32_t)ST_OTHER;
                        const uint32_t hit_node = kind == NK_INNER ? node + 1u : (is_leaf ? nd.skip : node);
                        node = miss ? nd.skip : hit_node;
                        stage = miss ? (uint32_t)ST_BOX : hit_stage;
                        if (stage == ST_BOX && node >= n_nodes) stage = ST_SHADE;
                    }
                }
                in_box = (uint32_t)__popcll(__ballot(stage == ST_BOX));
                if (COUNT && in_box * 64u >= th_box * live && in_box > 0) { prof_add(ST_BOX, 0u, 1); prof_add(ST_BOX, 1u, in_box); }
            } while (in_box * 64u >= th_box * live && in_box > 0);
        } else if (HAS_SPHERES && run == ST_SPHERE) {
            // ---------------- Sphere::hit (src/sphere.rs:58-83), one sphere per round ----------------
            if (stage == ST_SPHERE) {
                if (COUNT) cn.sphere_tests++;
                const uint32_t q = prim_cur;
                const Sphere *s = &sphere_tab[q];
                V3 center = ld3(s->center);
                if ((s->seq_moving & 1u)) center = center + ld3(s->center_vec) * time;
                const V3 oc = o - center;
                const double half_b = dot(oc, d);
                const double c = len2(oc) - s->radius * s->radius;
                const double discriminant = half_b * half_b - a * c;
                if (!(discriminant < 0.0)) {
                    const double sqrtd = __builtin_sqrt(discriminant);
                    // Interval::surrounds (src/interval.rs:44-46); the ordered walk also settles ties (see wins_tie)
                    auto inside = [&](double root) {
                        if (cur_tmin < root && root < cur_tmax) return true;
                        if constexpr (ORDERED)
                            return cur_tmin < root && root == (HAS_MEDIA ? best_t : cur_tmax) && best_prim != PRIM_NONE && (!HAS_MEDIA || (mode & 3u) == 0) &&
                                   wins_tie(s->seq_moving >> 1, false); // (a boundary query only wants t: ties are moot)
                        return false;
                    };
                    double root = (-half_b - sqrtd) / a;
                    bool ok = inside(root);
                    if (!ok) {
                        root = (-half_b + sqrtd) / a;
                        ok = inside(root);
                    }
                    if (ok) {
                        cur_tmax = root;
                        tmax32 = ORDERED ? f32_above(root) : (float)root;
                        if (!HAS_MEDIA || (mode & 3u) == 0) { if (HAS_MEDIA) best_t = root; best_prim = PRIM_SPHERE | q; best_inst = cur_inst; }
                        else mode |= 0x100u;
                    }
                }
                prim_cur = q + 1;
                if (prim_cur >= prim_end) {
                    if constexpr (ORDERED) o_next(false, 0u);
                    else stage = node >= n_nodes ? ST_SHADE : ST_BOX;
                }
            }
        } else if (HAS_QUADS && run == ST_QUAD) {
            // ---------------- Quad::hit (src/quad.rs:96-127): all quads of the leaf (HittableList order) ----------------
            if (stage == ST_QUAD) {
                auto quad_hit = [&](uint32_t q, bool inside_known) { // inside_known: the filter has shown alpha, beta in [0, 1] for this t
                    if (COUNT) cn.quad_tests++;
                    const Quad *qd = &quad_tab[q];
                    const V3 normal = ld3(qd->normal);
                    const double denom = dot(normal, d);
                    if (__builtin_fabs(denom) < 1e-8) return;
                    const double t = (qd->d - dot(normal, o)) / denom;
                    if (!(cur_tmin <= t && t <= cur_tmax)) return; // Interval::contains (src/interval.rs:40-42)
                    if constexpr (ORDERED) // the ordered walk settles ties explicitly (see wins_tie)
                        if (t == (HAS_MEDIA ? best_t : cur_tmax) && best_prim != PRIM_NONE && (!HAS_MEDIA || (mode & 3u) == 0) && !wins_tie(qd->seq, true)) return;
                    if (!inside_known) { // (behind a branch the wave skips when every lane's survivor is a certain one — nearly always)
                        const V3 intersection = o + d * t;
                        const V3 php = intersection - ld3(qd->q);
                        const V3 qw = ld3(qd->w);
                        const double alpha = dot(qw, cross(php, ld3(qd->v)));
                        const double beta = dot(qw, cross(ld3(qd->u), php));
                        if (alpha < 0.0 || alpha > 1.0 || beta < 0.0 || beta > 1.0) return;
                    }
                    cur_tmax = t;
                    tmax32 = ORDERED ? f32_above(t) : (float)t;
                    if (!HAS_MEDIA || (mode & 3u) == 0) { if (HAS_MEDIA) best_t = t; best_prim = PRIM_QUAD | q; best_inst = cur_inst; }
                    else mode |= 0x100u;
                };
                bool filtered = false;
                if constexpr (QFILT) {
                    // A leaf of several quads (flat leaves: Cornell's walls, the faces of a box): the conservative f32 filter looks at
                    // all of them, two at a time, against the interval as it is now, and the exact test runs for the lane's own
                    // survivors only, in list order (the wave iterates as often as its worst lane has survivors: once or twice, not six
                    // times).  A quad the filter drops is one the exact test would reject whatever the interval has shrunk to by then.
                    if (P.lds_off_qfilt != 0xffffffffu) { // (wave-uniform)
                        filtered = true;
                        const uint32_t count = prim_end - prim_cur;
                        uint32_t keep = 1u; // bits 0-7: survivors; bits 8-15: of them, those whose alpha and beta are known to be inside
                        if (count > 1u) {
                            const QRay32 qr = make_qray32(o, d);
                            const QFiltPair *rec = reinterpret_cast<const QFiltPair *>(lds_raw + P.lds_off_qfilt) + prim_cur;
                            keep = 0u;
#pragma unroll 1
                            for (uint32_t k = 0; k < count; k += 2u) keep |= quad_pair_keep(rec + k, qr, tmin32, tmax32) << k;
                            keep &= ((1u << count) - 1u) * 0x101u;
                        }
                        while ((keep & 0xffu) != 0u) {
                            const uint32_t k = (uint32_t)__builtin_ctz(keep);
                            const bool inside_known = (keep >> (8u + k) & 1u) != 0u;
                            keep &= keep - 1u;
                            quad_hit(prim_cur + k, inside_known);
                        }
                    }
                }
                if (!filtered)
                    for (uint32_t q = prim_cur; q < prim_end; ++q) quad_hit(q, false);
                prim_cur = prim_end;
                if constexpr (ORDERED) o_next(false, 0u);
                else stage = node >= n_nodes ? ST_SHADE : ST_BOX;
            }
        } else if (HAS_OTHER && run == ST_OTHER) {
            // ---------------- frame changes and ConstantMedium steps ----------------
            if constexpr (ORDERED) {
                if (HAS_MEDIA && stage == ST_OTHER && node == NODE_SEQ_NEXT) {
                    // ---- the world frame's sequence (rt_layout.h OSeq): a tree or a boundary query has ended ----
                    bool again = false;
                    if ((mode & 3u) != 0) { // a boundary query of the medium at the previous step
                        const OSeq *rec = &seq_tab[seq_pc - 1u];
                        again = medium_boundary_done(rec->a, rec->neg_inv_density);
                        if (again) { node = rec->b | W_FULL; sp = 0; stage = ST_BOX; }
                    }
                    if (!again) seq_advance();
                    refresh_interval32();
                } else
                if (stage == ST_OTHER) { // enter the frame of instance `node`, or leave the current one
                    const bool leaving = node == NODE_FRAME_EXIT;
                    if (leaving) {
                        cur_inst = inst_tab[cur_inst].parent;
                        restore_world_ray(o, d);
                        ray_to_frame(inst_tab, cur_inst, o, d);
                    } else {
                        if (COUNT) cn.instance_enters++;
                        const bool from_world = DEFER && (node & NODE_DEFERRED) != 0u; // a deferred instance: entered from the world frame,
                        node &= ~NODE_DEFERRED;                               // ... never left (no S_EXIT entry)
                        if (cur_inst < 0) park_world_ray(o, d); // leaving the world frame
                        else if (from_world) restore_world_ray(o, d); // ... from the deferred instance walked before this one
                        apply_instance(inst_tab[node], o, d);
                        cur_inst = (int32_t)node;
                        if (!from_world) {
                            stack[sp * THREADS] = (StackT)S_EXIT;
                            sp++;
                        }
                    }
                    // (a frame whose tree is one leaf — a box's faces: straight to the leaf's primitives; the frame's box has been tested
                    // where the walk met the instance, and what ends the leaf pops the stack as any leaf does.  Such a walk tests no box
                    // inside the frame — its next box test, if any, comes after the next frame change — so the ray's f32 copy is not
                    // rebuilt for it.)
                    // (LDS-resident scenes without media only: in the kernel that gathers final_scene from global memory — whose one instance
                    // holds a tree of a thousand spheres — the mere presence of this branch cost 3 %, and cornell_smoke's kernel, where it
                    // applies, lost 2.8 % with it)
                    const uint32_t start = (LDS == 3 && !HAS_MEDIA && !leaving && P.inst_shortcut != 0u) ? inst_tab[cur_inst].start_ref : 0u;
                    if (start == 0u) refresh_ray32();
                    a = len2(d);
                    // (... but the box that was tested there is the frame's box in the PARENT's coordinates — around a rotated box it is a
                    // fifth wider than the box —, and against the interval of that moment: the leaf's own box, in the frame's coordinates,
                    // is tested here, and a ray that misses it goes on as if the leaf had been looked at)
                    bool nothing = false;
                    if (start != 0u) {
                        const float *sb = inst_tab[cur_inst].start_box;
                        const float lo[3] = {sb[0], sb[2], sb[4]}, hi[3] = {sb[1], sb[3], sb[5]};
                        nothing = box_miss_f32(lo, hi, make_ray32(o, d), tmin32, tmax32);
                        if (COUNT && nothing) cn.instance_enters--; // (the counter: frames entered AND looked at)
                    }
                    if (leaving || nothing) o_next(false, 0u);
                    else {
                        node = inst_tab[cur_inst].root | W_FULL;
                        stage = ST_BOX;
                        if (start != 0u) {
                            stage = start >> OREF_KIND_SHIFT; // (OrderedKind SPHERES / QUADS = Stage ST_SPHERE / ST_QUAD)
                            prim_cur = start & OREF_INDEX_MASK;
                            prim_end = prim_cur + ((start >> OREF_COUNT_SHIFT) & OREF_COUNT_MASK) + 1u;
                        }
                    }
                }
            } else
            if (stage == ST_OTHER) {
                const NodeData nd = load_node<LDS>(P, lds_raw, node);
                const uint32_t kind = nd.packed & N32_KIND_MASK;
                const uint32_t na = nd.packed >> N32_A_SHIFT;
                if (HAS_FRAMES && kind == NK_INST_ENTER) {
                    if (COUNT) cn.instance_enters++;
                    if (cur_inst < 0) park_world_ray(o, d); // leaving the world frame
                    apply_instance(inst_tab[na], o, d);
                    cur_inst = (int32_t)na;
                    node = node + 1;
                } else if (HAS_FRAMES && kind == NK_INST_EXIT) {
                    cur_inst = inst_tab[na].parent;
                    restore_world_ray(o, d);
                    ray_to_frame(inst_tab, cur_inst, o, d);
                    node = node + 1;
                } else if (HAS_MEDIA && kind == NK_MEDIUM_ENTER) { // boundary.hit(r, UNIVERSE) (src/constant_medium.rs:35)
                    if (COUNT) cn.medium_visits++;
                    mode = 1;
                    cur_tmin = -INF;
                    cur_tmax = INF;
                    node = node + 1;
                } else if (HAS_MEDIA && kind == NK_MEDIUM_SPHERE) {
                    const Medium md = media_tab[na];
                    const Sphere *s = &sphere_tab[md.first_node];
                    medium_sphere_hit(na, ld3(s->center), ld3(s->center_vec), (s->seq_moving & 1u) != 0, s->radius, md.neg_inv_density);
                    node = nd.skip;
                } else if (HAS_MEDIA) { // NK_MEDIUM_EXIT
                    if (medium_boundary_done(na, media_tab[na].neg_inv_density)) node = media_tab[na].first_node;
                    else node = node + 1;
                }
                if (kind == NK_INST_ENTER || kind == NK_INST_EXIT) {
                    refresh_ray32();
                    a = len2(d);
                } else {
                    refresh_interval32();
                }
                stage = node >= n_nodes ? ST_SHADE : ST_BOX;
            }
        }
        bool start_query = false;
        if (run == ST_SHADE) {
            // ---------------- shade a closest hit: ray_color (src/renderer.rs:139-155), one level of the recursion per visit ----
            // In this codebase a material that scatters emits nothing and the one that emits never scatters, so the recursion
            // unrolls to  A_1 * (A_2 * ( ... (A_n * terminal)))  evaluated innermost first; attenuations are parked in att_stack
            // and multiplied back when the path ends.  (A query that hit nothing went straight to ST_NEWJOB.)
            if (stage == ST_SHADE) {
                V3 result = v3(0.0, 0.0, 0.0);
                bool path_done = false;
                // rebuild the HitRecord in its own frame, then carry it to the world
                if constexpr (DEFER) {
                    if (cur_inst >= 0) { restore_world_ray(o, d); cur_inst = -1; } // the query ended inside a deferred instance
                }
                V3 lo = o, ld = d; // all frames are closed at this point: (o, d) is the world ray
                if (HAS_FRAMES) ray_to_frame(inst_tab, best_inst, lo, ld);
                // (without media the interval's upper end IS the closest hit's t: one value less to keep per lane)
                V3 p = lo + ld * (HAS_MEDIA ? best_t : cur_tmax); // Ray::at (src/ray.rs:30-32)
                V3 outward_normal;
                uint32_t mat;
                double u = 0.0, v = 0.0;
                const uint32_t pk = best_prim & PRIM_KIND_MASK, pi = best_prim & PRIM_INDEX_MASK;
                bool uv_from_sphere = false;
                if (HAS_SPHERES && (pk == PRIM_SPHERE || (!HAS_QUADS && !HAS_MEDIA))) { // src/sphere.rs:85-88
                    const Sphere *s = &sphere_tab[pi];
                    V3 center = ld3(s->center);
                    if ((s->seq_moving & 1u)) center = center + ld3(s->center_vec) * time;
                    outward_normal = div(p - center, s->radius);
                    mat = s->material;
                    uv_from_sphere = true;
                } else if (HAS_QUADS && (pk == PRIM_QUAD || !HAS_MEDIA)) { // src/quad.rs:118-132
                    const Quad *qd = &quad_tab[pi];
                    outward_normal = ld3(qd->normal);
                    mat = qd->material;
                    if (HAS_TEXTURES && mats_tab[mat].needs_uv) {
                        const V3 php = p - ld3(qd->q);
                        const V3 qw = ld3(qd->w);
                        u = dot(qw, cross(php, ld3(qd->v)));
                        v = dot(qw, cross(ld3(qd->u), php));
                    }
                } else { // ConstantMedium: normal := r.direction (src/constant_medium.rs:52-58)
                    outward_normal = ld;
                    mat = media_tab[pi].phase_material;
                }
                const DMaterial *m = &mats_tab[mat];
                // A hit on a noise texture is rare (final_scene: 0.065 per sample) and dear (Perlin turbulence: seven noise evaluations,
                // ~1500 instructions), and one such lane makes the whole wave run that code: nearly half of final_scene's shade rounds
                // did.  Such lanes wait — nothing of theirs has been touched yet, they stay queued for the stage — until slow_min of them
                // are here, or slow_age shade rounds have passed, or nothing else is: final_scene +2.3 % (4 lanes / 32 rounds; image
                // lookups, 0.076 per sample and far cheaper, lose by waiting).
                bool postpone = false;
                if constexpr (HAS_TEXTURES) {
                    const bool slow = m->slow != 0u;
                    const uint32_t n_slow = (uint32_t)__popcll(__ballot(slow));
                    if (n_slow != 0u) {
                        const bool go = n_slow >= P.slow_min || slow_waited >= P.slow_age || n_slow == (uint32_t)__popcll(__ballot(true));
                        slow_waited = go ? 0u : slow_waited + 1u;
                        postpone = slow && !go;
                    }
                }
                if (!postpone) {
                if (HAS_TEXTURES && uv_from_sphere && m->needs_uv) { // get_sphere_uv (src/sphere.rs:48-52), from the outward normal
                    const double PI = 3.14159265358979323846264338327950288;
                    const double theta = rt_acos(-outward_normal.y);
                    const double phi = rt_atan2(-outward_normal.z, outward_normal.x) + PI;
                    u = phi / (2.0 * PI);
                    v = theta / PI;
                }
                // HitRecord::new (src/hittable.rs:22-37)
                const bool front_face = dot(ld, outward_normal) < 0.0;
                V3 normal = front_face ? outward_normal : -outward_normal;
                if (HAS_FRAMES) hit_to_world(inst_tab, best_inst, p, normal);

                const uint32_t mk = m->kind;
                // Every material that reads a texture reads exactly one, after its random draws (textures draw
                // nothing): evaluate it at one place.  Likewise the unit-sphere rejection sample
                // (src/vec3.rs:54-61) is the first draw of Lambertian, Metal and Isotropic alike.
                V3 tex = v3(1.0, 1.0, 1.0);
                V3 rs = v3(0.0, 0.0, 0.0);
                PROF_MARK(6);
                if (mk != RT_MATERIAL_DIELECTRIC && mk != RT_MATERIAL_DIFFUSE_LIGHT) rs = random_in_unit_sphere<COUNT>(rng, cn);
                PROF_MARK(7);
                if (mk != RT_MATERIAL_DIELECTRIC && mk != RT_MATERIAL_METAL) {
                    if constexpr (HAS_TEXTURES) tex = m->solid ? ld3(m->albedo) : texture_value<COUNT>(P, texs_tab, perlin_tab, m->texture, u, v, p, cn);
                    else tex = ld3(m->albedo); // every texture is a SolidColor (src/texture.rs:32-36): the colour was copied here
                }
                PROF_MARK(8);
                V3 attenuation = tex;
                // what gets parked: the material's index where its attenuation is a constant of the material, else the colour
                const uint32_t park_id = (!HAS_TEXTURES || (P.ids_ok != 0u && (mk == RT_MATERIAL_METAL || m->solid != 0u))) ? mat : ID_COLOUR;
                bool unit_attenuation = false;
                V3 new_dir = normal;
                // Metal and Dielectric both start from the unit direction: one square root and one division for the two branches of a round
                V3 unit_d = d;
                if (mk == RT_MATERIAL_METAL || mk == RT_MATERIAL_DIELECTRIC) unit_d = normalize(d);
                if (mk == RT_MATERIAL_DIFFUSE_LIGHT) { // emitted, no scatter (src/material.rs:114-122)
                    result = tex;
                    path_done = true;
                } else if (mk == RT_MATERIAL_LAMBERTIAN) { // src/material.rs:26-42
                    const V3 scatter_direction = normal + normalize(rs);
                    new_dir = near_zero(scatter_direction) ? normal : scatter_direction;
                } else if (mk == RT_MATERIAL_METAL) { // src/material.rs:53-64
                    const V3 refl = reflect(unit_d, normal);
                    const V3 reflected = refl + rs * m->fuzz;
                    if (!(dot(reflected, normal) > 0.0)) path_done = true; // absorbed: emission (zero) only
                    new_dir = reflected;
                    attenuation = ld3(m->albedo);
                } else if (mk == RT_MATERIAL_DIELECTRIC) { // src/material.rs:80-104
                    const double refraction_ratio = front_face ? m->albedo[0] : m->ir; // (1.0 / ir and both r0 were divided out when the table was built)
                    const V3 unit_direction = unit_d;
                    const double cos_theta = __builtin_fmin(dot(-unit_direction, normal), 1.0);
                    const double sin_theta = __builtin_sqrt(1.0 - cos_theta * cos_theta);
                    bool do_reflect = refraction_ratio * sin_theta > 1.0;
                    if (!do_reflect) { // `||` short-circuit: draw only when refraction is possible
                        const double r0 = front_face ? m->albedo[1] : m->albedo[2]; // ((1 - ratio) / (1 + ratio))^2
                        const double reflectance = r0 + (1.0 - r0) * rt_pow5(1.0 - cos_theta);
                        if (COUNT) cn.rng_draws++;
                        do_reflect = reflectance > rng.random();
                    }
                    new_dir = do_reflect ? reflect(unit_direction, normal) : refract(unit_direction, normal, refraction_ratio);
                    unit_attenuation = true; // Color::ONE: multiplying by it is the identity, nothing to park
                } else { // RT_MATERIAL_ISOTROPIC, src/material.rs:132-138
                    new_dir = normalize(rs);
                }
                if (!path_done) {
                    if (!unit_attenuation) park(park_id, attenuation);
                    depth--;
                    if (depth <= 0) { // the next ray_color call returns Color::ZERO at once
                        path_done = true;
                    } else {
                        o = p;
                        d = new_dir;
                    }
                }
                PROF_MARK(9);
                if (path_done) {
                    if (result.x != 0.0 || result.y != 0.0 || result.z != 0.0) {
                        // a light: the emitted colour is parked like one more attenuation and the path ends on Color::ONE
                        // (emitted * 1.0 is emitted, bit for bit), so the chain of products is written once, in ST_NEWJOB
                        park(park_id, result);
                        stage = ST_NEWJOB + TERM_ONE;
                    } else {
                        stage = ST_NEWJOB + TERM_ZERO; // absorbed, out of depth, or a black emitter
                    }
                } else {
                    start_query = true; // of the scattered ray
                }
                } // (!postpone)
            }
        }
        if (run == ST_NEWJOB || (run == ST_SHADE && merged)) {
            // ---------------- finish a path, take the next job, Camera::get_ray ----------------
            // What this block alone reads of the parameter block — the camera (44 scalar registers' worth of doubles), the seed, the job
            // arithmetic — is fetched from the kernel-argument segment HERE, through a pointer the optimiser cannot look behind.  Read as
            // plain members of P these values are loaded once at the kernel's start and then have to sit in scalar registers through
            // every round of every other stage; there are ~100 of those registers, the spill goes to VGPR lanes, and what got evicted was
            // what the hot loops use: the every-feature kernel reloaded the record table's base address with a v_readlane per visit.
            // (SGPR spills: spheres kernel 78 -> 2, quads + frames 67 -> 8, every-feature 160 -> 46; C2 +3.0 %, C3 +3.8 %, C4 +1.1 %.)
            const KParams __attribute__((address_space(4))) *KP =
                (const KParams __attribute__((address_space(4))) *)(uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(KP));
            const bool here = stage - ST_NEWJOB < 4u;
            const uint32_t term = stage - ST_NEWJOB;
            if (here && term != TERM_STORED) {
                V3 result = v3(0.0, 0.0, 0.0);
                if (term != TERM_ZERO) {
                    result = term == TERM_BACKGROUND ? v3(KP->cam.background.x, KP->cam.background.y, KP->cam.background.z) : v3(1.0, 1.0, 1.0);
                    // the parked attenuations, last parked first.  The newest four indices sit at fixed places of the two registers:
                    // their colours are fetched together and multiplied on one after the other (a level the path does not have reads
                    // Color::ONE); older ones come from att_ids, up to CHAIN per trip.  A zero terminal stays zero (attenuations are finite).
                    if (result.x != 0.0 || result.y != 0.0 || result.z != 0.0) {
                        V3 newest[IDS_IN_REGS];
                    #pragma unroll
                        for (uint32_t k = 0; k < IDS_IN_REGS; ++k) {
                            const uint32_t id = (k & 1u) ? ids[k >> 1] >> 16 : ids[k >> 1] & 0xffffu;
                            newest[k] = parked_colour(n_att > k ? id : P.id_one, n_att > k ? n_att - 1u - k : 0u);
                        }
                    #pragma unroll
                        for (uint32_t k = 0; k < IDS_IN_REGS; ++k) result = newest[k] * result;
                        uint32_t level = n_att > IDS_IN_REGS ? n_att - IDS_IN_REGS : 0u; // levels [0, level) are in att_ids
                        while (level > 0) {
                            V3 parked[CHAIN];
                            // (the loads are unconditional — a level the path does not have reads its level 0 and drops the value —:
                            // behind `level > j ? load : id_one` each became a branch with its own wait, CHAIN memory latencies in a row)
                            uint32_t stored[CHAIN];
                    #pragma unroll
                            for (uint32_t j = 0; j < CHAIN; ++j) stored[j] = att_ids[(level > j ? level - 1u - j : 0u) * att_lanes + gtid];
                    #pragma unroll
                            for (uint32_t j = 0; j < CHAIN; ++j) {
                                const uint32_t lv = level > j ? level - 1u - j : 0u;
                                const uint32_t id = level > j ? stored[j] : P.id_one;
                                parked[j] = parked_colour(id, lv);
                            }
                    #pragma unroll
                            for (uint32_t j = 0; j < CHAIN; ++j) result = parked[j] * result;
                            level = level > CHAIN ? level - CHAIN : 0u;
                        }
                    }
                }
                double *dst = KP->samples + (size_t)job * 3u;
                __builtin_nontemporal_store(result.x, &dst[0]); __builtin_nontemporal_store(result.y, &dst[1]); __builtin_nontemporal_store(result.z, &dst[2]);
                stage = ST_NEWJOB + TERM_STORED;
            }
            PROF_MARK(10);
            // ---- hand out jobs to the lanes that need one (wave-level: ballot + prefix count) ----
            const uint64_t want = __ballot(here);
            const uint32_t n_want = (uint32_t)__popcll(want);
            if (n_want) {
                // what is left of the wave's current range goes out first; if that does not serve every asking lane, the next
                // range is reserved in the same round (a lane left without a job would idle until the next path-end round)
                const uint32_t old_next = job_next, old_avail = job_end - job_next;
                uint32_t new_base = 0, new_avail = 0;
                if (jobs_left && old_avail < n_want) {
                    // guided hand-out: the grabs shrink as the launch runs out (never more than taper x what was left when this
                    // wave last looked — its previous reservation told it —, in whole sample-rows of a tile), so that the slowest
                    // wave's last grab, the tail of the launch, is short although the early grabs are large.  Any sizes tile the job
                    // range, so a stale estimate is harmless.  (No extra look at the counter: 4096 waves reading and adding to one
                    // word already approach what one L2 line serves — MI355X_MICROARCH.md "dequeue": ~88 per microsecond.)
                    uint32_t grab = (uint32_t)((float)jobs_seen_left * KP->grab_taper) & ~63u;
                    grab = grab < KP->jobs_per_grab ? grab : KP->jobs_per_grab;
                    grab = grab < MIN_JOBS_PER_GRAB ? MIN_JOBS_PER_GRAB : grab;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(KP->job_counter, grab);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    jobs_seen_left = base + grab < KP->n_jobs ? KP->n_jobs - (base + grab) : 0u;
                    if (base >= KP->n_jobs) { jobs_left = false; }
                    else { new_base = base; new_avail = (base + grab < KP->n_jobs ? base + grab : KP->n_jobs) - base; }
                }
                const uint32_t avail = old_avail + new_avail;
                const uint32_t rank = (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
                if (here) {
                    if (rank < avail) {
                        job = rank < old_avail ? old_next + rank : new_base + (rank - old_avail);
                        // job -> (local tile, sample, pixel): ((lt * S + s_rel) * 64 + p)
                        const uint32_t p64 = job & 63u;
                        const uint32_t row = job >> 6;
                        // n / d for n < 2^27 as trunc((n + 0.5) * (1 / d)) in f64: (n + 0.5) / d is at least 0.5 / d away from an
                        // integer, far more than the 2^-52 relative error of the product — exact, and 4 instructions, not 25
                        const uint32_t lt = (uint32_t)(((double)row + 0.5) * KP->inv_n_samples), s_rel = row - lt * KP->n_samples;
                        const uint32_t k = lt * (uint32_t)KP->shard_count + (uint32_t)KP->shard_index;
                        const uint32_t tile_row = (uint32_t)(((double)k + 0.5) * KP->inv_tiles_x), tile_col = k - tile_row * (uint32_t)KP->tiles_x;
                        const int32_t i = (int32_t)tile_col * RT_TILE_W + (int32_t)(p64 & 7u);
                        const int32_t j = (int32_t)tile_row * RT_TILE_H + (int32_t)(p64 >> 3);
                        if (i < KP->cam.image_width && j < KP->cam.image_height) {
                            const uint32_t pixel = (uint32_t)j * (uint32_t)KP->cam.image_width + (uint32_t)i; // screen_pos (src/renderer.rs:32-33)
                            rng.start(KP->seed_mixed, pixel, (uint32_t)KP->sample_begin + s_rel);
                            // Camera::get_ray (src/camera.rs:112-137)
                            const rt_camera __attribute__((address_space(4))) &cam = KP->cam;
                            const V3 du = v3(cam.pixel_delta_u.x, cam.pixel_delta_u.y, cam.pixel_delta_u.z), dv = v3(cam.pixel_delta_v.x, cam.pixel_delta_v.y, cam.pixel_delta_v.z);
                            const V3 pixel_center = v3(cam.pixel00_loc.x, cam.pixel00_loc.y, cam.pixel00_loc.z) + du * (double)i + dv * (double)j;
                            const double px = -0.5 + rng.random();
                            const double py = -0.5 + rng.random();
                            if (COUNT) cn.rng_draws += 2;
                            const V3 pixel_sample = pixel_center + (du * px + dv * py);
                            V3 ro;
                            if (cam.defocus_angle <= 0.0) {
                                ro = v3(cam.center.x, cam.center.y, cam.center.z);
                            } else { // random_in_unit_disk (src/vec3.rs:77-88)
                                double dx, dy;
                                for (;;) {
                                    dx = rng.range(-1.0, 1.0);
                                    dy = rng.range(-1.0, 1.0);
                                    if (COUNT) cn.rng_draws += 2;
                                    if (dx * dx + dy * dy + 0.0 * 0.0 < 1.0) break;
                                }
                                ro = v3(cam.center.x, cam.center.y, cam.center.z) + v3(cam.defocus_disk_u.x, cam.defocus_disk_u.y, cam.defocus_disk_u.z) * dx + v3(cam.defocus_disk_v.x, cam.defocus_disk_v.y, cam.defocus_disk_v.z) * dy;
                            }
                            o = ro;
                            d = pixel_sample - ro;
                            time = rng.random();
                            if (COUNT) cn.rng_draws += 1;
                            depth = KP->max_depth;
                            n_att = 0;
                            if (COUNT) cn.samples++;
                            start_query = true; // of the camera ray
                        }
                        // a job of a padding pixel (edge tile) traces nothing; the lane asks again next round
                    } else if (!jobs_left) {
                        stage = ST_DONE;
                    }
                }
                const uint32_t taken = n_want < avail ? n_want : avail;
                if (new_avail) { job_next = new_base + (taken - old_avail); job_end = new_base + new_avail; }
                else job_next = old_next + taken;
            }
            PROF_MARK(11);
        }
        if (start_query) { // the closest-hit query of a scattered or camera ray: world.hit(r, (0.001, inf)) (src/renderer.rs:144)
            if (COUNT) cn.rays++;
            a = len2(d);
            cur_tmin = 0.001; cur_tmax = INF;
            refresh_ray32();
            refresh_interval32();
            best_t = INF; best_prim = PRIM_NONE; best_inst = -1; cur_inst = -1;
            mode = 0;
            node = first_node;
            sp = 0;
            stage = ST_BOX;
            if constexpr (ORDERED && !HAS_MEDIA) {
                // (Measured against visiting the root record right here, with its box tests: 5340 vs 5060 Msamples/s on C2.  The blind
                // test keeps the lanes that start a query together — ONE round of the primitive stage serves them all, then they all walk
                // —, and a round costs the same for 23 lanes as for 43: the 1.2 tests per sample spent on rays that miss the leaf's box are free.)
                if (P.o_start_stage != 0u) { // the root's big leaf first, its other child set aside (rt_api.cpp "start shortcut")
                    const uint32_t rest = P.o_start_rest;
                    if (WIDE ? rest != 0u : (rest >> OREF_KIND_SHIFT) != OK_EMPTY) {
                        if constexpr (WIDE) stack[0] = (StackT)((first_node & W_INDEX) | (rest << W_SHIFT)); // (rest: the mask of the root's other children)
                        else stack[0] = (StackT)(rest < (1u << OREF_KIND_SHIFT) ? rest : (first_node | (P.o_start_slot ? SKIP_CHILD1 : SKIP_CHILD0)));
                        sp = 1;
                    }
                    prim_cur = P.o_start_prim;
                    prim_end = P.o_start_end;
                    stage = P.o_start_stage;
                }
            }
            if constexpr (ORDERED && HAS_MEDIA) { // the world's sequence starts over
                seq_pc = 1; // (its first step is a tree: first_node)
                if (first_node == NODE_SEQ_NEXT) { // ... or a medium: taken here, while the lanes starting a query are together
                    seq_pc = 0;
                    seq_advance();
                    refresh_interval32();
                }
                // Look ahead, while the lanes that start a query are together: if the ray cannot reach the box of any later step
                // within the interval it has now (it only shrinks), the sequence ends with the tree that is about to be walked — and
                // the walk's end goes straight to shading, not through a round of ST_OTHER that finds nothing to do (final_scene: a
                // quarter of the sequence rounds, 0.86 per sample, did).  Short sequences only: the look-ahead is a loop over the steps.
                // If the first later step it can reach is a TREE, the walk's end goes on with that tree's root directly (SEQ_JUMP, o_next):
                // the steps in between stay out of reach — the interval only shrinks — and the tree's own box test is made good by its
                // root record's (final_scene: the rays that miss the smoke's box but not the second tree's).
                if (P.seq_lookahead && stage == ST_BOX && mode == 0 && P.n_oseq - seq_pc <= 4u) {
                    uint32_t first = P.n_oseq;
                    for (uint32_t k = P.n_oseq; k-- > seq_pc;) {
                        float enter;
                        bool miss0, miss1;
                        box_pair_f32(opair_of_box(seq_tab[k].box, r32), r32, tmin32, tmax32, miss0, miss1, enter, enter);
                        first = miss0 ? first : k;
                    }
                    if (first == P.n_oseq) seq_pc = P.n_oseq;
                    else if (seq_tab[first].kind == OSEQ_TREE) seq_pc = first | SEQ_JUMP;
                }
                // A ray that STARTS INSIDE the ball of the sphere-bounded medium that follows the tree about to be walked (a scatter inside
                // final_scene's smoke: a third of its rays) would walk that tree to infinity although the medium nearly always ends the ray
                // within a few units.  For such a ray the medium's draw happens whatever the tree finds — rec1.t = t_min < rec2.t
                // (src/constant_medium.rs:40-44) unless the tree hits at exactly t_min, and then the draw is taken back (medium_sphere_hit) —
                // and its verdict, hit_distance <= (min(exit, closest so far) - t_min) |d|, only turns from yes to no as the tree finds
                // something closer.  So the draw is made HERE and, where it lands inside the ball, the tree is walked to just beyond the
                // candidate t (a relative 2^-40: any hit further out leaves the verdict a yes whatever the roundings).  The medium's own step
                // then evaluates the reference's predicate with the tree's closest hit, which the clipped walk has found if it matters.
                if (HAS_SPHERES && P.medium_first != 0u && stage == ST_BOX && mode == 0 && seq_pc < P.n_oseq && seq_tab[seq_pc].kind == OSEQ_MEDIUM_SPHERE) {
                    const OSeq *rec = &seq_tab[seq_pc];
                    V3 center = ld3(rec->center);
                    if (rec->moving != 0) center = center + ld3(rec->center_vec) * time;
                    const V3 oc = o - center;
                    const double half_b = dot(oc, d);
                    const double c = len2(oc) - rec->radius * rec->radius;
                    const double discriminant = half_b * half_b - a * c;
                    if (!(discriminant < 0.0)) {
                        const double sqrtd = __builtin_sqrt(discriminant);
                        const double root_a = (-half_b - sqrtd) / a, root_b = (-half_b + sqrtd) / a;
                        const double lo2 = root_a + 0.0001;
                        // (exactly medium_sphere_hit's choices: t1 = root_a <= t_min, t2 = root_b, h1 = t_min)
                        if (-INF < root_a && root_a <= 0.001 && lo2 < root_b && root_b < INF && 0.001 < __builtin_fmin(root_b, best_t)) {
                            const double ray_length = __builtin_sqrt(len2(d));
                            if (COUNT) cn.rng_draws++;
                            pre_hd = rec->neg_inv_density * rt_log(rng.random());
                            med_t1 = __builtin_nan(""); // (free outside boundary queries: the candidate t, if the draw lands inside the ball)
                            if (pre_hd <= (__builtin_fmin(root_b, best_t) - 0.001) * ray_length) {
                                const double t_m = 0.001 + pre_hd / ray_length;
                                med_t1 = t_m;
                                cur_tmax = __builtin_fmin(cur_tmax, t_m + t_m * 0x1p-40);
                                refresh_interval32();
                            }
                            // where a TREE follows the medium, the medium's step is settled where the walk of the tree in front of it ends
                            // (o_next), without a round of ST_OTHER, unless that walk hit something
                            if (seq_pc + 1u < P.n_oseq && seq_tab[seq_pc + 1u].kind == OSEQ_TREE) seq_pc = (seq_pc + 1u) | SEQ_JUMP;
                        }
                    }
                }
            }
        }
    }

    if (COUNT && P.counters) {
        const uint32_t vals[10] = {cn.samples, cn.rays, cn.node_visits, cn.sphere_tests, cn.quad_tests,
                                   cn.medium_visits, cn.rng_draws, cn.noise_evals, cn.image_lookups,
                                   cn.instance_enters};
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            unsigned long long v = vals[q];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&P.counters[q], v);
        }
        if (lane == 0) {
            for (uint32_t q = 0; q < PROF_SLOTS * 3u; ++q) atomicAdd(&P.counters[10 + q], prof[q]);
        }
    }
}

// Per-pixel sum of the sample buffer in sample order (src/renderer.rs:35-40: `avg_color += new_color`).
// One thread per (local tile, pixel of the tile); consecutive threads read consecutive 24-byte samples.
__global__ void sum_samples_kernel(const KParams P) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lt = idx >> 6, p64 = idx & 63u;
    if (lt >= P.n_local_tiles) return;
    const int32_t w = P.cam.image_width, h = P.cam.image_height;
    const uint32_t k = lt * (uint32_t)P.shard_count + (uint32_t)P.shard_index;
    const int32_t i = (int32_t)(k % (uint32_t)P.tiles_x) * RT_TILE_W + (int32_t)(p64 & 7u);
    const int32_t j = (int32_t)(k / (uint32_t)P.tiles_x) * RT_TILE_H + (int32_t)(p64 >> 3);
    const bool valid = i < w && j < h;
    double *dst = nullptr;
    if (P.out_layout == RT_OUT_TILES) dst = P.out + ((size_t)lt * 64u + p64) * 3u;
    else if (valid) dst = P.out + ((size_t)j * (size_t)w + (size_t)i) * 3u;
    if (!valid) {
        if (dst) { dst[0] = 0.0; dst[1] = 0.0; dst[2] = 0.0; }
        return;
    }
    V3 acc = v3(0.0, 0.0, 0.0);
    if (P.accumulate) acc = v3(dst[0], dst[1], dst[2]);
    const double *src = P.samples + ((size_t)lt * P.n_samples * 64u + p64) * 3u;
    for (uint32_t s = 0; s < P.n_samples; ++s) {
        acc = acc + v3(src[0], src[1], src[2]);
        src += 64u * 3u;
    }
    dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z;
}

// frame-end reassembly: [shard][local tile][64][3] -> row-major frame (T = double: channel sums; uint8_t: resolved RGB8)
template <class T>
__global__ void tiles_to_frame_kernel(int32_t w, int32_t h, int32_t tiles_x, int32_t shard_count, int64_t shard_stride,
                                      const T *__restrict__ gathered, T *__restrict__ frame) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // one thread per pixel
    if (idx >= (int64_t)w * h) return;
    const int32_t i = (int32_t)(idx % w), j = (int32_t)(idx / w);
    const int64_t k = (int64_t)(j / RT_TILE_H) * tiles_x + (i / RT_TILE_W);
    const int64_t shard = k % shard_count, lt = k / shard_count;
    const int64_t src = shard * shard_stride + ((lt * RT_TILE_H + (j % RT_TILE_H)) * RT_TILE_W + (i % RT_TILE_W)) * 3;
    frame[idx * 3 + 0] = gathered[src + 0];
    frame[idx * 3 + 1] = gathered[src + 1];
    frame[idx * 3 + 2] = gathered[src + 2];
}

// color_to_rgb(c / spp) (src/renderer.rs:55-58, src/color.rs:12-19): the host library's code (rt_shared_math.h)
__global__ void resolve_rgb8_kernel(int64_t n_values, double inv_spp, const double *__restrict__ sum, uint8_t *__restrict__ rgb) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_values) return;
    rgb[idx] = rtm::rt_quantise(rtm::rt_gamma_encode(sum[idx] * inv_spp));
}

// test hook: the conservative f32 box test against the exact f64 one on caller-supplied rays and boxes
__global__ void debug_box_kernel(int64_t n, const double *__restrict__ rays, const double *__restrict__ boxes, double tmin,
                                 double tmax, uint8_t *__restrict__ exact_hit, uint8_t *__restrict__ f32_hit) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const double *r = rays + idx * 6, *b = boxes + idx * 6;
    const V3 o = v3(r[0], r[1], r[2]), d = v3(r[3], r[4], r[5]);
    const double lo[3] = {b[0], b[1], b[2]}, hi[3] = {b[3], b[4], b[5]};
    // outward rounding exactly as the scene compiler does it
    float lo32[3], hi32[3];
    for (int k = 0; k < 3; ++k) {
        lo32[k] = __double2float_rd(lo[k]);
        hi32[k] = __double2float_ru(hi[k]);
    }
    exact_hit[idx] = box_miss_f64(lo, hi, o, d, tmin, tmax) ? 0 : 1;
    const bool single = !box_miss_f32(lo32, hi32, make_ray32(o, d), (float)tmin, (float)tmax);
    // the ordered walk's pair test, with the box in both slots
    const float bb[6] = {lo32[0], hi32[0], lo32[1], hi32[1], lo32[2], hi32[2]};
    bool m0, m1;
    float e0, e1;
    float extent = 0.0f; // the scene compiler's B is at least the largest coordinate of this box
    for (int k = 0; k < 6; ++k) extent = __builtin_fmaxf(extent, __builtin_fabsf(bb[k]));
    const RayPair32 rp = make_ray_pair32(o, d, 0u, extent);
    box_pair_f32(opair_of_box(bb, rp), rp, __double2float_rd(tmin), __double2float_ru(tmax), m0, m1, e0, e1);
    f32_hit[idx] = (uint8_t)((single ? 1 : 0) | (m0 ? 0 : 2) | (m1 ? 0 : 4)); // bit 0: box_miss_f32, bits 1-2: box_pair_f32
}

// test hook: the quad stage's conservative f32 filter against the exact f64 Quad::hit on caller-supplied (ray, quad) pairs
__global__ void debug_quad_kernel(int64_t n, const double *__restrict__ rays, const Quad *__restrict__ quads, const QFiltPair *__restrict__ filt,
                                  double tmin, double tmax, uint8_t *__restrict__ exact_hit, uint8_t *__restrict__ keep) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const double *r = rays + idx * 6;
    const V3 o = v3(r[0], r[1], r[2]), d = v3(r[3], r[4], r[5]);
    const Quad *qd = &quads[idx];
    bool hit = false, inside = true; // src/quad.rs:96-127, as the quad stage runs it; inside: what the exact test finds for alpha, beta
    const V3 normal = ld3(qd->normal);
    const double denom = dot(normal, d);
    if (!(__builtin_fabs(denom) < 1e-8)) {
        const double t = (qd->d - dot(normal, o)) / denom;
        if (tmin <= t && t <= tmax) {
            const V3 php = (o + d * t) - ld3(qd->q);
            const V3 qw = ld3(qd->w);
            const double alpha = dot(qw, cross(php, ld3(qd->v)));
            const double beta = dot(qw, cross(ld3(qd->u), php));
            inside = !(alpha < 0.0 || alpha > 1.0 || beta < 0.0 || beta > 1.0);
            hit = inside;
        }
    }
    exact_hit[idx] = hit ? 1 : 0;
    const uint32_t bits = quad_pair_keep(&filt[idx], make_qray32(o, d), f32_below(tmin), f32_above(tmax));
    keep[idx] = (uint8_t)((bits & 3u) | ((bits >> 8 & 3u) << 2) | (inside ? 16u : 0u)); // bits 0-1 keep, 2-3 certainly inside, 4: the exact alpha, beta ARE inside // (the quad sits in both slots)
}

// test hook: evaluates one device-side scalar function over arrays (rt_debug_eval)
__global__ void debug_eval_kernel(int32_t op, int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                  double *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const double x = a[idx], y = b ? b[idx] : 0.0;
    double r = 0.0;
    switch (op) {
    case RT_DEBUG_LOG: r = rt_log(x); break;
    case RT_DEBUG_SIN: r = rt_sin(x); break;
    case RT_DEBUG_ACOS: r = rt_acos(x); break;
    case RT_DEBUG_ATAN2: r = rt_atan2(x, y); break;
    case RT_DEBUG_POW5: r = rt_pow5(x); break;
    case RT_DEBUG_SQRT: r = __builtin_sqrt(x); break;
    case RT_DEBUG_DIV: r = x / y; break;
    case RT_DEBUG_MUL_ADD: r = x * y + x; break; // must be two roundings (no contraction)
    case RT_DEBUG_RNG_RANDOM: { // x, y carry the key's and the draw number's bits
        Rng g; g.start_key(f2u(x)); for (uint64_t i = 0; i < f2u(y); ++i) g.next(); r = g.random(); break;
    }
    case RT_DEBUG_RNG_RANGE: {
        Rng g; g.start_key(f2u(x)); for (uint64_t i = 0; i < f2u(y); ++i) g.next(); r = g.range(-1.0, 1.0); break;
    }
    case RT_DEBUG_RNG_UNNEXT: { // draw y of the stream, after one draw too many has been taken back
        Rng g; g.start_key(f2u(x)); for (uint64_t i = 0; i < f2u(y); ++i) g.next(); g.next(); g.next(); g.unnext(); g.unnext(); r = g.random(); break;
    }
    case RT_DEBUG_F32_ABOVE: r = (double)f32_above(x); break;
    case RT_DEBUG_F32_BELOW: r = (double)f32_below(x); break;
    default: r = 0.0;
    }
    out[idx] = r;
}

} // namespace

// =====================================================================================================
// Host side of this file: which instantiation renders what, and the launches of the small kernels
// =====================================================================================================
namespace rtk {

uint32_t kernel_features_for(uint32_t scene_features, int lds, bool ordered) {
    (void)ordered;
    if (lds == 3) { // the specialised instantiations exist for scenes that live in the LDS whole
        for (uint32_t f : {FEAT_SPHERES_SOLID, FEAT_QUADS_FRAMES, FEAT_QUADS_FRAMES_MEDIA, FEAT_SPHERES_QUADS_TEXTURES})
            if ((scene_features & ~f) == 0) return f;
    }
    return F_ALL;
}
int kernel_threads_for(uint32_t kernel_features, int lds, bool ordered) {
    if (lds == 0) return GLOBAL_THREADS;
    if (kernel_features == FEAT_QUADS_FRAMES) return QUADS_FRAMES_THREADS;
    if (kernel_features == FEAT_SPHERES_QUADS_TEXTURES && !ordered) return REFERENCE_TEXTURES_THREADS;
    return kernel_features == FEAT_SPHERES_SOLID ? LDS_THREADS : LDS_THREADS_GENERAL;
}
const void *path_kernel_for(int lds, bool counted, uint32_t feat, bool ordered, bool aux, bool wide) {
#define RT_PICK(L, T, F, O, A) (counted ? (const void *)path_kernel<true, L, T, F, O, A> : (const void *)path_kernel<false, L, T, F, O, A>)
#define RT_PICK_W(L, T, F, A) (counted ? (const void *)path_kernel<true, L, T, F, true, A, true> : (const void *)path_kernel<false, L, T, F, true, A, true>)
#define RT_PICK_AUX(L, T, F) (wide ? (aux ? RT_PICK_W(L, T, F, true) : RT_PICK_W(L, T, F, false)) : (aux ? RT_PICK(L, T, F, true, true) : RT_PICK(L, T, F, true, false)))
    if (ordered) { // (AUX: the small tables in the LDS as well, wherever they fit — rt_api.cpp decides)
        if (lds == 3) {
            if (feat == FEAT_SPHERES_SOLID) return RT_PICK_AUX(3, LDS_THREADS, FEAT_SPHERES_SOLID);
            if (feat == FEAT_QUADS_FRAMES) return RT_PICK_AUX(3, QUADS_FRAMES_THREADS, FEAT_QUADS_FRAMES);
            if (feat == FEAT_QUADS_FRAMES_MEDIA) return RT_PICK_AUX(3, LDS_THREADS_GENERAL, FEAT_QUADS_FRAMES_MEDIA);
            if (feat == FEAT_SPHERES_QUADS_TEXTURES) return RT_PICK_AUX(3, LDS_THREADS_GENERAL, FEAT_SPHERES_QUADS_TEXTURES);
            return RT_PICK_AUX(3, LDS_THREADS_GENERAL, F_ALL);
        }
        if (lds == 1) return RT_PICK_AUX(1, LDS_THREADS_GENERAL, F_ALL);
        return RT_PICK_AUX(0, GLOBAL_THREADS, F_ALL);
    }
    if (lds == 3) {
        if (feat == FEAT_SPHERES_SOLID) return RT_PICK(3, LDS_THREADS, FEAT_SPHERES_SOLID, false, false);
        if (feat == FEAT_QUADS_FRAMES) return RT_PICK(3, QUADS_FRAMES_THREADS, FEAT_QUADS_FRAMES, false, false);
        if (feat == FEAT_QUADS_FRAMES_MEDIA) return RT_PICK(3, LDS_THREADS_GENERAL, FEAT_QUADS_FRAMES_MEDIA, false, false);
        if (feat == FEAT_SPHERES_QUADS_TEXTURES) return RT_PICK(3, REFERENCE_TEXTURES_THREADS, FEAT_SPHERES_QUADS_TEXTURES, false, false);
        return RT_PICK(3, LDS_THREADS_GENERAL, F_ALL, false, false);
    }
    if (lds == 2) return RT_PICK(2, LDS_THREADS_GENERAL, F_ALL, false, false);
    if (lds == 1) return RT_PICK(1, LDS_THREADS_GENERAL, F_ALL, false, false);
    return RT_PICK(0, GLOBAL_THREADS, F_ALL, false, false);
#undef RT_PICK_AUX
#undef RT_PICK_W
#undef RT_PICK
}

void launch_sum_samples(const KParams &K, unsigned grid, hipStream_t stream) {
    hipLaunchKernelGGL(sum_samples_kernel, dim3(grid), dim3(256), 0, stream, K);
}
void launch_tiles_to_frame(int32_t w, int32_t h, int32_t tiles_x, int32_t shard_count, int64_t shard_stride, const double *gathered,
                           double *frame, hipStream_t stream) {
    const int64_t n = (int64_t)w * h;
    hipLaunchKernelGGL(tiles_to_frame_kernel<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, h, tiles_x, shard_count,
                       shard_stride, gathered, frame);
}
void launch_tiles_to_frame_rgb8(int32_t w, int32_t h, int32_t tiles_x, int32_t shard_count, int64_t shard_stride, const uint8_t *gathered,
                                uint8_t *frame, hipStream_t stream) {
    const int64_t n = (int64_t)w * h;
    hipLaunchKernelGGL(tiles_to_frame_kernel<uint8_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, h, tiles_x, shard_count,
                       shard_stride, gathered, frame);
}
void launch_resolve_rgb8(int64_t n_values, double inv_spp, const double *sum, uint8_t *rgb, hipStream_t stream) {
    hipLaunchKernelGGL(resolve_rgb8_kernel, dim3((unsigned)((n_values + 255) / 256)), dim3(256), 0, stream, n_values, inv_spp, sum, rgb);
}
void launch_debug_box(int64_t n, const double *rays, const double *boxes, double tmin, double tmax, uint8_t *exact_hit, uint8_t *f32_hit) {
    hipLaunchKernelGGL(debug_box_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, n, rays, boxes, tmin, tmax, exact_hit, f32_hit);
}
void launch_debug_quad(int64_t n, const double *rays, const Quad *quads, const QFiltPair *filt, double tmin, double tmax, uint8_t *exact_hit, uint8_t *keep) {
    hipLaunchKernelGGL(debug_quad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, n, rays, quads, filt, tmin, tmax, exact_hit, keep);
}
void launch_debug_eval(int32_t op, int64_t n, const double *a, const double *b, double *out) {
    hipLaunchKernelGGL(debug_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, op, n, a, b, out);
}

} // namespace rtk

