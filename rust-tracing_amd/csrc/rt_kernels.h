// rt_kernels.h — the seam between the device code (rt_kernel.hip) and the host side of librt_amd (rt_api.cpp, rt_debug.cpp):
// the kernels' parameter block, the feature bits a kernel instantiation is compiled for, and the launch entry points.
#pragma once
#include "rt_amd.h"
#include "rt_layout.h"

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <cstdint>

namespace rtk {
using namespace rtd;

struct DMaterial { // 64 bytes
    uint32_t kind;
    uint32_t texture;
    uint32_t needs_uv; // the texture below reads (u, v): only ImageTexture does (src/texture.rs:83)
    uint32_t solid;    // the texture is a SolidColor: its colour is copied into `albedo` (one dependent load fewer per hit)
    double albedo[3];  // Metal's albedo, or the SolidColor's colour; a Dielectric's 1 / ir, r0 seen from outside, r0 seen from inside (rt_api.cpp)
    double fuzz;
    double ir;
    uint32_t slow;     // evaluating the texture is dear (Perlin turbulence): see the shade stage
    uint32_t _pad2;
};
static_assert(sizeof(DMaterial) == 64, "DMaterial must be 64 bytes");

struct KParams {
    const Node32 *nodes;
    const Sphere *spheres;
    const Quad *quads;
    const Instance *insts;
    const Medium *media;
    const DMaterial *mats;
    const rt_texture *texs;
    const rt_perlin *perlins;
    const ImageRef *images;
    const uint8_t *texels;
    const double *srgb_lut;
    double *out;
    double *samples;                // [local tile][sample of this launch][64 pixels][3]: one colour per camera path
    double *att_stack;              // [max_depth + 1][3][n_threads]: parked attenuations that are a texture's value
    uint32_t *att_ids;              // [max_depth + 1 - 8][n_threads]: parked material indices that no longer fit the lane's registers
    uint32_t ids_ok;                // 1: material indices fit 16 bits (else every attenuation is parked as a colour, by a kernel with textures)
    uint32_t id_one;                // index of the material table's extra last entry, whose albedo is Color::ONE
    uint32_t *job_counter;
    unsigned long long *counters;   // rt_counters as 10 u64, then per profile slot (9): rounds, active lanes, cycles; or null
    rt_camera cam;
    uint64_t seed_mixed;            // mix64(seed + gamma)
    uint32_t n_nodes;
    uint32_t n_threads;
    int32_t sample_begin;           // first sample of this launch
    uint32_t n_samples;             // samples per pixel in this launch
    uint32_t n_jobs;                // n_local_tiles * n_samples * 64
    uint32_t jobs_per_grab;
    float grab_taper;               // a grab takes at most this fraction of the jobs still to hand out (guided self-scheduling)
    double inv_n_samples, inv_tiles_x; // 1 / n_samples, 1 / tiles_x (job decode)
    int32_t max_depth, accumulate;
    int32_t shard_index, shard_count, out_layout;
    int32_t tiles_x;
    uint32_t n_local_tiles;
    uint32_t th_prim, th_other, th_shade, th_new; // scheduler thresholds, in 64ths of the live lanes
    uint32_t th_box;                // the box loop keeps running while this many 64ths of the live lanes are in it
    uint32_t th_pack;               // path_kernel reads the four above from here: prim | other << 8 | shade << 16 | box << 24 (each <= 255)
    uint32_t defer_instances;       // ordered walk: 1 = the world frame's Translate/RotateY subtrees are walked after the world's own tree (path_kernel)
    // LDS-resident scene (SCENE_IN_LDS kernels): image to copy in, and where its parts start (bytes)
    const uint4 *lds_image;
    uint32_t lds_image_bytes;
    uint32_t lds_off_node_b, lds_off_spheres, lds_off_quads;
    uint32_t lds_off_qfilt;         // the quads' f32 filter records (rt_layout.h QFiltPair), behind the quads; 0xffffffff: none, every quad gets the exact test
    double *world_slots;            // [6][n_threads] doubles: a lane's world-frame ray while it walks inside a frame
    uint32_t lds_world_off;         // ... or, where the LDS has room, [6][block threads] doubles there (0xffffffff: global memory)
    // ordered layout (rt_layout.h): records, the world frame's root, and where the per-lane stacks start in the LDS
    const uint4 *oimage;            // the seven tables of load_opair, in global memory (LDS kernels copy them in)
    const OSeq *oseq;               // the world frame's sequence of trees and media (rt_layout.h)
    float box_extent;               // the largest |coordinate| of any box of the ordered layout (box_pair_f32's B)
    const uint4 *aux_image;         // AUX kernels: materials | textures | frames | media | Perlin tables, to copy into the LDS
    uint32_t aux_bytes, lds_aux_off, aux_off_mats, aux_off_texs, aux_off_insts, aux_off_media, aux_off_perlins;
    uint32_t n_oseq;
    uint32_t o_root;
    // a query may start with the primitives of a leaf under the root instead of the root record (rt_api.cpp "start shortcut"): the stage of the
    // leaf's kind (ST_SPHERE / ST_QUAD; 0: no shortcut), its primitives [prim, end), the root's other child, which child of the root the leaf is
    uint32_t o_start_stage, o_start_prim, o_start_end, o_start_rest, o_start_slot;
    uint32_t inst_shortcut;         // 1: a walk that enters a frame whose tree is one leaf starts with the leaf's primitives (Instance::start_ref)
    uint32_t slow_min, slow_age;    // shade stage: lanes with a dear texture wait for this many of their kind, at most this many shade rounds
    uint32_t medium_first;          // 1: the draw of a sphere-bounded medium the ray starts inside is made before the tree in front of it is walked (path_kernel)
    uint32_t seq_lookahead;         // 1: a query that cannot reach any later step of the world's sequence ends it at its start (path_kernel)
    uint32_t lds_stack_off;
    uint32_t lds_seq_off;           // the world frame's sequence, copied in by the ordered kernels (after the stacks)
    uint32_t lds_prof_off;          // COUNT kernels: per-wave profile rows (last)
};

// What a scene can contain.  A kernel instantiated without a feature has that code compiled out, which matters for
// more than its size: the register allocation of the whole kernel is set by its hungriest path.
enum Feature : uint32_t {
    F_SPHERES = 1u,  // Sphere leaves
    F_QUADS = 2u,    // Quad leaves
    F_FRAMES = 4u,   // Translate / RotateY
    F_MEDIA = 8u,    // ConstantMedium
    F_TEXTURES = 16u // Checker / Image / Noise textures (without it every texture is a SolidColor)
};
constexpr uint32_t F_ALL = 31u;

constexpr uint32_t PROF_SLOTS = 12;      // COUNT kernels: profile slots per wave (6 stages + 6 parts of the shade / path-end rounds)
constexpr uint32_t COUNTER_WORDS = 10 + PROF_SLOTS * 3; // rt_counters as 10 u64, then per profile slot: rounds, active lanes, cycles
// Jobs a wave reserves at a time: a multiple of 64 (one sample-row of an 8x8 tile, so the lanes a wave starts together
// trace neighbouring pixels).  Large grabs mean few atomics; small ones a short tail (the last grab of the slowest wave
// is all that is left running at the end): launch_render picks the size so that every wave gets at least ~32 grabs.
// (one device counter serves every wave: reservations have to stay well under ~88 per microsecond machine-wide)
constexpr uint32_t MAX_JOBS_PER_GRAB = 1024, MIN_JOBS_PER_GRAB = 128, MIN_JOBS_PER_WAVE = 256;

constexpr int GLOBAL_THREADS = 256;             // scene gathered from global memory: 256-thread blocks
#ifndef RT_LDS_THREADS
#define RT_LDS_THREADS 1024 // 16 waves = 4 per SIMD (tools/tune.py: 512 -> 1381, 768 -> 1774, 1024 -> 1921 Msamples/s on C2 at 48 spp)
#endif
constexpr int LDS_THREADS = RT_LDS_THREADS;     // scene in LDS: one 16-wave workgroup per CU shares the copy
#ifndef RT_LDS_THREADS_GENERAL
#define RT_LDS_THREADS_GENERAL 1024 // the every-feature kernels: 16 waves = 4 per SIMD at 128 registers.  (Rounds 1-2: 768 threads, 3 per SIMD at 168 registers —
// at 128 they spilled 80-141 of them; with the parked indices, the parameters and the slot addresses read at use they spill 0-75, nearly all of it in cold
// code, and the fourth wave pays: cornell_smoke 1296 -> 1465 Msamples/s, two_perlin_spheres 4663 -> 5082, simple_light 5684 -> 5871; the reference-order texture kernel is the exception, below)
#endif
constexpr int LDS_THREADS_GENERAL = RT_LDS_THREADS_GENERAL;
#ifndef RT_QUADS_FRAMES_THREADS
#define RT_QUADS_FRAMES_THREADS RT_LDS_THREADS // the quads + frames kernel (Cornell): 128 registers with 6-10 spilled at 1024 threads; tools A/B: 768
#endif
constexpr int QUADS_FRAMES_THREADS = RT_QUADS_FRAMES_THREADS;
// spheres + quads + textures in the REFERENCE's order (earth, a one-sphere scene, renders that way by default; the fallback of two_spheres,
// two_perlin_spheres, simple_light): 12 waves = 3 per SIMD at 168 registers.  At 1024 threads that kernel spills 48 registers in its
// texture code, which a one-primitive scene runs all the time: earth 17.6 Gsamples/s at 1024 threads, 22.1 at 768 (256 spp; the other
// three in reference order 5.6 / 2.9 / 5.6 -> 6.7 / 3.2 / 6.0).  The same features on the library's own trees keep 1024 (8.3 / 5.1 / 7.7
// against 8.1 / 4.8 / 7.3 at 768), and so does every kernel with media (cornell_smoke 1.50 against 1.32; reference order 1.17 / 1.03).
constexpr int REFERENCE_TEXTURES_THREADS = 768;
constexpr size_t LDS_BUDGET_BYTES = 160 * 1024; // LDS per CU on MI355X

// The kernel instantiations that exist: the general one (every feature) at each LDS level, plus specialised
// ones for scenes that fit the LDS entirely and use a subset of the features (BASELINE configs 1/2 and 3).
constexpr uint32_t FEAT_SPHERES_SOLID = F_SPHERES;          // random-spheres: spheres, solid colours
constexpr uint32_t FEAT_QUADS_FRAMES = F_QUADS | F_FRAMES;  // Cornell box: quads, cubes in Translate/RotateY frames
// two more, at 768 threads (3 waves per SIMD), which need no spill there: 156 and 150 registers (the every-feature kernel: 168 + 16..38 spilled)
constexpr uint32_t FEAT_QUADS_FRAMES_MEDIA = F_QUADS | F_FRAMES | F_MEDIA;              // cornell_smoke
constexpr uint32_t FEAT_SPHERES_QUADS_TEXTURES = F_SPHERES | F_QUADS | F_TEXTURES;    // two_spheres, earth, two_perlin_spheres, simple_light
uint32_t kernel_features_for(uint32_t scene_features, int lds, bool ordered);
int kernel_threads_for(uint32_t kernel_features, int lds, bool ordered); // workgroup size of that instantiation
const void *path_kernel_for(int lds, bool counted, uint32_t feat, bool ordered, bool aux, bool wide);

// launches of the small kernels (all asynchronous on `stream`; errors through hipGetLastError)
void launch_sum_samples(const KParams &K, unsigned grid, hipStream_t stream);
void launch_tiles_to_frame(int32_t w, int32_t h, int32_t tiles_x, int32_t shard_count, int64_t shard_stride, const double *gathered,
                           double *frame, hipStream_t stream);
void launch_tiles_to_frame_rgb8(int32_t w, int32_t h, int32_t tiles_x, int32_t shard_count, int64_t shard_stride, const uint8_t *gathered,
                                uint8_t *frame, hipStream_t stream);
void launch_resolve_rgb8(int64_t n_values, double inv_spp, const double *sum, uint8_t *rgb, hipStream_t stream);
void launch_debug_box(int64_t n, const double *rays, const double *boxes, double tmin, double tmax, uint8_t *exact_hit, uint8_t *f32_hit);
void launch_debug_quad(int64_t n, const double *rays, const Quad *quads, const QFiltPair *filt, double tmin, double tmax, uint8_t *exact_hit, uint8_t *keep);
void launch_debug_eval(int32_t op, int64_t n, const double *a, const double *b, double *out);

} // namespace rtk
