// rt_api.hpp — internals shared by the host-side translation units of librt_amd (rt_api.cpp, rt_debug.cpp, rt_gather.cpp).
#pragma once
#include "rt_amd.h"
#include "rt_amd_debug.h"
#include "rt_compile.hpp"
#include "rt_kernels.h"
#include "rt_ordered.hpp"

#include <hip/hip_runtime_api.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace rtapi {
using namespace rtd;
using namespace rtk;

extern thread_local uint32_t g_last_launch[4]; // of the calling thread's last render: 0 (reserved), LDS level, workgroup threads, grid
int fail(int status, const std::string &msg); // sets rt_last_error() of the calling thread, returns `status`
#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess)                                                                                  \
            return rtapi::fail(_e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP,                  \
                               std::string(#expr) + ": " + hipGetErrorString(_e));                             \
    } while (0)

// Per-launch scratch.  A frame that needs several launches (sample sub-ranges, bounded by the sample buffer) alternates between
// two such sets on two internal streams, so that the next launch's workgroups move in as the current one's drain (its tail)
// instead of waiting behind its per-pixel summation.
struct LaunchScratch {
    double *att_stack = nullptr;   // parked colours (KParams::att_stack): full size only for scenes that park colours
    size_t att_bytes = 0;
    uint32_t *att_ids = nullptr;   // parked material indices beyond the eight a lane keeps in registers (KParams::att_ids)
    size_t att_ids_bytes = 0;
    double *world_slots = nullptr; // [6][n_threads]
    size_t world_bytes = 0;
    double *samples = nullptr; // sample buffer of one launch
    size_t sample_bytes = 0;
    uint32_t *job_counter = nullptr;
};
struct Workspace {
    LaunchScratch half[2];
    hipStream_t aux[2] = {nullptr, nullptr};      // created on first use (all of them or none)
    hipEvent_t ev_start = nullptr, ev_sum[2] = {nullptr, nullptr};
    hipEvent_t ev_done = nullptr;                 // recorded on the caller's stream behind every render's last enqueue: what an
                                                  // eviction waits for (the caller's stream handle itself is never touched again)
    size_t sample_budget = 0;                     // > 0: the sample buffer size an out-of-memory back-off arrived at
    unsigned long long *counters = nullptr;
};
void free_workspace(Workspace &w);
// One per (scene, stream).  rt_scene::mu guards only the table and `in_use`; everything slow — draining the stream before a buffer
// grows, hipFree / hipMalloc, creating the internal streams, the enqueues themselves — happens under the slot's own mutex, so renders
// of one scene on DIFFERENT streams never wait for each other (renders on one stream are ordered anyway).
struct WorkspaceSlot {
    std::mutex mu;
    Workspace w;
    int in_use = 0; // renders that hold this slot (from the table look-up to the end of their enqueues): such a slot is never evicted
};

// Scheduler knobs (64ths of the live lanes a deferred stage must have queued / the box loop needs to keep running).
// The best values depend on the stage mix, so there is one preset per kernel instantiation, each picked with
// tools/tune.py on MI355X (DESIGN.md "Scheduler"); a value >= 0 in `forced` (RT_TH_* variables, rt_debug_set_tuning)
// overrides all presets.
struct Thresholds { uint32_t prim, other, shade, box, newjob; };
struct Tuning {
    Thresholds general{8, 8, 48, 8, 0};
    Thresholds ordered_general{8, 12, 40, 8, 0}; // every feature, ordered walk (final_scene: 760 vs 745 Msamples/s at 60 spp)
    Thresholds ordered_global{4, 8, 48, 8, 0};   // ... the same walk with the scene gathered from global memory (final_scene; swept twice on round 3's kernels: +0.4 % over the
                                                // preset above, which stays for the LDS-resident scenes — cornell_smoke loses 5 % with this one)
    Thresholds spheres_solid{6, 16, 32, 6, 28};  // random-spheres (re-tuned with the start shortcut: 5070 vs 4900 Msamples/s at 50 spp for round 1's 8/16/24/16/32)
    Thresholds quads_frames{24, 16, 48, 2, 0}; // Cornell box (tools/tune.py; with instances walked last and flat leaves the merged path end wins: round 1's 8/16/40/4/8 is 7 % behind)
    Thresholds spheres_threaded{8, 16, 24, 16, 32}; // ... the same kernel walking the reference's order (no shortcut there)
    Thresholds quads_only{8, 16, 40, 4, 8};    // ... the same kernel on a scene without instances (quads: 13.4 vs 12.8 Gsamples/s with the preset above)
    int forced[5] = {-1, -1, -1, -1, -1};    // prim, other, shade, box, newjob
    int use_lds = 1; // 0: always gather the scene from global memory (tuning / A-B runs)
    int refit = 1;   // 0: walk the reference's own (looser) boxes
    int ordered = 1; // scenes created from now on: 0 always the reference-order walk, 1 the ordered walk where it pays
                     // (ordered_walk_pays), 2 the ordered walk wherever the scene allows it
    int jobs_per_grab = 0; // > 0: fixed grab size (RT_JOBS_PER_GRAB; tuning runs)
    int wide = -1;         // own trees with four-child records (rt_layout.h ONode4): 1 always, 0 never, -1 for scenes of 64 primitives or more (RT_WIDE)
    int quad_filter = 1;   // multi-quad leaves go through the conservative f32 filter before the exact test (RT_QUAD_FILTER; rt_scene_options.quad_filter)
    int medium_first = 1;  // a ray that starts inside a sphere-bounded medium: that medium's draw first, the tree in front of it clipped (RT_MEDIUM_FIRST)
    int overlap = 1;       // 1: a frame of several launches alternates between two scratch sets on two streams (RT_OVERLAP)
    int slow_min = 4, slow_age = 32; // KParams::slow_min / slow_age (RT_SLOW_MIN, RT_SLOW_AGE; slow_min 1: nobody waits)
    int seq_lookahead = 1;  // scenes with media: a query looks ahead at the boxes of the sequence's later steps when it starts (RT_SEQ_LOOKAHEAD)
    int start_shortcut = 1; // a root whose one child is a single sphere spanning the scene (random-spheres' ground): queries start with that sphere's test (RT_START_SHORTCUT)
    int defer = 1;         // ordered walk: the world frame's instances (<= 32) are walked after the world's own tree, one frame change each instead of two (RT_DEFER)
    int grab_taper = 8;    // guided hand-out: a grab takes at most 1 / (waves x this) of the jobs left (RT_GRAB_TAPER; 0: off; tools/sweep_grabs.sh)
    OrderedOptions ordered_options;
    size_t sample_buffer_bytes = (size_t)2 << 30; // per-(scene, stream) sample buffer at most (halved on out-of-memory)
    Tuning();
    Thresholds pick(const Thresholds &preset) const {
        Thresholds t = preset;
        if (forced[0] >= 0) t.prim = (uint32_t)forced[0];
        if (forced[1] >= 0) t.other = (uint32_t)forced[1];
        if (forced[2] >= 0) t.shade = (uint32_t)forced[2];
        if (forced[3] >= 0) t.box = (uint32_t)forced[3];
        if (forced[4] >= 0) t.newjob = (uint32_t)forced[4];
        return t;
    }
};
// The caller's rt_scene_options (any struct_size this library has ever shipped, or null) laid over the defaults: every field at or
// beyond the caller's struct_size is the default; RT_ERR_INVALID_ARGUMENT for a size or a walk nobody knows.  One helper for
// rt_scene_create_ex and the layout hooks of rt_debug.cpp, so that what a test inspects is what a scene gets.
int resolve_scene_options(const rt_scene_options *options, rt_scene_options &out, const char *who);
// the tree-building options a scene gets from them (on top of the process defaults)
// Process-wide defaults (RT_* environment variables, rt_debug_set_*): read and written under one mutex; a scene takes its
// copy when it is created (walk, refit, tree options) and every launch takes one (thresholds, LDS use).
Tuning tuning_snapshot();
void tuning_update(void (*fn)(Tuning &, const void *), const void *arg);

template <class T> struct DeviceArray {
    T *ptr = nullptr;
    size_t bytes = 0;
};

} // namespace rtapi

using namespace rtapi; // (an internal header of three translation units)

struct rt_scene {
    int device = 0;
    int n_cus = 0;
    int blocks_per_cu[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}; // [LDS level][counted?]
    rtapi::DeviceArray<uint4> lds_image;               // the LDS-resident copy of nodes / spheres / quads (if they fit)
    uint32_t lds_off_node_b = 0, lds_off_spheres = 0, lds_off_quads = 0, lds_image_bytes = 0;
    uint32_t lds_off_qfilt = 0;                 // the quads' f32 filter records in the LDS image (0: none)
    bool has_instances = false;
    bool parks_colours = false;                 // some attenuation is a texture's value: launches need the colour stack (KParams::att_stack)
    int lds_level = 0;                          // 0 nothing fits, 1 nodes, 2 nodes + spheres, 3 nodes + spheres + quads
    uint32_t features = F_ALL;                  // Feature bits the scene uses
    uint32_t lds_prefix_bytes[4] = {0, 0, 0, 0}; // image prefix each LDS level copies in
    rtapi::DeviceArray<Node32> nodes;
    rtapi::DeviceArray<Sphere> spheres;
    rtapi::DeviceArray<Quad> quads;
    rtapi::DeviceArray<Instance> insts;
    rtapi::DeviceArray<Medium> media;
    rtapi::DeviceArray<DMaterial> mats;
    rtapi::DeviceArray<rt_texture> texs;
    rtapi::DeviceArray<rt_perlin> perlins;
    rtapi::DeviceArray<ImageRef> images;
    rtapi::DeviceArray<uint8_t> texels;
    rtapi::DeviceArray<double> lut;
    uint32_t n_nodes = 0;
    bool ordered = false;                        // ordered layout (rt_ordered.hpp): onodes instead of nodes
    bool wide = false;                           // ... with four-child records (onodes4)
    rtapi::DeviceArray<uint4> oimage;                   // ordered layout: the tables of load_opair (global copy)
    rtapi::DeviceArray<OSeq> oseq;                      // ... and the world frame's sequence
    uint32_t n_oseq = 0;
    float box_extent = 0.0f;                     // largest |coordinate| of any box of the ordered layout
    rtapi::DeviceArray<uint4> aux_image;                // materials | textures | frames | media | Perlin for the AUX kernels (0 bytes: not used)
    uint32_t aux_bytes = 0, aux_off[5] = {0, 0, 0, 0, 0};
    uint32_t o_root = 0, o_stack = 0;            // world root record; stack entries per lane
    uint32_t o_start_stage = 0, o_start_prim = 0, o_start_end = 0, o_start_rest = 0, o_start_slot = 0; // KParams::o_start_*
    rt_scene_stats stats{};
    std::mutex mu;
    std::map<hipStream_t, std::unique_ptr<rtapi::WorkspaceSlot>> workspaces; // one per stream: launches on a stream are ordered
    rt_scene_options options;                    // the caller's per-scene options (rt_scene_create_ex); unset fields: process defaults
    std::mutex host_render_mu;                   // rt_render (host-buffer form) calls on one scene run one at a time
};

namespace rtapi {
// of the last counted render: per profile slot (rounds, active lanes, cycles)
extern std::mutex g_stage_profile_mu;
extern unsigned long long g_stage_profile[PROF_SLOTS * 3];
} // namespace rtapi
