// rt_api.cpp — the host side of librt_amd: the C ABI of include/rt_amd.h (scene upload, render launches, frame-end
// helpers).  Plain C++ over the HIP runtime API; the kernels live in rt_kernel.hip and are reached through rt_kernels.h.
#include "rt_api.hpp"
#include "rt_qfilt.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace rtapi;

namespace rtapi {

thread_local std::string g_last_error;
thread_local uint32_t g_last_launch[4] = {0, 0, 0, 0};
std::mutex g_stage_profile_mu;
unsigned long long g_stage_profile[PROF_SLOTS * 3] = {0};

int fail(int status, const std::string &msg) {
    g_last_error = msg;
    return status;
}

Tuning::Tuning() {
    auto env = [](const char *name, int &v) { if (const char *e = getenv(name)) v = atoi(e); };
    env("RT_TH_PRIM", forced[0]); env("RT_TH_OTHER", forced[1]); env("RT_TH_SHADE", forced[2]); env("RT_TH_BOX", forced[3]); env("RT_TH_NEW", forced[4]);
    env("RT_USE_LDS", use_lds); env("RT_REFIT", refit); env("RT_ORDERED", ordered); env("RT_JOBS_PER_GRAB", jobs_per_grab); env("RT_GRAB_TAPER", grab_taper); env("RT_DEFER", defer); env("RT_START_SHORTCUT", start_shortcut); env("RT_SEQ_LOOKAHEAD", seq_lookahead); env("RT_SLOW_MIN", slow_min); env("RT_SLOW_AGE", slow_age); env("RT_OVERLAP", overlap);
    env("RT_WIDE", wide); env("RT_QUAD_FILTER", quad_filter); env("RT_MEDIUM_FIRST", medium_first);
    if (const char *e = getenv("RT_SAH_LEAF")) ordered_options.leaf_max = (uint32_t)atoi(e);
    if (const char *e = getenv("RT_FLAT_MAX")) ordered_options.flat_max = (uint32_t)atoi(e);
    if (const char *e = getenv("RT_SAH_SPHERE")) ordered_options.cost_sphere = atof(e);
    if (const char *e = getenv("RT_SAH_QUAD")) ordered_options.cost_quad = atof(e);
    if (const char *e = getenv("RT_SAH_INSTANCE")) ordered_options.cost_instance = atof(e);
    if (const char *e = getenv("RT_SAMPLE_BUFFER_MB")) sample_buffer_bytes = (size_t)strtoull(e, nullptr, 10) << 20;
}
namespace {
std::mutex g_tuning_mu;
Tuning &tuning_locked() { static Tuning t; return t; } // call with g_tuning_mu held
} // namespace
Tuning tuning_snapshot() {
    std::lock_guard<std::mutex> lock(g_tuning_mu);
    return tuning_locked();
}
void tuning_update(void (*fn)(Tuning &, const void *), const void *arg) {
    std::lock_guard<std::mutex> lock(g_tuning_mu);
    fn(tuning_locked(), arg);
}

} // namespace rtapi

namespace {

uint64_t host_mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

uint32_t lds_image_bytes_for(const rt_scene *s, int lds) { return s->lds_prefix_bytes[lds]; }
// the LDS image, then (ordered walk) the per-lane stacks: 2-byte entries beside an LDS-resident scene, else 4-byte
int block_threads(const rt_scene *s, int lds) { return kernel_threads_for(kernel_features_for(s->features, lds, s->ordered), lds, s->ordered); }
size_t stack_bytes(const rt_scene *s, int lds) {
    if (!s->ordered) return 0;
    return (size_t)s->o_stack * (size_t)block_threads(s, lds) * (lds ? 2u : 4u);
}
size_t prof_bytes(const rt_scene *s, int lds) { return (size_t)(block_threads(s, lds) / 64) * PROF_SLOTS * 3u * sizeof(unsigned long long); }
// behind the image and the stacks: the world's sequence | the small tables (AUX kernels) | parked world rays | profile rows —
// the optional parts are taken in this order for as long as the CU's LDS has room (the instrumented kernel's rows included,
// so that it has the layout of the one it stands in for)
size_t align16(size_t x) { return (x + 15u) & ~(size_t)15u; }
size_t seq_offset(const rt_scene *s, int lds) { return align16(lds_image_bytes_for(s, lds) + stack_bytes(s, lds)); }
size_t aux_offset(const rt_scene *s, int lds) { return align16(seq_offset(s, lds) + (s->ordered ? (size_t)s->n_oseq * sizeof(OSeq) : 0u)); }
bool aux_in_lds(const rt_scene *s, int lds) {
    return s->ordered && s->aux_bytes != 0 && aux_offset(s, lds) + s->aux_bytes + prof_bytes(s, lds) <= LDS_BUDGET_BYTES;
}
size_t world_offset(const rt_scene *s, int lds) { return align16(aux_offset(s, lds) + (aux_in_lds(s, lds) ? s->aux_bytes : 0u)); }
size_t world_bytes(const rt_scene *s, int lds) { return (size_t)6 * sizeof(double) * (size_t)block_threads(s, lds); }
bool world_in_lds(const rt_scene *s, int lds) {
    return s->has_instances && world_offset(s, lds) + world_bytes(s, lds) + prof_bytes(s, lds) <= LDS_BUDGET_BYTES;
}
size_t prof_offset(const rt_scene *s, int lds) { return world_offset(s, lds) + (world_in_lds(s, lds) ? world_bytes(s, lds) : 0u); }
size_t dynamic_lds_bytes(const rt_scene *s, int lds, bool counted) {
    return prof_offset(s, lds) + (counted ? prof_bytes(s, lds) : 0);
}

template <class T> int upload(DeviceArray<T> &dst, const std::vector<T> &src) {
    dst.bytes = src.size() * sizeof(T);
    // never hand the kernel a null table: allocate at least one element
    const size_t alloc = dst.bytes ? dst.bytes : sizeof(T);
    HIP_TRY(hipMalloc((void **)&dst.ptr, alloc));
    if (dst.bytes) HIP_TRY(hipMemcpy(dst.ptr, src.data(), dst.bytes, hipMemcpyHostToDevice));
    else HIP_TRY(hipMemset(dst.ptr, 0, alloc));
    return RT_OK;
}

bool texture_needs_uv(const std::vector<rt_texture> &texs, int32_t t, int depth = 0) {
    if (t < 0 || depth > 16) return false;
    const rt_texture &x = texs[(size_t)t];
    if (x.kind == RT_TEXTURE_IMAGE) return true;
    if (x.kind == RT_TEXTURE_CHECKER) return texture_needs_uv(texs, x.even, depth + 1) || texture_needs_uv(texs, x.odd, depth + 1);
    return false;
}

// Perlin turbulence anywhere under texture t (seven noise evaluations per lookup: by far the dearest thing a hit can ask for)
bool texture_has_noise(const std::vector<rt_texture> &texs, int32_t t, int depth = 0) {
    if (t < 0 || depth > 16) return false;
    const rt_texture &x = texs[(size_t)t];
    if (x.kind == RT_TEXTURE_NOISE) return true;
    if (x.kind == RT_TEXTURE_CHECKER) return texture_has_noise(texs, x.even, depth + 1) || texture_has_noise(texs, x.odd, depth + 1);
    return false;
}

} // namespace

namespace rtapi {
void free_workspace(Workspace &w) {
    for (int h = 0; h < 2; ++h) {
        (void)hipFree(w.half[h].att_stack); (void)hipFree(w.half[h].att_ids); (void)hipFree(w.half[h].samples); (void)hipFree(w.half[h].world_slots); (void)hipFree(w.half[h].job_counter);
        if (w.aux[h]) (void)hipStreamDestroy(w.aux[h]);
        if (w.ev_sum[h]) (void)hipEventDestroy(w.ev_sum[h]);
    }
    if (w.ev_start) (void)hipEventDestroy(w.ev_start);
    if (w.ev_done) (void)hipEventDestroy(w.ev_done);
    (void)hipFree(w.counters);
    w = Workspace();
}
} // namespace rtapi

namespace {

void free_scene(rt_scene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (auto &kv : s->workspaces) free_workspace(kv.second->w);
    (void)hipFree(s->nodes.ptr); (void)hipFree(s->spheres.ptr); (void)hipFree(s->quads.ptr); (void)hipFree(s->insts.ptr);
    (void)hipFree(s->media.ptr); (void)hipFree(s->mats.ptr); (void)hipFree(s->texs.ptr); (void)hipFree(s->perlins.ptr);
    (void)hipFree(s->images.ptr); (void)hipFree(s->texels.ptr); (void)hipFree(s->lut.ptr); (void)hipFree(s->lds_image.ptr); (void)hipFree(s->oimage.ptr); (void)hipFree(s->oseq.ptr); (void)hipFree(s->aux_image.ptr);
    delete s;
}

int64_t tiles_total(int32_t w, int32_t h) {
    return (int64_t)((w + RT_TILE_W - 1) / RT_TILE_W) * ((h + RT_TILE_H - 1) / RT_TILE_H);
}
int64_t tiles_local(int32_t w, int32_t h, int32_t shard_index, int32_t shard_count) {
    return (tiles_total(w, h) - shard_index + shard_count - 1) / shard_count;
}

int normalise_params(const rt_camera *cam, rt_render_params &p) {
    if (cam->image_width <= 0 || cam->image_height <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: empty image");
    if ((int64_t)cam->image_width * cam->image_height > 0x7fffffffll)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: image has more than 2^31 pixels");
    if (p.sample_end <= 0) p.sample_end = cam->samples_per_pixel;
    if (p.max_depth <= 0) p.max_depth = cam->max_depth;
    if (p.shard_count <= 0) p.shard_count = 1;
    if (p.sample_begin < 0 || p.sample_end < p.sample_begin) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: bad sample range");
    if (p.shard_index < 0 || p.shard_index >= p.shard_count) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: shard_index out of range");
    if (p.out_layout != RT_OUT_FRAME && p.out_layout != RT_OUT_TILES) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: unknown out_layout");
    if (p.max_depth <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: max_depth must be positive");
    return RT_OK;
}

// the process defaults of this moment with the scene's own options (rt_scene_create_ex) on top
Tuning effective_tuning(const rt_scene *scene) {
    Tuning tn = tuning_snapshot();
    const rt_scene_options &o = scene->options;
    if (o.use_lds >= 0) tn.use_lds = o.use_lds;
    const int32_t th[5] = {o.th_prim, o.th_other, o.th_shade, o.th_box, o.th_new};
    for (int k = 0; k < 5; ++k)
        if (th[k] >= 0 && tn.forced[k] < 0) tn.forced[k] = th[k];
    if (o.sample_buffer_bytes > 0) tn.sample_buffer_bytes = (size_t)o.sample_buffer_bytes;
    if (o.start_shortcut >= 0) tn.start_shortcut = o.start_shortcut;
    if (o.defer_instances >= 0) tn.defer = o.defer_instances;
    if (o.seq_lookahead >= 0) tn.seq_lookahead = o.seq_lookahead;
    if (o.slow_min >= 1) tn.slow_min = o.slow_min;
    if (o.slow_age >= 0) tn.slow_age = o.slow_age;
    if (o.quad_filter >= 0) tn.quad_filter = o.quad_filter;
    if (o.medium_first >= 0) tn.medium_first = o.medium_first;
    return tn;
}

constexpr size_t MAX_WORKSPACES = 4; // per scene: one per stream in use; beyond that the idle ones are released

int launch_render(rt_scene *scene, const rt_camera *camera, rt_render_params p, double *d_out, hipStream_t stream,
                  rt_counters *out_counters) {
    int rc = normalise_params(camera, p);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipSetDevice(scene->device));
    const bool counted = out_counters != nullptr;
    const int64_t n_local = tiles_local(camera->image_width, camera->image_height, p.shard_index, p.shard_count);
    const int64_t n_samples_total = (int64_t)p.sample_end - p.sample_begin;
    if (n_local <= 0 || n_samples_total <= 0) {
        if (out_counters) *out_counters = rt_counters{};
        return RT_OK;
    }
    const Tuning tn = effective_tuning(scene);

    // samples per launch: bounded by the sample buffer and by the 32-bit job index.  A frame that needs more than one launch is
    // pipelined over two scratch sets (half the budget each) unless that is switched off or the kernel is the instrumented one.
    const int64_t bytes_per_sample_row = n_local * 64 * 3 * (int64_t)sizeof(double);
    const int64_t max_by_index = ((int64_t)1 << 31) / (n_local * 64);
    auto chunk_for = [&](size_t budget) {
        int64_t c = (int64_t)(budget / (size_t)bytes_per_sample_row);
        if (c > max_by_index) c = max_by_index;
        if (c > n_samples_total) c = n_samples_total;
        return c;
    };
    int64_t chunk = chunk_for(tn.sample_buffer_bytes);
    if (chunk < 1) return fail(RT_ERR_UNSUPPORTED, "rt_render: one sample per pixel does not fit the sample buffer");
    bool pipelined = !counted && tn.overlap != 0 && chunk < n_samples_total && chunk_for(tn.sample_buffer_bytes / 2) >= 1;
    if (pipelined) chunk = chunk_for(tn.sample_buffer_bytes / 2);

    const int lds = tn.use_lds != 0 ? scene->lds_level : 0;
    const int threads = block_threads(scene, lds);
    const int bpc = scene->blocks_per_cu[lds][counted ? 1 : 0];
    const size_t dyn_lds = dynamic_lds_bytes(scene, lds, counted);
    // persistent grid: every resident wave pulls jobs until none are left
    int64_t grid = (int64_t)scene->n_cus * bpc;
    const int64_t waves_per_block = threads / 64;
    // (a small frame gets a smaller grid: a wave with fewer than MIN_JOBS_PER_WAVE jobs costs more to start than it adds)
    const int64_t max_useful = (n_local * 64 * chunk + MIN_JOBS_PER_WAVE * waves_per_block - 1) / (MIN_JOBS_PER_WAVE * waves_per_block);
    if (grid > max_useful) grid = max_useful;
    if (grid < 1) grid = 1;
    const uint32_t n_threads = (uint32_t)(grid * threads);

    // The scratch of this (scene, stream).  Under the scene's lock: the table look-up, the in-use mark, and — a new stream and a full
    // table — taking the idle slots OUT of the table.  Everything that can wait (an evicted slot's last render, a drain before a buffer
    // grows, hipFree / hipMalloc) and the enqueues happen outside it, under the slot's own mutex: renders on other streams go on.
    // Only slots nobody holds are ever evicted, after waiting for the event their last render recorded (a library-owned handle:
    // the caller's stream may be gone by then).
    Workspace ws;
    WorkspaceSlot *slot = nullptr;
    std::vector<std::unique_ptr<WorkspaceSlot>> evicted;
    {
        std::lock_guard<std::mutex> lock(scene->mu);
        if (scene->workspaces.find(stream) == scene->workspaces.end() && scene->workspaces.size() >= MAX_WORKSPACES) {
            // (slots in use stay: the table then grows past its nominal size rather than pull scratch from under a render)
            for (auto it = scene->workspaces.begin(); it != scene->workspaces.end();) {
                if (it->second->in_use != 0) { ++it; continue; }
                evicted.push_back(std::move(it->second));
                it = scene->workspaces.erase(it);
            }
        }
        std::unique_ptr<WorkspaceSlot> &entry = scene->workspaces[stream];
        if (!entry) entry.reset(new WorkspaceSlot());
        slot = entry.get();
        slot->in_use++;
    }
    struct Release { // (declared before the slot's lock: runs after it is dropped)
        rt_scene *scene; WorkspaceSlot *slot; hipStream_t stream;
        bool enqueued = false, completed = false;
        ~Release() {
            // a render that failed after some launches were enqueued: what is in flight is waited for and the slot's event recorded all the
            // same, so that a later eviction never frees scratch a kernel still uses
            if (enqueued && !completed) {
                for (int h = 0; h < 2; ++h) if (slot->w.aux[h]) (void)hipStreamSynchronize(slot->w.aux[h]);
                if (slot->w.ev_done) (void)hipEventRecord(slot->w.ev_done, stream);
                (void)hipGetLastError();
            }
            std::lock_guard<std::mutex> lock(scene->mu);
            if (slot->in_use > 0) slot->in_use--;
        }
    } release{scene, slot, stream};
    for (auto &old : evicted) {
        if (old->w.ev_done) (void)hipEventSynchronize(old->w.ev_done);
        free_workspace(old->w);
    }
    if (!evicted.empty()) (void)hipGetLastError();
    evicted.clear();
    const bool colours = scene->parks_colours;
    std::unique_lock<std::mutex> slot_lock(slot->mu); // held to the end of the enqueues
    {
        Workspace &w = slot->w;
        const size_t need_att = colours ? ((size_t)p.max_depth + 1u) * n_threads * 3u * sizeof(double) : sizeof(double); // + a light's emitted colour
        if (need_att / sizeof(double) >= ((size_t)1 << 32)) return fail(RT_ERR_UNSUPPORTED, "rt_render: max_depth too large for the attenuation stack's 32-bit indices");
        const size_t need_ids = ((size_t)p.max_depth + 1u) * n_threads * sizeof(uint32_t);
        if (need_ids / sizeof(uint32_t) >= ((size_t)1 << 32)) return fail(RT_ERR_UNSUPPORTED, "rt_render: max_depth too large for the index stack's 32-bit indices");
        const size_t need_world = (size_t)6 * n_threads * sizeof(double);
        if (!w.ev_done) HIP_TRY(hipEventCreateWithFlags(&w.ev_done, hipEventDisableTiming));
        if (pipelined && !w.aux[0]) { // both streams and all three events, or none of them
            hipStream_t st[2] = {nullptr, nullptr};
            hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
            hipError_t e = hipSuccess;
            for (int h = 0; h < 2 && e == hipSuccess; ++h) e = hipStreamCreateWithFlags(&st[h], hipStreamNonBlocking);
            for (int h = 0; h < 3 && e == hipSuccess; ++h) e = hipEventCreateWithFlags(&ev[h], hipEventDisableTiming);
            if (e != hipSuccess) {
                for (int h = 0; h < 2; ++h) if (st[h]) (void)hipStreamDestroy(st[h]);
                for (int h = 0; h < 3; ++h) if (ev[h]) (void)hipEventDestroy(ev[h]);
                return fail(RT_ERR_HIP, std::string("rt_render: internal streams: ") + hipGetErrorString(e));
            }
            w.aux[0] = st[0]; w.aux[1] = st[1]; w.ev_sum[0] = ev[0]; w.ev_sum[1] = ev[1]; w.ev_start = ev[2];
        }
        // (an earlier out-of-memory back-off on this stream is remembered: the buffer it arrived at is the budget from then on)
        if (w.sample_budget > 0) {
            const int64_t c = chunk_for(w.sample_budget);
            if (c >= 1 && c < chunk) chunk = c;
        }
        // (every earlier launch on this stream ends with the stream waiting for the internal ones: draining it drains them)
        bool drained = false;
        auto drain = [&]() -> int { if (!drained) { HIP_TRY(hipStreamSynchronize(stream)); drained = true; } return RT_OK; };
        for (int h = 0; h < (pipelined ? 2 : 1); ++h) {
            LaunchScratch &x = w.half[h];
            if (x.att_bytes < need_att) {
                if ((rc = drain()) != RT_OK) return rc;
                if (x.att_stack) HIP_TRY(hipFree(x.att_stack));
                x.att_stack = nullptr; x.att_bytes = 0;
                HIP_TRY(hipMalloc((void **)&x.att_stack, need_att));
                x.att_bytes = need_att;
            }
            if (x.att_ids_bytes < need_ids) {
                if ((rc = drain()) != RT_OK) return rc;
                if (x.att_ids) HIP_TRY(hipFree(x.att_ids));
                x.att_ids = nullptr; x.att_ids_bytes = 0;
                HIP_TRY(hipMalloc((void **)&x.att_ids, need_ids));
                x.att_ids_bytes = need_ids;
            }
            if (x.world_bytes < need_world) {
                if ((rc = drain()) != RT_OK) return rc;
                if (x.world_slots) HIP_TRY(hipFree(x.world_slots));
                x.world_slots = nullptr; x.world_bytes = 0;
                HIP_TRY(hipMalloc((void **)&x.world_slots, need_world));
                x.world_bytes = need_world;
            }
            size_t need_samples = (size_t)bytes_per_sample_row * (size_t)chunk;
            if (x.sample_bytes < need_samples) {
                if ((rc = drain()) != RT_OK) return rc;
                if (x.samples) HIP_TRY(hipFree(x.samples));
                x.samples = nullptr; x.sample_bytes = 0;
                // a device short of memory gets a smaller chunk (more launches), not an error
                for (;;) {
                    const hipError_t e = hipMalloc((void **)&x.samples, need_samples);
                    if (e == hipSuccess) break;
                    (void)hipGetLastError();
                    x.samples = nullptr;
                    if (e != hipErrorOutOfMemory || chunk <= 1)
                        return fail(e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, std::string("rt_render: sample buffer: ") + hipGetErrorString(e));
                    chunk = (chunk + 1) / 2;
                    need_samples = (size_t)bytes_per_sample_row * (size_t)chunk;
                    w.sample_budget = need_samples;
                }
                x.sample_bytes = need_samples;
            }
            if (!x.job_counter) HIP_TRY(hipMalloc((void **)&x.job_counter, sizeof(uint32_t)));
        }
        if (!w.counters) HIP_TRY(hipMalloc((void **)&w.counters, COUNTER_WORDS * sizeof(unsigned long long)));
        ws = w;
    }
    release.enqueued = true; // (from here on something may be in flight on `stream` or the internal streams)
    if (counted) HIP_TRY(hipMemsetAsync(ws.counters, 0, COUNTER_WORDS * sizeof(unsigned long long), stream));

    KParams K{};
    K.nodes = scene->nodes.ptr; K.spheres = scene->spheres.ptr; K.quads = scene->quads.ptr; K.insts = scene->insts.ptr;
    K.media = scene->media.ptr; K.mats = scene->mats.ptr; K.texs = scene->texs.ptr; K.perlins = scene->perlins.ptr;
    K.images = scene->images.ptr; K.texels = scene->texels.ptr; K.srgb_lut = scene->lut.ptr;
    K.out = d_out;
    K.counters = counted ? ws.counters : nullptr;
    K.cam = *camera;
    K.seed_mixed = host_mix64(p.seed + 0x9E3779B97F4A7C15ull);
    K.n_nodes = scene->n_nodes;
    K.n_threads = n_threads;
    K.id_one = (uint32_t)(scene->mats.bytes / sizeof(DMaterial)) - 1u;
    K.ids_ok = K.id_one < 0xffffu ? 1u : 0u;
    K.max_depth = p.max_depth;
    K.shard_index = p.shard_index; K.shard_count = p.shard_count; K.out_layout = p.out_layout;
    K.tiles_x = (camera->image_width + RT_TILE_W - 1) / RT_TILE_W;
    K.inv_tiles_x = 1.0 / (double)K.tiles_x;
    K.n_local_tiles = (uint32_t)n_local;
    K.lds_image = scene->lds_image.ptr; K.lds_image_bytes = lds_image_bytes_for(scene, lds);
    K.lds_off_node_b = scene->lds_off_node_b;
    K.lds_off_spheres = scene->lds_off_spheres; K.lds_off_quads = scene->lds_off_quads;
    K.lds_off_qfilt = (lds == 3 && tn.quad_filter != 0 && scene->lds_off_qfilt != 0) ? scene->lds_off_qfilt : 0xffffffffu;
    K.lds_world_off = world_in_lds(scene, lds) ? (uint32_t)world_offset(scene, lds) : 0xffffffffu;
    K.box_extent = scene->box_extent;
    K.seq_lookahead = tn.seq_lookahead ? 1u : 0u;
    K.medium_first = tn.medium_first ? 1u : 0u;
    K.inst_shortcut = tn.start_shortcut ? 1u : 0u;
    K.slow_min = (uint32_t)tn.slow_min; K.slow_age = (uint32_t)tn.slow_age;
    K.o_start_stage = tn.start_shortcut ? scene->o_start_stage : 0u; K.o_start_prim = scene->o_start_prim; K.o_start_end = scene->o_start_end;
    K.o_start_rest = scene->o_start_rest; K.o_start_slot = scene->o_start_slot;
    K.oimage = scene->oimage.ptr; K.o_root = scene->o_root; K.oseq = scene->oseq.ptr; K.n_oseq = scene->n_oseq; K.lds_stack_off = lds_image_bytes_for(scene, lds);
    K.lds_seq_off = (uint32_t)seq_offset(scene, lds);
    K.aux_image = scene->aux_image.ptr; K.aux_bytes = scene->aux_bytes; K.lds_aux_off = (uint32_t)aux_offset(scene, lds);
    K.aux_off_mats = scene->aux_off[0]; K.aux_off_texs = scene->aux_off[1]; K.aux_off_insts = scene->aux_off[2];
    K.aux_off_media = scene->aux_off[3]; K.aux_off_perlins = scene->aux_off[4];
    K.lds_prof_off = (uint32_t)prof_offset(scene, lds);
    {
        const uint32_t kf = kernel_features_for(scene->features, lds, scene->ordered);
        const Thresholds th = tn.pick(kf == FEAT_SPHERES_SOLID ? (scene->ordered ? tn.spheres_solid : tn.spheres_threaded) : (kf == FEAT_QUADS_FRAMES ? (scene->insts.bytes ? tn.quads_frames : tn.quads_only) : (scene->ordered ? (lds == 0 ? tn.ordered_global : tn.ordered_general) : tn.general)));
        K.th_prim = th.prim; K.th_other = th.other; K.th_shade = th.shade; K.th_box = th.box; K.th_new = th.newjob;
        auto byte = [](uint32_t v) { return v > 255u ? 255u : v; };
        K.th_pack = byte(th.prim) | (byte(th.other) << 8) | (byte(th.shade) << 16) | (byte(th.box) << 24);
    }
    // (a lane keeps the instances it has yet to walk as one 32-bit mask of their indices)
    K.defer_instances = (scene->ordered && tn.defer != 0 && scene->insts.bytes / sizeof(Instance) <= 32u) ? 1u : 0u;

    // Launch k renders samples [sb, sb + ns) into its scratch set's sample buffer; sum_samples_kernel then adds them onto `out`
    // in sample order.  Pipelined: launch k runs on internal stream k % 2; its summation waits for launch k - 1's (the sums
    // stay in order) while the render kernel of launch k + 1, on the other stream, takes the CUs that launch k's tail frees.
    if (pipelined) {
        HIP_TRY(hipEventRecord(ws.ev_start, stream));
        for (int h = 0; h < 2; ++h) HIP_TRY(hipStreamWaitEvent(ws.aux[h], ws.ev_start, 0));
    }
    const unsigned sum_grid = (unsigned)((n_local * 64 + 255) / 256);
    int64_t k = 0;
    for (int64_t sb = p.sample_begin; sb < p.sample_end; sb += chunk, ++k) {
        const int64_t ns = (p.sample_end - sb) < chunk ? (p.sample_end - sb) : chunk;
        const int h = pipelined ? (int)(k & 1) : 0;
        const hipStream_t s = pipelined ? ws.aux[h] : stream;
        const LaunchScratch &x = ws.half[h];
        K.samples = x.samples; K.att_stack = x.att_stack; K.att_ids = x.att_ids; K.job_counter = x.job_counter; K.world_slots = x.world_slots;
        K.sample_begin = (int32_t)sb;
        K.n_samples = (uint32_t)ns;
        K.n_jobs = (uint32_t)(n_local * 64 * ns);
        K.inv_n_samples = 1.0 / (double)ns;
        {
            // ~32 grabs per wave or more, rounded down to a multiple of 64 within [MIN, MAX]
            const int64_t waves = grid * waves_per_block;
            int64_t per_grab = (int64_t)K.n_jobs / (waves * 32);
            if (tn.jobs_per_grab > 0) per_grab = tn.jobs_per_grab;
            per_grab = per_grab / 64 * 64;
            if (per_grab > MAX_JOBS_PER_GRAB) per_grab = MAX_JOBS_PER_GRAB;
            if (per_grab < MIN_JOBS_PER_GRAB) per_grab = MIN_JOBS_PER_GRAB;
            K.jobs_per_grab = (uint32_t)per_grab;
            // (RT_GRAB_TAPER=0 switches the taper off: every grab is jobs_per_grab)
            K.grab_taper = tn.grab_taper > 0 ? 1.0f / (float)(waves * tn.grab_taper) : 1.0f; // (default 8, tools/sweep_grabs.sh)
        }
        K.accumulate = (p.accumulate || sb > p.sample_begin) ? 1 : 0;
        HIP_TRY(hipMemsetAsync(x.job_counter, 0, sizeof(uint32_t), s));
        {
            void *args[] = {(void *)&K};
            const uint32_t kf = kernel_features_for(scene->features, lds, scene->ordered);
            const void *fn = path_kernel_for(lds, counted, kf, scene->ordered, aux_in_lds(scene, lds), scene->wide);
            HIP_TRY(hipLaunchKernel(fn, dim3((unsigned)grid), dim3(threads), args, dyn_lds, s));
        }
        HIP_TRY(hipGetLastError());
        if (pipelined && k > 0) HIP_TRY(hipStreamWaitEvent(s, ws.ev_sum[h ^ 1], 0));
        launch_sum_samples(K, sum_grid, s);
        HIP_TRY(hipGetLastError());
        if (pipelined) HIP_TRY(hipEventRecord(ws.ev_sum[h], s));
    }
    if (pipelined) HIP_TRY(hipStreamWaitEvent(stream, ws.ev_sum[(k - 1) & 1], 0)); // (the last sum waited for all before it)
    HIP_TRY(hipEventRecord(ws.ev_done, stream));
    release.completed = true;
    g_last_launch[0] = 0u; g_last_launch[1] = (uint32_t)lds; g_last_launch[2] = (uint32_t)threads; g_last_launch[3] = (uint32_t)grid;

    if (counted) {
        unsigned long long host[COUNTER_WORDS];
        HIP_TRY(hipMemcpyAsync(host, ws.counters, sizeof host, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        {
            std::lock_guard<std::mutex> lock(g_stage_profile_mu);
            for (uint32_t q = 0; q < PROF_SLOTS * 3u; ++q) g_stage_profile[q] = host[10 + q];
        }
        out_counters->samples = host[0]; out_counters->rays = host[1]; out_counters->node_visits = host[2];
        out_counters->sphere_tests = host[3]; out_counters->quad_tests = host[4]; out_counters->medium_visits = host[5];
        out_counters->rng_draws = host[6]; out_counters->noise_evals = host[7]; out_counters->image_lookups = host[8];
        out_counters->instance_enters = host[9];
    }
    return RT_OK;
}

} // namespace

namespace rtapi {
int resolve_scene_options(const rt_scene_options *options, rt_scene_options &opt, const char *who) {
    rt_scene_options_init(&opt);
    if (options) {
        const uint32_t size = options->struct_size;
        if (size < 8 || size > sizeof opt || size % 4 != 0) return fail(RT_ERR_INVALID_ARGUMENT, std::string(who) + ": rt_scene_options.struct_size is not one this library knows");
        memcpy(&opt, options, size); // (an older, shorter struct: the fields it lacks keep their defaults)
        opt.struct_size = (uint32_t)sizeof opt;
        // the 56-byte struct of round 2 ended in a reserved word that its init zeroed, where flat_max (0: off) now is
        if (size <= 56) opt.flat_max = -1;
    }
    if (opt.walk < RT_WALK_DEFAULT || opt.walk > RT_WALK_OWN_TREES) return fail(RT_ERR_INVALID_ARGUMENT, std::string(who) + ": unknown walk");
    return RT_OK;
}
} // namespace rtapi

extern "C" {

const char *rt_last_error(void) { return g_last_error.c_str(); }
const char *rt_version(void) { return "rt_amd 0.4 (gfx950, abi 2: rt_scene_options grew to 88 bytes, older sizes accepted)"; }

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int64_t rt_out_size(int32_t width, int32_t height, int32_t out_layout, int32_t shard_index, int32_t shard_count) {
    if (width <= 0 || height <= 0) return -1;
    if (shard_count <= 0) shard_count = 1;
    if (shard_index < 0 || shard_index >= shard_count) return -1;
    if (out_layout == RT_OUT_FRAME) return (int64_t)width * height * 3;
    if (out_layout == RT_OUT_TILES) return tiles_local(width, height, shard_index, shard_count) * RT_TILE_W * RT_TILE_H * 3;
    return -1;
}

// Which walk for this scene?  Measured on MI355X at the in-code cameras (tools/scene_speed.py, profiles/r02_scene_speed.txt), own trees
// vs reference order, Msamples/s: random_balls 4416 / 2128, two_spheres 8739 / 7281, earth 22557 / 22523 (one primitive: a tree and a
// stack are pure overhead), two_perlin_spheres 4821 / 3288, quads 12860 / 12612, simple_light 5845 / 4894, cornell_box 3224 / 2102,
// cornell_smoke 1261 / 966 (a tie until its frames' primitives went into flat leaves, rt_ordered.hpp flat_max), final_scene 983 / 643.
static bool ordered_walk_pays(const CompiledScene &cs) {
    return cs.spheres.size() + cs.quads.size() > 1;
}

void rt_scene_options_init(rt_scene_options *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->struct_size = (uint32_t)sizeof *o;
    o->walk = RT_WALK_DEFAULT; o->leaf_max = 0; o->refit = -1; o->use_lds = -1;
    o->th_prim = o->th_other = o->th_shade = o->th_box = o->th_new = -1;
    o->sample_buffer_bytes = 0;
    o->reserved_pool = -1;
    o->flat_max = o->start_shortcut = o->defer_instances = o->seq_lookahead = o->slow_min = o->slow_age = o->wide = -1;
    o->quad_filter = -1;
    o->medium_first = -1;
}

// (for a caller compiled against an older, shorter struct: nothing beyond ITS size is written)
int rt_scene_options_init_sized(rt_scene_options *o, uint32_t struct_size) {
    if (!o) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_options_init_sized: null argument");
    if (struct_size < 8 || struct_size > sizeof(rt_scene_options) || struct_size % 4 != 0)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_options_init_sized: struct_size is not one this library knows");
    rt_scene_options full;
    rt_scene_options_init(&full);
    full.struct_size = struct_size;
    memcpy(o, &full, struct_size);
    return RT_OK;
}

int rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out_scene) { return rt_scene_create_ex(desc, device, nullptr, out_scene); }

int rt_scene_create_ex(const rt_scene_desc *desc, int device, const rt_scene_options *options, rt_scene **out_scene) {
    if (!desc || !out_scene) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: null argument");
    *out_scene = nullptr;
    rt_scene_options opt;
    if (int orc = resolve_scene_options(options, opt, "rt_scene_create_ex")) return orc;
    // the process defaults in force now (RT_* variables, rt_debug_set_*), then the caller's options
    const Tuning tn = tuning_snapshot();
    const int walk = opt.walk != RT_WALK_DEFAULT ? opt.walk : tn.ordered;
    const bool refit = opt.refit >= 0 ? opt.refit != 0 : tn.refit != 0;
    OrderedOptions oopt = tn.ordered_options;
    if (opt.leaf_max > 0) oopt.leaf_max = (uint32_t)opt.leaf_max < OREF_MAX_LEAF ? (uint32_t)opt.leaf_max : OREF_MAX_LEAF;
    if (opt.flat_max >= 0) oopt.flat_max = (uint32_t)opt.flat_max;
    const int want_wide = opt.wide >= 0 ? opt.wide : tn.wide; // (-1: where it measured faster, below)
    CompiledScene cs;
    try {
        cs = compile_scene(*desc, refit);
        // Four children per record where the trees are big enough for the halved number of visits to pay for the dearer visit
        // (MI355X, Msamples/s two / four children: random-spheres 5428 / 5437, final_scene 1049 / 1088; Cornell's 4 records: 2438 / 2345)
        oopt.wide = want_wide >= 0 ? want_wide != 0 : cs.spheres.size() + cs.quads.size() >= 64;
        if (walk == RT_WALK_OWN_TREES || (walk == RT_WALK_AUTO && ordered_walk_pays(cs))) build_ordered(cs, oopt);
    } catch (const CompileError &e) {
        return fail(e.status, e.what());
    } catch (const std::exception &e) {
        return fail(RT_ERR_INVALID_ARGUMENT, std::string("rt_scene_create: ") + e.what());
    }
    const int ndev = rt_device_count();
    if (ndev <= 0) return fail(RT_ERR_NO_DEVICE, "rt_scene_create: no HIP device is visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(RT_ERR_NO_DEVICE, "rt_scene_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(device));

    rt_scene *s = new rt_scene();
    s->device = device;
    s->options = opt;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete s; return fail(RT_ERR_HIP, "hipGetDeviceProperties failed"); }
    s->n_cus = prop.multiProcessorCount;
    s->has_instances = !cs.instances.empty();
    s->features = (cs.spheres.empty() ? 0u : F_SPHERES) | (cs.quads.empty() ? 0u : F_QUADS) | (cs.instances.empty() ? 0u : F_FRAMES) |
                  (cs.media.empty() ? 0u : F_MEDIA);
    for (const auto &t : cs.textures)
        if (t.kind != RT_TEXTURE_SOLID) s->features |= F_TEXTURES;
    // attenuations are parked as material indices unless the colour is a texture's value, or the indices do not fit 16 bits: such a
    // scene is rendered by the kernels that can park colours, i.e. those with textures
    if (cs.materials.size() >= 0xffffu) s->features |= F_TEXTURES;
    s->parks_colours = (s->features & F_TEXTURES) != 0;
    s->ordered = cs.ordered;
    // a walk starts in the first step's tree; a sequence that starts with a medium goes through ST_OTHER first
    s->o_root = cs.ordered && cs.oseq[0].kind == OSEQ_TREE ? cs.oseq[0].a : 0xfffffffeu;
    s->n_oseq = (uint32_t)cs.oseq.size();
    // Start shortcut.  random-spheres' ground sphere (r = 1000) spans the scene, so the sweep puts it directly under the root and every
    // ray's origin lies inside its box — the walk visits the root, finds that leaf "nearest", tests the sphere, comes back for the other
    // child; a frame of a few primitives keeps them in one leaf under its root (rt_ordered.hpp flat_max: Cornell's walls).  When the root
    // has such a child (a leaf whose box has at least half the area of the root's), a query starts in the leaf's primitive stage with the
    // other child already set aside: the same tests, minus the visit of the root record, and the lanes that start together stay together.
    s->wide = cs.ordered && cs.wide;
    if (cs.ordered && cs.media.empty() && cs.oseq.size() == 1 && cs.oseq[0].kind == OSEQ_TREE) {
        auto half_area = [](const float *b) { const double x = (double)b[1] - b[0], y = (double)b[3] - b[2], z = (double)b[5] - b[4]; return x * y + y * z + z * x; };
        // the root's children: (box, reference) in slot order — two of a binary record, up to four of a wide one
        std::vector<std::pair<const float *, uint32_t>> kids;
        if (s->wide) {
            const ONode4 &root = cs.onodes4[s->o_root];
            for (int k = 0; k < 4; ++k) kids.emplace_back(root.b[k], root.c[k]);
        } else {
            const ONode &root = cs.onodes[s->o_root];
            kids.emplace_back(root.b0, root.c[0]);
            kids.emplace_back(root.b1, root.c[1]);
        }
        float all[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
        for (const auto &kid : kids)
            if ((kid.second >> OREF_KIND_SHIFT) != OK_EMPTY)
                for (int k = 0; k < 6; k += 2) { all[k] = std::fmin(all[k], kid.first[k]); all[k + 1] = std::fmax(all[k + 1], kid.first[k + 1]); }
        for (uint32_t slot = 0; slot < kids.size(); ++slot) {
            const uint32_t ref = kids[slot].second, kind = ref >> OREF_KIND_SHIFT;
            if ((kind == OK_SPHERES || kind == OK_QUADS) && half_area(kids[slot].first) >= 0.5 * half_area(all)) {
                s->o_start_stage = kind; // (OrderedKind SPHERES / QUADS = Stage ST_SPHERE / ST_QUAD)
                s->o_start_prim = ref & OREF_INDEX_MASK;
                s->o_start_end = s->o_start_prim + ((ref >> OREF_COUNT_SHIFT) & OREF_COUNT_MASK) + 1u;
                s->o_start_slot = slot;
                if (s->wide) { // what is set aside: the root record with the mask of its other children
                    uint32_t mask = 0;
                    for (uint32_t k = 0; k < 4; ++k)
                        if (k != slot && (kids[k].second >> OREF_KIND_SHIFT) != OK_EMPTY) mask |= 1u << k;
                    s->o_start_rest = mask;
                } else {
                    s->o_start_rest = kids[slot ^ 1u].second;
                }
                break;
            }
        }
    }
    // The same for a Translate / RotateY frame whose tree is a single leaf (a box's six faces in a flat leaf): a walk that enters the frame
    // starts with the leaf's primitives; the visit of a root record with one child is skipped (Cornell: 2.6 of 9.2 record visits per sample).
    if (cs.ordered)
        for (Instance &in : cs.instances) {
            uint32_t only = 0, n = 0;
            const float *box = nullptr;
            auto look = [&](uint32_t ref, const float *b) { if ((ref >> OREF_KIND_SHIFT) != OK_EMPTY) { ++n; only = ref; box = b; } };
            if (s->wide) for (int k = 0; k < 4; ++k) look(cs.onodes4[in.root].c[k], cs.onodes4[in.root].b[k]);
            else { look(cs.onodes[in.root].c[0], cs.onodes[in.root].b0); look(cs.onodes[in.root].c[1], cs.onodes[in.root].b1); }
            const uint32_t kind = only >> OREF_KIND_SHIFT;
            in.start_ref = (n == 1 && (kind == OK_SPHERES || kind == OK_QUADS)) ? only : 0u;
            if (in.start_ref) memcpy(in.start_box, box, sizeof in.start_box);
        }
    for (const ONode &nd : cs.onodes)
        for (int k = 0; k < 6; ++k) {
            if ((nd.c[0] >> OREF_KIND_SHIFT) != OK_EMPTY) s->box_extent = std::fmax(s->box_extent, std::fabs(nd.b0[k]));
            if ((nd.c[1] >> OREF_KIND_SHIFT) != OK_EMPTY) s->box_extent = std::fmax(s->box_extent, std::fabs(nd.b1[k]));
        }
    for (const OSeq &st : cs.oseq)
        for (int k = 0; k < 6; ++k) s->box_extent = std::fmax(s->box_extent, std::fabs(st.box[k]));
    s->o_stack = cs.ordered_stack;
    // Node tables (load_node / load_opair): threaded records as two 16-byte halves, ordered records as six 16-byte plane
    // tables and an 8-byte reference table.  LDS image = node tables | spheres | quads; every LDS level copies a prefix.
    {
        const bool wide = cs.ordered && cs.wide;
        const size_t n = wide ? cs.onodes4.size() : (cs.ordered ? cs.onodes.size() : cs.nodes32.size());
        const size_t off_b = wide ? n * 32 : n * 16; // bytes of one plane table (16 bytes per record; wide records: 32)
        const size_t off_sph = wide ? n * (6 * 32 + 16) : (cs.ordered ? ((n * (6 * 16 + 8) + 15u) & ~(size_t)15u) : n * 32);
        std::vector<uint4> tables(off_sph / 16);
        {
            unsigned char *base = reinterpret_cast<unsigned char *>(tables.data());
            if (wide) { // (rt_device_scene.h load_oquad)
                for (size_t i = 0; i < n; ++i) {
                    const ONode4 &nd = cs.onodes4[i];
                    for (int ax = 0; ax < 3; ++ax) {
                        float plus[8], minus[8];
                        for (int k = 0; k < 4; ++k) {
                            plus[k] = nd.b[k][2 * ax]; plus[4 + k] = nd.b[k][2 * ax + 1];   // enter through lo, leave through hi
                            minus[k] = nd.b[k][2 * ax + 1]; minus[4 + k] = nd.b[k][2 * ax];
                        }
                        memcpy(base + (size_t)(2 * ax) * off_b + i * 32, plus, 32);
                        memcpy(base + (size_t)(2 * ax + 1) * off_b + i * 32, minus, 32);
                    }
                    memcpy(base + 6 * off_b + i * 16, nd.c, 16);
                }
            } else if (cs.ordered) {
                for (size_t i = 0; i < n; ++i) {
                    const ONode &nd = cs.onodes[i];
                    for (int ax = 0; ax < 3; ++ax) {
                        const float lo0 = nd.b0[2 * ax], hi0 = nd.b0[2 * ax + 1], lo1 = nd.b1[2 * ax], hi1 = nd.b1[2 * ax + 1];
                        const float plus[4] = {lo0, lo1, hi0, hi1}, minus[4] = {hi0, hi1, lo0, lo1};
                        memcpy(base + (size_t)(2 * ax) * off_b + i * 16, plus, 16);
                        memcpy(base + (size_t)(2 * ax + 1) * off_b + i * 16, minus, 16);
                    }
                    memcpy(base + 6 * off_b + i * 8, nd.c, 8);
                }
            } else {
                const unsigned char *src = reinterpret_cast<const unsigned char *>(cs.nodes32.data());
                for (size_t i = 0; i < n; ++i)
                    for (size_t q = 0; q < 2; ++q) memcpy(base + q * off_b + i * 16, src + (i * 2 + q) * 16, 16);
            }
        }
        if (wide) { // the global copy: 256 bytes per record (load_oquad<0>)
            std::vector<uint4> lines(n * 16);
            unsigned char *dst = reinterpret_cast<unsigned char *>(lines.data());
            const unsigned char *tab = reinterpret_cast<const unsigned char *>(tables.data());
            for (size_t i = 0; i < n; ++i) {
                for (size_t q = 0; q < 6; ++q) memcpy(dst + i * 256 + q * 32, tab + q * off_b + i * 32, 32);
                memcpy(dst + i * 256 + 192, tab + 6 * off_b + i * 16, 16);
            }
            int urc = upload(s->oimage, lines);
            if (urc != RT_OK) { free_scene(s); return urc; }
        } else if (cs.ordered) { // the global copy: one 128-byte line per record (load_opair<0>)
            std::vector<uint4> lines(n * 8);
            unsigned char *dst = reinterpret_cast<unsigned char *>(lines.data());
            const unsigned char *tab = reinterpret_cast<const unsigned char *>(tables.data());
            for (size_t i = 0; i < n; ++i) {
                for (size_t q = 0; q < 6; ++q) memcpy(dst + i * 128 + q * 16, tab + q * off_b + i * 16, 16);
                memcpy(dst + i * 128 + 96, tab + 6 * off_b + i * 8, 8);
            }
            int urc = upload(s->oimage, lines);
            if (urc != RT_OK) { free_scene(s); return urc; }
        }
        s->lds_off_node_b = (uint32_t)off_b;
        const size_t off_quads = off_sph + cs.spheres.size() * sizeof(Sphere);
        // behind the quads: their f32 filter records (rt_qfilt.hpp) — if some leaf of the library's own trees holds more than one quad
        // (a single quad's exact test runs in its round anyway) and the LDS has room for them
        bool flat_quads = false;
        if (cs.ordered) {
            auto multi = [](uint32_t ref) { return (ref >> OREF_KIND_SHIFT) == OK_QUADS && ((ref >> OREF_COUNT_SHIFT) & OREF_COUNT_MASK) != 0u; };
            for (const ONode &nd : cs.onodes) flat_quads = flat_quads || multi(nd.c[0]) || multi(nd.c[1]);
            for (const ONode4 &nd : cs.onodes4) for (int k = 0; k < 4; ++k) flat_quads = flat_quads || multi(nd.c[k]);
        }
        const size_t off_qfilt = (off_quads + cs.quads.size() * sizeof(Quad) + 15u) & ~(size_t)15u;
        const size_t total_plain = off_qfilt, total_filt = off_qfilt + cs.quads.size() * sizeof(QFiltPair);
        const size_t stack3 = stack_bytes(s, 3), stack1 = stack_bytes(s, 1); // (the two levels' kernels differ in workgroup size)
        // level 2 (nodes + spheres) exists but is not selected: on final_scene it measured 10 % slower than level 1
        // (behind the stacks: the world's sequence, and the instrumented kernels' profile rows)
        const size_t budget = LDS_BUDGET_BYTES - 4096 - cs.oseq.size() * sizeof(OSeq);
        const bool with_filter = flat_quads && total_filt + stack3 <= budget;
        const size_t total = with_filter ? total_filt : total_plain;
        s->lds_level = total + stack3 <= budget ? 3 : (off_sph + stack1 <= budget ? 1 : 0);
        if (cs.ordered && n >= (wide ? (size_t)WIDE_MAX_LDS_RECORDS : (size_t)0x3fffu)) s->lds_level = 0; // 2-byte stack entries: a record index in 14 bits + two skip bits (wide: 12 + 4 mask bits)
        if (s->lds_level) {
            const size_t used = s->lds_level == 3 ? total : (s->lds_level == 2 ? ((off_quads + 15u) & ~(size_t)15u) : off_sph);
            std::vector<uint4> img(used / 16);
            unsigned char *base = reinterpret_cast<unsigned char *>(img.data());
            memcpy(base, tables.data(), off_sph);
            if (s->lds_level >= 2 && !cs.spheres.empty()) memcpy(base + off_sph, cs.spheres.data(), cs.spheres.size() * sizeof(Sphere));
            if (s->lds_level == 3 && !cs.quads.empty()) memcpy(base + off_quads, cs.quads.data(), cs.quads.size() * sizeof(Quad));
            if (s->lds_level == 3 && with_filter) {
                const std::vector<QFiltPair> qf = qfilt_table(cs.quads);
                memcpy(base + off_qfilt, qf.data(), qf.size() * sizeof(QFiltPair));
                s->lds_off_qfilt = (uint32_t)off_qfilt;
            }
            int urc = upload(s->lds_image, img);
            if (urc != RT_OK) { free_scene(s); return urc; }
            s->lds_off_spheres = (uint32_t)off_sph; s->lds_off_quads = (uint32_t)off_quads;
            s->lds_image_bytes = (uint32_t)used;
            s->lds_prefix_bytes[1] = (uint32_t)off_sph;
            s->lds_prefix_bytes[2] = (uint32_t)((off_quads + 15u) & ~(size_t)15u);
            s->lds_prefix_bytes[3] = (uint32_t)total;
            for (int l = s->lds_level + 1; l < 4; ++l) s->lds_prefix_bytes[l] = 0;
        }
    }
    // (one entry more than the scene has materials: Color::ONE, what the path end multiplies by for a level a path does not have;
    // two more where the one's index would be 0xffff, the mark of a parked colour)
    const size_t n_mats = cs.materials.size();
    std::vector<DMaterial> mats(n_mats + (n_mats == 0xffffu ? 2u : 1u));
    for (size_t i = n_mats; i < mats.size(); ++i) {
        mats[i].albedo[0] = mats[i].albedo[1] = mats[i].albedo[2] = 1.0;
        mats[i].solid = 1u;
    }
    for (size_t i = 0; i < n_mats; ++i) {
        const rt_material &m = cs.materials[i];
        DMaterial d{};
        d.kind = (uint32_t)m.kind;
        d.texture = m.texture >= 0 ? (uint32_t)m.texture : 0u;
        d.needs_uv = texture_needs_uv(cs.textures, m.texture) ? 1u : 0u;
        d.slow = (m.kind != RT_MATERIAL_METAL && m.kind != RT_MATERIAL_DIELECTRIC && texture_has_noise(cs.textures, m.texture)) ? 1u : 0u;
        d.albedo[0] = m.albedo.x; d.albedo[1] = m.albedo.y; d.albedo[2] = m.albedo.z;
        if (m.kind != RT_MATERIAL_METAL && m.kind != RT_MATERIAL_DIELECTRIC && m.texture >= 0 &&
            cs.textures[(size_t)m.texture].kind == RT_TEXTURE_SOLID) {
            const rt_vec3 &c = cs.textures[(size_t)m.texture].color;
            d.solid = 1u;
            d.albedo[0] = c.x; d.albedo[1] = c.y; d.albedo[2] = c.z;
        }
        d.fuzz = m.fuzz;
        d.ir = m.ir;
        if (m.kind == RT_MATERIAL_DIELECTRIC) {
            // a Dielectric attenuates by Color::ONE and reads no albedo: the three doubles carry what its scatter() divides out on every
            // hit (src/material.rs:86, :75-77) — the same IEEE operations, done once here: 1 / ir, and Schlick's r0 for either face
            const double inv_ir = 1.0 / m.ir;
            double r0_front = (1.0 - inv_ir) / (1.0 + inv_ir), r0_back = (1.0 - m.ir) / (1.0 + m.ir);
            r0_front = r0_front * r0_front; r0_back = r0_back * r0_back;
            d.albedo[0] = inv_ir; d.albedo[1] = r0_front; d.albedo[2] = r0_back;
        }
        mats[i] = d;
    }

    // AUX image (path_kernel's AUX): the small tables an ordered scene's kernel reads per hit — materials, frames, media, and
    // (only when some texture is not a SolidColor: otherwise the colours sit in the material records) textures and Perlin
    // tables; copied into the LDS wherever it fits (aux_in_lds)
    {
        const bool with_textures = (s->features & F_TEXTURES) != 0;
        const size_t o_mats = 0, o_texs = align16(o_mats + mats.size() * sizeof(DMaterial)),
                     o_insts = align16(o_texs + (with_textures ? cs.textures.size() * sizeof(rt_texture) : 0u)),
                     o_media = align16(o_insts + cs.instances.size() * sizeof(Instance)),
                     o_perlins = align16(o_media + cs.media.size() * sizeof(Medium)),
                     total = align16(o_perlins + (with_textures ? cs.perlins.size() * sizeof(rt_perlin) : 0u));
        if (s->ordered && total > 0 && total <= 48 * 1024) {
            std::vector<uint4> img(total / 16);
            unsigned char *base = reinterpret_cast<unsigned char *>(img.data());
            if (!mats.empty()) memcpy(base + o_mats, mats.data(), mats.size() * sizeof(DMaterial));
            if (with_textures && !cs.textures.empty()) memcpy(base + o_texs, cs.textures.data(), cs.textures.size() * sizeof(rt_texture));
            if (!cs.instances.empty()) memcpy(base + o_insts, cs.instances.data(), cs.instances.size() * sizeof(Instance));
            if (!cs.media.empty()) memcpy(base + o_media, cs.media.data(), cs.media.size() * sizeof(Medium));
            if (with_textures && !cs.perlins.empty()) memcpy(base + o_perlins, cs.perlins.data(), cs.perlins.size() * sizeof(rt_perlin));
            int urc = upload(s->aux_image, img);
            if (urc != RT_OK) { free_scene(s); return urc; }
            s->aux_bytes = (uint32_t)total;
            s->aux_off[0] = (uint32_t)o_mats; s->aux_off[1] = (uint32_t)o_texs; s->aux_off[2] = (uint32_t)o_insts;
            s->aux_off[3] = (uint32_t)o_media; s->aux_off[4] = (uint32_t)o_perlins;
        }
    }
    for (int lds = 0; lds < 4; ++lds)
        for (int counted = 0; counted < 2; ++counted) {
            if (lds && lds != s->lds_level) { s->blocks_per_cu[lds][counted] = 0; continue; }
            const void *fn = path_kernel_for(lds, counted != 0, kernel_features_for(s->features, lds, s->ordered), s->ordered, aux_in_lds(s, lds), s->wide);
            const int threads = block_threads(s, lds);
            const size_t dyn = dynamic_lds_bytes(s, lds, counted != 0);
            if (dyn > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            int b = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, fn, threads, dyn) != hipSuccess || b < 1) b = 1;
            s->blocks_per_cu[lds][counted] = b;
        }

    int rc = RT_OK;
    if ((rc = upload(s->nodes, cs.nodes32)) != RT_OK || (rc = upload(s->oseq, cs.oseq)) != RT_OK ||
        (rc = upload(s->spheres, cs.spheres)) != RT_OK ||
        (rc = upload(s->quads, cs.quads)) != RT_OK || (rc = upload(s->insts, cs.instances)) != RT_OK ||
        (rc = upload(s->media, cs.media)) != RT_OK || (rc = upload(s->mats, mats)) != RT_OK ||
        (rc = upload(s->texs, cs.textures)) != RT_OK || (rc = upload(s->perlins, cs.perlins)) != RT_OK ||
        (rc = upload(s->images, cs.images)) != RT_OK || (rc = upload(s->texels, cs.texels)) != RT_OK ||
        (rc = upload(s->lut, cs.srgb_lut)) != RT_OK) {
        free_scene(s);
        return rc;
    }
    s->n_nodes = (uint32_t)cs.nodes.size();
    rt_scene_stats &st = s->stats;
    st.node_bytes = cs.ordered ? s->oimage.bytes : s->nodes.bytes; st.sphere_bytes = s->spheres.bytes; st.quad_bytes = s->quads.bytes;
    st.instance_bytes = s->insts.bytes; st.medium_bytes = s->media.bytes; st.material_bytes = s->mats.bytes;
    st.texture_bytes = s->texs.bytes; st.perlin_bytes = s->perlins.bytes; st.image_bytes = s->texels.bytes;
    st.n_nodes = (uint32_t)(cs.ordered ? (cs.wide ? cs.onodes4.size() : cs.onodes.size()) : cs.nodes.size()); st.n_spheres = (uint32_t)cs.spheres.size(); st.n_quads = (uint32_t)cs.quads.size();
    st.n_instances = (uint32_t)cs.instances.size(); st.n_media = (uint32_t)cs.media.size();
    st.max_instance_depth = cs.max_instance_depth;
    st.lds_nodes = s->lds_level ? st.n_nodes : 0; st.lds_bytes = s->lds_level ? (uint32_t)dynamic_lds_bytes(s, s->lds_level, false) : 0;
    st.ordered = cs.ordered ? 1u : 0u; st.stack_entries = cs.ordered ? cs.ordered_stack : 0u;
    *out_scene = s;
    return RT_OK;
}

void rt_scene_destroy(rt_scene *scene) { free_scene(scene); }

int rt_scene_get_stats(const rt_scene *scene, rt_scene_stats *out) {
    if (!scene || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_get_stats: null argument");
    *out = scene->stats;
    return RT_OK;
}

int rt_render_device(const rt_scene *scene, const rt_camera *camera, const rt_render_params *params, double *d_out_rgb_sum,
                     void *hip_stream) {
    if (!scene || !camera || !params || !d_out_rgb_sum) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_device: null argument");
    return launch_render(const_cast<rt_scene *>(scene), camera, *params, d_out_rgb_sum, (hipStream_t)hip_stream, nullptr);
}

int rt_render_device_counted(const rt_scene *scene, const rt_camera *camera, const rt_render_params *params,
                             double *d_out_rgb_sum, void *hip_stream, rt_counters *out_counters) {
    if (!scene || !camera || !params || !d_out_rgb_sum || !out_counters)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_device_counted: null argument");
    return launch_render(const_cast<rt_scene *>(scene), camera, *params, d_out_rgb_sum, (hipStream_t)hip_stream, out_counters);
}

int rt_render(const rt_scene *scene, const rt_camera *camera, const rt_render_params *params, double *out_rgb_sum) {
    if (!scene || !camera || !params || !out_rgb_sum) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: null argument");
    rt_render_params p = *params;
    int rc = normalise_params(camera, p);
    if (rc != RT_OK) return rc;
    rt_scene *s = const_cast<rt_scene *>(scene);
    std::lock_guard<std::mutex> serial(s->host_render_mu);
    HIP_TRY(hipSetDevice(s->device));
    const int32_t w = camera->image_width, h = camera->image_height;
    // on the device the shard always renders into its compact tile buffer; the requested layout is produced on the host
    rt_render_params dp = p;
    dp.out_layout = RT_OUT_TILES;
    const int64_t n_tiles_vals = rt_out_size(w, h, RT_OUT_TILES, p.shard_index, p.shard_count);
    if (n_tiles_vals <= 0) return RT_OK;
    const int32_t tiles_x = (w + RT_TILE_W - 1) / RT_TILE_W;
    const int64_t n_local = n_tiles_vals / (RT_TILE_W * RT_TILE_H * 3);
    std::vector<double> tiles((size_t)n_tiles_vals);
    auto for_each_pixel = [&](auto &&fn) {
        for (int64_t lt = 0; lt < n_local; ++lt) {
            const int64_t k = lt * p.shard_count + p.shard_index;
            const int32_t x0 = (int32_t)(k % tiles_x) * RT_TILE_W, y0 = (int32_t)(k / tiles_x) * RT_TILE_H;
            for (int32_t ty = 0; ty < RT_TILE_H; ++ty)
                for (int32_t tx = 0; tx < RT_TILE_W; ++tx) {
                    const int32_t i = x0 + tx, j = y0 + ty;
                    if (i >= w || j >= h) continue;
                    fn(&tiles[(size_t)((lt * RT_TILE_H + ty) * RT_TILE_W + tx) * 3u], ((size_t)j * w + i) * 3u);
                }
        }
    };
    if (p.accumulate) { // seed the device buffer with the caller's running sums
        if (p.out_layout == RT_OUT_TILES) std::copy(out_rgb_sum, out_rgb_sum + n_tiles_vals, tiles.begin());
        else for_each_pixel([&](double *t, size_t f) { t[0] = out_rgb_sum[f]; t[1] = out_rgb_sum[f + 1]; t[2] = out_rgb_sum[f + 2]; });
    }
    double *d_tiles = nullptr;
    HIP_TRY(hipMalloc((void **)&d_tiles, (size_t)n_tiles_vals * sizeof(double)));
    hipStream_t stream = nullptr;
    rc = RT_OK;
    do {
        if (p.accumulate && hipMemcpy(d_tiles, tiles.data(), tiles.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(RT_ERR_HIP, "rt_render: upload of running sums failed");
            break;
        }
        rc = launch_render(s, camera, dp, d_tiles, stream, nullptr);
        if (rc != RT_OK) break;
        hipError_t e = hipMemcpy(tiles.data(), d_tiles, tiles.size() * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = fail(RT_ERR_HIP, std::string("rt_render: ") + hipGetErrorString(e)); break; }
    } while (0);
    (void)hipFree(d_tiles);
    if (rc != RT_OK) return rc;
    if (p.out_layout == RT_OUT_TILES) std::copy(tiles.begin(), tiles.end(), out_rgb_sum);
    else for_each_pixel([&](double *t, size_t f) { out_rgb_sum[f] = t[0]; out_rgb_sum[f + 1] = t[1]; out_rgb_sum[f + 2] = t[2]; });
    return RT_OK;
}

// the frame-end kernels run on the device that owns the buffers, whichever device the calling thread had selected
static int select_device_of(const void *device_ptr, const char *who) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, device_ptr) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RT_ERR_INVALID_ARGUMENT, std::string(who) + ": not a device pointer");
    }
    HIP_TRY(hipSetDevice(attr.device));
    return RT_OK;
}

int rt_tiles_to_frame_device(int32_t width, int32_t height, int32_t shard_count, const double *d_gathered, double *d_frame,
                             void *hip_stream) {
    if (!d_gathered || !d_frame || width <= 0 || height <= 0 || shard_count <= 0)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_tiles_to_frame_device: bad argument");
    if (int rc = select_device_of(d_frame, "rt_tiles_to_frame_device")) return rc;
    const int64_t stride = rt_out_size(width, height, RT_OUT_TILES, 0, shard_count);
    const int32_t tiles_x = (width + RT_TILE_W - 1) / RT_TILE_W;
    launch_tiles_to_frame(width, height, tiles_x, shard_count, stride, d_gathered, d_frame, (hipStream_t)hip_stream);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}
int rt_device_malloc(int device, int64_t bytes, void **out_device_ptr) {
    if (!out_device_ptr || bytes <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_device_malloc: bad argument");
    *out_device_ptr = nullptr;
    if (device < 0 || device >= rt_device_count()) return fail(RT_ERR_NO_DEVICE, "rt_device_malloc: device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(out_device_ptr, (size_t)bytes));
    return RT_OK;
}

int rt_device_free(int device, void *device_ptr) {
    if (!device_ptr) return RT_OK;
    if (device < 0 || device >= rt_device_count()) return fail(RT_ERR_NO_DEVICE, "rt_device_free: device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(device_ptr));
    return RT_OK;
}

int rt_device_download(int device, void *dst_host, const void *src_device, int64_t bytes, void *hip_stream) {
    if (!dst_host || !src_device || bytes <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_device_download: bad argument");
    if (device < 0 || device >= rt_device_count()) return fail(RT_ERR_NO_DEVICE, "rt_device_download: device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpyAsync(dst_host, src_device, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    return RT_OK;
}

int rt_tiles_to_frame_rgb8_device(int32_t width, int32_t height, int32_t shard_count, const uint8_t *d_gathered, uint8_t *d_frame,
                                  void *hip_stream) {
    if (!d_gathered || !d_frame || width <= 0 || height <= 0 || shard_count <= 0)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_tiles_to_frame_rgb8_device: bad argument");
    if (int rc = select_device_of(d_frame, "rt_tiles_to_frame_rgb8_device")) return rc;
    const int64_t stride = rt_out_size(width, height, RT_OUT_TILES, 0, shard_count);
    const int32_t tiles_x = (width + RT_TILE_W - 1) / RT_TILE_W;
    launch_tiles_to_frame_rgb8(width, height, tiles_x, shard_count, stride, d_gathered, d_frame, (hipStream_t)hip_stream);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_resolve_rgb8_values_device(int64_t n_values, int32_t spp, const double *d_sum, uint8_t *d_rgb8, void *hip_stream) {
    if (!d_sum || !d_rgb8 || n_values <= 0 || spp <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_resolve_rgb8_values_device: bad argument");
    if (int rc = select_device_of(d_rgb8, "rt_resolve_rgb8_values_device")) return rc;
    launch_resolve_rgb8(n_values, 1.0 / (double)spp, d_sum, d_rgb8, (hipStream_t)hip_stream);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_resolve_rgb8_device(int32_t width, int32_t height, int32_t spp, const double *d_frame_sum, uint8_t *d_rgb8, void *hip_stream) {
    if (width <= 0 || height <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_resolve_rgb8_device: bad argument");
    return rt_resolve_rgb8_values_device((int64_t)width * height * 3, spp, d_frame_sum, d_rgb8, hip_stream);
}

} // extern "C"
