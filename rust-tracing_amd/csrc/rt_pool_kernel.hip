// rt_pool_kernel.hip — the render kernel with ray compaction across stages (north_star: "wavefront ballot/prefix for active-ray
// compaction across bounces"; the bounce loop it regroups is src/renderer.rs:139-155).
//
// path_kernel (rt_kernel.hip) ties a path to a lane for its whole life: a lane whose query has ended waits — idle — until
// enough lanes of its wave wait for the same stage (shade, path end), and those rounds then run at half a wave's width.  Here a
// path moves: the waves of a workgroup have ROLES.  Traversal waves only walk (box / sphere / quad / frame and medium steps);
// service waves only shade and end paths.  Between them sits a pool of path slots in the LDS, 64 to a word, with one bitmap
// per slot state:
//     FREE (an id, nothing else) -> BOX (a fresh ray) -> SHADE (a finished query that hit) | END (one that hit nothing, or a
//     path that ended) -> BOX | FREE ...
//  * a traversal lane whose query has ended SWAPS: by wave-level ballot / prefix count the lanes that finished pick slots of one
//    word that hold fresh rays, take those rays and leave their finished queries in the same slots — one bitmap claim, no queue
//    pointers, nobody waits on anybody (a claim that loses a race is simply retried the next round);
//  * lanes choose a word that holds no entries of the OTHER kind, so a word fills up with one kind; a service wave claims a
//    whole word at once (lane L <-> slot L: no index shuffling) and shades, or ends, 64 paths in one full-width round; the
//    results are written back in place as fresh rays.
// Every path keeps its own RNG stream and its own column of the attenuation stack (an id that travels with it), and its
// colour goes to the sample buffer by job index, so the frame is bit for bit what path_kernel (and the CPU oracle) produce —
// only who computes what, and when, differs.
//
// No wave ever waits for another: idle waves poll the bitmaps (s_sleep between polls) and leave when the workgroup's `done`
// flag is up — set once every service wave has run out of jobs and the count of live paths is zero.
#include "rt_device_scene.h"

namespace {

constexpr uint32_t POOL_MAX_WORDS = 16;
constexpr uint32_t POOL_IDLE_LIMIT = 4000000u; // consecutive fruitless polls (about a second) before a wave gives up and flags the launch
struct alignas(16) PoolWord {      // the four bitmaps of one word of 64 slots, side by side: one 32-byte read shows a word's state
    unsigned long long box;   // slot holds a fresh ray: a query to start
    unsigned long long shade; // slot holds a finished query that hit something: to shade
    unsigned long long end;   // slot holds a path that has ended (or a query that hit nothing): products, store, next job
    unsigned long long free_; // slot holds only an id
};
struct PoolCtl {
    PoolWord w[POOL_MAX_WORDS];
    uint32_t live;      // paths in existence in this workgroup (created - ended)
    uint32_t exhausted; // service waves that have no job left to hand out
    uint32_t done;      // every wave leaves
    uint32_t progress;  // service rounds completed (idle waves watch it: no progress anywhere for long = something is wrong)
};
static_assert(sizeof(PoolCtl) == 4 * POOL_MAX_WORDS * 8 + 16, "PoolCtl layout");
RT_DEV PoolWord load_word(const PoolWord *p) { // two 16-byte LDS reads (relaxed: the claims that follow are the atomics that count)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 a = *reinterpret_cast<const volatile u32x4 *>(p), b = *(reinterpret_cast<const volatile u32x4 *>(p) + 1);
    PoolWord r;
    r.box = ((uint64_t)a.y << 32) | a.x; r.shade = ((uint64_t)a.w << 32) | a.z;
    r.end = ((uint64_t)b.y << 32) | b.x; r.free_ = ((uint64_t)b.w << 32) | b.z;
    return r;
}

// traversal-lane states beyond the walk's own stages (rt_kernel.hip Stage: 0 box, 1 sphere, 2 quad, 3 other)
enum PoolStage : uint32_t { PS_BOX = 0, PS_SPHERE = 1, PS_QUAD = 2, PS_OTHER = 3, PS_XSHADE = 4, PS_XEND = 5, PS_EMPTY = 6 };
enum PoolTerm : uint32_t { PT_BACKGROUND = 0, PT_ONE = 1, PT_ZERO = 2 };

RT_DEV uint32_t mbcnt64(uint64_t m) { // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
RT_DEV uint64_t uniform64(uint64_t v) { // lane 0's value, as a wave-uniform
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
RT_DEV uint64_t readlane64(uint64_t v, uint32_t from) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)from);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)from);
    return ((uint64_t)hi << 32) | lo;
}

// One path as a pool slot holds it: six 16-byte fields, field-major in the LDS (field f of slot s at (f * slots + s) * 16), so
// that the 64 lanes of a service wave read 64 consecutive 16-byte pieces and swapping lanes spread over the banks.
struct SlotView {
    uint4 *base;
    uint32_t slots;
    RT_DEV uint4 *field(uint32_t f, uint32_t s) const { return base + f * slots + s; }
};
RT_DEV uint4 pack2(double x, double y) {
    const uint64_t a = f2u(x), b = f2u(y);
    return uint4{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
}
RT_DEV void unpack2(uint4 v, double &x, double &y) {
    x = u2f(((uint64_t)v.y << 32) | v.x);
    y = u2f(((uint64_t)v.w << 32) | v.z);
}
// field 5: job | depth (16) n_att (16) | best_prim | id (16) best_inst + 1 (8) term (8)
RT_DEV uint4 pack_meta(uint32_t job, uint32_t depth, uint32_t n_att, uint32_t best_prim, uint32_t id, int32_t best_inst, uint32_t term) {
    return uint4{job, (depth & 0xffffu) | (n_att << 16), best_prim, (id & 0xffffu) | (((uint32_t)(best_inst + 1) & 0xffu) << 16) | (term << 24)};
}

// PROF: per section, shader cycles / rounds / lanes served, summed over waves into P.counters (RT_POOL_PROF=1, tools/sweep_pool.sh)
//   0 box  1 sphere  2 quad  3 other  4 exchange (progress)  5 exchange (nothing to swap with)  6 idle traversal rounds
//   7 shade service  8 end service  9 service polls
template <int LDS, int THREADS, uint32_t FEAT, bool AUX, bool PROF = false>
__global__ __launch_bounds__(THREADS, 1) void pool_kernel(const KParams P) {
    static_assert(LDS == 3, "the pool kernel walks LDS-resident ordered scenes");
    constexpr bool HAS_SPHERES = (FEAT & F_SPHERES) != 0, HAS_QUADS = (FEAT & F_QUADS) != 0, HAS_FRAMES = (FEAT & F_FRAMES) != 0,
                   HAS_MEDIA = (FEAT & F_MEDIA) != 0, HAS_TEXTURES = (FEAT & F_TEXTURES) != 0;
    constexpr bool HAS_OTHER = HAS_FRAMES || HAS_MEDIA;
    constexpr bool COUNT = false;
    const double INF = __builtin_inf();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw);
        for (uint32_t k = threadIdx.x; k < P.lds_image_bytes / 16u; k += THREADS) dst[k] = P.lds_image[k];
    }
    const OSeq *const seq_tab = reinterpret_cast<const OSeq *>(lds_raw + P.lds_seq_off);
    if constexpr (HAS_MEDIA) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw + P.lds_seq_off);
        const uint4 *src = reinterpret_cast<const uint4 *>(P.oseq);
        for (uint32_t k = threadIdx.x; k < P.n_oseq * (uint32_t)(sizeof(OSeq) / 16u); k += THREADS) dst[k] = src[k];
    }
    if constexpr (AUX) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw + P.lds_aux_off);
        for (uint32_t k = threadIdx.x; k < P.aux_bytes / 16u; k += THREADS) dst[k] = P.aux_image[k];
    }
    const Sphere *const sphere_tab = reinterpret_cast<const Sphere *>(lds_raw + P.lds_off_spheres);
    const Quad *const quad_tab = reinterpret_cast<const Quad *>(lds_raw + P.lds_off_quads);
    const DMaterial *const mats_tab = AUX ? reinterpret_cast<const DMaterial *>(lds_raw + P.lds_aux_off + P.aux_off_mats) : P.mats;
    const rt_texture *const texs_tab = AUX ? reinterpret_cast<const rt_texture *>(lds_raw + P.lds_aux_off + P.aux_off_texs) : P.texs;
    const Instance *const inst_tab = AUX ? reinterpret_cast<const Instance *>(lds_raw + P.lds_aux_off + P.aux_off_insts) : P.insts;
    const Medium *const media_tab = AUX ? reinterpret_cast<const Medium *>(lds_raw + P.lds_aux_off + P.aux_off_media) : P.media;
    const rt_perlin *const perlin_tab = AUX ? reinterpret_cast<const rt_perlin *>(lds_raw + P.lds_aux_off + P.aux_off_perlins) : P.perlins;

    // ---- the pool ----
    PoolCtl *const ctl = reinterpret_cast<PoolCtl *>(lds_raw + P.pool_off);
    const uint32_t n_slots = P.pool_slots, n_words = n_slots >> 6;
    const SlotView pool{reinterpret_cast<uint4 *>(lds_raw + P.pool_off + sizeof(PoolCtl)), n_slots};
    const uint32_t n_service = P.pool_service_waves;
    const uint32_t ids_per_block = n_slots + THREADS;
    if (threadIdx.x < POOL_MAX_WORDS) {
        ctl->w[threadIdx.x].box = 0; ctl->w[threadIdx.x].shade = 0; ctl->w[threadIdx.x].end = 0;
        ctl->w[threadIdx.x].free_ = threadIdx.x < n_words ? ~0ull : 0ull;
    }
    if (threadIdx.x == 0) { ctl->live = 0; ctl->exhausted = 0; ctl->done = 0; ctl->progress = 0; }
    for (uint32_t s = threadIdx.x; s < n_slots; s += THREADS) *pool.field(5, s) = uint4{0u, 0u, PRIM_NONE, s}; // a free slot holds its id
    __syncthreads();

    Counts cn{};
    unsigned long long pf_cyc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pf_rounds[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pf_lanes[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pf_t = PROF ? __builtin_amdgcn_s_memtime() : 0;
    auto pf_mark = [&](uint32_t slot, uint32_t lanes) {
        if constexpr (PROF) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            pf_cyc[slot] += t - pf_t; pf_rounds[slot] += 1; pf_lanes[slot] += lanes;
            pf_t = t;
        }
    };
    auto pf_flush = [&]() {
        if constexpr (PROF) {
            if (lane == 0 && P.counters)
                for (uint32_t q = 0; q < 10; ++q) { atomicAdd(&P.counters[q * 3], pf_cyc[q]); atomicAdd(&P.counters[q * 3 + 1], pf_rounds[q]); atomicAdd(&P.counters[q * 3 + 2], pf_lanes[q]); }
        }
    };
    const uint32_t att_stride = P.n_threads * 3u; // (pool launches: n_threads = workgroups x (slots + threads) path ids)
    double *const att_block = P.att_stack + (size_t)blockIdx.x * ids_per_block * 3u;
    const int32_t w = P.cam.image_width, h = P.cam.image_height;

    if (wave < n_service) {
        // =====================================================================================================
        // Service wave: whole words of finished queries / ended paths, 64 at a time
        // =====================================================================================================
        uint32_t job_next = 0, job_end = 0, jobs_seen_left = P.n_jobs;
        bool jobs_left = true, counted_exhausted = false;
        uint32_t idle_polls = 0, stuck_polls = 0, seen_progress = 0;
        for (;;) {
            // ---- look at the bitmaps: a full word of one kind, else (after a few idle polls) the fullest one ----
            unsigned long long m_shade = 0, m_end = 0, m_free = 0;
            if (lane < n_words) {
                const PoolWord pw = load_word(&ctl->w[lane]);
                m_shade = pw.shade; m_end = pw.end; m_free = pw.free_;
            }
            const uint32_t c_shade = (uint32_t)__popcll(m_shade), c_end = (uint32_t)__popcll(m_end) + (jobs_left ? (uint32_t)__popcll(m_free) : 0u);
            const uint32_t want = idle_polls >= P.pool_patience ? 1u : P.pool_full;
            const uint64_t ok_end = __ballot(lane < n_words && c_end >= want), ok_shade = __ballot(lane < n_words && c_shade >= want);
            uint32_t pick = 0xffffffffu;
            bool do_end = false;
            if (ok_end | ok_shade) {
                // the fullest word wins (6 steps of a max over the first 16 lanes)
                uint32_t best = c_end >= c_shade ? (c_end << 8) | (1u << 7) | lane : (c_shade << 8) | lane;
                if (lane >= n_words) best = 0;
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {
                    const uint32_t other = (uint32_t)__shfl_xor((int)best, off);
                    best = other > best ? other : best;
                }
                best = (uint32_t)__builtin_amdgcn_readfirstlane((int)best);
                if ((best >> 8) >= want) { pick = best & 0x7fu & 63u; do_end = (best & 0x80u) != 0; }
            }
            if (pick == 0xffffffffu) {
                if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ctl->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) break;
                if (!jobs_left && !counted_exhausted) {
                    if (lane == 0) atomicAdd(&ctl->exhausted, 1u);
                    counted_exhausted = true;
                }
                if (counted_exhausted && lane == 0 && __hip_atomic_load(&ctl->exhausted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == n_service &&
                    __hip_atomic_load(&ctl->live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
                    __hip_atomic_store(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                idle_polls++;
                const uint32_t now_progress = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ctl->progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                stuck_polls = now_progress != seen_progress ? 0u : stuck_polls + 1u;
                seen_progress = now_progress;
                if (stuck_polls > POOL_IDLE_LIMIT && __hip_atomic_load(&ctl->live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) stuck_polls = 0; // (nothing alive: waiting for the others to run out of jobs)
                if (stuck_polls > POOL_IDLE_LIMIT) {
                    if (lane == 0) { atomicOr(P.job_counter + 1, 1u); __hip_atomic_store(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                pf_mark(9, 0);
                continue;
            }
            idle_polls = 0;

            if (!do_end) {
                // ---------------- shade 64 closest hits: ray_color (src/renderer.rs:139-155), one level of the recursion ----------------
                uint64_t got = 0;
                if (lane == 0) got = atomicAnd(&ctl->w[pick].shade, 0ull);
                got = uniform64(got);
                if (got == 0) continue;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); // (the slots' contents were written before their bits were set)
                const bool mine = (got >> lane) & 1ull;
                const uint32_t slot = pick * 64u + lane;
                bool to_end = false;
                if (mine) {
                    V3 o, d;
                    double time, t_hit;
                    Rng rng;
                    {
                        const uint4 f0 = *pool.field(0, slot), f1 = *pool.field(1, slot), f2 = *pool.field(2, slot), f3 = *pool.field(3, slot), f4 = *pool.field(4, slot);
                        unpack2(f0, o.x, o.y); unpack2(f1, o.z, d.x); unpack2(f2, d.y, d.z); unpack2(f3, time, t_hit);
                        rng.x = ((uint64_t)f4.y << 32) | f4.x; rng.y = ((uint64_t)f4.w << 32) | f4.z;
                    }
                    const uint4 meta = *pool.field(5, slot);
                    // (a slot is only ever read after its bit was set: these checks cannot fail; should they, the launch is flagged
                    // and stopped rather than let a wild index reach global memory)
                    const bool sane = (meta.w & 0xffffu) < ids_per_block && meta.x < P.n_jobs && (meta.y >> 16) <= (uint32_t)P.max_depth + 1u;
                    if (!sane) { atomicOr(P.job_counter + 1, 4u); __hip_atomic_store(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    const uint32_t job = sane ? meta.x : 0u;
                    int32_t depth = (int32_t)(meta.y & 0xffffu);
                    uint32_t n_att = sane ? meta.y >> 16 : 0u;
                    const uint32_t best_prim = meta.z, id = sane ? meta.w & 0xffffu : 0u;
                    const int32_t best_inst = (int32_t)((meta.w >> 16) & 0xffu) - 1;
                    double *const att = att_block + (size_t)id * 3u;

                    V3 result = v3(0.0, 0.0, 0.0);
                    bool path_done = false;
                    // rebuild the HitRecord in its own frame, then carry it to the world ((o, d) is the world ray)
                    V3 lo = o, ld = d;
                    if (HAS_FRAMES) ray_to_frame(inst_tab, best_inst, lo, ld);
                    V3 p = lo + ld * t_hit; // Ray::at (src/ray.rs:30-32)
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
                    V3 tex = v3(1.0, 1.0, 1.0);
                    V3 rs = v3(0.0, 0.0, 0.0);
                    if (mk != RT_MATERIAL_DIELECTRIC && mk != RT_MATERIAL_DIFFUSE_LIGHT) rs = random_in_unit_sphere<COUNT>(rng, cn);
                    if (mk != RT_MATERIAL_DIELECTRIC && mk != RT_MATERIAL_METAL) {
                        if constexpr (HAS_TEXTURES) tex = m->solid ? ld3(m->albedo) : texture_value<COUNT>(P, texs_tab, perlin_tab, m->texture, u, v, p, cn);
                        else tex = ld3(m->albedo);
                    }
                    V3 attenuation = tex;
                    bool unit_attenuation = false;
                    V3 new_dir = normal;
                    if (mk == RT_MATERIAL_DIFFUSE_LIGHT) { // emitted, no scatter (src/material.rs:114-122)
                        result = tex;
                        path_done = true;
                    } else if (mk == RT_MATERIAL_LAMBERTIAN) { // src/material.rs:26-42
                        const V3 scatter_direction = normal + normalize(rs);
                        new_dir = near_zero(scatter_direction) ? normal : scatter_direction;
                    } else if (mk == RT_MATERIAL_METAL) { // src/material.rs:53-64
                        const V3 refl = reflect(normalize(d), normal);
                        const V3 reflected = refl + rs * m->fuzz;
                        if (!(dot(reflected, normal) > 0.0)) path_done = true; // absorbed: emission (zero) only
                        new_dir = reflected;
                        attenuation = ld3(m->albedo);
                    } else if (mk == RT_MATERIAL_DIELECTRIC) { // src/material.rs:80-104
                        const double refraction_ratio = front_face ? 1.0 / m->ir : m->ir;
                        const V3 unit_direction = normalize(d);
                        const double cos_theta = __builtin_fmin(dot(-unit_direction, normal), 1.0);
                        const double sin_theta = __builtin_sqrt(1.0 - cos_theta * cos_theta);
                        bool do_reflect = refraction_ratio * sin_theta > 1.0;
                        if (!do_reflect) { // `||` short-circuit: draw only when refraction is possible
                            double r0 = (1.0 - refraction_ratio) / (1.0 + refraction_ratio);
                            r0 = r0 * r0;
                            const double reflectance = r0 + (1.0 - r0) * rt_pow5(1.0 - cos_theta);
                            do_reflect = reflectance > rng.random();
                        }
                        new_dir = do_reflect ? reflect(unit_direction, normal) : refract(unit_direction, normal, refraction_ratio);
                        unit_attenuation = true; // Color::ONE: nothing to park
                    } else { // RT_MATERIAL_ISOTROPIC, src/material.rs:132-138
                        new_dir = normalize(rs);
                    }
                    uint32_t term = PT_ZERO;
                    if (!path_done) {
                        if (!unit_attenuation) {
                            double *slot_att = att + n_att * att_stride;
                            slot_att[0] = attenuation.x; slot_att[1] = attenuation.y; slot_att[2] = attenuation.z;
                            n_att++;
                        }
                        depth--;
                        if (depth <= 0) path_done = true; // the next ray_color call returns Color::ZERO at once
                    } else if (result.x != 0.0 || result.y != 0.0 || result.z != 0.0) {
                        // a light: the emitted colour is parked like one more attenuation and the path ends on Color::ONE
                        double *slot_att = att + n_att * att_stride;
                        slot_att[0] = result.x; slot_att[1] = result.y; slot_att[2] = result.z;
                        n_att++;
                        term = PT_ONE;
                    }
                    if (path_done) { // -> an END entry in place (rare: a light, an absorbing bounce, the depth limit)
                        *pool.field(5, slot) = pack_meta(job, 0u, n_att, PRIM_NONE, id, -1, term);
                        to_end = true;
                    } else {     // -> a fresh ray in place
                        *pool.field(0, slot) = pack2(p.x, p.y);
                        *pool.field(1, slot) = pack2(p.z, new_dir.x);
                        *pool.field(2, slot) = pack2(new_dir.y, new_dir.z);
                        *pool.field(4, slot) = uint4{(uint32_t)rng.x, (uint32_t)(rng.x >> 32), (uint32_t)rng.y, (uint32_t)(rng.y >> 32)};
                        *pool.field(5, slot) = pack_meta(job, (uint32_t)depth, n_att, PRIM_NONE, id, -1, 0u);
                    }
                }
                const uint64_t m_to_end = __ballot(mine && to_end), m_to_box = got & ~m_to_end;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the slot writes are in the LDS before the bitmaps say so
                if (lane == 0) {
                    if (m_to_box) atomicOr(&ctl->w[pick].box, m_to_box);
                    if (m_to_end) atomicOr(&ctl->w[pick].end, m_to_end);
                    atomicAdd(&ctl->progress, 1u);
                }
                pf_mark(7, (uint32_t)__popcll(got));
            } else {
                // ---------------- end 64 paths: products of the parked attenuations, sample store, next job, Camera::get_ray ----------------
                uint64_t got_end = 0, got_free = 0;
                if (lane == 0) {
                    got_end = atomicAnd(&ctl->w[pick].end, 0ull);
                    if (jobs_left) got_free = atomicAnd(&ctl->w[pick].free_, 0ull);
                }
                got_end = uniform64(got_end); got_free = uniform64(got_free);
                if ((got_end | got_free) == 0) continue;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const bool ended = (got_end >> lane) & 1ull, here = ((got_end | got_free) >> lane) & 1ull;
                const uint32_t slot = pick * 64u + lane;
                uint32_t id = 0;
                if (here) {
                    const uint4 meta = *pool.field(5, slot);
                    const bool sane = (meta.w & 0xffffu) < ids_per_block && (!ended || (meta.x < P.n_jobs && (meta.y >> 16) <= (uint32_t)P.max_depth + 1u));
                    if (!sane) { atomicOr(P.job_counter + 1, 8u); __hip_atomic_store(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    id = sane ? meta.w & 0xffffu : 0u;
                    if (ended) {
                        const uint32_t job = sane ? meta.x : 0u, term = meta.w >> 24;
                        uint32_t n_att = sane ? meta.y >> 16 : 0u;
                        const double *att = att_block + (size_t)id * 3u;
                        V3 result = v3(0.0, 0.0, 0.0);
                        if (term != PT_ZERO) {
                            result = term == PT_BACKGROUND ? from(P.cam.background) : v3(1.0, 1.0, 1.0);
                            if (result.x != 0.0 || result.y != 0.0 || result.z != 0.0) {
                                while (n_att > 0) { // last parked first; four levels' loads issued together
                                    V3 parked[4];
#pragma unroll
                                    for (uint32_t j = 0; j < 4; ++j) {
                                        const uint32_t level = n_att > j ? n_att - 1u - j : 0u;
                                        const double *sa = att + level * att_stride;
                                        parked[j] = v3(sa[0], sa[1], sa[2]);
                                    }
#pragma unroll
                                    for (uint32_t j = 0; j < 4; ++j)
                                        if (n_att > j) result = parked[j] * result;
                                    n_att = n_att > 4u ? n_att - 4u : 0u;
                                }
                            }
                        }
                        double *dst = P.samples + (size_t)job * 3u;
                        dst[0] = result.x; dst[1] = result.y; dst[2] = result.z;
                    }
                }
                // ---- hand out jobs to all the claimed slots (wave-level: ballot + prefix count; guided reservations) ----
                const uint64_t want_jobs = got_end | got_free;
                const uint32_t n_want = (uint32_t)__popcll(want_jobs);
                const uint32_t old_next = job_next, old_avail = job_end - job_next;
                uint32_t new_base = 0, new_avail = 0;
                if (jobs_left && old_avail < n_want) {
                    uint32_t grab = (uint32_t)((float)jobs_seen_left * P.grab_taper) & ~63u;
                    grab = grab < P.jobs_per_grab ? grab : P.jobs_per_grab;
                    grab = grab < MIN_JOBS_PER_GRAB ? MIN_JOBS_PER_GRAB : grab;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(P.job_counter, grab);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    jobs_seen_left = base + grab < P.n_jobs ? P.n_jobs - (base + grab) : 0u;
                    if (base >= P.n_jobs) jobs_left = false;
                    else { new_base = base; new_avail = (base + grab < P.n_jobs ? base + grab : P.n_jobs) - base; }
                }
                const uint32_t avail = old_avail + new_avail;
                const uint32_t rank = mbcnt64(want_jobs);
                bool created = false;
                if (here && rank < avail) {
                    const uint32_t job = rank < old_avail ? old_next + rank : new_base + (rank - old_avail);
                    // job -> (local tile, sample, pixel): ((lt * S + s_rel) * 64 + p)
                    const uint32_t p64 = job & 63u;
                    const uint32_t row = job >> 6;
                    const uint32_t lt = (uint32_t)(((double)row + 0.5) * P.inv_n_samples), s_rel = row - lt * P.n_samples;
                    const uint32_t k = lt * (uint32_t)P.shard_count + (uint32_t)P.shard_index;
                    const uint32_t tile_row = (uint32_t)(((double)k + 0.5) * P.inv_tiles_x), tile_col = k - tile_row * (uint32_t)P.tiles_x;
                    const int32_t i = (int32_t)tile_col * RT_TILE_W + (int32_t)(p64 & 7u);
                    const int32_t j = (int32_t)tile_row * RT_TILE_H + (int32_t)(p64 >> 3);
                    if (i < w && j < h) {
                        const uint32_t pixel = (uint32_t)j * (uint32_t)w + (uint32_t)i; // screen_pos (src/renderer.rs:32-33)
                        Rng rng;
                        rng.start(P.seed_mixed, pixel, (uint32_t)P.sample_begin + s_rel);
                        // Camera::get_ray (src/camera.rs:112-137)
                        const rt_camera &cam = P.cam;
                        const V3 du = from(cam.pixel_delta_u), dv = from(cam.pixel_delta_v);
                        const V3 pixel_center = from(cam.pixel00_loc) + du * (double)i + dv * (double)j;
                        const double px = -0.5 + rng.random();
                        const double py = -0.5 + rng.random();
                        const V3 pixel_sample = pixel_center + (du * px + dv * py);
                        V3 ro;
                        if (cam.defocus_angle <= 0.0) {
                            ro = from(cam.center);
                        } else { // random_in_unit_disk (src/vec3.rs:77-88)
                            double dx, dy;
                            for (;;) {
                                dx = rng.range(-1.0, 1.0);
                                dy = rng.range(-1.0, 1.0);
                                if (dx * dx + dy * dy + 0.0 * 0.0 < 1.0) break;
                            }
                            ro = from(cam.center) + from(cam.defocus_disk_u) * dx + from(cam.defocus_disk_v) * dy;
                        }
                        const V3 rd = pixel_sample - ro;
                        const double time = rng.random();
                        *pool.field(0, slot) = pack2(ro.x, ro.y);
                        *pool.field(1, slot) = pack2(ro.z, rd.x);
                        *pool.field(2, slot) = pack2(rd.y, rd.z);
                        *pool.field(3, slot) = pack2(time, 0.0);
                        *pool.field(4, slot) = uint4{(uint32_t)rng.x, (uint32_t)(rng.x >> 32), (uint32_t)rng.y, (uint32_t)(rng.y >> 32)};
                        *pool.field(5, slot) = pack_meta(job, (uint32_t)P.max_depth, 0u, PRIM_NONE, id, -1, 0u);
                        created = true;
                    }
                    // (a job of a padding pixel of an edge tile traces nothing: the slot goes back to the free ones)
                }
                if (here && !created) *pool.field(5, slot) = uint4{0u, 0u, PRIM_NONE, id};
                const uint32_t taken = n_want < avail ? n_want : avail;
                if (new_avail) { job_next = new_base + (taken - old_avail); job_end = new_base + new_avail; }
                else job_next = old_next + taken;
                const uint64_t m_created = __ballot(created), m_freed = (got_end | got_free) & ~m_created;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) {
                    const int32_t delta = (int32_t)__popcll(m_created) - (int32_t)__popcll(got_end);
                    if (delta != 0) atomicAdd(&ctl->live, (uint32_t)delta);
                    if (m_created) atomicOr(&ctl->w[pick].box, m_created);
                    if (m_freed) atomicOr(&ctl->w[pick].free_, m_freed);
                    atomicAdd(&ctl->progress, 1u);
                }
                pf_mark(8, (uint32_t)__popcll(got_end | got_free));
            }
        }
        pf_flush();
        return;
    }

    // =========================================================================================================
    // Traversal wave: the walk of path_kernel's ordered layout; a lane whose query has ended swaps it for a fresh ray
    // =========================================================================================================
    const uint32_t tthread = threadIdx.x - n_service * 64u; // (stacks and parked world rays are per traversal thread)
    const bool world_in_lds = P.lds_world_off != 0xffffffffu;
    double *const world_lds = reinterpret_cast<double *>(lds_raw + (world_in_lds ? P.lds_world_off : 0u)) + tthread;
    double *const world_glb = P.world_slots + gtid;
    auto park_world_ray = [&](V3 po, V3 pd) {
        if (world_in_lds) {
            world_lds[0] = po.x; world_lds[THREADS] = po.y; world_lds[2 * THREADS] = po.z;
            world_lds[3 * THREADS] = pd.x; world_lds[4 * THREADS] = pd.y; world_lds[5 * THREADS] = pd.z;
        } else {
            const size_t ws = (size_t)gridDim.x * THREADS;
            world_glb[0] = po.x; world_glb[ws] = po.y; world_glb[2 * ws] = po.z;
            world_glb[3 * ws] = pd.x; world_glb[4 * ws] = pd.y; world_glb[5 * ws] = pd.z;
        }
    };
    auto restore_world_ray = [&](V3 &ro, V3 &rd) {
        if (world_in_lds) {
            ro = v3(world_lds[0], world_lds[THREADS], world_lds[2 * THREADS]);
            rd = v3(world_lds[3 * THREADS], world_lds[4 * THREADS], world_lds[5 * THREADS]);
        } else {
            const size_t ws = (size_t)gridDim.x * THREADS;
            ro = v3(world_glb[0], world_glb[ws], world_glb[2 * ws]);
            rd = v3(world_glb[3 * ws], world_glb[4 * ws], world_glb[5 * ws]);
        }
    };

    Rng rng;
    rng.x = 0; rng.y = 0;
    V3 o = v3(0, 0, 0), d = v3(0, 0, 1);
    double a = 1.0, time = 0.0;
    RayPair32 r32 = make_ray_pair32(o, d, P.lds_off_node_b, P.box_extent);
    float tmin32 = 0, tmax32 = 0;
    uint32_t job = 0, depth = 0, n_att = 0;
    uint32_t my_id = n_slots + threadIdx.x; // an empty lane holds a spare id; a lane with a path, the path's
    double cur_tmin = 0.001, cur_tmax = INF;
    double best_t = INF, med_t1 = 0.0;
    uint32_t best_prim = PRIM_NONE;
    int32_t best_inst = -1, cur_inst = -1;
    uint32_t node = 0, prim_cur = 0, prim_end = 0;
    uint32_t mode = 0;
    uint32_t stage = PS_EMPTY;
    using StackT = uint16_t;
    constexpr uint32_t S_EXIT = 0xffffu;
    constexpr uint32_t SKIP_CHILD0 = 0x4000u, SKIP_CHILD1 = 0x8000u, NODE_INDEX = SKIP_CHILD0 - 1u;
    constexpr uint32_t NODE_FRAME_EXIT = 0xffffffffu, NODE_SEQ_NEXT = 0xfffffffeu;
    StackT *const stack = reinterpret_cast<StackT *>(lds_raw + P.lds_stack_off) + tthread;
    uint32_t sp = 0;
    uint32_t seq_pc = 0;
    const uint32_t first_node = P.o_root;

    auto finish_query = [&]() { stage = best_prim == PRIM_NONE ? (uint32_t)PS_XEND : (uint32_t)PS_XSHADE; };
    auto o_next = [&](bool have, uint32_t ref) {
        uint32_t new_stage, new_node = node, new_cur = prim_cur, new_end = prim_end;
        if (have) {
            const uint32_t kind = ref >> OREF_KIND_SHIFT, index = ref & OREF_INDEX_MASK;
            const bool leaf = kind == OK_SPHERES || kind == OK_QUADS;
            new_stage = kind; // OrderedKind INNER / SPHERES / QUADS / INSTANCE = PS_BOX / PS_SPHERE / PS_QUAD / PS_OTHER
            new_node = leaf ? node : index;
            new_cur = leaf ? index : prim_cur;
            new_end = leaf ? index + ((ref >> OREF_COUNT_SHIFT) & OREF_COUNT_MASK) + 1u : prim_end;
        } else if (sp != 0) {
            sp--;
            const uint32_t e = stack[sp * THREADS];
            const bool leave_frame = HAS_FRAMES && e == S_EXIT;
            new_node = leave_frame ? NODE_FRAME_EXIT : e;
            new_stage = leave_frame ? (uint32_t)PS_OTHER : (uint32_t)PS_BOX;
        } else { // this tree is done
            new_stage = best_prim == PRIM_NONE ? (uint32_t)PS_XEND : (uint32_t)PS_XSHADE;
            if constexpr (HAS_MEDIA) {
                const bool more = seq_pc < P.n_oseq || (mode & 3u) != 0;
                new_node = more ? NODE_SEQ_NEXT : node;
                new_stage = more ? (uint32_t)PS_OTHER : new_stage;
            }
        }
        stage = new_stage; node = new_node; prim_cur = new_cur; prim_end = new_end;
    };
    auto wins_tie = [&](uint32_t my_seq, bool i_am_quad) -> bool { // rt_kernel.hip "ties"
        if (HAS_MEDIA && (best_prim & PRIM_KIND_MASK) == PRIM_MEDIUM) return i_am_quad;
        const bool best_is_quad = (best_prim & PRIM_KIND_MASK) == PRIM_QUAD;
        const uint32_t bi = best_prim & PRIM_INDEX_MASK;
        const uint32_t best_seq = best_is_quad ? quad_tab[bi].seq : (sphere_tab[bi].seq_moving >> 1);
        return my_seq > best_seq ? i_am_quad : !best_is_quad;
    };
    auto refresh_ray32 = [&]() { r32 = make_ray_pair32(o, d, P.lds_off_node_b, P.box_extent); };
    auto refresh_interval32 = [&]() { tmin32 = __double2float_rd(cur_tmin); tmax32 = __double2float_ru(cur_tmax); };

    // ---- ConstantMedium::hit (src/constant_medium.rs:33-71): as in path_kernel ----
    auto medium_sphere_hit = [&](uint32_t na, V3 center, V3 center_vec, bool moving, double radius, double neg_inv_density) {
        if (moving) center = center + center_vec * time;
        const V3 oc = o - center;
        const double half_b = dot(oc, d);
        const double c = len2(oc) - radius * radius;
        const double discriminant = half_b * half_b - a * c;
        if (!(discriminant < 0.0)) {
            const double sqrtd = __builtin_sqrt(discriminant);
            const double root_a = (-half_b - sqrtd) / a, root_b = (-half_b + sqrtd) / a;
            const bool a1 = -INF < root_a && root_a < INF, b1 = -INF < root_b && root_b < INF;
            if (a1 || b1) {
                const double t1 = a1 ? root_a : root_b;
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
                        const double hit_distance = neg_inv_density * rt_log(rng.random());
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
    };
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
    auto seq_advance = [&]() {
        if constexpr (HAS_MEDIA) {
            finish_query();
            while (seq_pc < P.n_oseq) {
                const OSeq *rec = &seq_tab[seq_pc];
                seq_pc++;
                float enter;
                bool miss0, miss1;
                box_pair_f32(opair_of_box(rec->box, r32), r32, __double2float_rd(cur_tmin), __double2float_ru(cur_tmax), miss0, miss1, enter, enter);
                if (miss0) continue;
                if (rec->kind == OSEQ_TREE) {
                    node = rec->a; sp = 0; stage = PS_BOX;
                } else if (rec->kind == OSEQ_MEDIUM_SPHERE) {
                    medium_sphere_hit(rec->a, ld3(rec->center), ld3(rec->center_vec), rec->moving != 0, rec->radius, rec->neg_inv_density);
                    finish_query(); // (the medium may be the first thing hit)
                    continue;
                } else { // boundary.hit(r, UNIVERSE) (src/constant_medium.rs:35)
                    mode = 1;
                    cur_tmin = -INF;
                    cur_tmax = INF;
                    node = rec->b; sp = 0; stage = PS_BOX;
                }
                break;
            }
        }
    };
    // the closest-hit query of the ray just taken out of a slot: world.hit(r, (0.001, inf)) (src/renderer.rs:144)
    auto start_query = [&]() {
        a = len2(d);
        cur_tmin = 0.001; cur_tmax = INF;
        refresh_ray32();
        refresh_interval32();
        best_t = INF; best_prim = PRIM_NONE; best_inst = -1; cur_inst = -1;
        mode = 0;
        node = first_node;
        sp = 0;
        stage = PS_BOX;
        if constexpr (HAS_MEDIA) {
            seq_pc = 1;
            if (first_node == NODE_SEQ_NEXT) {
                seq_pc = 0;
                seq_advance();
                refresh_interval32();
            }
        }
    };

    uint32_t idle_rounds = 0, seen_progress = 0, x_backoff = 0, x_rounds = 0;
    for (;;) {
        // ---------------- scheduler ----------------
        const uint32_t c_box = (uint32_t)__popcll(__ballot(stage == PS_BOX));
        const uint32_t c_sph = HAS_SPHERES ? (uint32_t)__popcll(__ballot(stage == PS_SPHERE)) : 0u;
        const uint32_t c_quad = HAS_QUADS ? (uint32_t)__popcll(__ballot(stage == PS_QUAD)) : 0u;
        const uint32_t c_oth = HAS_OTHER ? (uint32_t)__popcll(__ballot(stage == PS_OTHER)) : 0u;
        const uint64_t m_xs = __ballot(stage == PS_XSHADE), m_xe = __ballot(stage == PS_XEND), m_em = __ballot(stage == PS_EMPTY);
        const uint32_t c_x = (uint32_t)__popcll(m_xs | m_xe | m_em);
        const uint32_t walking = c_box + c_sph + c_quad + c_oth;
        // (after an exchange that found nothing to swap with, the walking lanes get their rounds first: the pool does not change
        // faster than the service waves work)
        const uint32_t cx_eff = (x_backoff != 0 && walking != 0) ? 0u : c_x;
        if (x_backoff) x_backoff--;
        uint32_t run = PS_BOX, best_c = 0;
        // (thresholds in LANES here: a traversal wave's lanes all hold a path, or wait for one)
        if (c_sph >= P.th_prim && c_sph > best_c) { run = PS_SPHERE; best_c = c_sph; }
        if (c_quad >= P.th_prim && c_quad > best_c) { run = PS_QUAD; best_c = c_quad; }
        if (c_oth >= P.th_other && c_oth > best_c) { run = PS_OTHER; best_c = c_oth; }
        if (cx_eff >= P.th_shade && cx_eff > best_c) { run = PS_XSHADE; best_c = cx_eff; }
        if (best_c == 0 && c_box == 0) { // nothing is queued deep enough and nobody walks boxes: whatever has most lanes
            run = PS_SPHERE; best_c = c_sph;
            if (c_quad > best_c) { run = PS_QUAD; best_c = c_quad; }
            if (c_oth > best_c) { run = PS_OTHER; best_c = c_oth; }
            if (cx_eff > best_c || best_c == 0) { run = PS_XSHADE; best_c = c_x; }
        }

        if (run == PS_BOX) {
            uint32_t in_box;
            do {
                if (stage == PS_BOX) {
                    const uint32_t nid = node & NODE_INDEX;
                    const OPair nd = load_opair<LDS>(P, lds_raw, nid, r32.offx, r32.offy, r32.offz);
                    float e0, e1;
                    bool m0, m1;
                    box_pair_f32(nd, r32, tmin32, tmax32, m0, m1, e0, e1);
                    bool h0 = !m0, h1 = !m1;
                    h0 = h0 & ((node & SKIP_CHILD0) == 0u) & (nd.c0 < (OK_EMPTY << OREF_KIND_SHIFT));
                    h1 = h1 & ((node & SKIP_CHILD1) == 0u) & (nd.c1 < (OK_EMPTY << OREF_KIND_SHIFT));
                    const bool one_first = h1 && (!h0 || e1 < e0);
                    if (h0 && h1) {
                        const uint32_t far_ref = one_first ? nd.c0 : nd.c1;
                        const uint32_t entry = far_ref < (1u << OREF_KIND_SHIFT) ? far_ref : (nid | (one_first ? SKIP_CHILD1 : SKIP_CHILD0));
                        stack[sp * THREADS] = (StackT)entry;
                        sp++;
                    }
                    o_next(h0 || h1, one_first ? nd.c1 : nd.c0);
                }
                in_box = (uint32_t)__popcll(__ballot(stage == PS_BOX));
            } while (in_box >= P.th_box && in_box > 0);
            pf_mark(0, c_box);
        } else if (HAS_SPHERES && run == PS_SPHERE) {
            // ---------------- Sphere::hit (src/sphere.rs:58-83), one sphere per round ----------------
            if (stage == PS_SPHERE) {
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
                    auto inside = [&](double root) {
                        if (cur_tmin < root && root < cur_tmax) return true;
                        return cur_tmin < root && root == cur_tmax && best_prim != PRIM_NONE && (!HAS_MEDIA || (mode & 3u) == 0) &&
                               wins_tie(s->seq_moving >> 1, false);
                    };
                    double root = (-half_b - sqrtd) / a;
                    bool ok = inside(root);
                    if (!ok) {
                        root = (-half_b + sqrtd) / a;
                        ok = inside(root);
                    }
                    if (ok) {
                        cur_tmax = root;
                        tmax32 = __double2float_ru(root);
                        if (!HAS_MEDIA || (mode & 3u) == 0) { best_t = root; best_prim = PRIM_SPHERE | q; best_inst = cur_inst; }
                        else mode |= 0x100u;
                    }
                }
                prim_cur = q + 1;
                if (prim_cur >= prim_end) o_next(false, 0u);
            }
            pf_mark(1, c_sph);
        } else if (HAS_QUADS && run == PS_QUAD) {
            // ---------------- Quad::hit (src/quad.rs:96-127): all quads of the leaf (HittableList order) ----------------
            if (stage == PS_QUAD) {
                for (uint32_t q = prim_cur; q < prim_end; ++q) {
                    const Quad *qd = &quad_tab[q];
                    const V3 normal = ld3(qd->normal);
                    const double denom = dot(normal, d);
                    if (__builtin_fabs(denom) < 1e-8) continue;
                    const double t = (qd->d - dot(normal, o)) / denom;
                    if (!(cur_tmin <= t && t <= cur_tmax)) continue; // Interval::contains (src/interval.rs:40-42)
                    if (t == cur_tmax && best_prim != PRIM_NONE && (!HAS_MEDIA || (mode & 3u) == 0) && !wins_tie(qd->seq, true)) continue;
                    const V3 intersection = o + d * t;
                    const V3 php = intersection - ld3(qd->q);
                    const V3 qw = ld3(qd->w);
                    const double alpha = dot(qw, cross(php, ld3(qd->v)));
                    const double beta = dot(qw, cross(ld3(qd->u), php));
                    if (alpha < 0.0 || alpha > 1.0 || beta < 0.0 || beta > 1.0) continue;
                    cur_tmax = t;
                    tmax32 = __double2float_ru(t);
                    if (!HAS_MEDIA || (mode & 3u) == 0) { best_t = t; best_prim = PRIM_QUAD | q; best_inst = cur_inst; }
                    else mode |= 0x100u;
                }
                prim_cur = prim_end;
                o_next(false, 0u);
            }
            pf_mark(2, c_quad);
        } else if (HAS_OTHER && run == PS_OTHER) {
            // ---------------- frame changes and the world sequence's steps ----------------
            if (HAS_MEDIA && stage == PS_OTHER && node == NODE_SEQ_NEXT) {
                bool again = false;
                if ((mode & 3u) != 0) { // a boundary query of the medium at the previous step
                    const OSeq *rec = &seq_tab[seq_pc - 1u];
                    again = medium_boundary_done(rec->a, rec->neg_inv_density);
                    if (again) { node = rec->b; sp = 0; stage = PS_BOX; }
                }
                if (!again) seq_advance();
                refresh_interval32();
            } else if (stage == PS_OTHER) { // enter the frame of instance `node`, or leave the current one
                const bool leaving = node == NODE_FRAME_EXIT;
                if (leaving) {
                    cur_inst = inst_tab[cur_inst].parent;
                    restore_world_ray(o, d);
                    ray_to_frame(inst_tab, cur_inst, o, d);
                } else {
                    if (cur_inst < 0) park_world_ray(o, d); // leaving the world frame
                    apply_instance(inst_tab[node], o, d);
                    cur_inst = (int32_t)node;
                    stack[sp * THREADS] = (StackT)S_EXIT;
                    sp++;
                }
                refresh_ray32();
                a = len2(d);
                if (leaving) o_next(false, 0u);
                else { node = inst_tab[cur_inst].root; stage = PS_BOX; }
            }
            pf_mark(3, c_oth);
        } else {
            // ---------------- exchange: finished queries out, fresh rays in ----------------
            // One look at the bitmaps, ONE claiming instruction (lane 0 claims for the lanes with a hit to shade, lane 1 for those
            // whose query or path ended — in different words, so that a word fills up with one kind), the slot reads, the slot
            // writes, one publishing instruction.  Empty lanes (start and end of a launch) are served in rounds of their own.
            bool progress = false;
            PoolWord pw{0, 0, 0, 0};
            if (lane < n_words) pw = load_word(&ctl->w[lane]);
            const uint32_t kS = (uint32_t)__popcll(m_xs), kE = (uint32_t)__popcll(m_xe);
            // (lanes left empty by a push into a free slot must not starve behind the pushers: every other round is theirs)
            x_rounds++;
            const bool pushers = (kS | kE) != 0 && !(m_em != 0 && (x_rounds & 1u));
            // word per group: fresh rays to swap with (else free slots to push into), no entries of the other kind; preferably one
            // that already collects this kind.  starving (nobody walks): any word with room.
            auto choose = [&](unsigned long long mine_k, unsigned long long other_k, uint32_t taken) -> uint32_t {
                const bool usable = lane < n_words && lane != taken;
                const bool has_box = pw.box != 0, has_room = (pw.box | pw.free_) != 0;
                uint64_t c1 = __ballot(usable && has_box && other_k == 0 && mine_k != 0);
                if (!c1) c1 = __ballot(usable && has_box && other_k == 0);
                if (!c1) c1 = __ballot(usable && has_room && other_k == 0);
                if (!c1 && walking == 0) c1 = __ballot(usable && has_room);
                return c1 ? (uint32_t)__builtin_ctzll(c1) : 0xffffffffu;
            };
            uint32_t pickS = 0xffffffffu, pickE = 0xffffffffu;
            if (pushers) {
                if (kS >= kE) { if (kS) pickS = choose(pw.shade, pw.end, 0xffffffffu); if (kE) pickE = choose(pw.end, pw.shade, pickS); }
                else { if (kE) pickE = choose(pw.end, pw.shade, 0xffffffffu); if (kS) pickS = choose(pw.shade, pw.end, pickE); }
            } else if (m_em) {
                const uint64_t c1 = __ballot(lane < n_words && pw.box != 0);
                if (c1) pickS = (uint32_t)__builtin_ctzll(c1); // (the empty lanes borrow group S's plumbing)
            }
            if ((pickS & pickE) != 0xffffffffu) {
                const uint64_t xmS = pushers ? m_xs : m_em, xmE = pushers ? m_xe : 0ull;
                const uint32_t nS = (uint32_t)__popcll(xmS), nE = (uint32_t)__popcll(xmE);
                const uint32_t wS = pickS != 0xffffffffu ? pickS : 0u, wE = pickE != 0xffffffffu ? pickE : 0u;
                const uint64_t boxS = readlane64(pw.box, wS), freeS = readlane64(pw.free_, wS), boxE = readlane64(pw.box, wE), freeE = readlane64(pw.free_, wE);
                const bool fromboxS = boxS != 0 || !pushers, fromboxE = boxE != 0; // fresh rays first: a swap; without any, a push into free slots
                const uint64_t candS = pickS == 0xffffffffu ? 0ull : (fromboxS ? boxS : freeS), candE = pickE == 0xffffffffu ? 0ull : (fromboxE ? boxE : freeE);
                const uint64_t mS = __ballot(((candS >> lane) & 1ull) && mbcnt64(candS) < nS), mE = __ballot(((candE >> lane) & 1ull) && mbcnt64(candE) < nE);
                // the claim: lane 0 for group S, lane 1 for group E, one LDS instruction
                unsigned long long old = 0;
                if (lane < 2) {
                    unsigned long long *addr = lane == 0 ? (fromboxS ? &ctl->w[wS].box : &ctl->w[wS].free_) : (fromboxE ? &ctl->w[wE].box : &ctl->w[wE].free_);
                    const unsigned long long mask = lane == 0 ? mS : mE;
                    if (mask) old = atomicAnd(addr, ~mask);
                }
                const uint64_t gotS = readlane64(old, 0) & mS, gotE = readlane64(old, 1) & mE;
                if (gotS | gotE) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    // the i-th lane of a group takes the group's i-th claimed slot
                    const uint32_t gS = (uint32_t)__popcll(gotS), gE = (uint32_t)__popcll(gotE);
                    const uint32_t sentS = (uint32_t)__builtin_amdgcn_ds_permute((int)((((gotS >> lane) & 1ull) ? mbcnt64(gotS) : 63u) << 2), (int)lane);
                    const uint32_t sentE = (uint32_t)__builtin_amdgcn_ds_permute((int)((((gotE >> lane) & 1ull) ? mbcnt64(gotE) : 63u) << 2), (int)lane);
                    const bool inS = (xmS >> lane) & 1ull, inE = (xmE >> lane) & 1ull;
                    const uint32_t my_rank = inS ? mbcnt64(xmS) : mbcnt64(xmE);
                    const uint32_t laneS = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((my_rank & 63u) << 2), (int)sentS);
                    const uint32_t laneE = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((my_rank & 63u) << 2), (int)sentE);
                    const bool valid = (inS && my_rank < gS) || (inE && my_rank < gE);
                    if (valid) {
                        const uint32_t slot = inS ? wS * 64u + laneS : wE * 64u + laneE;
                        const bool from_box = inS ? fromboxS : fromboxE;
                        const uint4 meta_in = *pool.field(5, slot);
                        uint4 f0{}, f1{}, f2{}, f3{}, f4{};
                        if (from_box) { f0 = *pool.field(0, slot); f1 = *pool.field(1, slot); f2 = *pool.field(2, slot); f3 = *pool.field(3, slot); f4 = *pool.field(4, slot); }
                        if (pushers) { // leave the finished query in the slot
                            *pool.field(0, slot) = pack2(o.x, o.y);
                            *pool.field(1, slot) = pack2(o.z, d.x);
                            *pool.field(2, slot) = pack2(d.y, d.z);
                            *pool.field(3, slot) = pack2(time, HAS_MEDIA ? best_t : cur_tmax);
                            *pool.field(4, slot) = uint4{(uint32_t)rng.x, (uint32_t)(rng.x >> 32), (uint32_t)rng.y, (uint32_t)(rng.y >> 32)};
                            *pool.field(5, slot) = pack_meta(job, depth, n_att, best_prim, my_id, best_inst, PT_BACKGROUND);
                        } else {        // an empty lane leaves its spare id
                            *pool.field(5, slot) = uint4{0u, 0u, PRIM_NONE, my_id};
                        }
                        my_id = meta_in.w & 0xffffu;
                        if (from_box) {
                            double unused;
                            unpack2(f0, o.x, o.y); unpack2(f1, o.z, d.x); unpack2(f2, d.y, d.z); unpack2(f3, time, unused);
                            rng.x = ((uint64_t)f4.y << 32) | f4.x; rng.y = ((uint64_t)f4.w << 32) | f4.z;
                            job = meta_in.x; depth = meta_in.y & 0xffffu; n_att = meta_in.y >> 16;
                            start_query();
                        } else {
                            stage = PS_EMPTY;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane < 2) { // publish: lane 0 group S's slots, lane 1 group E's (empty lanes: the slots they emptied are free)
                        const unsigned long long bits = lane == 0 ? gotS : gotE;
                        unsigned long long *addr = lane == 0 ? (pushers ? &ctl->w[wS].shade : &ctl->w[wS].free_) : &ctl->w[wE].end;
                        if (bits) atomicOr(addr, bits);
                    }
                    progress = true;
                }
            }
            pf_mark(progress ? 4u : 5u, c_x);
            if (!progress && walking == 0) {
                if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ctl->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) break;
                const uint32_t now_progress = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ctl->progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                idle_rounds = now_progress != seen_progress ? 0u : idle_rounds + 1u;
                seen_progress = now_progress;
                if (idle_rounds > POOL_IDLE_LIMIT && (c_x == (uint32_t)__popcll(m_em))) idle_rounds = 0; // (only empty lanes: waiting for `done`)
                if (idle_rounds > POOL_IDLE_LIMIT) { // (cannot happen; a wave that would wait for ever says so and lets everyone leave)
                    if (lane == 0) { atomicOr(P.job_counter + 1, 2u); __hip_atomic_store(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
                pf_mark(6, 0);
            } else {
                idle_rounds = 0;
                if (!progress) x_backoff = 6; // (lanes are walking: let them)
            }
        }
    }
    pf_flush();
}

} // namespace

namespace rtk {

const void *pool_kernel_for(uint32_t feat, bool aux, bool prof) {
#define RT_POOL_PICK(T, F) (prof ? (aux ? (const void *)pool_kernel<3, T, F, true, true> : (const void *)pool_kernel<3, T, F, false, true>) \
                                 : (aux ? (const void *)pool_kernel<3, T, F, true, false> : (const void *)pool_kernel<3, T, F, false, false>))
    if (feat == FEAT_SPHERES_SOLID) return RT_POOL_PICK(LDS_THREADS, FEAT_SPHERES_SOLID);
    if (feat == FEAT_QUADS_FRAMES) return RT_POOL_PICK(QUADS_FRAMES_THREADS, FEAT_QUADS_FRAMES);
    return RT_POOL_PICK(LDS_THREADS_GENERAL, F_ALL); // (also for the two 768-thread specialisations: a superset of their features)
#undef RT_POOL_PICK
}
size_t pool_ctl_bytes() { return sizeof(PoolCtl); }

} // namespace rtk
