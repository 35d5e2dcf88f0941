// rt_debug.cpp — the test and tuning hooks of include/rt_amd_debug.h (not part of the drop-in boundary).
#include "rt_api.hpp"
#include "rt_qfilt.hpp"

#include <cmath>
#include <cstring>

using namespace rtapi;

extern "C" {

int rt_debug_set_tuning(int32_t th_prim, int32_t th_other, int32_t th_shade, int32_t th_box, int32_t use_lds, int32_t th_new) {
    const int32_t v[6] = {th_prim, th_other, th_shade, th_box, th_new, use_lds};
    tuning_update([](Tuning &t, const void *arg) {
        const int32_t *a = static_cast<const int32_t *>(arg);
        for (int k = 0; k < 5; ++k) t.forced[k] = a[k];
        if (a[5] >= 0) t.use_lds = a[5];
    }, v);
    return RT_OK;
}

int rt_debug_set_traversal(int32_t ordered, int32_t leaf_max) {
    const int32_t v[2] = {ordered, leaf_max};
    tuning_update([](Tuning &t, const void *arg) {
        const int32_t *a = static_cast<const int32_t *>(arg);
        if (a[0] >= 0) t.ordered = a[0];
        if (a[1] > 0) t.ordered_options.leaf_max = (uint32_t)a[1] < OREF_MAX_LEAF ? (uint32_t)a[1] : OREF_MAX_LEAF;
        else if (a[1] == 0) t.ordered_options.leaf_max = OrderedOptions().leaf_max;
    }, v);
    return RT_OK;
}

int rt_debug_set_walk_shortcuts(int32_t flat_max, int32_t start_shortcut, int32_t defer_instances, int32_t seq_lookahead, int32_t slow_min,
                                int32_t slow_age) {
    const int32_t v[6] = {flat_max, start_shortcut, defer_instances, seq_lookahead, slow_min, slow_age};
    tuning_update([](Tuning &t, const void *arg) {
        const int32_t *a = static_cast<const int32_t *>(arg);
        if (a[0] >= 0) t.ordered_options.flat_max = (uint32_t)a[0];
        if (a[1] >= 0) t.start_shortcut = a[1];
        if (a[2] >= 0) t.defer = a[2];
        if (a[3] >= 0) t.seq_lookahead = a[3];
        if (a[4] >= 1) t.slow_min = a[4];
        if (a[5] >= 0) t.slow_age = a[5];
    }, v);
    return RT_OK;
}

int rt_debug_ordered_layout(const rt_scene_desc *desc, rt_debug_ordered *io) { return rt_debug_ordered_layout_ex(desc, nullptr, io); }

int rt_debug_ordered_layout_ex(const rt_scene_desc *desc, const rt_scene_options *options, rt_debug_ordered *io) {
    if (!desc || !io) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_ordered_layout: null argument");
    rt_scene_options opt; // (checked and defaulted exactly as rt_scene_create_ex does)
    if (int orc = resolve_scene_options(options, opt, "rt_debug_ordered_layout_ex")) return orc;
    CompiledScene cs;
    try {
        cs = compile_scene(*desc, true);
        OrderedOptions oopt = tuning_snapshot().ordered_options;
        if (opt.leaf_max > 0) oopt.leaf_max = (uint32_t)opt.leaf_max < OREF_MAX_LEAF ? (uint32_t)opt.leaf_max : OREF_MAX_LEAF;
        if (opt.flat_max >= 0) oopt.flat_max = (uint32_t)opt.flat_max;
        build_ordered(cs, oopt);
    } catch (const CompileError &e) {
        return fail(e.status, e.what());
    } catch (const std::exception &e) {
        return fail(RT_ERR_INVALID_ARGUMENT, std::string("rt_debug_ordered_layout: ") + e.what());
    }
    io->ordered = cs.ordered ? 1u : 0u;
    io->root = cs.ordered && cs.oseq[0].kind == OSEQ_TREE ? cs.oseq[0].a : 0u;
    io->n_steps = (int64_t)cs.oseq.size();
    if (io->steps) {
        if (io->cap_steps < io->n_steps) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_ordered_layout: steps buffer too small");
        if (io->n_steps) memcpy(io->steps, cs.oseq.data(), cs.oseq.size() * sizeof(OSeq));
    }
    io->n_media = (int64_t)cs.media.size();
    if (io->media) {
        if (io->cap_media < io->n_media) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_ordered_layout: media buffer too small");
        for (size_t i = 0; i < cs.media.size(); ++i) io->media[i] = cs.media[i].first_node;
    }
    io->stack_entries = cs.ordered_stack;
    io->n_nodes = (int64_t)cs.onodes.size(); io->n_spheres = (int64_t)cs.spheres.size();
    io->n_quads = (int64_t)cs.quads.size(); io->n_instances = (int64_t)cs.instances.size();
    if (io->nodes) {
        if (io->cap_nodes < io->n_nodes) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_ordered_layout: nodes buffer too small");
        if (io->n_nodes) memcpy(io->nodes, cs.onodes.data(), cs.onodes.size() * sizeof(ONode));
    }
    if (io->spheres) {
        if (io->cap_spheres < io->n_spheres) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_ordered_layout: spheres buffer too small");
        for (size_t i = 0; i < cs.spheres.size(); ++i) {
            const Sphere &sp = cs.spheres[i];
            double *o = io->spheres + i * 9;
            for (int k = 0; k < 3; ++k) { o[k] = sp.center[k]; o[4 + k] = sp.center_vec[k]; }
            o[3] = sp.radius; o[7] = (double)(sp.seq_moving >> 1); o[8] = (double)(sp.seq_moving & 1u);
        }
    }
    if (io->quads) {
        if (io->cap_quads < io->n_quads) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_ordered_layout: quads buffer too small");
        for (size_t i = 0; i < cs.quads.size(); ++i) {
            const Quad &qd = cs.quads[i];
            double *o = io->quads + i * 10;
            for (int k = 0; k < 3; ++k) { o[k] = qd.q[k]; o[3 + k] = qd.u[k]; o[6 + k] = qd.v[k]; }
            o[9] = (double)qd.seq;
        }
    }
    if (io->instances) {
        if (io->cap_instances < io->n_instances) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_ordered_layout: instances buffer too small");
        for (size_t i = 0; i < cs.instances.size(); ++i) {
            const Instance &in = cs.instances[i];
            double *o = io->instances + i * 8;
            for (int k = 0; k < 3; ++k) o[k] = in.offset[k];
            o[3] = in.sin_theta; o[4] = in.cos_theta; o[5] = (double)in.parent; o[6] = (double)in.flags; o[7] = (double)in.root;
        }
    }
    return RT_OK;
}

// Structure check of the wide layout (no device): walks every tree from its root and reports
//   out[0] records, out[1] primitives found in leaves, out[2] primitives of the scene's tables, out[3] stack entries a walk needs,
//   out[4] violations (a primitive in no leaf or in two, a child's box not inside its parent's slot, a reference out of range),
//   out[5] deepest chain of records
int rt_debug_wide_layout(const rt_scene_desc *desc, const rt_scene_options *options, uint64_t out[6]) {
    if (!desc || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_wide_layout: null argument");
    rt_scene_options opt;
    if (int orc = resolve_scene_options(options, opt, "rt_debug_wide_layout")) return orc;
    CompiledScene cs;
    try {
        cs = compile_scene(*desc, true);
        OrderedOptions oopt = tuning_snapshot().ordered_options;
        if (opt.leaf_max > 0) oopt.leaf_max = (uint32_t)opt.leaf_max < OREF_MAX_LEAF ? (uint32_t)opt.leaf_max : OREF_MAX_LEAF;
        if (opt.flat_max >= 0) oopt.flat_max = (uint32_t)opt.flat_max;
        oopt.wide = true;
        build_ordered(cs, oopt);
    } catch (const CompileError &e) {
        return fail(e.status, e.what());
    } catch (const std::exception &e) {
        return fail(RT_ERR_INVALID_ARGUMENT, std::string("rt_debug_wide_layout: ") + e.what());
    }
    for (int k = 0; k < 6; ++k) out[k] = 0;
    if (!cs.ordered || !cs.wide) return RT_OK;
    std::vector<uint32_t> seen_s(cs.spheres.size(), 0), seen_q(cs.quads.size(), 0), seen_r(cs.onodes4.size(), 0);
    uint64_t violations = 0, deepest = 0;
    struct Walk { uint32_t rec; uint64_t depth; };
    std::vector<Walk> todo;
    for (const OSeq &st : cs.oseq) {
        if (st.kind == OSEQ_TREE) todo.push_back({st.a, 1});
        else if (st.kind == OSEQ_MEDIUM) todo.push_back({st.b, 1});
    }
    for (const Instance &in : cs.instances) todo.push_back({in.root, 1});
    while (!todo.empty()) {
        const Walk w = todo.back();
        todo.pop_back();
        if (w.rec >= cs.onodes4.size()) { violations++; continue; }
        if (seen_r[w.rec]++) continue; // (an instance's root is reached from the table above and, as a child, never: each record once)
        deepest = std::max(deepest, w.depth);
        const ONode4 &nd = cs.onodes4[w.rec];
        for (int k = 0; k < 4; ++k) {
            const uint32_t kind = nd.c[k] >> OREF_KIND_SHIFT, index = nd.c[k] & OREF_INDEX_MASK, count = ((nd.c[k] >> OREF_COUNT_SHIFT) & OREF_COUNT_MASK) + 1u;
            if (kind == OK_EMPTY) { if (!(nd.b[k][0] > nd.b[k][1])) violations++; continue; } // an empty slot's box must admit no ray
            if (kind == OK_INNER) {
                if (index >= cs.onodes4.size()) { violations++; continue; }
                const ONode4 &ch = cs.onodes4[index];
                for (int q = 0; q < 4; ++q)
                    if ((ch.c[q] >> OREF_KIND_SHIFT) != OK_EMPTY)
                        for (int ax = 0; ax < 3; ++ax)
                            if (ch.b[q][2 * ax] < nd.b[k][2 * ax] || ch.b[q][2 * ax + 1] > nd.b[k][2 * ax + 1]) violations++;
                todo.push_back({index, w.depth + 1});
            } else if (kind == OK_SPHERES) {
                for (uint32_t i = 0; i < count; ++i) { if (index + i >= seen_s.size()) violations++; else seen_s[index + i]++; }
            } else if (kind == OK_QUADS) {
                for (uint32_t i = 0; i < count; ++i) { if (index + i >= seen_q.size()) violations++; else seen_q[index + i]++; }
            } else if (kind == OK_INSTANCE) {
                if (index >= cs.instances.size()) violations++;
            } else violations++;
        }
    }
    uint64_t found = 0;
    // (the boundary spheres of media solved in place are in no leaf: they sit behind the leaves' spheres, cs.media[].first_node)
    std::vector<bool> boundary(cs.spheres.size(), false);
    for (const OSeq &st : cs.oseq)
        if (st.kind == OSEQ_MEDIUM_SPHERE) boundary[cs.media[st.a].first_node] = true;
    for (size_t i = 0; i < seen_s.size(); ++i) { found += seen_s[i]; if (seen_s[i] != (boundary[i] ? 0u : 1u)) violations++; }
    for (size_t i = 0; i < seen_q.size(); ++i) { found += seen_q[i]; if (seen_q[i] != 1u) violations++; }
    out[0] = cs.onodes4.size(); out[1] = found; out[2] = cs.spheres.size() + cs.quads.size(); out[3] = cs.ordered_stack; out[4] = violations; out[5] = deepest;
    return RT_OK;
}

int rt_debug_compiled_nodes(const rt_scene_desc *desc, int32_t refit, rt_debug_node *out_nodes, int64_t capacity, int64_t *out_count) {
    if (!desc || !out_count) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_compiled_nodes: null argument");
    CompiledScene cs;
    try {
        cs = compile_scene(*desc, refit != 0);
    } catch (const CompileError &e) {
        return fail(e.status, e.what());
    } catch (const std::exception &e) {
        return fail(RT_ERR_INVALID_ARGUMENT, std::string("rt_debug_compiled_nodes: ") + e.what());
    }
    *out_count = (int64_t)cs.nodes.size();
    if (!out_nodes) return RT_OK;
    if ((int64_t)cs.nodes.size() > capacity) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_compiled_nodes: buffer too small");
    for (size_t i = 0; i < cs.nodes.size(); ++i) {
        const Node &n = cs.nodes[i];
        const Node32 &m = cs.nodes32[i];
        rt_debug_node &o = out_nodes[i];
        for (int k = 0; k < 3; ++k) { o.lo[k] = n.lo[k]; o.hi[k] = n.hi[k]; }
        o.lo32[0] = m.bx[0]; o.hi32[0] = m.bx[1]; o.lo32[1] = m.by[0]; o.hi32[1] = m.by[1]; o.lo32[2] = m.bz[0]; o.hi32[2] = m.bz[1];
        o.skip = n.skip; o.kind = n.kind & NODE_KIND_MASK; o.no_bbox = (n.kind & NODE_NO_BBOX) ? 1u : 0u; o.a = n.a; o.b = n.b;
        o.prim_lo[0] = o.prim_lo[1] = o.prim_lo[2] = INFINITY;
        o.prim_hi[0] = o.prim_hi[1] = o.prim_hi[2] = -INFINITY;
        // bound of the record's own primitives (leaves), in the frame the record lives in
        auto grow = [&](double x, double y, double z) {
            const double p[3] = {x, y, z};
            for (int k = 0; k < 3; ++k) { o.prim_lo[k] = std::fmin(o.prim_lo[k], p[k]); o.prim_hi[k] = std::fmax(o.prim_hi[k], p[k]); }
        };
        if (o.kind == NK_SPHERES || o.kind == NK_MEDIUM_SPHERE) {
            const uint32_t first = o.kind == NK_SPHERES ? n.a : cs.media[n.a].first_node, count = o.kind == NK_SPHERES ? n.b : 1u;
            for (uint32_t q = first; q < first + count; ++q) {
                const Sphere &sp = cs.spheres[q];
                for (int e = 0; e < ((sp.seq_moving & 1u) ? 2 : 1); ++e) {
                    const double c[3] = {sp.center[0] + e * sp.center_vec[0], sp.center[1] + e * sp.center_vec[1], sp.center[2] + e * sp.center_vec[2]};
                    grow(c[0] - sp.radius, c[1] - sp.radius, c[2] - sp.radius);
                    grow(c[0] + sp.radius, c[1] + sp.radius, c[2] + sp.radius);
                }
            }
        } else if (o.kind == NK_QUADS) {
            for (uint32_t q = n.a; q < n.a + n.b; ++q) {
                const Quad &qd = cs.quads[q];
                for (int i2 = 0; i2 < 2; ++i2)
                    for (int j2 = 0; j2 < 2; ++j2)
                        grow(qd.q[0] + i2 * qd.u[0] + j2 * qd.v[0], qd.q[1] + i2 * qd.u[1] + j2 * qd.v[1], qd.q[2] + i2 * qd.u[2] + j2 * qd.v[2]);
            }
        }
    }
    return RT_OK;
}

int rt_debug_last_launch(uint32_t out[4]) {
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_last_launch: null argument");
    for (int k = 0; k < 4; ++k) out[k] = g_last_launch[k];
    return RT_OK;
}

int rt_debug_stage_profile(uint64_t out[36]) {
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_stage_profile: null argument");
    std::lock_guard<std::mutex> lock(g_stage_profile_mu);
    for (uint32_t q = 0; q < PROF_SLOTS * 3u; ++q) out[q] = g_stage_profile[q];
    return RT_OK;
}

int rt_debug_box_tests(int64_t n, const double *rays, const double *boxes, double tmin, double tmax, uint8_t *out_exact_hit,
                       uint8_t *out_f32_hit, int device) {
    if (n <= 0 || !rays || !boxes || !out_exact_hit || !out_f32_hit) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_box_tests: bad argument");
    if (rt_device_count() <= device || device < 0) return fail(RT_ERR_NO_DEVICE, "rt_debug_box_tests: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    double *dr = nullptr, *db = nullptr;
    uint8_t *de = nullptr, *df = nullptr;
    const size_t bytes = (size_t)n * 6 * sizeof(double);
    int rc = RT_OK;
    do {
        if (hipMalloc((void **)&dr, bytes) != hipSuccess || hipMalloc((void **)&db, bytes) != hipSuccess ||
            hipMalloc((void **)&de, (size_t)n) != hipSuccess || hipMalloc((void **)&df, (size_t)n) != hipSuccess) {
            rc = fail(RT_ERR_OUT_OF_MEMORY, "rt_debug_box_tests: hipMalloc failed"); break;
        }
        if (hipMemcpy(dr, rays, bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(db, boxes, bytes, hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(RT_ERR_HIP, "rt_debug_box_tests: upload failed"); break;
        }
        launch_debug_box(n, dr, db, tmin, tmax, de, df);
        if (hipMemcpy(out_exact_hit, de, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(out_f32_hit, df, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(RT_ERR_HIP, "rt_debug_box_tests: download failed"); break;
        }
    } while (0);
    (void)hipFree(dr); (void)hipFree(db); (void)hipFree(de); (void)hipFree(df);
    return rc;
}

int rt_debug_quad_filter_tests(int64_t n, const double *rays, const double *quads, double tmin, double tmax, uint8_t *out_exact_hit,
                               uint8_t *out_keep, int device) {
    if (n <= 0 || !rays || !quads || !out_exact_hit || !out_keep) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_quad_filter_tests: bad argument");
    if (rt_device_count() <= device || device < 0) return fail(RT_ERR_NO_DEVICE, "rt_debug_quad_filter_tests: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    std::vector<Quad> hq((size_t)n);
    std::vector<QFiltPair> hf((size_t)n);
    for (int64_t i = 0; i < n; ++i) { // Quad::new (src/quad.rs:24-27)
        const double *s = quads + i * 9;
        Quad &q = hq[(size_t)i];
        memset(&q, 0, sizeof q);
        for (int k = 0; k < 3; ++k) { q.q[k] = s[k]; q.u[k] = s[3 + k]; q.v[k] = s[6 + k]; }
        const double nn[3] = {q.u[1] * q.v[2] - q.u[2] * q.v[1], q.u[2] * q.v[0] - q.u[0] * q.v[2], q.u[0] * q.v[1] - q.u[1] * q.v[0]};
        const double n2 = nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2];
        const double rl = 1.0 / std::sqrt(n2), rn = 1.0 / n2;
        for (int k = 0; k < 3; ++k) { q.normal[k] = nn[k] * rl; q.w[k] = nn[k] * rn; }
        q.d = q.normal[0] * q.q[0] + q.normal[1] * q.q[1] + q.normal[2] * q.q[2];
        memset(&hf[(size_t)i], 0, sizeof(QFiltPair));
    }
    for (int64_t i = 0; i < n; ++i) { // (the quad sits in both slots of its record: the pair's bounds are its own)
        qfilt_fill(hq[(size_t)i], hf[(size_t)i], 0);
        qfilt_fill(hq[(size_t)i], hf[(size_t)i], 1);
    }
    double *dr = nullptr;
    Quad *dq = nullptr;
    QFiltPair *df = nullptr;
    uint8_t *de = nullptr, *dk = nullptr;
    int rc = RT_OK;
    do {
        if (hipMalloc((void **)&dr, (size_t)n * 6 * sizeof(double)) != hipSuccess || hipMalloc((void **)&dq, (size_t)n * sizeof(Quad)) != hipSuccess ||
            hipMalloc((void **)&df, (size_t)n * sizeof(QFiltPair)) != hipSuccess || hipMalloc((void **)&de, (size_t)n) != hipSuccess ||
            hipMalloc((void **)&dk, (size_t)n) != hipSuccess) { rc = fail(RT_ERR_OUT_OF_MEMORY, "rt_debug_quad_filter_tests: hipMalloc failed"); break; }
        if (hipMemcpy(dr, rays, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dq, hq.data(), (size_t)n * sizeof(Quad), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(df, hf.data(), (size_t)n * sizeof(QFiltPair), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(RT_ERR_HIP, "rt_debug_quad_filter_tests: upload failed"); break; }
        launch_debug_quad(n, dr, dq, df, tmin, tmax, de, dk);
        if (hipMemcpy(out_exact_hit, de, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(out_keep, dk, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(RT_ERR_HIP, "rt_debug_quad_filter_tests: download failed"); break;
        }
    } while (0);
    (void)hipFree(dr); (void)hipFree(dq); (void)hipFree(df); (void)hipFree(de); (void)hipFree(dk);
    return rc;
}

int rt_debug_eval(int32_t op, int64_t n, const double *a, const double *b, double *out, int device) {
    if (n <= 0 || !a || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_eval: bad argument");
    if (rt_device_count() <= device || device < 0) return fail(RT_ERR_NO_DEVICE, "rt_debug_eval: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    const size_t bytes = (size_t)n * sizeof(double);
    int rc = RT_OK;
    do {
        if (hipMalloc((void **)&da, bytes) != hipSuccess || hipMalloc((void **)&dout, bytes) != hipSuccess ||
            (b && hipMalloc((void **)&db, bytes) != hipSuccess)) { rc = fail(RT_ERR_OUT_OF_MEMORY, "rt_debug_eval: hipMalloc failed"); break; }
        if (hipMemcpy(da, a, bytes, hipMemcpyHostToDevice) != hipSuccess ||
            (b && hipMemcpy(db, b, bytes, hipMemcpyHostToDevice) != hipSuccess)) { rc = fail(RT_ERR_HIP, "rt_debug_eval: upload failed"); break; }
        launch_debug_eval(op, n, da, db, dout);
        hipError_t e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rc = fail(RT_ERR_HIP, std::string("rt_debug_eval: ") + hipGetErrorString(e)); break; }
    } while (0);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return rc;
}

} // extern "C"
