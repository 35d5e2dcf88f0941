// Scene compiler: rt_scene_desc (object graph, include/rt_amd.h) -> threaded preorder device layout
// (rt_layout.h).  Host code, no HIP: it runs inside rt_scene_create before the upload.
#pragma once
#include "rt_layout.h"
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace rtd {

struct CompileError : std::runtime_error {
    int status;
    CompileError(int st, const std::string &m) : std::runtime_error(m), status(st) {}
};

// ---- bounds of the geometry (box refit, ordered trees) ---------------------------------------------------
struct Bound {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool empty() const { return !(lo[0] <= hi[0]); }
    void add_point(double x, double y, double z) {
        const double p[3] = {x, y, z};
        for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], p[k]); hi[k] = std::fmax(hi[k], p[k]); }
    }
    void add(const Bound &b) {
        if (b.empty()) return;
        for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], b.lo[k]); hi[k] = std::fmax(hi[k], b.hi[k]); }
    }
};
inline Bound sphere_bound(const Sphere &s) {
    Bound b;
    for (int e = 0; e < 2; ++e) { // both ends of the motion (time in [0, 1), src/sphere.rs:34-46)
        const double c[3] = {s.center[0] + (e ? s.center_vec[0] : 0.0), s.center[1] + (e ? s.center_vec[1] : 0.0),
                             s.center[2] + (e ? s.center_vec[2] : 0.0)};
        const double r = std::fabs(s.radius);
        b.add_point(c[0] - r, c[1] - r, c[2] - r);
        b.add_point(c[0] + r, c[1] + r, c[2] + r);
        if (!(s.seq_moving & 1u)) break;
    }
    return b;
}
inline Bound quad_bound(const Quad &q) {
    Bound b;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            b.add_point(q.q[0] + i * q.u[0] + j * q.v[0], q.q[1] + i * q.u[1] + j * q.v[1], q.q[2] + i * q.u[2] + j * q.v[2]);
    return b;
}
// The box a record carries for geometry bounded by `b`: an axis the geometry is flat on (a quad) gets the thickness the
// reference's AABB::pad gives it (src/aabb.rs:35-53), and every face moves out by a few ulps because hit points are
// computed with rounding and may sit a hair outside the exact shape.
inline void pad_axis(double &lo, double &hi) {
    if (hi - lo < 0.0001) { lo -= 0.00005; hi += 0.00005; }
    const double slack = 8.0 * 2.220446049250313e-16 * std::fmax(std::fabs(lo), std::fabs(hi));
    lo -= slack; hi += slack;
}
// a frame's bound seen from the enclosing frame: rotate the eight corners, then shift (src/hittable.rs:126-149,:88)
inline Bound instance_bound(const Instance &in, const Bound &inner) {
    Bound b;
    if (inner.empty()) return b;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int l = 0; l < 2; ++l) {
                double x = i ? inner.hi[0] : inner.lo[0], y = j ? inner.hi[1] : inner.lo[1], z = l ? inner.hi[2] : inner.lo[2];
                if (in.flags & INST_ROTATE) {
                    const double nx = in.cos_theta * x + in.sin_theta * z, nz = -in.sin_theta * x + in.cos_theta * z;
                    x = nx; z = nz;
                }
                if (in.flags & INST_TRANSLATE) { x += in.offset[0]; y += in.offset[1]; z += in.offset[2]; }
                b.add_point(x, y, z);
            }
    return b;
}
// f64 -> f32 rounded towards -inf / +inf
inline float round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -INFINITY);
    return f;
}
inline float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, INFINITY);
    return f;
}

struct CompiledScene {
    std::vector<Node> nodes;     // f64 boxes (build-time)
    std::vector<Node32> nodes32; // what the device walks
    // ordered layout (rt_ordered.hpp), when the scene allows it: the primitive tables are then in leaf order
    bool ordered = false;
    std::vector<ONode> onodes;
    bool wide = false;           // ... walked through the four-child records below instead (rt_layout.h ONode4)
    std::vector<ONode4> onodes4;
    std::vector<OSeq> oseq;     // the world frame's sequence of trees and media
    uint32_t ordered_stack = 0; // stack entries a lane needs at most
    std::vector<Sphere> spheres;
    std::vector<Quad> quads;
    std::vector<Instance> instances;
    std::vector<Medium> media;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_perlin> perlins;
    std::vector<ImageRef> images;
    std::vector<uint8_t> texels;
    std::vector<double> srgb_lut; // 256 entries: pow(c/255, 2.2) (src/color.rs:8-10,:21-26), host libm
    uint32_t max_instance_depth = 0;
    bool uses_image = false, uses_noise = false;
};

class Compiler {
  public:
    explicit Compiler(const rt_scene_desc &d, bool refit = true) : d_(d), refit_(refit) {}

    CompiledScene run() {
        validate_tables();
        emit_object(d_.world, nullptr, -1, false, 0);
        if (out_.nodes.empty()) throw CompileError(RT_ERR_INVALID_ARGUMENT, "scene has no geometry");
        if (out_.nodes.size() > 0x3fffffffu) throw CompileError(RT_ERR_UNSUPPORTED, "too many nodes");
        if (refit_) refit_range(0, (uint32_t)out_.nodes.size());
        pack_nodes();
        out_.materials.assign(d_.materials, d_.materials + d_.n_materials);
        out_.textures.assign(d_.textures, d_.textures + d_.n_textures);
        if (d_.n_perlins) out_.perlins.assign(d_.perlins, d_.perlins + d_.n_perlins);
        for (int32_t i = 0; i < d_.n_images; ++i) {
            const rt_image &im = d_.images[i];
            if (im.width <= 0 || im.height <= 0 || !im.rgb) throw CompileError(RT_ERR_INVALID_ARGUMENT, "image without texels");
            const uint32_t tiles_x = ((uint32_t)im.width + TEXEL_TILE - 1u) / TEXEL_TILE, tiles_y = ((uint32_t)im.height + TEXEL_TILE - 1u) / TEXEL_TILE;
            ImageRef r{};
            r.width = (uint32_t)im.width; r.height = (uint32_t)im.height; r.tiles_x = tiles_x; r.offset = (uint64_t)out_.texels.size();
            // row-major RGB8 (rt_image) -> 8x8 tiles (rt_layout.h ImageRef); padding texels are never addressed
            out_.texels.resize(out_.texels.size() + (size_t)tiles_x * tiles_y * TEXEL_TILE * TEXEL_TILE * 3u, 0);
            uint8_t *dst = out_.texels.data() + r.offset;
            for (uint32_t j = 0; j < r.height; ++j)
                for (uint32_t i = 0; i < r.width; ++i) {
                    const uint8_t *src = im.rgb + ((size_t)j * r.width + i) * 3u;
                    uint8_t *px = dst + texel_index(tiles_x, i, j) * 3u;
                    px[0] = src[0]; px[1] = src[1]; px[2] = src[2];
                }
            // keep every image 16-byte aligned in the pool
            out_.texels.resize((out_.texels.size() + 15u) & ~(size_t)15u, 0);
            out_.images.push_back(r);
        }
        out_.srgb_lut.resize(256);
        for (int c = 0; c < 256; ++c) out_.srgb_lut[c] = std::pow((double)c / 255.0, 2.2);
        for (const auto &t : out_.textures) {
            if (t.kind == RT_TEXTURE_IMAGE) out_.uses_image = true;
            if (t.kind == RT_TEXTURE_NOISE) out_.uses_noise = true;
        }
        return std::move(out_);
    }

  private:
    const rt_scene_desc &d_;
    bool refit_;
    uint32_t next_seq_ = 0; // spheres and quads are created in the reference's scan order
    CompiledScene out_;

    // ---- box refit ------------------------------------------------------------------------------------------
    // The boxes the reference carries can be much larger than the geometry under them: a HittableList starts from
    // an all-zero box (derive(Default), src/hittable.rs:50-57), so every Quad::cube list — and every Translate /
    // RotateY / BVH node above it — also spans the origin (final_scene: 400 such boxes all overlap there).  The
    // walk only needs boxes that CONTAIN their geometry (a tighter box rejects more rays whose line or interval
    // cannot reach anything inside, never one that can: DESIGN.md "Box test"), so every record's box is replaced
    // by the intersection of the reference's box with the bound of the primitives actually below it.
    // bound of the records [begin, end) (siblings), in the frame they live in
    Bound refit_range(uint32_t begin, uint32_t end) {
        Bound all;
        for (uint32_t k = begin; k < end;) {
            all.add(refit_node(k));
            const uint32_t next = out_.nodes[k].skip;
            k = next > k ? next : k + 1;
        }
        return all;
    }
    Bound refit_node(uint32_t k) {
        Node &n = out_.nodes[k];
        const uint32_t kind = n.kind & NODE_KIND_MASK;
        Bound b;
        switch (kind) {
        case NK_INNER: b = refit_range(k + 1, n.skip); break;
        case NK_SPHERES: for (uint32_t i = 0; i < n.b; ++i) b.add(sphere_bound(out_.spheres[n.a + i])); break;
        case NK_QUADS: for (uint32_t i = 0; i < n.b; ++i) b.add(quad_bound(out_.quads[n.a + i])); break;
        case NK_MEDIUM_ENTER: b = refit_range(k + 1, n.skip - 1); break; // the boundary's geometry
        case NK_MEDIUM_SPHERE: b = sphere_bound(out_.spheres[out_.media[n.a].first_node]); break;
        case NK_INST_ENTER: {
            b = instance_bound(out_.instances[n.a], refit_range(k + 1, n.skip - 1)); // back to the enclosing frame
            break;
        }
        default: break; // NK_INST_EXIT, NK_MEDIUM_EXIT: no geometry of their own
        }
        if (!(n.kind & NODE_NO_BBOX) && !b.empty()) {
            for (int ax = 0; ax < 3; ++ax) {
                double lo = b.lo[ax], hi = b.hi[ax];
                pad_axis(lo, hi);
                n.lo[ax] = std::fmax(n.lo[ax], lo);
                n.hi[ax] = std::fmin(n.hi[ax], hi);
            }
        }
        return b;
    }

    void pack_nodes() {
        out_.nodes32.resize(out_.nodes.size());
        for (size_t i = 0; i < out_.nodes.size(); ++i) {
            const Node &n = out_.nodes[i];
            Node32 &m = out_.nodes32[i];
            m.bx[0] = round_down(n.lo[0]); m.bx[1] = round_up(n.hi[0]);
            m.by[0] = round_down(n.lo[1]); m.by[1] = round_up(n.hi[1]);
            m.bz[0] = round_down(n.lo[2]); m.bz[1] = round_up(n.hi[2]);
            m.skip = n.skip;
            const uint32_t kind = n.kind & NODE_KIND_MASK;
            const uint32_t count = (kind == NK_SPHERES || kind == NK_QUADS) ? n.b : 0u;
            if (count > N32_MAX_COUNT || n.a > N32_MAX_A)
                throw CompileError(RT_ERR_UNSUPPORTED, "rt_scene_create: scene exceeds the packed record limits");
            m.packed = kind | ((n.kind & NODE_NO_BBOX) ? N32_NO_BBOX : 0u) | (count << N32_COUNT_SHIFT) | (n.a << N32_A_SHIFT);
        }
    }

    [[noreturn]] static void bad(const std::string &m) { throw CompileError(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: " + m); }

    static void need(bool ok, const char *what) {
        if (!ok) bad(what);
    }

    void check_texture(int32_t t, int depth) const {
        need(t >= 0 && t < d_.n_textures, "texture index out of range");
        need(depth < 16, "texture nesting too deep (cycle?)");
        const rt_texture &x = d_.textures[t];
        switch (x.kind) {
        case RT_TEXTURE_SOLID: break;
        case RT_TEXTURE_CHECKER:
            check_texture(x.even, depth + 1);
            check_texture(x.odd, depth + 1);
            break;
        case RT_TEXTURE_IMAGE: need(x.image >= 0 && x.image < d_.n_images, "image index out of range"); break;
        case RT_TEXTURE_NOISE: need(x.perlin >= 0 && x.perlin < d_.n_perlins, "perlin index out of range"); break;
        default: bad("unknown texture kind");
        }
    }

    void validate_tables() const {
        need(d_.abi_version == RT_ABI_VERSION, "abi_version mismatch");
        auto arr = [](int32_t n, const void *p) { return n >= 0 && (n == 0 || p != nullptr); };
        need(arr(d_.n_spheres, d_.spheres) && arr(d_.n_quads, d_.quads) && arr(d_.n_lists, d_.lists) &&
                 arr(d_.n_list_items, d_.list_items) && arr(d_.n_translates, d_.translates) &&
                 arr(d_.n_rotates, d_.rotates) && arr(d_.n_bvh_nodes, d_.bvh_nodes) && arr(d_.n_bvhs, d_.bvhs) &&
                 arr(d_.n_media, d_.media) && arr(d_.n_materials, d_.materials) &&
                 arr(d_.n_textures, d_.textures) && arr(d_.n_perlins, d_.perlins) && arr(d_.n_images, d_.images),
             "null array with non-zero count");
        for (int32_t i = 0; i < d_.n_materials; ++i) {
            const rt_material &m = d_.materials[i];
            switch (m.kind) {
            case RT_MATERIAL_LAMBERTIAN:
            case RT_MATERIAL_DIFFUSE_LIGHT:
            case RT_MATERIAL_ISOTROPIC: check_texture(m.texture, 0); break;
            case RT_MATERIAL_METAL:
            case RT_MATERIAL_DIELECTRIC: break;
            default: bad("unknown material kind");
            }
        }
        for (int32_t i = 0; i < d_.n_textures; ++i) check_texture(i, 0);
        for (int32_t i = 0; i < d_.n_images; ++i)
            need(d_.images[i].width > 0 && d_.images[i].height > 0 && d_.images[i].rgb, "bad image");
        for (int32_t i = 0; i < d_.n_perlins; ++i)
            for (int k = 0; k < RT_PERLIN_POINTS; ++k)
                need((uint32_t)d_.perlins[i].perm_x[k] < 256u && (uint32_t)d_.perlins[i].perm_y[k] < 256u &&
                         (uint32_t)d_.perlins[i].perm_z[k] < 256u,
                     "perlin permutation entry out of range");
    }

    void check_material(int32_t m) const { need(m >= 0 && m < d_.n_materials, "material index out of range"); }

    uint32_t push_node(uint32_t kind, const rt_aabb *bbox, uint32_t a, uint32_t b) {
        Node n{};
        if (bbox) {
            for (int k = 0; k < 3; ++k) { n.lo[k] = bbox->lo[k]; n.hi[k] = bbox->hi[k]; }
            n.kind = kind;
        } else {
            for (int k = 0; k < 3; ++k) { n.lo[k] = -INFINITY; n.hi[k] = INFINITY; }
            n.kind = kind | NODE_NO_BBOX;
        }
        n.a = a;
        n.b = b;
        n.skip = 0;
        out_.nodes.push_back(n);
        return (uint32_t)out_.nodes.size() - 1u;
    }
    void close(uint32_t idx) { out_.nodes[idx].skip = (uint32_t)out_.nodes.size(); }

    uint32_t add_sphere(int32_t i) {
        need(i >= 0 && i < d_.n_spheres, "sphere index out of range");
        const rt_sphere &s = d_.spheres[i];
        check_material(s.material);
        Sphere o{};
        o.center[0] = s.center.x; o.center[1] = s.center.y; o.center[2] = s.center.z;
        o.radius = s.radius;
        o.center_vec[0] = s.center_vec.x; o.center_vec[1] = s.center_vec.y; o.center_vec[2] = s.center_vec.z;
        o.material = (uint32_t)s.material;
        o.seq_moving = (next_seq_++ << 1) | (s.is_moving ? 1u : 0u);
        out_.spheres.push_back(o);
        return (uint32_t)out_.spheres.size() - 1u;
    }
    uint32_t add_quad(int32_t i) {
        need(i >= 0 && i < d_.n_quads, "quad index out of range");
        const rt_quad &q = d_.quads[i];
        check_material(q.material);
        Quad o{};
        const rt_vec3 *src[5] = {&q.normal, &q.q, &q.w, &q.u, &q.v};
        double *dst[5] = {o.normal, o.q, o.w, o.u, o.v};
        for (int k = 0; k < 5; ++k) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
        o.d = q.d;
        o.material = (uint32_t)q.material;
        o.seq = next_seq_++;
        out_.quads.push_back(o);
        return (uint32_t)out_.quads.size() - 1u;
    }

    // `bbox`: the box the reference tests before calling this object's hit() (the BVH leaf's box), or null if
    // the reference reaches the object without a box test.
    void emit_object(rt_ref obj, const rt_aabb *bbox, int32_t cur_inst, bool in_medium, int depth) {
        need(depth < 64, "object graph too deep (cycle?)");
        switch (obj.kind) {
        case RT_HITTABLE_SPHERE: {
            uint32_t s = add_sphere(obj.index);
            close(push_node(NK_SPHERES, bbox, s, 1));
            break;
        }
        case RT_HITTABLE_QUAD: {
            uint32_t q = add_quad(obj.index);
            close(push_node(NK_QUADS, bbox, q, 1));
            break;
        }
        case RT_HITTABLE_LIST: {
            need(obj.index >= 0 && obj.index < d_.n_lists, "list index out of range");
            const rt_list &l = d_.lists[obj.index];
            need(l.first >= 0 && l.count >= 0 && (int64_t)l.first + l.count <= d_.n_list_items, "list range out of bounds");
            // HittableList::hit is a linear scan with a shrinking tmax and no box tests (src/hittable.rs:61-74).
            // Runs of quads (Quad::cube, src/quad.rs:45-93) or spheres collapse into one leaf record.
            bool wrapped = false;
            uint32_t wrapper = 0;
            int32_t i = 0;
            auto kind_of = [&](int32_t k) { return d_.list_items[l.first + k].kind; };
            // does the whole list collapse into a single leaf?  then the leaf itself can carry the box
            bool single_run = l.count > 0;
            for (int32_t k = 1; k < l.count && single_run; ++k) single_run = kind_of(k) == kind_of(0);
            single_run = single_run && (kind_of(0) == RT_HITTABLE_QUAD || kind_of(0) == RT_HITTABLE_SPHERE);
            if (bbox && !single_run) {
                wrapper = push_node(NK_INNER, bbox, 0, 0);
                wrapped = true;
            }
            while (i < l.count) {
                int32_t k0 = kind_of(i);
                if (k0 == RT_HITTABLE_QUAD || k0 == RT_HITTABLE_SPHERE) {
                    int32_t j = i;
                    uint32_t first = 0;
                    while (j < l.count && kind_of(j) == k0) {
                        uint32_t id = k0 == RT_HITTABLE_QUAD ? add_quad(d_.list_items[l.first + j].index)
                                                             : add_sphere(d_.list_items[l.first + j].index);
                        if (j == i) first = id;
                        ++j;
                    }
                    // one leaf record holds at most N32_MAX_COUNT primitives; a longer run continues in the next record
                    // (the box, if any, is simply tested again)
                    for (uint32_t done = 0, total = (uint32_t)(j - i); done < total;) {
                        const uint32_t take = total - done > N32_MAX_COUNT ? N32_MAX_COUNT : total - done;
                        close(push_node(k0 == RT_HITTABLE_QUAD ? NK_QUADS : NK_SPHERES, single_run ? bbox : nullptr,
                                        first + done, take));
                        done += take;
                    }
                    i = j;
                } else {
                    emit_object(d_.list_items[l.first + i], nullptr, cur_inst, in_medium, depth + 1);
                    ++i;
                }
            }
            if (wrapped) close(wrapper);
            break;
        }
        case RT_HITTABLE_TRANSLATE:
        case RT_HITTABLE_ROTATE_Y: {
            Instance inst{};
            inst.parent = cur_inst;
            inst.depth = cur_inst < 0 ? 0u : out_.instances[(size_t)cur_inst].depth + 1u;
            if (inst.depth >= MAX_INSTANCE_DEPTH)
                throw CompileError(RT_ERR_UNSUPPORTED, "rt_scene_create: Translate/RotateY nesting deeper than 4");
            rt_ref child;
            if (obj.kind == RT_HITTABLE_TRANSLATE) {
                need(obj.index >= 0 && obj.index < d_.n_translates, "translate index out of range");
                const rt_translate &t = d_.translates[obj.index];
                inst.flags |= INST_TRANSLATE;
                inst.offset[0] = t.offset.x; inst.offset[1] = t.offset.y; inst.offset[2] = t.offset.z;
                child = t.object;
                // Translate(RotateY(x)) — the only pairing the reference's scenes use — fuses into one frame change
                if (child.kind == RT_HITTABLE_ROTATE_Y) {
                    need(child.index >= 0 && child.index < d_.n_rotates, "rotate index out of range");
                    const rt_rotate_y &r = d_.rotates[child.index];
                    inst.flags |= INST_ROTATE;
                    inst.sin_theta = r.sin_theta;
                    inst.cos_theta = r.cos_theta;
                    child = r.object;
                }
            } else {
                need(obj.index >= 0 && obj.index < d_.n_rotates, "rotate index out of range");
                const rt_rotate_y &r = d_.rotates[obj.index];
                inst.flags |= INST_ROTATE;
                inst.sin_theta = r.sin_theta;
                inst.cos_theta = r.cos_theta;
                child = r.object;
            }
            out_.instances.push_back(inst);
            const uint32_t id = (uint32_t)out_.instances.size() - 1u;
            if (inst.depth + 1u > out_.max_instance_depth) out_.max_instance_depth = inst.depth + 1u;
            uint32_t enter = push_node(NK_INST_ENTER, bbox, id, 0);
            emit_object(child, nullptr, (int32_t)id, in_medium, depth + 1);
            close(push_node(NK_INST_EXIT, nullptr, id, 0));
            close(enter);
            break;
        }
        case RT_HITTABLE_BVH: {
            need(obj.index >= 0 && obj.index < d_.n_bvhs, "bvh index out of range");
            // BVHNode::hit = root.hit (src/bvh.rs:115-118).  When this BVH is itself a leaf of an outer BVH the outer
            // leaf's box IS the root's box (BVHNode::bounding_box, src/bvh.rs:120-122), so one test suffices.
            emit_bvh_node(d_.bvhs[obj.index].root, cur_inst, in_medium, depth + 1);
            break;
        }
        case RT_HITTABLE_CONSTANT_MEDIUM: {
            need(obj.index >= 0 && obj.index < d_.n_media, "medium index out of range");
            if (in_medium)
                throw CompileError(RT_ERR_UNSUPPORTED, "rt_scene_create: a ConstantMedium inside a ConstantMedium boundary");
            const rt_constant_medium &m = d_.media[obj.index];
            check_material(m.phase_material);
            Medium dm{};
            dm.neg_inv_density = m.neg_inv_density;
            dm.phase_material = (uint32_t)m.phase_material;
            out_.media.push_back(dm);
            const uint32_t id = (uint32_t)out_.media.size() - 1u;
            if (m.boundary.kind == RT_HITTABLE_SPHERE) {
                // boundary.hit() twice on one sphere (src/constant_medium.rs:35-38) needs no walk: one record
                out_.media[id].first_node = add_sphere(m.boundary.index);
                close(push_node(NK_MEDIUM_SPHERE, bbox, id, 0));
                break;
            }
            uint32_t enter = push_node(NK_MEDIUM_ENTER, bbox, id, 0);
            const uint32_t first_child = (uint32_t)out_.nodes.size();
            out_.media[id].first_node = first_child;
            emit_object(m.boundary, nullptr, cur_inst, true, depth + 1);
            need(out_.nodes.size() > first_child, "medium boundary is empty");
            close(push_node(NK_MEDIUM_EXIT, nullptr, id, first_child));
            close(enter);
            break;
        }
        default: bad("unknown hittable kind");
        }
    }

    // one `(Node, AABB)` pair (src/bvh.rs:90-113)
    void emit_bvh_node(int32_t idx, int32_t cur_inst, bool in_medium, int depth) {
        need(idx >= 0 && idx < d_.n_bvh_nodes, "bvh node index out of range");
        need(depth < 96, "BVH too deep (cycle?)");
        const rt_bvh_node &n = d_.bvh_nodes[idx];
        if (n.is_leaf) {
            if (n.object.kind == RT_HITTABLE_BVH) {
                // nested BVHNode as a leaf (src/main.rs:533): its root box equals this leaf's box
                emit_object(n.object, nullptr, cur_inst, in_medium, depth + 1);
            } else {
                emit_object(n.object, &n.bbox, cur_inst, in_medium, depth + 1);
            }
        } else {
            uint32_t me = push_node(NK_INNER, &n.bbox, 0, 0);
            emit_bvh_node(n.left, cur_inst, in_medium, depth + 1);
            emit_bvh_node(n.right, cur_inst, in_medium, depth + 1);
            close(me);
        }
    }
};

inline CompiledScene compile_scene(const rt_scene_desc &d, bool refit = true) { return Compiler(d, refit).run(); }

} // namespace rtd
