// Ordered layout builder: CompiledScene (threaded records + primitive tables) -> one SAH tree per frame
// (rt_layout.h "ordered layout").  Host code, no HIP.
//
// Why the kernel may walk a tree of its own: in a scene without a ConstantMedium, hit() draws no random number
// and `world.hit(r, (0.001, inf))` (src/renderer.rs:144) is a pure minimum over the primitives — BVHNode /
// HittableList only decide which tests are skipped (src/bvh.rs:97-108, src/hittable.rs:61-74).  The reference's
// tree (random axis, median split, src/bvh.rs:31-66) must be walked left child first to reproduce its tie
// handling; with ties settled explicitly by `seq` (rt_kernel.hip "ties") neither the tree nor the order matters,
// and a surface-area-heuristic tree walked nearest child first needs a fraction of the box tests.
// A ConstantMedium breaks this only locally: its hit() consumes a draw iff its boundary is crossed before the closest hit
// found SO FAR in the scan (src/constant_medium.rs:33-61), so what matters is which primitives come before it in scan
// order, not how they are searched.  The world frame therefore becomes a SEQUENCE: maximal runs of medium-free objects
// (each searched through a tree of its own, in any order) alternating with the media, in the reference's scan order;
// a medium's boundary is again a closest-hit query over a medium-free subtree and gets its own tree.  (Whether the
// reference would have culled the medium by one of its ancestors' boxes does not matter: those boxes contain the
// boundary, and a ray that cannot reach the boundary before the current closest hit draws nothing either way.)
// Only a medium inside a Translate / RotateY frame or inside another medium's boundary keeps the threaded layout.
#pragma once
#include "rt_compile.hpp"
#include <algorithm>
#include <cstring>
#include <numeric>

namespace rtd {

struct OrderedOptions {
    uint32_t leaf_max = 1;        // primitives per leaf at most (<= OREF_MAX_LEAF); 1 measured best (Cornell: 1392 vs 1328 Msamples/s at 4)
    double cost_node = 1.0;       // one record visit (two box tests) ...
    double cost_sphere = 1.6;     // ... against one Sphere::hit,
    double cost_quad = 0.8;       // one Quad::hit (most end at the plane test),
    double cost_instance = 6.0;   // and one frame change plus the walk inside
    uint32_t world_depth = 24;    // inner records on a root-to-leaf path at most ...
    uint32_t flat_max = 8;        // a frame with at most this many primitives (all spheres or all quads; <= OREF_MAX_LEAF) keeps them in ONE leaf
                                  // under its root, beside the tree of its instances: the lanes that enter the frame test them together (0: off)
    uint32_t frame_slack = 4;     // ... and at most this many more than an even split of the tree's items needs
                                  // (the kernel's stack holds ORDERED_MAX_STACK entries in all, frames below included)
    bool wide = false;            // four-child records (rt_layout.h ONode4): every binary record's grandchildren pulled up
};

class OrderedBuilder {
  public:
    OrderedBuilder(CompiledScene &cs, const OrderedOptions &opt) : cs_(cs), opt_(opt) {}

    // false: the scene must keep the threaded layout (a medium inside a frame, or a tree too deep for the stack)
    bool run() {
        if (cs_.spheres.size() > OREF_INDEX_MASK || cs_.quads.size() > OREF_INDEX_MASK) return false;
        if (!collect()) return false;
        const size_t n_frames = frames_.size();
        need_.assign(n_frames, 0);
        bound_.assign(n_frames, Bound());
        root_.assign(n_frames, 0);
        // innermost frames first (a frame is always created after the one that refers to it): an instance's box in its
        // parent's frame comes from the tree built for it
        for (size_t f = n_frames; f-- > 0;) {
            std::vector<Item> &items = frames_[f];
            for (Item &it : items)
                if (it.kind == OK_INSTANCE) {
                    Bound b = instance_bound(cs_.instances[it.index], bound_[frame_of_inst_[it.index]]);
                    if (!b.empty())
                        for (int ax = 0; ax < 3; ++ax) pad_axis(b.lo[ax], b.hi[ax]);
                    it.b = b;
                }
            // geometry-free instances (an empty list inside a Translate) can never be hit: drop them
            items.erase(std::remove_if(items.begin(), items.end(), [](const Item &it) { return it.b.empty(); }), items.end());
            for (Item &it : items) {
                for (int ax = 0; ax < 3; ++ax) it.c[ax] = 0.5 * (it.b.lo[ax] + it.b.hi[ax]);
                bound_[f].add(it.b);
            }
            build_frame(f);
        }
        // the world frame's sequence: trees and media in scan order
        uint32_t need = 1;
        std::vector<OSeq> seq;
        std::vector<std::pair<uint32_t, uint32_t>> new_first_node; // (medium, index of its boundary sphere in new_spheres_)
        for (const SeqItem &si : sequence_) {
            OSeq o{};
            o.kind = si.kind;
            Bound b;
            if (si.kind == OSEQ_TREE) {
                if (frames_[si.frame].empty()) continue;
                o.a = root_[si.frame];
                need = std::max(need, need_[si.frame]);
                b = bound_[si.frame];
            } else if (si.kind == OSEQ_MEDIUM_SPHERE) {
                o.a = si.medium;
                // the boundary sphere is in no leaf: it moves to the end of the (reordered) sphere table
                const uint32_t old_index = cs_.media[si.medium].first_node;
                const Sphere &sp = cs_.spheres[old_index];
                b = sphere_bound(sp);
                for (int ax = 0; ax < 3; ++ax) { o.center[ax] = sp.center[ax]; o.center_vec[ax] = sp.center_vec[ax]; }
                o.radius = sp.radius;
                o.moving = sp.seq_moving & 1u;
                o.neg_inv_density = cs_.media[si.medium].neg_inv_density;
                new_first_node.emplace_back(si.medium, (uint32_t)new_spheres_.size()); // applied only if the build succeeds
                new_spheres_.push_back(cs_.spheres[old_index]);
            } else {
                if (frames_[si.frame].empty()) continue; // a boundary without geometry: the medium can never be hit
                o.a = si.medium;
                o.b = root_[si.frame];
                o.neg_inv_density = cs_.media[si.medium].neg_inv_density;
                need = std::max(need, need_[si.frame]);
                b = bound_[si.frame];
            }
            for (int ax = 0; ax < 3; ++ax) {
                pad_axis(b.lo[ax], b.hi[ax]);
                o.box[2 * ax] = round_down(b.lo[ax]);
                o.box[2 * ax + 1] = round_up(b.hi[ax]);
            }
            seq.push_back(o);
        }
        // (nothing of cs_ has been touched up to here: a scene that is turned down keeps its threaded layout intact)
        if (need > ORDERED_MAX_STACK || seq.size() > ORDERED_MAX_STEPS) return false;
        if (seq.empty()) { // nothing can be hit: one tree whose root has only empty children
            OSeq o{};
            o.kind = OSEQ_TREE;
            o.a = alloc_node();
            if (opt_.wide) o.a = widen(o.a).ref;
            seq.push_back(o);
        }
        for (size_t i = 0; i < cs_.instances.size(); ++i) cs_.instances[i].root = root_[frame_of_inst_[i]];
        for (const auto &mf : new_first_node) cs_.media[mf.first].first_node = mf.second;
        cs_.spheres.swap(new_spheres_);
        cs_.quads.swap(new_quads_);
        cs_.onodes.swap(nodes_);
        cs_.onodes4.swap(wnodes_);
        cs_.wide = opt_.wide;
        cs_.oseq.swap(seq);
        cs_.ordered_stack = need;
        cs_.ordered = true;
        return true;
    }

  private:
    struct Item {
        Bound b;
        double c[3];
        uint32_t kind;  // OK_SPHERES / OK_QUADS / OK_INSTANCE
        uint32_t index; // in cs_.spheres / cs_.quads / cs_.instances
        uint32_t seq;
    };
    struct Built {
        uint32_t ref;
        Bound b;
        uint32_t need; // stack entries a walk below this child can hold at once
    };

    CompiledScene &cs_;
    OrderedOptions opt_;
    std::vector<std::vector<Item>> frames_; // one per tree: runs of the world frame, instances, medium boundaries
    std::vector<uint32_t> need_, root_;
    std::vector<Bound> bound_;
    std::vector<ONode> nodes_;
    std::vector<ONode4> wnodes_;
    std::vector<Sphere> new_spheres_;
    std::vector<Quad> new_quads_;

    struct SeqItem {
        uint32_t kind, frame, medium;
    };
    std::vector<SeqItem> sequence_;
    std::vector<uint32_t> frame_of_inst_;

    uint32_t new_frame() {
        frames_.emplace_back();
        return (uint32_t)frames_.size() - 1u;
    }
    // The threaded records list every primitive, frame change and medium in the reference's scan order.
    bool collect() {
        frames_.clear();
        sequence_.clear();
        frame_of_inst_.assign(cs_.instances.size(), 0u);
        enum Where : uint32_t { WORLD, INSTANCE, BOUNDARY };
        struct Open { uint32_t frame, where; };
        const uint32_t NO_FRAME = 0xffffffffu;
        std::vector<Open> open{{NO_FRAME, WORLD}}; // the world's current run of medium-free objects gets its frame lazily
        auto current = [&]() -> std::vector<Item> & {
            if (open.back().frame == NO_FRAME) {
                open.back().frame = new_frame();
                sequence_.push_back({OSEQ_TREE, open.back().frame, 0u});
            }
            return frames_[open.back().frame];
        };
        for (const Node &n : cs_.nodes) {
            const uint32_t kind = n.kind & NODE_KIND_MASK;
            if (kind == NK_SPHERES || kind == NK_QUADS) {
                for (uint32_t i = 0; i < n.b; ++i) {
                    Item it{};
                    it.kind = kind == NK_SPHERES ? OK_SPHERES : OK_QUADS;
                    it.index = n.a + i;
                    it.b = kind == NK_SPHERES ? sphere_bound(cs_.spheres[it.index]) : quad_bound(cs_.quads[it.index]);
                    it.seq = kind == NK_SPHERES ? cs_.spheres[it.index].seq_moving >> 1 : cs_.quads[it.index].seq;
                    for (int ax = 0; ax < 3; ++ax) {
                        if (!std::isfinite(it.b.lo[ax]) || !std::isfinite(it.b.hi[ax])) return false; // not a shape a box can hold
                        pad_axis(it.b.lo[ax], it.b.hi[ax]);
                    }
                    current().push_back(it);
                }
            } else if (kind == NK_INST_ENTER) {
                Item it{};
                it.kind = OK_INSTANCE;
                it.index = n.a;
                current().push_back(it);
                frame_of_inst_[n.a] = new_frame();
                open.push_back({frame_of_inst_[n.a], INSTANCE});
            } else if (kind == NK_INST_EXIT) {
                open.pop_back();
            } else if (kind == NK_MEDIUM_SPHERE || kind == NK_MEDIUM_ENTER) {
                if (open.size() != 1) return false; // a medium inside a frame or a boundary: threaded layout
                open.back().frame = NO_FRAME;       // the run of medium-free objects ends here
                if (kind == NK_MEDIUM_SPHERE) {
                    const Sphere &sp = cs_.spheres[cs_.media[n.a].first_node];
                    if (!std::isfinite(sp.radius) || !std::isfinite(sp.center[0]) || !std::isfinite(sp.center[1]) || !std::isfinite(sp.center[2]))
                        return false;
                    sequence_.push_back({OSEQ_MEDIUM_SPHERE, 0u, n.a});
                } else {
                    const uint32_t f = new_frame();
                    sequence_.push_back({OSEQ_MEDIUM, f, n.a});
                    open.push_back({f, BOUNDARY});
                }
            } else if (kind == NK_MEDIUM_EXIT) {
                open.pop_back();
            } else if (kind != NK_INNER) {
                return false;
            }
        }
        return true;
    }

    double item_cost(const Item &it) const {
        return it.kind == OK_SPHERES ? opt_.cost_sphere : (it.kind == OK_QUADS ? opt_.cost_quad : opt_.cost_instance);
    }
    static double half_area(const Bound &b) {
        if (b.empty()) return 0.0;
        const double x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
        return x * y + y * z + z * x;
    }
    static uint32_t levels_for(size_t n) { // inner records on a path when n single-item leaves are split evenly
        uint32_t l = 0;
        while (((size_t)1 << l) < n) ++l;
        return l;
    }

    uint32_t alloc_node() {
        ONode n{};
        n.c[0] = n.c[1] = OK_EMPTY << OREF_KIND_SHIFT;
        nodes_.push_back(n);
        return (uint32_t)nodes_.size() - 1u;
    }
    static void set_child(ONode &n, int slot, const Built &c) {
        float *b = slot ? n.b1 : n.b0;
        for (int ax = 0; ax < 3; ++ax) {
            b[2 * ax] = round_down(c.b.lo[ax]);
            b[2 * ax + 1] = round_up(c.b.hi[ax]);
        }
        n.c[slot] = c.ref;
    }

    // ---- four-child records: the binary tree of frame f, collapsed ------------------------------------------------------------
    // A record's children are the binary record's two; as long as there is room, the inner child with the largest box is replaced
    // by its own two children.  The mask scheme of the walk (rt_layout.h) keeps ONE stack entry per record on the path, so a walk
    // below a record needs 1 + the deepest child's entries.
    struct Wide { uint32_t ref; uint32_t need; };
    Wide widen(uint32_t bin_id) {
        struct Child { float b[6]; uint32_t ref; };
        std::vector<Child> ch;
        auto add_children_of = [&](uint32_t id) {
            const ONode &n = nodes_[id];
            for (int k = 0; k < 2; ++k)
                if ((n.c[k] >> OREF_KIND_SHIFT) != OK_EMPTY) {
                    Child c;
                    memcpy(c.b, k ? n.b1 : n.b0, sizeof c.b);
                    c.ref = n.c[k];
                    ch.push_back(c);
                }
        };
        add_children_of(bin_id);
        auto area = [](const float *b) { const double x = (double)b[1] - b[0], y = (double)b[3] - b[2], z = (double)b[5] - b[4]; return x * y + y * z + z * x; };
        for (;;) {
            int best = -1;
            for (size_t i = 0; i < ch.size(); ++i)
                if ((ch[i].ref >> OREF_KIND_SHIFT) == OK_INNER) {
                    const ONode &n = nodes_[ch[i].ref];
                    const size_t grand = ((n.c[0] >> OREF_KIND_SHIFT) != OK_EMPTY) + ((n.c[1] >> OREF_KIND_SHIFT) != OK_EMPTY);
                    if (ch.size() - 1 + grand <= 4 && (best < 0 || area(ch[i].b) > area(ch[(size_t)best].b))) best = (int)i;
                }
            if (best < 0) break;
            const uint32_t id = ch[(size_t)best].ref;
            ch.erase(ch.begin() + best);
            add_children_of(id);
        }
        const uint32_t wid = (uint32_t)wnodes_.size();
        wnodes_.emplace_back();
        uint32_t need = 0;
        ONode4 rec{};
        for (int k = 0; k < 4; ++k) {
            rec.c[k] = OK_EMPTY << OREF_KIND_SHIFT;
            for (int ax = 0; ax < 3; ++ax) { rec.b[k][2 * ax] = INFINITY; rec.b[k][2 * ax + 1] = -INFINITY; }
        }
        for (size_t i = 0; i < ch.size(); ++i) {
            memcpy(rec.b[i], ch[i].b, sizeof ch[i].b);
            const uint32_t kind = ch[i].ref >> OREF_KIND_SHIFT;
            if (kind == OK_INNER) {
                const Wide w = widen(ch[i].ref);
                rec.c[i] = w.ref;
                need = std::max(need, w.need);
            } else {
                rec.c[i] = ch[i].ref;
                if (kind == OK_INSTANCE) need = std::max(need, 1u + need_[frame_of_inst_[ch[i].ref & OREF_INDEX_MASK]]); // exit marker + the walk inside
            }
        }
        wnodes_[wid] = rec;
        return Wide{wid, 1u + need};
    }

    void build_frame(size_t f) {
        build_binary_frame(f);
        if (opt_.wide && need_[f] <= ORDERED_MAX_STACK) {
            const Wide w = widen(root_[f]);
            root_[f] = w.ref;
            need_[f] = w.need;
        }
    }

    void build_binary_frame(size_t f) {
        std::vector<Item> &items = frames_[f];
        // a frame's own depth, the exit marker and the deepest frame below it share the ORDERED_MAX_STACK stack entries
        uint32_t below = 0;
        for (const Item &it : items)
            if (it.kind == OK_INSTANCE) below = std::max(below, 1u + need_[frame_of_inst_[it.index]]);
        uint32_t budget = std::min(opt_.world_depth, levels_for(items.size()) + opt_.frame_slack);
        if (below >= ORDERED_MAX_STACK) { need_[f] = ORDERED_MAX_STACK + 1; return; }
        budget = std::min(budget, ORDERED_MAX_STACK - below);
        if (items.empty()) { // nothing to hit: a record with two empty children
            root_[f] = alloc_node();
            need_[f] = 1;
            return;
        }
        if (levels_for(items.size()) > budget) { need_[f] = ORDERED_MAX_STACK + 1; return; }
        // A frame of a few primitives (Cornell's six walls, the six faces of its boxes): a tree over them culls next to nothing — most rays
        // visit most of its records, each on its own schedule — whereas the lanes that enter the frame together (a shade or frame-change
        // round has just served them) can test ONE leaf of them all in one round of the primitive stage.  The root's other child is the
        // tree of the frame's instances.
        {
            const size_t n_prims = (size_t)std::count_if(items.begin(), items.end(), [](const Item &it) { return it.kind != OK_INSTANCE; });
            std::stable_partition(items.begin(), items.end(), [](const Item &it) { return it.kind != OK_INSTANCE; });
            bool same = n_prims >= 2 && n_prims <= std::min<size_t>(opt_.flat_max, OREF_MAX_LEAF);
            for (size_t i = 1; i < n_prims && same; ++i) same = items[i].kind == items[0].kind;
            if (same && budget >= 1u + (items.size() > n_prims ? levels_for(items.size() - n_prims) : 0u)) {
                const uint32_t id = alloc_node();
                const Built l = make_leaf(items, 0, n_prims);
                set_child(nodes_[id], 0, l);
                uint32_t need = 1;
                if (items.size() > n_prims) {
                    const Built r = build(items, n_prims, items.size(), budget - 1);
                    set_child(nodes_[id], 1, r);
                    need = 1u + r.need;
                }
                root_[f] = id;
                need_[f] = need;
                return;
            }
        }
        Built r = build(items, 0, items.size(), budget);
        if ((r.ref >> OREF_KIND_SHIFT) != OK_INNER) { // a frame's root is always a record
            const uint32_t id = alloc_node();
            set_child(nodes_[id], 0, r);
            r.ref = id;
            r.need += 1;
        }
        root_[f] = r.ref;
        need_[f] = r.need;
    }

    Built make_leaf(std::vector<Item> &items, size_t lo, size_t hi) {
        Built out{};
        for (size_t i = lo; i < hi; ++i) out.b.add(items[i].b);
        const Item &first = items[lo];
        if (first.kind == OK_INSTANCE) {
            out.ref = (OK_INSTANCE << OREF_KIND_SHIFT) | first.index;
            out.need = 1u + need_[frame_of_inst_[first.index]]; // the frame-exit marker, then the walk inside
            return out;
        }
        // primitives of a leaf sit next to each other in the table, in scan order
        std::sort(items.begin() + (ptrdiff_t)lo, items.begin() + (ptrdiff_t)hi, [](const Item &a, const Item &b) { return a.seq < b.seq; });
        uint32_t start;
        if (first.kind == OK_SPHERES) {
            start = (uint32_t)new_spheres_.size();
            for (size_t i = lo; i < hi; ++i) new_spheres_.push_back(cs_.spheres[items[i].index]);
        } else {
            start = (uint32_t)new_quads_.size();
            for (size_t i = lo; i < hi; ++i) new_quads_.push_back(cs_.quads[items[i].index]);
        }
        out.ref = (first.kind << OREF_KIND_SHIFT) | ((uint32_t)(hi - lo - 1) << OREF_COUNT_SHIFT) | start;
        out.need = 0;
        return out;
    }

    // SAH split of items [lo, hi): returns the position to cut at after reordering the range, or lo if none was found
    size_t sah_split(std::vector<Item> &items, size_t lo, size_t hi, double &best_cost) {
        const size_t n = hi - lo;
        Bound all;
        for (size_t i = lo; i < hi; ++i) all.add(items[i].b);
        const double inv_area = 1.0 / std::fmax(half_area(all), 1e-300);
        best_cost = INFINITY;
        int best_axis = -1;
        size_t best_cut = 0;
        std::vector<double> right_area(n), right_cost(n);
        std::vector<uint32_t> order(n);
        for (int ax = 0; ax < 3; ++ax) {
            std::iota(order.begin(), order.end(), 0u);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return items[lo + a].c[ax] < items[lo + b].c[ax]; });
            Bound acc;
            double cost = 0.0;
            for (size_t k = n; k-- > 0;) {
                acc.add(items[lo + order[k]].b);
                cost += item_cost(items[lo + order[k]]);
                right_area[k] = half_area(acc);
                right_cost[k] = cost;
            }
            acc = Bound();
            cost = 0.0;
            for (size_t k = 0; k + 1 < n; ++k) { // cut after position k
                acc.add(items[lo + order[k]].b);
                cost += item_cost(items[lo + order[k]]);
                const double c = opt_.cost_node + (half_area(acc) * cost + right_area[k + 1] * right_cost[k + 1]) * inv_area;
                if (c < best_cost) { best_cost = c; best_axis = ax; best_cut = k + 1; }
            }
        }
        if (best_axis < 0) return lo;
        const int ax = best_axis; // the same stable sort again: the order the cut was found in
        std::stable_sort(items.begin() + (ptrdiff_t)lo, items.begin() + (ptrdiff_t)hi, [ax](const Item &a, const Item &b) { return a.c[ax] < b.c[ax]; });
        return lo + best_cut;
    }
    // even split along the widest axis of the centroids: what is left when the depth budget is nearly used up
    size_t median_split(std::vector<Item> &items, size_t lo, size_t hi) {
        double cl[3] = {INFINITY, INFINITY, INFINITY}, ch[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = lo; i < hi; ++i)
            for (int ax = 0; ax < 3; ++ax) { cl[ax] = std::fmin(cl[ax], items[i].c[ax]); ch[ax] = std::fmax(ch[ax], items[i].c[ax]); }
        int ax = 0;
        if (ch[1] - cl[1] > ch[ax] - cl[ax]) ax = 1;
        if (ch[2] - cl[2] > ch[ax] - cl[ax]) ax = 2;
        std::stable_sort(items.begin() + (ptrdiff_t)lo, items.begin() + (ptrdiff_t)hi, [ax](const Item &a, const Item &b) { return a.c[ax] < b.c[ax]; });
        return lo + (hi - lo + 1) / 2;
    }

    Built build(std::vector<Item> &items, size_t lo, size_t hi, uint32_t budget) {
        const size_t n = hi - lo;
        if (n == 1) return make_leaf(items, lo, hi);
        bool can_leaf = n <= opt_.leaf_max && n <= OREF_MAX_LEAF && items[lo].kind != OK_INSTANCE;
        double leaf_cost = 0.0;
        for (size_t i = lo; i < hi; ++i) {
            can_leaf = can_leaf && items[i].kind == items[lo].kind;
            leaf_cost += item_cost(items[i]);
        }
        size_t cut = lo;
        if (budget > levels_for(n)) {
            double split_cost;
            cut = sah_split(items, lo, hi, split_cost);
            if (can_leaf && (cut == lo || leaf_cost <= split_cost)) return make_leaf(items, lo, hi);
        } else if (can_leaf) {
            return make_leaf(items, lo, hi);
        }
        if (cut == lo) cut = median_split(items, lo, hi);
        const uint32_t id = alloc_node();
        const Built l = build(items, lo, cut, budget - 1);
        const Built r = build(items, cut, hi, budget - 1);
        set_child(nodes_[id], 0, l);
        set_child(nodes_[id], 1, r);
        Built out{};
        out.ref = id; // OK_INNER << 29 == 0
        out.b = l.b;
        out.b.add(r.b);
        out.need = 1u + std::max(l.need, r.need);
        return out;
    }
};

inline bool build_ordered(CompiledScene &cs, const OrderedOptions &opt = OrderedOptions()) { return OrderedBuilder(cs, opt).run(); }

} // namespace rtd
