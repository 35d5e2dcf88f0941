// BVHNode: the host-side acceleration-structure build.  Traversal runs on the GPU (csrc/rt_kernel.hip).
//   reference: src/bvh.rs:12-84
// Default policy = the reference's: one random axis per call (drawn for leaves too, src/bvh.rs:32), objects
// sorted by bounding-box minimum on that axis, split at len/2, spans of 1 and 2 special-cased (:42-57).
// The `sah` policy (binned surface-area heuristic) is this build's own alternative: it
// yields the same closest hits with far fewer box tests; the tree it produces is consumed identically by
// the GPU path and by the oracle, so parity tests cover both policies.
#pragma once
#include "hittable.hpp"
#include <algorithm>
#include <memory>
#include <stdexcept>

namespace rt {

enum class BvhPolicy { Reference, Sah };
BvhPolicy &bvh_policy(); // the calling thread's; default Reference

class BVHNode : public Hittable {
  public:
    explicit BVHNode(HittableList &list) : BVHNode(list.objects) {}
    explicit BVHNode(std::vector<std::shared_ptr<Hittable>> &objects) {
        if (objects.empty()) throw std::runtime_error("BVHNode: empty object list"); // the reference would recurse forever
        root = node_from_list(objects.data(), objects.size());
    }
    AABB bounding_box() const override { return root->bbox; }

  protected:
    rt_ref record(SceneDescriber &sd) const override {
        rt_bvh b{emit(sd, *root), 0};
        sd.bvhs.push_back(b);
        return rt_ref{RT_HITTABLE_BVH, (int32_t)sd.bvhs.size() - 1};
    }

  private:
    // one `(Node, AABB)` pair (src/bvh.rs:16-19)
    struct NodeBox {
        AABB bbox;
        std::unique_ptr<NodeBox> left, right; // Branch
        std::shared_ptr<Hittable> leaf;       // Leaf
    };
    std::unique_ptr<NodeBox> root;

    static int32_t emit(SceneDescriber &sd, const NodeBox &n) {
        rt_bvh_node r{};
        r.bbox = n.bbox.pod();
        if (n.leaf) {
            r.is_leaf = 1;
            r.left = r.right = -1;
            r.object = n.leaf->describe(sd);
        } else {
            r.is_leaf = 0;
            r.object = rt_ref{RT_HITTABLE_NONE, -1};
        }
        int32_t idx = (int32_t)sd.bvh_nodes.size();
        sd.bvh_nodes.push_back(r);
        if (!n.leaf) {
            int32_t l = emit(sd, *n.left);
            int32_t rr = emit(sd, *n.right);
            sd.bvh_nodes[idx].left = l;
            sd.bvh_nodes[idx].right = rr;
        }
        return idx;
    }

    static std::unique_ptr<NodeBox> make_leaf(const std::shared_ptr<Hittable> &obj) {
        auto n = std::make_unique<NodeBox>();
        n->bbox = obj->bounding_box();
        n->leaf = obj;
        return n;
    }
    static std::unique_ptr<NodeBox> make_branch(std::unique_ptr<NodeBox> l, std::unique_ptr<NodeBox> r) {
        auto n = std::make_unique<NodeBox>();
        n->bbox = AABB::from_aabbs(l->bbox, r->bbox);
        n->left = std::move(l);
        n->right = std::move(r);
        return n;
    }

    // box_compare (src/bvh.rs:68-74) answers only Less/Greater; as a strict "less" predicate that is a.min < b.min
    static bool box_less(const std::shared_ptr<Hittable> &a, const std::shared_ptr<Hittable> &b, int axis) {
        return a->bounding_box().axis(axis).min < b->bounding_box().axis(axis).min;
    }

    static std::unique_ptr<NodeBox> node_from_list(std::shared_ptr<Hittable> *objects, size_t span) {
        if (bvh_policy() == BvhPolicy::Sah) return node_from_list_sah(objects, span);

        const int axis = (int)thread_rng().gen_range_inclusive_usize(0, 2);
        if (span == 1) {
            return make_leaf(objects[0]);
        } else if (span == 2) {
            const std::shared_ptr<Hittable> *left = &objects[0], *right = &objects[1];
            if (!box_less(*left, *right, axis)) std::swap(left, right);
            return make_branch(make_leaf(*left), make_leaf(*right));
        } else {
            // sort_unstable_by in the reference; ties are implementation-defined there, stable here
            std::stable_sort(objects, objects + span,
                             [axis](const auto &a, const auto &b) { return box_less(a, b, axis); });
            const size_t mid = span / 2;
            auto l = node_from_list(objects, mid);
            auto r = node_from_list(objects + mid, span - mid);
            return make_branch(std::move(l), std::move(r));
        }
    }

    static FP half_area(const AABB &b) {
        const FP dx = b.x.size(), dy = b.y.size(), dz = b.z.size();
        return dx * dy + dy * dz + dz * dx;
    }
    static AABB merge_range(std::shared_ptr<Hittable> *o, size_t n) {
        AABB b = o[0]->bounding_box();
        for (size_t i = 1; i < n; ++i) b = AABB::from_aabbs(b, o[i]->bounding_box());
        return b;
    }
    // Full-sweep SAH over the three axes (object counts here are <= a few thousand, so O(n log n) per level
    // is fine).  Children are ordered by centroid so that traversal stays "left first" along the split axis.
    static std::unique_ptr<NodeBox> node_from_list_sah(std::shared_ptr<Hittable> *objects, size_t span) {
        if (span == 1) return make_leaf(objects[0]);
        FP best_cost = __builtin_inf();
        int best_axis = 0;
        size_t best_split = span / 2;
        std::vector<FP> right_area(span);
        for (int axis = 0; axis < 3; ++axis) {
            std::stable_sort(objects, objects + span, [axis](const auto &a, const auto &b) {
                const Interval &ia = a->bounding_box().axis(axis), &ib = b->bounding_box().axis(axis);
                return ia.min + ia.max < ib.min + ib.max;
            });
            AABB acc = objects[span - 1]->bounding_box();
            for (size_t i = span - 1; i >= 1; --i) {
                acc = AABB::from_aabbs(acc, objects[i]->bounding_box());
                right_area[i] = half_area(acc);
            }
            acc = objects[0]->bounding_box();
            for (size_t i = 1; i < span; ++i) {
                // split: [0, i) | [i, span)
                const FP cost = half_area(acc) * (FP)i + right_area[i] * (FP)(span - i);
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = i; }
                acc = AABB::from_aabbs(acc, objects[i]->bounding_box());
            }
        }
        std::stable_sort(objects, objects + span, [best_axis](const auto &a, const auto &b) {
            const Interval &ia = a->bounding_box().axis(best_axis), &ib = b->bounding_box().axis(best_axis);
            return ia.min + ia.max < ib.min + ib.max;
        });
        auto l = node_from_list_sah(objects, best_split);
        auto r = node_from_list_sah(objects + best_split, span - best_split);
        return make_branch(std::move(l), std::move(r));
    }
};

} // namespace rt
