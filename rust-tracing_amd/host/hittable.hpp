// The Hittable trait, HittableList, Translate and RotateY, host side: construction, bounding boxes and
// describe().  `hit()` runs on the GPU (csrc/rt_kernel.hip).
//   reference: src/hittable.rs:45-193
#pragma once
#include "describe.hpp"
#include "material.hpp"
#include "vec3.hpp"
#include <memory>
#include <vector>

namespace rt {

class Hittable {
  public:
    virtual ~Hittable() = default;
    virtual AABB bounding_box() const = 0;
    rt_ref describe(SceneDescriber &sd) const {
        auto it = sd.seen_hittables.find(this);
        if (it != sd.seen_hittables.end()) return it->second;
        rt_ref r = record(sd);
        sd.seen_hittables.emplace(this, r);
        return r;
    }

  protected:
    virtual rt_ref record(SceneDescriber &sd) const = 0;
};

// src/hittable.rs:50-79
class HittableList : public Hittable {
  public:
    std::vector<std::shared_ptr<Hittable>> objects;

    void add(std::shared_ptr<Hittable> object) {
        // note: starts from the all-zero default box, as the reference's derive(Default) does (src/hittable.rs:50-57)
        bbox = AABB::from_aabbs(bbox, object->bounding_box());
        objects.push_back(std::move(object));
    }
    AABB bounding_box() const override { return bbox; }

  protected:
    rt_ref record(SceneDescriber &sd) const override {
        std::vector<rt_ref> items;
        items.reserve(objects.size());
        for (const auto &o : objects) items.push_back(o->describe(sd));
        rt_list l{(int32_t)sd.list_items.size(), (int32_t)items.size()};
        sd.list_items.insert(sd.list_items.end(), items.begin(), items.end());
        sd.lists.push_back(l);
        return rt_ref{RT_HITTABLE_LIST, (int32_t)sd.lists.size() - 1};
    }

  private:
    AABB bbox;
};

// src/hittable.rs:81-111
class Translate : public Hittable {
  public:
    Translate(std::shared_ptr<Hittable> object_, const Vec3 &offset_)
        : object(std::move(object_)), offset(offset_), bbox(object->bounding_box() + offset_) {}
    AABB bounding_box() const override { return bbox; }

  protected:
    rt_ref record(SceneDescriber &sd) const override {
        rt_translate t{object->describe(sd), offset.pod()};
        sd.translates.push_back(t);
        return rt_ref{RT_HITTABLE_TRANSLATE, (int32_t)sd.translates.size() - 1};
    }

  private:
    std::shared_ptr<Hittable> object;
    Vec3 offset;
    AABB bbox;
};

// src/hittable.rs:113-193
class RotateY : public Hittable {
  public:
    RotateY(std::shared_ptr<Hittable> object_, FP angle) : object(std::move(object_)) {
        const FP theta = degrees_to_radians(angle);
        sin_theta = std::sin(theta);
        cos_theta = std::cos(theta);
        const AABB b = object->bounding_box();

        Point3 min = Point3::INFINITY_();
        Point3 max = Point3::NEG_INFINITY_();
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    const FP x = (FP)i * b.x.max + (1.0 - (FP)i) * b.x.min;
                    const FP y = (FP)j * b.y.max + (1.0 - (FP)j) * b.y.min;
                    const FP z = (FP)k * b.z.max + (1.0 - (FP)k) * b.z.min;

                    const FP new_x = cos_theta * x + sin_theta * z;
                    const FP new_z = -sin_theta * x + cos_theta * z;

                    const Vec3 tester(new_x, y, new_z);
                    for (int c = 0; c < 3; ++c) {
                        min[c] = std::fmin(min[c], tester[c]);
                        max[c] = std::fmax(max[c], tester[c]);
                    }
                }
        bbox = AABB::from_points(min, max);
    }
    AABB bounding_box() const override { return bbox; }

  protected:
    rt_ref record(SceneDescriber &sd) const override {
        rt_rotate_y r{object->describe(sd), sin_theta, cos_theta};
        sd.rotates.push_back(r);
        return rt_ref{RT_HITTABLE_ROTATE_Y, (int32_t)sd.rotates.size() - 1};
    }

  private:
    std::shared_ptr<Hittable> object;
    FP sin_theta, cos_theta;
    AABB bbox;
};

} // namespace rt
