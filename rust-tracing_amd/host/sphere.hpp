// Sphere (static and moving), host side.  reference: src/sphere.rs:13-46,:91-93
#pragma once
#include "hittable.hpp"

namespace rt {

class Sphere : public Hittable {
  public:
    Sphere(const Point3 &center_, FP radius_, std::shared_ptr<Material> material_)
        : center(center_), radius(radius_), material(std::move(material_)), center_vec(Vec3::ZERO()),
          is_moving(false) {
        const Vec3 rvec = Vec3::splat(radius);
        bbox = AABB::from_points(center - rvec, center + rvec);
    }
    // builder-style, like `Sphere::new(..).with_target(target)` (src/sphere.rs:34-46)
    Sphere with_target(const Point3 &target) const {
        Sphere s = *this;
        const Vec3 rvec = Vec3::splat(radius);
        const AABB box1 = AABB::from_points(center - rvec, center + rvec);
        const AABB box2 = AABB::from_points(target - rvec, target + rvec);
        s.center_vec = target - center;
        s.is_moving = true;
        s.bbox = AABB::from_aabbs(box1, box2);
        return s;
    }
    AABB bounding_box() const override { return bbox; }

  protected:
    rt_ref record(SceneDescriber &sd) const override {
        rt_sphere s{};
        s.center = center.pod();
        s.radius = radius;
        s.center_vec = center_vec.pod();
        s.is_moving = is_moving ? 1 : 0;
        s.material = material->describe(sd);
        sd.spheres.push_back(s);
        return rt_ref{RT_HITTABLE_SPHERE, (int32_t)sd.spheres.size() - 1};
    }

  private:
    Point3 center;
    FP radius;
    std::shared_ptr<Material> material;
    Vec3 center_vec;
    bool is_moving;
    AABB bbox;
};

} // namespace rt
