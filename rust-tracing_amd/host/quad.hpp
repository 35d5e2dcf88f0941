// Quad and Quad::cube, host side.  reference: src/quad.rs:11-93,:135-137
#pragma once
#include "hittable.hpp"

namespace rt {

class Quad : public Hittable {
  public:
    Quad(const Vec3 &q_, const Vec3 &u_, const Vec3 &v_, std::shared_ptr<Material> mat_)
        : q(q_), u(u_), v(v_), mat(std::move(mat_)) {
        const Vec3 n = u.cross(v);
        normal = n.normalize();
        d = normal.dot(q);
        w = n / n.length_squared();
        bbox = AABB::from_points(q, q + u + v).pad();
    }
    AABB bounding_box() const override { return bbox; }

    // the six sides in the reference's order: front, right, back, left, top, bottom (src/quad.rs:45-93)
    static std::shared_ptr<HittableList> cube(const Point3 &a, const Point3 &b, std::shared_ptr<Material> mat) {
        auto sides = std::make_shared<HittableList>();

        const Point3 min(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z));
        const Point3 max(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z));

        const Vec3 dx(max.x - min.x, 0.0, 0.0);
        const Vec3 dy(0.0, max.y - min.y, 0.0);
        const Vec3 dz(0.0, 0.0, max.z - min.z);

        sides->add(std::make_shared<Quad>(Point3(min.x, min.y, max.z), dx, dy, mat));
        sides->add(std::make_shared<Quad>(Point3(max.x, min.y, max.z), -dz, dy, mat));
        sides->add(std::make_shared<Quad>(Point3(max.x, min.y, min.z), -dx, dy, mat));
        sides->add(std::make_shared<Quad>(Point3(min.x, min.y, min.z), dz, dy, mat));
        sides->add(std::make_shared<Quad>(Point3(min.x, max.y, max.z), dx, -dz, mat));
        sides->add(std::make_shared<Quad>(Point3(min.x, min.y, min.z), dx, dz, mat));
        return sides;
    }

  protected:
    rt_ref record(SceneDescriber &sd) const override {
        rt_quad r{};
        r.q = q.pod(); r.u = u.pod(); r.v = v.pod(); r.w = w.pod(); r.normal = normal.pod();
        r.d = d;
        r.material = mat->describe(sd);
        sd.quads.push_back(r);
        return rt_ref{RT_HITTABLE_QUAD, (int32_t)sd.quads.size() - 1};
    }

  private:
    Point3 q;
    Vec3 u, v, w;
    std::shared_ptr<Material> mat;
    AABB bbox;
    FP d;
    Vec3 normal;
};

} // namespace rt
