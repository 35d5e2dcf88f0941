// ConstantMedium, host side.  reference: src/constant_medium.rs:14-31,:73-75
#pragma once
#include "hittable.hpp"

namespace rt {

class ConstantMedium : public Hittable {
  public:
    ConstantMedium(std::shared_ptr<Hittable> boundary_, FP density, std::shared_ptr<Texture> albedo)
        : boundary(std::move(boundary_)), neg_inv_density(-1.0 / density),
          phase_function(std::make_shared<Isotropic>(std::move(albedo))) {}
    static std::shared_ptr<ConstantMedium> new_from_color(std::shared_ptr<Hittable> boundary, FP density,
                                                          const Color &albedo) {
        return std::make_shared<ConstantMedium>(std::move(boundary), density, std::make_shared<SolidColor>(albedo));
    }
    AABB bounding_box() const override { return boundary->bounding_box(); }

  protected:
    rt_ref record(SceneDescriber &sd) const override {
        rt_constant_medium m{};
        m.boundary = boundary->describe(sd);
        m.neg_inv_density = neg_inv_density;
        m.phase_material = phase_function->describe(sd);
        sd.media.push_back(m);
        return rt_ref{RT_HITTABLE_CONSTANT_MEDIUM, (int32_t)sd.media.size() - 1};
    }

  private:
    std::shared_ptr<Hittable> boundary;
    FP neg_inv_density;
    std::shared_ptr<Material> phase_function;
};

} // namespace rt
