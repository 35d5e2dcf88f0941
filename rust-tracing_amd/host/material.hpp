// The Material trait and its five implementations, host side: construction + describe().
// `scatter()`/`emitted()` run on the GPU (csrc/rt_kernel.hip).
//   reference: src/material.rs:11-138
#pragma once
#include "texture.hpp"

namespace rt {

class Material {
  public:
    virtual ~Material() = default;
    int32_t describe(SceneDescriber &sd) const {
        auto it = sd.seen_materials.find(this);
        if (it != sd.seen_materials.end()) return it->second;
        rt_material m = record(sd);
        int32_t idx = (int32_t)sd.materials.size();
        sd.materials.push_back(m);
        sd.seen_materials.emplace(this, idx);
        return idx;
    }

  protected:
    virtual rt_material record(SceneDescriber &sd) const = 0;
    static rt_material blank(int32_t kind) {
        rt_material m{};
        m.kind = kind;
        m.texture = -1;
        return m;
    }
};

// src/material.rs:18-42
class Lambertian : public Material {
  public:
    explicit Lambertian(std::shared_ptr<Texture> albedo_) : albedo(std::move(albedo_)) {}
    std::shared_ptr<Texture> albedo;

  protected:
    rt_material record(SceneDescriber &sd) const override {
        rt_material m = blank(RT_MATERIAL_LAMBERTIAN);
        m.texture = albedo->describe(sd);
        return m;
    }
};

// src/material.rs:44-64 (fuzz is stored as given, not clamped)
class Metal : public Material {
  public:
    Metal(const Color &albedo_, FP fuzz_) : albedo(albedo_), fuzz(fuzz_) {}
    Color albedo;
    FP fuzz;

  protected:
    rt_material record(SceneDescriber &) const override {
        rt_material m = blank(RT_MATERIAL_METAL);
        m.albedo = albedo.pod();
        m.fuzz = fuzz;
        return m;
    }
};

// src/material.rs:66-104
class Dielectric : public Material {
  public:
    explicit Dielectric(FP ir_) : ir(ir_) {}
    FP ir;

  protected:
    rt_material record(SceneDescriber &) const override {
        rt_material m = blank(RT_MATERIAL_DIELECTRIC);
        m.ir = ir;
        return m;
    }
};

// src/material.rs:106-122
class DiffuseLight : public Material {
  public:
    explicit DiffuseLight(std::shared_ptr<Texture> emit_) : emit(std::move(emit_)) {}
    std::shared_ptr<Texture> emit;

  protected:
    rt_material record(SceneDescriber &sd) const override {
        rt_material m = blank(RT_MATERIAL_DIFFUSE_LIGHT);
        m.texture = emit->describe(sd);
        return m;
    }
};

// src/material.rs:124-138
class Isotropic : public Material {
  public:
    explicit Isotropic(std::shared_ptr<Texture> albedo_) : albedo(std::move(albedo_)) {}
    std::shared_ptr<Texture> albedo;

  protected:
    rt_material record(SceneDescriber &sd) const override {
        rt_material m = blank(RT_MATERIAL_ISOTROPIC);
        m.texture = albedo->describe(sd);
        return m;
    }
};

} // namespace rt
