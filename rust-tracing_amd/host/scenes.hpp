// The nine scene builders.  reference: src/main.rs:56-639 (one function each, same names, same objects in the
// same order, same in-code camera settings); scene selection as `match args.scene` (src/main.rs:645-656).
#pragma once
#include "bvh.hpp"
#include "camera.hpp"
#include "constant_medium.hpp"
#include "hittable.hpp"
#include "quad.hpp"
#include "sphere.hpp"
#include <string>
#include <utility>

namespace rt {

// What BASELINE.json's configs change relative to the in-code settings: image size, aspect, spp, depth
// (values <= 0 keep the scene's own), plus where the earth texture comes from.
struct SceneOptions {
    int64_t image_width = 0;
    FP aspect_ratio = 0.0;
    int32_t samples_per_pixel = 0;
    int32_t max_depth = 0;
    std::string earth_image = "assets/earth-large.jpg"; // src/main.rs:179,:591
};

using SceneResult = std::pair<HittableList, Camera>;

SceneResult random_balls(const SceneOptions &o = {});
SceneResult two_spheres(const SceneOptions &o = {});
SceneResult earth(const SceneOptions &o = {});
SceneResult two_perlin_spheres(const SceneOptions &o = {});
SceneResult quads(const SceneOptions &o = {});
SceneResult simple_light(const SceneOptions &o = {});
SceneResult cornell_box(const SceneOptions &o = {});
SceneResult cornell_smoke(const SceneOptions &o = {});
SceneResult final_scene(const SceneOptions &o = {});

// scene index -> builder, unknown index -> random_balls (src/main.rs:645-656)
SceneResult build_scene(int scene, const SceneOptions &o = {});
const char *scene_name(int scene);

} // namespace rt
