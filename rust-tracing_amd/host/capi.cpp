// C entry points of librt_host.so (include/rt_host.h).
#include "rt_host.h"
#include "color.hpp"
#include "image_io.hpp"
#include "renderer.hpp"
#include "scenes.hpp"
#include <cstring>
#include <exception>
#include <string>

using namespace rt;

struct rth_scene {
    HittableList world;             // the list main hands to BVHNode::new (kept alive: the BVH shares its objects)
    std::shared_ptr<BVHNode> bvh;
    std::unique_ptr<Camera> camera;
    SceneDescriber sd;
    rt_scene_desc desc;
    rt_camera cam;
};

static thread_local std::string g_error;
static int fail(const std::string &msg) {
    g_error = msg;
    return -1;
}

extern "C" {

const char *rth_last_error(void) { return g_error.c_str(); }

int rth_scene_build(const rth_scene_options *options, rth_scene **out_scene) {
    if (!options || !out_scene) return fail("rth_scene_build: null argument");
    *out_scene = nullptr;
    try {
        SceneOptions so;
        so.image_width = options->image_width;
        so.aspect_ratio = options->aspect_ratio;
        so.samples_per_pixel = options->samples_per_pixel;
        so.max_depth = options->max_depth;
        so.earth_image = options->earth_image ? options->earth_image : "synthetic:1024x512";

        // (the policy is this thread's and is put back afterwards: the caller's later builds see what they saw before)
        struct PolicyScope {
            BvhPolicy saved = bvh_policy();
            ~PolicyScope() { bvh_policy() = saved; }
        } policy_scope;
        bvh_policy() = options->bvh_policy == 1 ? BvhPolicy::Sah : BvhPolicy::Reference;
        seed_rng(options->scene_seed);

        auto s = std::make_unique<rth_scene>();
        auto built = build_scene(options->scene, so);
        s->world = std::move(built.first);
        s->camera = std::make_unique<Camera>(built.second);
        s->bvh = std::make_shared<BVHNode>(s->world);
        const rt_ref root = s->bvh->describe(s->sd);
        s->desc = s->sd.desc(root);
        s->cam = s->camera->pod();
        *out_scene = s.release();
        return 0;
    } catch (const std::exception &e) {
        return fail(std::string("rth_scene_build: ") + e.what());
    }
}

void rth_scene_destroy(rth_scene *scene) { delete scene; }
const rt_scene_desc *rth_scene_desc(const rth_scene *scene) { return scene ? &scene->desc : nullptr; }
const rt_camera *rth_scene_camera(const rth_scene *scene) { return scene ? &scene->cam : nullptr; }

int rth_resolve_rgb8(int32_t width, int32_t height, int32_t spp, const double *rgb_sum, uint8_t *out_rgb8) {
    if (!rgb_sum || !out_rgb8 || width <= 0 || height <= 0 || spp <= 0) return fail("rth_resolve_rgb8: bad argument");
    const size_t n = (size_t)width * (size_t)height;
    const FP inv = 1.0 / (FP)spp;
    for (size_t i = 0; i < n; ++i) {
        const auto rgb = color_to_rgb(Color(rgb_sum[3 * i] * inv, rgb_sum[3 * i + 1] * inv, rgb_sum[3 * i + 2] * inv));
        out_rgb8[3 * i] = rgb[0]; out_rgb8[3 * i + 1] = rgb[1]; out_rgb8[3 * i + 2] = rgb[2];
    }
    return 0;
}

int rth_write_png(const char *path, int32_t width, int32_t height, const uint8_t *rgb8) {
    if (!path || !rgb8) return fail("rth_write_png: null argument");
    return write_png_rgb8(path, width, height, rgb8) ? 0 : fail(std::string("rth_write_png: cannot write ") + path);
}

int rth_synthetic_earth(int32_t width, int32_t height, uint8_t *out_rgb8) {
    if (!out_rgb8) return fail("rth_synthetic_earth: null argument");
    try {
        const ImageRGB8 img = synthetic_earth(width, height);
        memcpy(out_rgb8, img.pixels->data(), img.pixels->size());
        return 0;
    } catch (const std::exception &e) {
        return fail(e.what());
    }
}

int rth_load_image(const char *path, int32_t *out_width, int32_t *out_height, uint8_t *out_rgb8, int64_t capacity) {
    if (!path || !out_width || !out_height) return fail("rth_load_image: null argument");
    try {
        const ImageRGB8 img = load_image_rgb8(path);
        *out_width = img.width;
        *out_height = img.height;
        if (out_rgb8) {
            if ((int64_t)img.pixels->size() > capacity) return fail("rth_load_image: buffer too small");
            memcpy(out_rgb8, img.pixels->data(), img.pixels->size());
        }
        return 0;
    } catch (const std::exception &e) {
        return fail(e.what());
    }
}

} // extern "C"
