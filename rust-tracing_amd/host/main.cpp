// Command-line front end.  reference: src/main.rs:40-54 (Args) and :641-669 (main)
//   -s/--scene N, -o/--output NAME as in the reference; -l/--live is accepted and refused (no window system
//   on a GPU node).  Added: --width/--aspect/--spp/--depth (BASELINE.json's configs change these),
//   --seed/--scene-seed, --gpus, --earth PATH|synthetic:WxH, --bvh reference|sah, --progressive N (rewrite the PNG
//   every N samples per pixel: what -l/--live shows in a window, written to the file instead).
#include "renderer.hpp"
#include "scenes.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

using namespace rt;

static void usage(const char *argv0) {
    fprintf(stderr,
            "Usage: %s [-s SCENE] [-o OUTPUT] [--width W] [--aspect A] [--spp N] [--depth D]\n"
            "          [--seed S] [--scene-seed S] [--gpus N] [--progressive SPP_PER_PASS] [--earth PATH|synthetic:WxH] [--bvh reference|sah]\n"
            "  scenes: 0 random balls, 1 two spheres, 2 earth, 3 perlin spheres, 4 quads, 5 simple light,\n"
            "          6 cornell box, 7 cornell smoke, 8 final scene\n",
            argv0);
}

int main(int argc, char **argv) {
    int scene = 0;
    bool live = false;
    std::string output = "output";
    SceneOptions so;
    so.earth_image = "assets/earth-large.jpg"; // the reference's default (src/main.rs:179,:591); --earth synthetic:WxH needs no file
    RenderOptions ro;
    uint64_t scene_seed = 1;

    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char *name) -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", name); usage(argv[0]); exit(2); }
            return argv[++i];
        };
        if (a == "-l" || a == "--live") live = true;
        else if (a == "-s" || a == "--scene") scene = atoi(need("--scene"));
        else if (a == "-o" || a == "--output") output = need("--output");
        else if (a == "--width") so.image_width = atoll(need("--width"));
        else if (a == "--aspect") so.aspect_ratio = atof(need("--aspect"));
        else if (a == "--spp") so.samples_per_pixel = atoi(need("--spp"));
        else if (a == "--depth") so.max_depth = atoi(need("--depth"));
        else if (a == "--seed") ro.seed = strtoull(need("--seed"), nullptr, 10);
        else if (a == "--scene-seed") scene_seed = strtoull(need("--scene-seed"), nullptr, 10);
        else if (a == "--gpus") ro.gpus = atoi(need("--gpus"));
        else if (a == "--progressive") ro.progressive_spp = atoi(need("--progressive"));
        else if (a == "--earth") so.earth_image = need("--earth");
        else if (a == "--bvh") bvh_policy() = std::string(need("--bvh")) == "sah" ? BvhPolicy::Sah : BvhPolicy::Reference;
        else if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); usage(argv[0]); return 2; }
    }
    printf("Args: { live: %s, scene: %d, output: \"%s\" }\n", live ? "true" : "false", scene, output.c_str());
    if (live) {
        fprintf(stderr, "live rendering needs a window system and is not available in this build\n");
        return 2;
    }

    try {
        seed_rng(scene_seed);
        auto [world, camera] = build_scene(scene, so);

        auto now = std::chrono::steady_clock::now();
        auto bvh = std::make_shared<BVHNode>(world);
        printf("Building BVH: %.2fms\n",
               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - now).count());

        render(std::make_shared<Camera>(camera), bvh, output, ro);
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
