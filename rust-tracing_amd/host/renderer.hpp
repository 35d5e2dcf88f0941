// render(): the batch entry point.  reference: src/renderer.rs:12-75
// Same contract as the reference's `render(camera, world, output_file_name)`: trace every pixel, then
// divide by spp, gamma-encode, quantise and write `<output_file_name>.png` — but the pixel loop
// (src/renderer.rs:26-49) is one call into librt_amd (include/rt_amd.h) instead of a rayon par_iter.
// live_render (src/renderer.rs:77-137) needs a window system and is not provided.
#pragma once
#include "camera.hpp"
#include "hittable.hpp"
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace rt {

struct RenderOptions {
    uint64_t seed = 1; // render seed (the reference's RNG is unseeded; see include/rt_amd.h "RNG")
    int gpus = 1;      // framebuffer tiles are dealt round-robin to this many devices
    bool quiet = false;
    // > 0: render in passes of this many samples per pixel and rewrite the PNG after every pass — the batch
    // counterpart of the reference's live_render (src/renderer.rs:77-137: one more sample per frame, running mean),
    // without the window.  The final image is bit-identical to a single-pass render (ranges accumulate exactly).
    int progressive_spp = 0;
};

// Returns the per-pixel sums (w*h*3 doubles, row-major) exactly like the reference's `raw_pixels`
// (src/renderer.rs:26-49).  Throws std::runtime_error if the GPU library reports an error.
// `on_pass(sums, samples_done)`, if given, is called after every progressive pass.
std::vector<double> render_sums(const Camera &camera, const Hittable &world, const RenderOptions &opt = {},
                                const std::function<void(const std::vector<double> &, int)> &on_pass = nullptr);

// color_to_rgb(c / spp) over the whole frame (src/renderer.rs:55-58)
std::vector<uint8_t> resolve_rgb8(const std::vector<double> &sums, int32_t spp);

void render(std::shared_ptr<Camera> camera, std::shared_ptr<Hittable> world, const std::string &output_file_name,
            const RenderOptions &opt = {});

} // namespace rt
