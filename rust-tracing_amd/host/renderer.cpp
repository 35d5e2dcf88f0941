#include "renderer.hpp"
#include "color.hpp"
#include "image_io.hpp"
#include "rt_amd.h"
#include <chrono>
#include <cstdio>
#include <stdexcept>
#include <thread>

namespace rt {

std::vector<double> render_sums(const Camera &camera, const Hittable &world, const RenderOptions &opt,
                                const std::function<void(const std::vector<double> &, int)> &on_pass) {
    SceneDescriber sd;
    const rt_ref root = world.describe(sd);
    const rt_scene_desc desc = sd.desc(root);
    const rt_camera cam = camera.pod();

    const int ngpu = opt.gpus < 1 ? 1 : opt.gpus;
    if (rt_device_count() < ngpu)
        throw std::runtime_error("render: " + std::to_string(ngpu) + " GPU(s) requested, " +
                                 std::to_string(rt_device_count()) + " visible");

    std::vector<rt_scene *> scenes((size_t)ngpu, nullptr);
    auto destroy_all = [&]() { for (rt_scene *s : scenes) rt_scene_destroy(s); };
    for (int g = 0; g < ngpu; ++g)
        if (rt_scene_create(&desc, g, &scenes[(size_t)g]) != RT_OK) {
            const std::string msg = rt_last_error();
            destroy_all();
            throw std::runtime_error("render: " + msg);
        }

    std::vector<double> sums((size_t)cam.image_width * (size_t)cam.image_height * 3u, 0.0);
    const int spp = cam.samples_per_pixel;
    const int pass = opt.progressive_spp > 0 ? opt.progressive_spp : spp;
    const int32_t w = cam.image_width, h = cam.image_height;

    // Device route (include/rt_amd.h "frame-end gather"): every device renders its tiles (k % ngpu == g) into its own tile buffer in
    // HBM; at the end of a pass ONE grouped RCCL exchange brings the buffers to device 0 over xGMI, which reassembles the frame and
    // hands it to the host in one copy.  If RCCL cannot be loaded the frame travels through host memory instead (below).
    std::vector<rt_comm *> comms((size_t)ngpu, nullptr);
    // (one GPU: nothing to gather — no communicator, no RCCL: the frame comes back through rt_render's own copy, below)
    const bool device_route = ngpu > 1 && rt_comm_create_all(ngpu, nullptr, comms.data()) == RT_OK;
    if (device_route) {
        const int64_t stride = rt_out_size(w, h, RT_OUT_TILES, 0, ngpu);
        std::vector<void *> tiles((size_t)ngpu, nullptr);
        void *gathered = nullptr, *frame = nullptr;
        auto cleanup = [&]() {
            for (int g = 0; g < ngpu; ++g) { rt_device_free(g, tiles[(size_t)g]); rt_comm_destroy(comms[(size_t)g]); }
            rt_device_free(0, gathered); rt_device_free(0, frame);
            destroy_all();
        };
        bool ok = rt_device_malloc(0, stride * ngpu * (int64_t)sizeof(double), &gathered) == RT_OK &&
                  rt_device_malloc(0, (int64_t)sums.size() * (int64_t)sizeof(double), &frame) == RT_OK;
        for (int g = 0; g < ngpu && ok; ++g) ok = rt_device_malloc(g, stride * (int64_t)sizeof(double), &tiles[(size_t)g]) == RT_OK;
        if (!ok) { const std::string msg = rt_last_error(); cleanup(); throw std::runtime_error("render: " + msg); }
        for (int begin = 0; begin < spp; begin += pass) {
            const int end = begin + pass < spp ? begin + pass : spp;
            std::vector<std::string> errors((size_t)ngpu);
            auto on_every_device = [&](auto &&fn) { // one host thread per device
                std::vector<std::thread> workers;
                for (int g = 0; g < ngpu; ++g) workers.emplace_back([&, g]() { if (fn(g) != RT_OK) errors[(size_t)g] = rt_last_error(); });
                for (auto &t : workers) t.join();
                for (const auto &e : errors)
                    if (!e.empty()) { cleanup(); throw std::runtime_error("render: " + e); }
            };
            // Two phases with a join in between: a rank enters the grouped exchange only when EVERY rank has rendered — a rank that
            // failed and stayed out would leave its peers' ncclSend / ncclRecv pending for ever (and cleanup() waiting on them).
            on_every_device([&](int g) {
                rt_render_params p{};
                p.seed = opt.seed; p.sample_begin = begin; p.sample_end = end; p.max_depth = cam.max_depth;
                p.accumulate = begin > 0 ? 1 : 0; // the tile buffers keep the running sums between passes
                p.shard_index = g; p.shard_count = ngpu; p.out_layout = RT_OUT_TILES; p.device = g;
                return rt_render_device(scenes[(size_t)g], &cam, &p, static_cast<double *>(tiles[(size_t)g]), nullptr);
            });
            on_every_device([&](int g) {
                return rt_gather_tiles_device(comms[(size_t)g], w, h, 8, tiles[(size_t)g], g == 0 ? gathered : nullptr, 0, nullptr);
            });
            if (rt_tiles_to_frame_device(w, h, ngpu, static_cast<const double *>(gathered), static_cast<double *>(frame), nullptr) != RT_OK ||
                rt_device_download(0, sums.data(), frame, (int64_t)sums.size() * (int64_t)sizeof(double), nullptr) != RT_OK) {
                const std::string msg = rt_last_error(); cleanup(); throw std::runtime_error("render: " + msg);
            }
            if (on_pass) on_pass(sums, end);
        }
        cleanup();
        return sums;
    }

    for (int begin = 0; begin < spp; begin += pass) {
        const int end = begin + pass < spp ? begin + pass : spp;
        std::vector<std::string> errors((size_t)ngpu);
        std::vector<std::thread> workers;
        // one host thread per device; each renders the tiles k with k % ngpu == g into its own pixels of `sums`
        for (int g = 0; g < ngpu; ++g) {
            workers.emplace_back([&, g]() {
                rt_render_params p{};
                p.seed = opt.seed;
                p.sample_begin = begin;
                p.sample_end = end;
                p.max_depth = cam.max_depth;
                p.accumulate = begin > 0 ? 1 : 0;
                p.shard_index = g;
                p.shard_count = ngpu;
                p.out_layout = RT_OUT_FRAME;
                p.device = g;
                if (rt_render(scenes[(size_t)g], &cam, &p, sums.data()) != RT_OK) errors[(size_t)g] = rt_last_error();
            });
        }
        for (auto &w : workers) w.join();
        for (const auto &e : errors)
            if (!e.empty()) {
                destroy_all();
                throw std::runtime_error("render: " + e);
            }
        if (on_pass) on_pass(sums, end);
    }
    destroy_all();
    return sums;
}

std::vector<uint8_t> resolve_rgb8(const std::vector<double> &sums, int32_t spp) {
    std::vector<uint8_t> px(sums.size());
    const FP inv = 1.0 / (FP)spp; // `c / spp as FP` is a multiply by the reciprocal (src/vec3.rs:244-249)
    for (size_t i = 0; i + 2 < sums.size(); i += 3) {
        const auto rgb = color_to_rgb(Color(sums[i] * inv, sums[i + 1] * inv, sums[i + 2] * inv));
        px[i] = rgb[0]; px[i + 1] = rgb[1]; px[i + 2] = rgb[2];
    }
    return px;
}

void render(std::shared_ptr<Camera> camera, std::shared_ptr<Hittable> world, const std::string &output_file_name,
            const RenderOptions &opt) {
    using clock = std::chrono::steady_clock;
    auto now = clock::now();
    const int32_t w = (int32_t)camera->image_width, h = (int32_t)camera->image_height;
    // progressive passes: the PNG on disk always shows the mean of the samples traced so far
    auto on_pass = [&](const std::vector<double> &partial, int done) {
        if (opt.progressive_spp <= 0 || done >= camera->samples_per_pixel) return;
        write_png_rgb8(output_file_name + ".png", w, h, resolve_rgb8(partial, done).data());
        if (!opt.quiet) printf("  %d / %d spp\n", done, camera->samples_per_pixel);
    };
    const std::vector<double> sums = render_sums(*camera, *world, opt, on_pass);
    const double render_s = std::chrono::duration<double>(clock::now() - now).count();
    if (!opt.quiet) {
        const double msamples = (double)camera->image_width * (double)camera->image_height *
                                (double)camera->samples_per_pixel / 1e6;
        printf("Render time: %.2fs (%.1f Msamples/s)\n", render_s, msamples / render_s);
    }

    now = clock::now();
    const std::vector<uint8_t> px = resolve_rgb8(sums, camera->samples_per_pixel);
    if (!write_png_rgb8(output_file_name + ".png", (int32_t)camera->image_width, (int32_t)camera->image_height,
                        px.data()))
        throw std::runtime_error("Should've encoded the image into a file."); // src/renderer.rs:72
    if (!opt.quiet)
        printf("PNG encoding: %.2fs\n", std::chrono::duration<double>(clock::now() - now).count());
}

} // namespace rt
