/*
 * rt_host.h — C entry points of the host library (librt_host.so): the reference's scene builders, BVH build,
 * camera construction and output stage, exposed so that non-C++ callers (the Python test-suite and bench.py,
 * a Rust binding) can obtain an rt_scene_desc / rt_camera pair to hand to librt_amd (rt_amd.h).
 * Everything here is CPU-only host logic; no function in this header traces a ray.
 *
 * reference: scene functions src/main.rs:56-639, BVHNode::new src/bvh.rs:21-66 (called at src/main.rs:659),
 *            Camera::new src/camera.rs:54-110, color_to_rgb src/color.rs:12-19, PNG output src/renderer.rs:53-74.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene; /* opaque: owns the object graph and its POD description */

typedef struct rth_scene_options {
    int32_t scene;             /* 0..8 as the reference's --scene (src/main.rs:47-49); other values -> 0 */
    int32_t bvh_policy;        /* 0 = reference (random-axis median split, src/bvh.rs:31-66), 1 = SAH   */
    uint64_t scene_seed;       /* seeds every build-time draw (object placement, Perlin tables, BVH axes) */
    int64_t image_width;       /* <= 0: the scene's in-code value                                        */
    double aspect_ratio;       /* <= 0: the scene's in-code value                                        */
    int32_t samples_per_pixel; /* <= 0: the scene's in-code value                                        */
    int32_t max_depth;         /* <= 0: the scene's in-code value                                        */
    const char *earth_image;   /* NULL: "synthetic:1024x512"; else a path (PPM, JPEG, PNG) or "synthetic:WxH"  */
} rth_scene_options;

/* Builds scene + top-level BVH + camera exactly as `main` does (src/main.rs:645-660) and describes them. */
int rth_scene_build(const rth_scene_options *options, rth_scene **out_scene);
void rth_scene_destroy(rth_scene *scene);
/* Borrowed pointers, valid until rth_scene_destroy. */
const rt_scene_desc *rth_scene_desc(const rth_scene *scene);
const rt_camera *rth_scene_camera(const rth_scene *scene);

/* color_to_rgb(sum / spp) over a frame of per-pixel sums (src/renderer.rs:55-58, src/color.rs:12-19). */
int rth_resolve_rgb8(int32_t width, int32_t height, int32_t spp, const double *rgb_sum, uint8_t *out_rgb8);
/* RGB8 PNG writer (src/renderer.rs:59-72). */
int rth_write_png(const char *path, int32_t width, int32_t height, const uint8_t *rgb8);
/* Deterministic procedural RGB8 map used where assets/earth-large.jpg is unavailable. */
int rth_synthetic_earth(int32_t width, int32_t height, uint8_t *out_rgb8);
/* Image ingest used by ImageTexture (PPM / JPEG, sequential or progressive / PNG / synthetic:WxH); out_rgb8 may be NULL to query size. */
int rth_load_image(const char *path, int32_t *out_width, int32_t *out_height, uint8_t *out_rgb8, int64_t capacity);

const char *rth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
