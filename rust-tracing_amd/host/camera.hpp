// CameraSettings and Camera::new, host side.  get_ray runs on the GPU (csrc/rt_kernel.hip).
//   reference: src/camera.rs:8-110
#pragma once
#include "vec3.hpp"

namespace rt {

struct CameraSettings { // defaults: src/camera.rs:21-37
    FP aspect_ratio = 16.0 / 9.0;
    size_t image_width = 400;
    int32_t samples_per_pixel = 100;
    int32_t max_depth = 50;
    FP vfov = 90.0;
    Point3 look_from = Point3::ZERO();
    Point3 look_at = Point3(0.0, 0.0, -1.0);
    Vec3 vup = Vec3::UP();
    FP defocus_angle = 0.0;
    FP focus_dist = 10.0;
    Color background = Color::ZERO();
};

class Camera {
  public:
    size_t image_width, image_height;
    int32_t samples_per_pixel, max_depth;
    Color background;

    explicit Camera(const CameraSettings &s) {
        image_width = s.image_width;
        samples_per_pixel = s.samples_per_pixel;
        max_depth = s.max_depth;
        background = s.background;

        image_height = (size_t)((FP)image_width / s.aspect_ratio); // `as usize` truncation (src/camera.rs:69)

        const FP theta = degrees_to_radians(s.vfov);
        const FP h = std::tan(theta / 2.0);

        const FP viewport_height = 2.0 * h * s.focus_dist;
        const FP viewport_width = viewport_height * ((FP)image_width / (FP)image_height);

        const Vec3 w = (s.look_from - s.look_at).normalize();
        const Vec3 u = s.vup.cross(w).normalize();
        const Vec3 v = w.cross(u);

        const Vec3 viewport_u = viewport_width * u;
        const Vec3 viewport_v = -viewport_height * v;

        center = s.look_from;
        pixel_delta_u = viewport_u / (FP)image_width;
        pixel_delta_v = viewport_v / (FP)image_height;

        const Vec3 viewport_upper_left = center - s.focus_dist * w - viewport_u * 0.5 - viewport_v * 0.5;
        pixel00_loc = viewport_upper_left + 0.5 * (pixel_delta_u + pixel_delta_v);

        defocus_angle = s.defocus_angle;
        const FP defocus_radius = s.focus_dist * std::tan(degrees_to_radians(s.defocus_angle / 2.0));
        defocus_disk_u = u * defocus_radius;
        defocus_disk_v = v * defocus_radius;
    }

    rt_camera pod() const {
        rt_camera c{};
        c.image_width = (int32_t)image_width;
        c.image_height = (int32_t)image_height;
        c.samples_per_pixel = samples_per_pixel;
        c.max_depth = max_depth;
        c.background = background.pod();
        c.center = center.pod();
        c.pixel00_loc = pixel00_loc.pod();
        c.pixel_delta_u = pixel_delta_u.pod();
        c.pixel_delta_v = pixel_delta_v.pod();
        c.defocus_angle = defocus_angle;
        c.defocus_disk_u = defocus_disk_u.pod();
        c.defocus_disk_v = defocus_disk_v.pod();
        return c;
    }

  private:
    Point3 center, pixel00_loc;
    Vec3 pixel_delta_u, pixel_delta_v;
    FP defocus_angle;
    Vec3 defocus_disk_u, defocus_disk_v;
};

} // namespace rt
