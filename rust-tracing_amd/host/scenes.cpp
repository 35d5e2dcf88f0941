#include "scenes.hpp"

namespace rt {

HostRng &thread_rng() {
    thread_local HostRng rng(1);
    return rng;
}
BvhPolicy &bvh_policy() { // per thread, like thread_rng(): concurrent scene builds do not see each other's choice
    thread_local BvhPolicy p = BvhPolicy::Reference;
    return p;
}

namespace {

template <class T, class... A> std::shared_ptr<T> mk(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }

std::shared_ptr<Lambertian> lambertian(FP r, FP g, FP b) { return mk<Lambertian>(mk<SolidColor>(r, g, b)); }

Camera make_camera(CameraSettings s, const SceneOptions &o) {
    if (o.image_width > 0) s.image_width = (size_t)o.image_width;
    if (o.aspect_ratio > 0.0) s.aspect_ratio = o.aspect_ratio;
    if (o.samples_per_pixel > 0) s.samples_per_pixel = o.samples_per_pixel;
    if (o.max_depth > 0) s.max_depth = o.max_depth;
    return Camera(s);
}

const Color SKY(0.7, 0.8, 1.0);

} // namespace

// src/main.rs:56-138
SceneResult random_balls(const SceneOptions &o) {
    HittableList world;

    auto ground_material = mk<Lambertian>(mk<SolidColor>(Color::splat(0.5)));
    world.add(mk<Sphere>(Point3::DOWN() * 1000.0, 1000.0, ground_material));

    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            const FP choose_mat = random();
            const FP cx = (FP)a + 0.9 * random();
            const FP cz = (FP)b + 0.9 * random();
            const Point3 center(cx, 0.2, cz);

            if ((center - Point3(4.0, 0.2, 0.0)).length() > 0.9) {
                if (choose_mat < 0.8) {
                    const Color c1 = Color::random();
                    const Color c2 = Color::random();
                    auto color = mk<SolidColor>(c1 * c2);
                    const Sphere s(center, 0.2, mk<Lambertian>(color));
                    const FP lift = random();
                    world.add(mk<Sphere>(s.with_target(center + Vec3::UP() * lift * 0.5)));
                } else if (choose_mat < 0.95) {
                    const Color albedo = Color::random_range(0.5, 1.0);
                    const FP fuzz = thread_rng().gen_range(0.0, 0.5);
                    world.add(mk<Sphere>(center, 0.2, mk<Metal>(albedo, fuzz)));
                } else {
                    world.add(mk<Sphere>(center, 0.2, mk<Dielectric>(1.5)));
                }
            }
        }
    }

    world.add(mk<Sphere>(Point3(0.0, 1.0, 0.0), 1.0, mk<Dielectric>(1.5)));
    world.add(mk<Sphere>(Point3(-4.0, 1.0, 0.0), 1.0, lambertian(0.4, 0.2, 0.1)));
    world.add(mk<Sphere>(Point3(4.0, 1.0, 0.0), 1.0, mk<Metal>(Color(0.7, 0.6, 0.5), 0.0)));

    CameraSettings s;
    s.aspect_ratio = 16.0 / 9.0;
    s.image_width = 600;
    s.samples_per_pixel = 128;
    s.max_depth = 8;
    s.background = SKY;
    s.vfov = 20.0;
    s.look_from = Point3(13.0, 2.0, 3.0);
    s.look_at = Point3::ZERO();
    s.defocus_angle = 0.6;
    s.focus_dist = 10.0;
    return {std::move(world), make_camera(s, o)};
}

// src/main.rs:140-173
SceneResult two_spheres(const SceneOptions &o) {
    HittableList world;

    std::shared_ptr<Material> checker =
        mk<Lambertian>(CheckerTexture::new_from_colors(0.32, Color(0.2, 0.3, 0.1), Color::splat(0.9)));

    world.add(mk<Sphere>(Point3(0.0, -10.0, 0.0), 10.0, checker));
    world.add(mk<Sphere>(Point3(0.0, 10.0, 0.0), 10.0, checker));

    CameraSettings s;
    s.aspect_ratio = 16.0 / 9.0;
    s.image_width = 1200;
    s.samples_per_pixel = 128;
    s.max_depth = 8;
    s.background = SKY;
    s.vfov = 20.0;
    s.look_from = Point3(13.0, 2.0, 3.0);
    s.look_at = Point3::ZERO();
    return {std::move(world), make_camera(s, o)};
}

// src/main.rs:175-203
SceneResult earth(const SceneOptions &o) {
    HittableList world;

    auto earth_texture = mk<Lambertian>(mk<ImageTexture>(o.earth_image));
    world.add(mk<Sphere>(Point3(0.0, 0.0, 0.0), 2.0, earth_texture));

    CameraSettings s;
    s.aspect_ratio = 16.0 / 9.0;
    s.image_width = 1200;
    s.samples_per_pixel = 128;
    s.max_depth = 8;
    s.background = SKY;
    s.vfov = 20.0;
    s.look_from = Point3(12.0, 0.0, 0.0);
    s.look_at = Point3::ZERO();
    return {std::move(world), make_camera(s, o)};
}

// src/main.rs:205-237
SceneResult two_perlin_spheres(const SceneOptions &o) {
    HittableList world;

    std::shared_ptr<Material> perlin_texture = mk<Lambertian>(mk<NoiseTexture>(4.0));

    world.add(mk<Sphere>(Point3(0.0, -1000.0, 0.0), 1000.0, perlin_texture));
    world.add(mk<Sphere>(Point3(0.0, 2.0, 0.0), 2.0, perlin_texture));

    CameraSettings s;
    s.aspect_ratio = 16.0 / 9.0;
    s.image_width = 1200;
    s.samples_per_pixel = 128;
    s.max_depth = 8;
    s.background = SKY;
    s.vfov = 20.0;
    s.look_from = Point3(13.0, 2.0, 3.0);
    s.look_at = Point3::ZERO();
    return {std::move(world), make_camera(s, o)};
}

// src/main.rs:239-294
SceneResult quads(const SceneOptions &o) {
    HittableList world;

    auto left_red = lambertian(1.0, 0.2, 0.2);
    auto back_green = lambertian(0.2, 1.0, 0.2);
    auto right_blue = lambertian(0.2, 0.2, 1.0);
    auto upper_orange = lambertian(1.0, 0.5, 0.0);
    auto lower_teal = lambertian(0.2, 0.8, 0.8);

    world.add(mk<Quad>(Point3(-3.0, -2.0, 5.0), Vec3::BACKWARD() * 4.0, Vec3::UP() * 4.0, left_red));
    world.add(mk<Quad>(Point3(-2.0, -2.0, 0.0), Vec3::RIGHT() * 4.0, Vec3::UP() * 4.0, back_green));
    world.add(mk<Quad>(Point3(3.0, -2.0, 1.0), Vec3::FORWARD() * 4.0, Vec3::UP() * 4.0, right_blue));
    world.add(mk<Quad>(Point3(-2.0, 3.0, 1.0), Vec3::RIGHT() * 4.0, Vec3::FORWARD() * 4.0, upper_orange));
    world.add(mk<Quad>(Point3(-2.0, -3.0, 5.0), Vec3::RIGHT() * 4.0, Vec3::BACKWARD() * 4.0, lower_teal));

    CameraSettings s;
    s.aspect_ratio = 1.0;
    s.image_width = 1200;
    s.samples_per_pixel = 128;
    s.max_depth = 8;
    s.background = SKY;
    s.vfov = 80.0;
    s.look_from = Point3::FORWARD() * 9.0;
    s.look_at = Point3::ZERO();
    return {std::move(world), make_camera(s, o)};
}

// src/main.rs:296-342
SceneResult simple_light(const SceneOptions &o) {
    HittableList world;

    std::shared_ptr<Texture> perlin_texture = mk<NoiseTexture>(4.0);

    world.add(mk<Sphere>(Point3(0.0, -1000.0, 0.0), 1000.0, mk<Lambertian>(perlin_texture)));
    world.add(mk<Sphere>(Point3(0.0, 2.0, 0.0), 2.0, mk<Lambertian>(perlin_texture)));

    std::shared_ptr<Material> diffuse_light = mk<DiffuseLight>(mk<SolidColor>(4.0, 4.0, 4.0));

    world.add(mk<Quad>(Point3(3.0, 1.0, -2.0), Vec3::RIGHT() * 2.0, Vec3::UP() * 2.0, diffuse_light));
    world.add(mk<Sphere>(Point3(0.0, 7.0, 0.0), 2.0, diffuse_light));

    CameraSettings s;
    s.aspect_ratio = 16.0 / 9.0;
    s.image_width = 600;
    s.samples_per_pixel = 1024;
    s.max_depth = 8;
    s.background = Color::ZERO();
    s.vfov = 20.0;
    s.look_from = Point3(26.0, 3.0, 6.0);
    s.look_at = Point3::UP() * 2.0;
    return {std::move(world), make_camera(s, o)};
}

namespace {
CameraSettings cornell_camera() {
    CameraSettings s;
    s.aspect_ratio = 1.0;
    s.image_width = 600;
    s.samples_per_pixel = 4096;
    s.max_depth = 8;
    s.background = Color::ZERO();
    s.vfov = 40.0;
    s.look_from = Point3(278.0, 278.0, -800.0);
    s.look_at = Point3(278.0, 278.0, 0.0);
    return s;
}
} // namespace

// src/main.rs:344-421
SceneResult cornell_box(const SceneOptions &o) {
    HittableList world;

    auto red = lambertian(0.65, 0.05, 0.05);
    std::shared_ptr<Material> white = lambertian(0.73, 0.73, 0.73);
    auto green = lambertian(0.12, 0.45, 0.15);
    auto light = mk<DiffuseLight>(mk<SolidColor>(15.0, 15.0, 15.0));

    world.add(mk<Quad>(Point3(555.0, 0.0, 555.0), Vec3::UP() * 555.0, Vec3::BACKWARD() * 555.0, green));
    world.add(mk<Quad>(Point3::ZERO(), Vec3::UP() * 555.0, Vec3::FORWARD() * 555.0, red));
    world.add(mk<Quad>(Point3(343.0, 554.0, 332.0), Vec3::LEFT() * 130.0, Vec3::BACKWARD() * 105.0, light));
    world.add(mk<Quad>(Point3::FORWARD() * 555.0, Vec3::RIGHT() * 555.0, Vec3::BACKWARD() * 555.0, white));
    world.add(mk<Quad>(Point3::ONE() * 555.0, Vec3::LEFT() * 555.0, Vec3::BACKWARD() * 555.0, white));
    world.add(mk<Quad>(Point3(555.0, 0.0, 555.0), Vec3::LEFT() * 555.0, Vec3::UP() * 555.0, white));

    std::shared_ptr<Hittable> box1 = Quad::cube(Point3::ZERO(), Point3(165.0, 330.0, 165.0), white);
    box1 = mk<RotateY>(box1, 15.0);
    box1 = mk<Translate>(box1, Vec3(265.0, 0.0, 295.0));
    world.add(box1);

    std::shared_ptr<Hittable> box2 = Quad::cube(Point3::ZERO(), Point3::splat(165.0), white);
    box2 = mk<RotateY>(box2, -18.0);
    box2 = mk<Translate>(box2, Vec3(130.0, 0.0, 65.0));
    world.add(box2);

    return {std::move(world), make_camera(cornell_camera(), o)};
}

// src/main.rs:423-506
SceneResult cornell_smoke(const SceneOptions &o) {
    HittableList world;

    auto red = lambertian(0.65, 0.05, 0.05);
    std::shared_ptr<Material> white = lambertian(0.73, 0.73, 0.73);
    auto green = lambertian(0.12, 0.45, 0.15);
    auto light = mk<DiffuseLight>(mk<SolidColor>(7.0, 7.0, 7.0));

    world.add(mk<Quad>(Point3(555.0, 0.0, 555.0), Vec3::UP() * 555.0, Vec3::BACKWARD() * 555.0, green));
    world.add(mk<Quad>(Point3::ZERO(), Vec3::UP() * 555.0, Vec3::FORWARD() * 555.0, red));
    world.add(mk<Quad>(Point3(113.0, 554.0, 127.0), Vec3::RIGHT() * 330.0, Vec3::FORWARD() * 305.0, light));
    world.add(mk<Quad>(Point3::FORWARD() * 555.0, Vec3::RIGHT() * 555.0, Vec3::BACKWARD() * 555.0, white));
    world.add(mk<Quad>(Point3::ONE() * 555.0, Vec3::LEFT() * 555.0, Vec3::BACKWARD() * 555.0, white));
    world.add(mk<Quad>(Point3(555.0, 0.0, 555.0), Vec3::LEFT() * 555.0, Vec3::UP() * 555.0, white));

    std::shared_ptr<Hittable> box1 = Quad::cube(Point3::ZERO(), Point3(165.0, 330.0, 165.0), white);
    box1 = mk<RotateY>(box1, 15.0);
    box1 = mk<Translate>(box1, Vec3(265.0, 0.0, 295.0));
    world.add(ConstantMedium::new_from_color(box1, 0.01, Color::ZERO()));

    std::shared_ptr<Hittable> box2 = Quad::cube(Point3::ZERO(), Point3::splat(165.0), white);
    box2 = mk<RotateY>(box2, -18.0);
    box2 = mk<Translate>(box2, Vec3(130.0, 0.0, 65.0));
    world.add(ConstantMedium::new_from_color(box2, 0.01, Color::ONE()));

    return {std::move(world), make_camera(cornell_camera(), o)};
}

// src/main.rs:508-639
SceneResult final_scene(const SceneOptions &o) {
    HittableList world;

    std::shared_ptr<Material> ground = lambertian(0.48, 0.83, 0.53);
    HittableList boxes1;

    for (int i = 0; i < 20; ++i) {
        for (int j = 0; j < 20; ++j) {
            const FP side = 100.0;
            const FP x0 = -1000.0 + (FP)i * side;
            const FP x1 = x0 + side;
            const FP z0 = -1000.0 + (FP)j * side;
            const FP z1 = z0 + side;
            const FP y0 = 0.0;
            const FP y1 = thread_rng().gen_range(1.0, 101.0);
            boxes1.add(Quad::cube(Point3(x0, y0, z0), Point3(x1, y1, z1), ground));
        }
    }

    // Green Ground Boxes
    world.add(mk<BVHNode>(boxes1));

    // Light source
    world.add(mk<Quad>(Point3(123.0, 554.0, 147.0), Vec3::RIGHT() * 300.0, Vec3::FORWARD() * 265.0,
                       mk<DiffuseLight>(mk<SolidColor>(7.0, 7.0, 7.0))));

    // Motion blurred sphere
    const Point3 center1(400.0, 400.0, 200.0);
    const Point3 center2 = center1 + Vec3::RIGHT() * 30.0;
    auto sphere_material = lambertian(0.7, 0.3, 0.1);
    world.add(mk<Sphere>(Sphere(center1, 50.0, sphere_material).with_target(center2)));

    // Glass Sphere
    world.add(mk<Sphere>(Point3(260.0, 150.0, 45.0), 50.0, mk<Dielectric>(1.5)));
    // Fuzzy Metal Sphere
    world.add(mk<Sphere>(Point3(0.0, 150.0, 145.0), 50.0, mk<Metal>(Color(0.8, 0.8, 0.9), 1.0)));

    // Subsurface Scattering Sphere: the same sphere object is both a glass surface and a medium boundary
    std::shared_ptr<Hittable> boundary = mk<Sphere>(Point3(360.0, 150.0, 145.0), 70.0, mk<Dielectric>(1.5));
    world.add(boundary);
    world.add(ConstantMedium::new_from_color(boundary, 0.2, Color(0.2, 0.4, 0.9)));

    // Global Scene Fog
    std::shared_ptr<Hittable> fog_boundary = mk<Sphere>(Point3::ZERO(), 5000.0, mk<Dielectric>(1.5));
    world.add(ConstantMedium::new_from_color(fog_boundary, 0.0001, Color::ONE()));

    // Earth Sphere
    auto earth_material = mk<Lambertian>(mk<ImageTexture>(o.earth_image));
    world.add(mk<Sphere>(Point3(400.0, 200.0, 400.0), 100.0, earth_material));

    // Noise Sphere
    auto perlin_texture = mk<NoiseTexture>(0.1);
    world.add(mk<Sphere>(Point3(220.0, 280.0, 300.0), 80.0, mk<Lambertian>(perlin_texture)));

    // Box of Spheres
    HittableList boxes2;
    std::shared_ptr<Material> white = lambertian(0.73, 0.73, 0.73);
    for (int k = 0; k < 1000; ++k) {
        const Point3 c = Point3::random_range(0.0, 165.0);
        boxes2.add(mk<Sphere>(c, 10.0, white));
    }

    world.add(mk<Translate>(mk<RotateY>(mk<BVHNode>(boxes2), 15.0), Vec3(-100.0, 270.0, 395.0)));

    CameraSettings s;
    s.aspect_ratio = 1.0;
    s.image_width = 800;
    s.samples_per_pixel = 8192;
    s.max_depth = 40;
    s.background = Color::ZERO();
    s.vfov = 40.0;
    s.look_from = Point3(478.0, 278.0, -600.0);
    s.look_at = Point3(278.0, 278.0, 0.0);
    return {std::move(world), make_camera(s, o)};
}

SceneResult build_scene(int scene, const SceneOptions &o) {
    switch (scene) {
    case 0: return random_balls(o);
    case 1: return two_spheres(o);
    case 2: return earth(o);
    case 3: return two_perlin_spheres(o);
    case 4: return quads(o);
    case 5: return simple_light(o);
    case 6: return cornell_box(o);
    case 7: return cornell_smoke(o);
    case 8: return final_scene(o);
    default: return random_balls(o);
    }
}

const char *scene_name(int scene) {
    static const char *names[] = {"random_balls", "two_spheres", "earth",       "two_perlin_spheres", "quads",
                                  "simple_light", "cornell_box", "cornell_smoke", "final_scene"};
    return (scene >= 0 && scene <= 8) ? names[scene] : names[0];
}

} // namespace rt
