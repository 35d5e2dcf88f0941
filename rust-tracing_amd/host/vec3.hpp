// Vec3 / Point3 / Color, Interval and AABB as the host side needs them (construction of scenes, cameras
// and bounding boxes).  Operation order follows the reference so that derived quantities (quad normals,
// camera basis, bounding boxes) are bit-identical to what the Rust code would produce from the same inputs.
//   reference: src/vec3.rs:8-289, src/interval.rs:5-69, src/aabb.rs:9-105
#pragma once
#include "common.hpp"
#include "rt_amd.h"
#include <cassert>

namespace rt {

struct Vec3 {
    FP x = 0, y = 0, z = 0;

    constexpr Vec3() = default;
    constexpr Vec3(FP x_, FP y_, FP z_) : x(x_), y(y_), z(z_) {}
    static constexpr Vec3 splat(FP v) { return Vec3(v, v, v); }

    static constexpr Vec3 ZERO() { return splat(0.0); }
    static constexpr Vec3 ONE() { return splat(1.0); }
    static constexpr Vec3 INFINITY_() { return splat(__builtin_inf()); }
    static constexpr Vec3 NEG_INFINITY_() { return splat(-__builtin_inf()); }
    static constexpr Vec3 RIGHT() { return Vec3(1, 0, 0); }
    static constexpr Vec3 UP() { return Vec3(0, 1, 0); }
    static constexpr Vec3 FORWARD() { return Vec3(0, 0, 1); }
    static constexpr Vec3 LEFT() { return Vec3(-1, 0, 0); }
    static constexpr Vec3 DOWN() { return Vec3(0, -1, 0); }
    static constexpr Vec3 BACKWARD() { return Vec3(0, 0, -1); }

    // src/vec3.rs:42-52
    static Vec3 random() {
        FP a = rt::random(), b = rt::random(), c = rt::random();
        return Vec3(a, b, c);
    }
    static Vec3 random_range(FP lo, FP hi) {
        FP a = thread_rng().gen_range(lo, hi), b = thread_rng().gen_range(lo, hi),
           c = thread_rng().gen_range(lo, hi);
        return Vec3(a, b, c);
    }

    FP dot(const Vec3 &r) const { return x * r.x + y * r.y + z * r.z; }
    FP length_squared() const { return dot(*this); }
    FP length() const { return std::sqrt(dot(*this)); }
    FP length_recip() const { return 1.0 / length(); }                 // f64::recip
    Vec3 normalize() const { return *this * length_recip(); }          // src/vec3.rs:128-131
    Vec3 cross(const Vec3 &r) const {
        return Vec3(y * r.z - z * r.y, z * r.x - x * r.z, x * r.y - y * r.x);
    }

    FP operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    FP &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }

    Vec3 operator-() const { return Vec3(-x, -y, -z); }
    Vec3 operator+(const Vec3 &r) const { return Vec3(x + r.x, y + r.y, z + r.z); }
    Vec3 operator-(const Vec3 &r) const { return Vec3(x - r.x, y - r.y, z - r.z); }
    Vec3 operator*(const Vec3 &r) const { return Vec3(x * r.x, y * r.y, z * r.z); }
    Vec3 operator*(FP s) const { return Vec3(x * s, y * s, z * s); }
    Vec3 operator/(FP s) const { return *this * (1.0 / s); }           // src/vec3.rs:244-249: times reciprocal
    Vec3 &operator+=(const Vec3 &r) { x += r.x; y += r.y; z += r.z; return *this; }

    rt_vec3 pod() const { return rt_vec3{x, y, z}; }
};
inline Vec3 operator*(FP s, const Vec3 &v) { return v * s; }

using Point3 = Vec3;
using Color = Vec3;

struct Interval {
    FP min = 0, max = 0; // #[derive(Default)]
    constexpr Interval() = default;
    constexpr Interval(FP a, FP b) : min(a), max(b) {}
    static Interval from_intervals(const Interval &a, const Interval &b) {
        return Interval(std::fmin(a.min, b.min), std::fmax(a.max, b.max));
    }
    Interval expand(FP delta) const { return Interval(min - delta * 0.5, max + delta * 0.5); }
    FP size() const { return max - min; }
    Interval operator+(FP r) const { return Interval(min + r, max + r); }
};

struct AABB {
    Interval x, y, z; // Default: all zero, exactly as the reference's derive(Default) (src/aabb.rs:9)

    AABB() = default;
    AABB(Interval x_, Interval y_, Interval z_) : x(x_), y(y_), z(z_) {}
    static AABB from_points(const Point3 &a, const Point3 &b) {
        return AABB(Interval(std::fmin(a.x, b.x), std::fmax(a.x, b.x)),
                    Interval(std::fmin(a.y, b.y), std::fmax(a.y, b.y)),
                    Interval(std::fmin(a.z, b.z), std::fmax(a.z, b.z)));
    }
    static AABB from_aabbs(const AABB &a, const AABB &b) {
        return AABB(Interval::from_intervals(a.x, b.x), Interval::from_intervals(a.y, b.y),
                    Interval::from_intervals(a.z, b.z));
    }
    AABB pad() const { // src/aabb.rs:35-53
        const FP delta = 0.0001;
        AABB r = *this;
        if (r.x.size() < delta) r.x = r.x.expand(delta);
        if (r.y.size() < delta) r.y = r.y.expand(delta);
        if (r.z.size() < delta) r.z = r.z.expand(delta);
        return r;
    }
    const Interval &axis(int n) const {
        assert(n >= 0 && n <= 2);
        return n == 0 ? x : (n == 1 ? y : z);
    }
    AABB operator+(const Vec3 &o) const { return AABB(x + o.x, y + o.y, z + o.z); }

    rt_aabb pod() const { return rt_aabb{{x.min, y.min, z.min}, {x.max, y.max, z.max}}; }
};

} // namespace rt
