// Perlin table generation (host side of the reference's Perlin; the noise evaluation itself runs on the GPU).
//   reference: src/perlin.rs:5-24 (new), :66-79 (perlin_generate_perm, permute)
#pragma once
#include "vec3.hpp"
#include <array>
#include <utility>

namespace rt {

constexpr int POINT_COUNT = 256;

class Perlin {
  public:
    // Perlin::new (src/perlin.rs:15-25): 256 vectors uniform in [-1,1)^3 (NOT normalised), then the x, y and
    // z permutations, drawn in that order.
    Perlin() {
        for (int i = 0; i < POINT_COUNT; ++i) ranvec[i] = Vec3::random_range(-1.0, 1.0);
        perm_x = perlin_generate_perm();
        perm_y = perlin_generate_perm();
        perm_z = perlin_generate_perm();
    }

    rt_perlin pod() const {
        rt_perlin p;
        for (int i = 0; i < POINT_COUNT; ++i) {
            p.ranvec[i] = ranvec[i].pod();
            p.perm_x[i] = perm_x[i];
            p.perm_y[i] = perm_y[i];
            p.perm_z[i] = perm_z[i];
        }
        return p;
    }

    std::array<Vec3, POINT_COUNT> ranvec;
    std::array<int32_t, POINT_COUNT> perm_x, perm_y, perm_z;

  private:
    static std::array<int32_t, POINT_COUNT> perlin_generate_perm() {
        std::array<int32_t, POINT_COUNT> p;
        for (int i = 0; i < POINT_COUNT; ++i) p[i] = i;
        permute(p, POINT_COUNT);
        return p;
    }
    // Fisher-Yates from the top (src/perlin.rs:73-79)
    static void permute(std::array<int32_t, POINT_COUNT> &p, int n) {
        for (int i = n - 1; i >= 1; --i) {
            int32_t target = thread_rng().gen_range_inclusive_i32(0, i);
            std::swap(p[i], p[target]);
        }
    }
};

} // namespace rt
