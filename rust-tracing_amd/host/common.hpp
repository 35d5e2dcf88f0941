// Host-side mirror of the reference's scalar basics and of the `rand` calls its scene builders make.
//   reference: src/common.rs:1-8 (FP, PI, degrees_to_radians)
//              rand 0.8.5 call sites used at scene-build time: src/main.rs:70-91,:523, src/bvh.rs:32,
//              src/perlin.rs:18-23,:76, src/vec3.rs:42-52
// The reference's generator is OS-seeded; here every build-time draw comes from one explicitly seeded
// stream so that a scene is a pure function of its seed (SURVEY.md §8d "scene_seed").
#pragma once
#include <cmath>
#include <cstdint>

namespace rt {

using FP = double;

constexpr FP PI = 3.14159265358979323846264338327950288;

inline FP degrees_to_radians(FP degrees) { return degrees * PI / 180.0; }

inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

// Build-time generator (one stream per thread, like thread_rng(), but seeded by seed_rng()).
class HostRng {
  public:
    explicit HostRng(uint64_t seed = 1) { reseed(seed); }
    void reseed(uint64_t seed) { state_ = mix64(seed + 0x9E3779B97F4A7C15ull) ^ 0x5CE4E5B95CE4E5B9ull; }
    uint64_t next_u64() {
        state_ += 0x9E3779B97F4A7C15ull;
        return mix64(state_);
    }
    uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }

    // rand::random::<f64>(): 53 random bits scaled to [0, 1).
    FP random() { return (FP)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }

    // thread_rng().gen_range(lo..hi) for f64 (rand 0.8 UniformFloat::sample_single): a 52-bit fraction in
    // [1, 2) minus one, scaled; redrawn in the (rounding-only) case that the result reaches `hi`.
    FP gen_range(FP lo, FP hi) {
        const FP scale = hi - lo;
        for (;;) {
            uint64_t bits = (next_u64() >> 12) | 0x3FF0000000000000ull;
            FP value1_2;
            __builtin_memcpy(&value1_2, &bits, 8);
            FP res = (value1_2 - 1.0) * scale + lo;
            if (res < hi) return res;
        }
    }

    // gen_range(lo..=hi) on a 32-bit integer type (rand 0.8 UniformInt::sample_single_inclusive:
    // widening multiply with a rejection zone).
    int32_t gen_range_inclusive_i32(int32_t lo, int32_t hi) {
        uint32_t range = (uint32_t)(hi - lo) + 1u;
        if (range == 0) return (int32_t)next_u32();
        uint32_t zone = (range << __builtin_clz(range)) - 1u;
        for (;;) {
            uint64_t m = (uint64_t)next_u32() * (uint64_t)range;
            if ((uint32_t)m <= zone) return lo + (int32_t)(m >> 32);
        }
    }
    // the same on usize (64-bit arithmetic), as `let axis: usize = gen_range(0..=2)` uses (src/bvh.rs:32)
    uint64_t gen_range_inclusive_usize(uint64_t lo, uint64_t hi) {
        uint64_t range = hi - lo + 1u;
        if (range == 0) return next_u64();
        uint64_t zone = (range << __builtin_clzll(range)) - 1u;
        for (;;) {
            unsigned __int128 m = (unsigned __int128)next_u64() * range;
            if ((uint64_t)m <= zone) return lo + (uint64_t)(m >> 64);
        }
    }

  private:
    uint64_t state_;
};

// The process-wide build-time stream (one per thread) and the free functions the scene code calls, named
// after what they replace.
HostRng &thread_rng();
inline void seed_rng(uint64_t seed) { thread_rng().reseed(seed); }
inline FP random() { return thread_rng().random(); }

} // namespace rt
