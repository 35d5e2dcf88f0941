// Output-stage conversion.  reference: src/color.rs:3-19
#pragma once
#include "vec3.hpp"
#include "../csrc/rt_shared_math.h"
#include <array>

namespace rt {

// x.powf(1.0 / 2.2) by the ABI's fixed algorithm (rt_shared_math.h; within 4 ulp of libm's pow): the same code runs on the
// device (rt_resolve_rgb8_device), so a frame resolved there is byte-identical to one resolved here
inline FP linear_to_gamma(FP linear_component) { return rtm::rt_gamma_encode(linear_component); }

// `(256.0 * x.clamp(0.0, 0.999)) as u8`: f64::clamp keeps NaN, and `NaN as u8` is 0 (Rust saturating cast)
inline uint8_t quantise(FP gamma_component) { return rtm::rt_quantise(gamma_component); }

inline std::array<uint8_t, 3> color_to_rgb(const Color &rgb) {
    return {quantise(linear_to_gamma(rgb.x)), quantise(linear_to_gamma(rgb.y)), quantise(linear_to_gamma(rgb.z))};
}

} // namespace rt
